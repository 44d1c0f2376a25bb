// Arguments of the query-side row-tile chain kernels (chain.hip); shared with the host sequence in api.cpp.
#pragma once
#include <hip/hip_runtime.h>

struct ChainArgs {
  int kind;                  // 0 = A, 1 = C, 2 = D1, 3 = D2
  int M;                     // rows
  int tile_p;                // row tiles: 0 = chosen per launch (chain.hip), 1 = 32 rows, 2 = 64 rows, 3 = 16-row units (chain16.hip)
  const float* x; int ldx;   // chain input (M, C)
  // kind A, x == nullptr: the chain input is the cross attention's context, combined here from the attention kernel's
  // split-key partials (attn.hip: part_o fragment images + (m, l) per row) instead of by a separate attn_combine launch
  const float4* part_o; const float2* part_ml;
  int pc_heads, pc_chunks, pc_nq;            // heads, key chunks (<= 4), queries per sample
  float pc_kc2;                              // log2(e) / sqrt(head dim)
  const float4* w1; const float* b1;            // first Linear (C x C packed)
  const float* res; int ldres; int res_mod;     // residual rows (row % res_mod when res_mod > 0: one copy shared by all samples)
  const float* ln_g; const float* ln_b; float eps;   // kind A: LayerNorm after the first Linear
  float* y1; int ldy1;       // result of the first stage (h / f), row-major
  const float4* w2; const float* b2; int n2;    // trailing Linear: n2 C-wide column passes (0 = none), packed (n2*C x C)
  float* y2; int ldy2;
  // kind D
  const float4* wf4; const float* bf4;          // packed (5C x C): reg_branch.0 | intermediate.dense ; bias (5C)
  const float* wreg2; const float* breg2;       // (3, C) raw, (3)
  const float* xyz_in; float* xyz_out;          // (M, 3)
  const float4* wout; const float* bout;        // packed (C x 4C)
  const float* ln2_g; const float* ln2_b;
  float* y3; int ldy3;       // feats
  // chain16's one-unit tiles: every packed weight above has a native 16x16x4 image this many bytes away (handle.cpp native16)
  long long native_delta;
};

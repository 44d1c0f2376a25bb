// Row-tile chain kernels for the query side of a decoder block (point_METRO_layer.forward + pointer_layer.forward,
// lib/models/bricks/pt_metro_transformer.py:56-91,34-40 upstream): the Linears, residual adds and LayerNorms that sit BETWEEN
// the big kernels (cross attention, vector attention) act on M = B * 799 rows of C channels -- as separate launches they
// were ~14 kernels per block of 40-200 us each, every one a panel load + an HBM round trip of the activations.  Here one
// block owns a tile of XS rows and walks a whole chain with the activations resident in LDS:
//
//   kind A  (after a BERT cross attention)   x = ctx
//           t = x Wo^T + bo + residual ; h = LayerNorm(t) -> y1 ; [y2 = h W2^T + b2]        (W2: next query projection, or
//                                                                                             the fused (w_qs|w_ks|w_vs) o fc1)
//   kind C  (after the vector self attention) x = r
//           f = x Wfc2^T + b + residual -> y1 ; y2 = f W2^T + b2                             (W2: composed cross query)
//   kind D1 (after the vector cross attention) x = r
//           f = x Wfc2^T + b + residual -> y1
//           u = relu(f Wreg0^T + b) ; xyz' = xyz + u Wreg2^T + b                             (reg_branch, 3 outputs)
//           -- its own launch: the next block's neighbour searches wait for xyz' only and overlap the rest of the tail
//   kind D2 x = f
//           o = sum_s gelu(f Wint_s^T + b_s) Wout_s^T  (four C-wide slabs of the 4C intermediate, never materialised:
//               the K = 4C contraction accumulates slab by slab in k order) ; t = o + bout + f ; g = LayerNorm(t) -> y3
//           y2 = g W2^T + b2                                                                  (W2: next block's embedding | query)
//
// Layout: X[channel][row] in LDS (row stride XS + 1: conflict-free for the transposed fills as well as for the MFMA
// operand reads); every GEMM is D[c'][j] = sum_c W[c'][c] X[c][j] with the packed weight fragment as A operand (1 KiB
// wave loads from L2 through a buffer descriptor) and an X row as B operand -- the result layout (lane = row, registers
// = channels) is what the next GEMM's X wants, so epilogues are register-local.  LayerNorm: per-row sums over the
// registers, the two half-waves and (through LDS) the waves; two-pass (mean, then centred squares) like the stand-alone
// kernel.  A sample's rows never meet another sample's: results do not depend on the batch.
#include "common.h"
#include <algorithm>

#include "chain.h"

#include "lds_gemm.h"

template <int C, int P, int NW, int KIND>
__global__ __launch_bounds__(NW * 64, KIND == 3 ? NW / 4 : NW / 2) void chain_kernel(ChainArgs A) {
  constexpr int XS = 32 * P, XSP = XS + 1, NTILE = C / 32, TPW = NTILE / NW, KCH = C / 8, NT = NW * 64;
  static_assert(NTILE % NW == 0, "waves must divide the channel tiles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X0 = smem;                       // C * XSP
  float* X1 = X0 + C * XSP;               // C * XSP (kind D2 only)
  float* red = KIND == 3 ? X1 + C * XSP : X0 + C * XSP;   // NW * XS partial row sums
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, h = lane >> 5;
  const int items = (A.M + XS - 1) / XS;
  const int tile0 = wv * TPW;             // this wave's first channel tile within a C-wide pass
  const unsigned CC4 = (unsigned)(C * C * 4);

  // ---- helpers (lambdas keep the register arrays in scope) ----------------------------------------------------------------
  // bias + optional residual (row-major, 16-byte loads per lane) + optional relu / gelu on acc; channel of register i of
  // tile tp: (tile0 + tp) * 32 + mfma_row(i, h); row of column tile p: row0 + 32 p + j
  auto add_bias = [&](f32x16 (&acc)[TPW][P], const float* bias, int act) {
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      const int cbase = (tile0 + tp) * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bb = *reinterpret_cast<const float4*>(bias + cbase + 8 * g);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[tp][p][4 * g + e] + (&bb.x)[e];
            if (act == 1) v = relu_nan(v);
            if (act == 2) v = gelu_erf(v);
            acc[tp][p][4 * g + e] = v;
          }
      }
    }
  };
  auto add_rows = [&](f32x16 (&acc)[TPW][P], const float* R, int ld, int mod, int row0) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      int row = min(row0 + 32 * p + j, A.M - 1);
      if (mod > 0) row %= mod;
      const float* rp = R + (size_t)row * ld + 4 * h;
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 rr = *reinterpret_cast<const float4*>(rp + (tile0 + tp) * 32 + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[tp][p][4 * g + e] += (&rr.x)[e];
        }
    }
  };
  auto to_lds = [&](const f32x16 (&acc)[TPW][P], float* X) {
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int i = 0; i < 16; ++i) X[((tile0 + tp) * 32 + mfma_row(i, h)) * XSP + 32 * p + j] = acc[tp][p][i];
  };
  auto to_global = [&](const f32x16 (&acc)[TPW][P], float* Y, int ld, int col0, int row0) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int row = row0 + 32 * p + j;
      if (row >= A.M) continue;
      float* yp = Y + (size_t)row * ld + col0 + 4 * h;
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(yp + (tile0 + tp) * 32 + 8 * g) =
              make_float4(acc[tp][p][4 * g], acc[tp][p][4 * g + 1], acc[tp][p][4 * g + 2], acc[tp][p][4 * g + 3]);
    }
  };
  // LayerNorm over the C channels of every row, in place on acc (all waves take part: two block barriers per pass)
  auto layer_norm = [&](f32x16 (&acc)[TPW][P], const float* gamma, const float* beta, float eps) {
    float mean[P], rstd[P];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float s[P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float t = 0.f;
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float d = pass == 0 ? acc[tp][p][i] : acc[tp][p][i] - mean[p];
            t = pass == 0 ? t + d : __builtin_fmaf(d, d, t);       // (explicit: chain16.hip must round the same way)
          }
        s[p] = half_sum(t);
      }
      __syncthreads();                        // previous readers of `red` are done
      if (h == 0)
#pragma unroll
        for (int p = 0; p < P; ++p) red[wv * XS + 32 * p + j] = s[p];
      __syncthreads();
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w * XS + 32 * p + j];
        if (pass == 0) mean[p] = t / (float)C;
        else rstd[p] = 1.0f / sqrtf(t / (float)C + eps);
      }
    }
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      const int cbase = (tile0 + tp) * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 gg = *reinterpret_cast<const float4*>(gamma + cbase + 8 * g);
        const float4 bb = *reinterpret_cast<const float4*>(beta + cbase + 8 * g);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[tp][p][4 * g + e] = __builtin_fmaf((acc[tp][p][4 * g + e] - mean[p]) * rstd[p], (&gg.x)[e], (&bb.x)[e]);
      }
    }
  };
  // trailing wide Linear: n C-wide passes over X, results straight to global
  auto wide_linear = [&](const float4* W, const float* bias, int n, const float* X, float* Y, int ld, int row0) {
    const __amdgpu_buffer_rsrc_t wrs = frag_rsrc(W, (unsigned)n * CC4);
    for (int pass = 0; pass < n; ++pass) {
      f32x16 acc[TPW][P];
      lds_gemm<KCH, XSP, P, TPW, true, TPW * P == 1>(wrs, __builtin_amdgcn_readfirstlane((pass * NTILE + tile0) * KCH * 1024), KCH * 1024, X, acc, lane);
      add_bias(acc, bias + pass * C, 0);
      to_global(acc, Y, ld, pass * C, row0);
    }
  };

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int row0 = item * XS;
    __syncthreads();                          // the previous item's readers of X0 / X1
    if (KIND == 0 && A.x == nullptr) {
      // ---- fill X0 with the cross attention's context, combined from the split-key partials (attn_combine_kernel's
      // arithmetic, same order: w_s = 2^((m_s - M) kc2), ctx = sum_s w_s O_s / sum_s w_s l_s).  Lanes run along the rows:
      // row r of query tile qt is lane r (+32 for the odd float4 of a channel octet) of a partial's fragment image, so a
      // wave's 16-byte loads are contiguous.
      constexpr int RS = NT / XS, DH4 = C / 4, HEADS = 4;      // row sets; float4 groups per row; heads (checked at launch)
      constexpr int G = DH4 / (HEADS * RS), DT = C / HEADS / 32;   // groups per (thread, head); channel tiles per head
      static_assert(G >= 1 && DH4 % (HEADS * RS) == 0 && DT >= 1, "shape");
      const int rr = tid % XS, cgw = tid / XS;
      const int i = min(row0 + rr, A.M - 1);
      const int b = i / A.pc_nq, q = i % A.pc_nq, qt = q >> 5, r = q & 31;
      const int nqt = (A.pc_nq + 31) >> 5, nch = A.pc_chunks;
      // (m, l) of every (head, chunk) of this row: all loads first
      float2 ml[HEADS][4];
#pragma unroll
      for (int hd = 0; hd < HEADS; ++hd)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          ml[hd][s] = A.part_ml[(((size_t)(b * HEADS + hd) * nch + min(s, nch - 1)) * nqt + qt) * 32 + r];
      float w[HEADS][4], rden[HEADS];
#pragma unroll
      for (int hd = 0; hd < HEADS; ++hd) {
        float mx = ml[hd][0].x;
#pragma unroll
        for (int s = 1; s < 4; ++s) mx = fmaxf(mx, ml[hd][s].x);
        float den = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          w[hd][s] = s < nch ? ((ml[hd][s].x == mx) ? 1.0f : __builtin_amdgcn_exp2f((ml[hd][s].x - mx) * A.pc_kc2)) : 0.f;
          den = s < nch ? fmaf(w[hd][s], ml[hd][s].y, den) : den;
        }
        rden[hd] = den;
      }
      // the partial outputs, one head (G groups x 4 chunks of 16-byte loads) in flight at a time
#pragma unroll
      for (int hd = 0; hd < HEADS; ++hd) {
        float4 p[G][4];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const int cg4 = hd * (DH4 / HEADS) + cgw + RS * gi;
          const int d = (cg4 >> 3) % DT, g = (cg4 & 7) >> 1, hh = cg4 & 1;
#pragma unroll
          for (int s = 0; s < 4; ++s)
            p[gi][s] = A.part_o[((((size_t)(b * HEADS + hd) * nch + min(s, nch - 1)) * nqt + qt) * (DT * 4) + d * 4 + g) * 64 + r + 32 * hh];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const int cg4 = hd * (DH4 / HEADS) + cgw + RS * gi;
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc.x = fmaf(w[hd][s], p[gi][s].x, acc.x); acc.y = fmaf(w[hd][s], p[gi][s].y, acc.y);
            acc.z = fmaf(w[hd][s], p[gi][s].z, acc.z); acc.w = fmaf(w[hd][s], p[gi][s].w, acc.w);
          }
          float* xo = X0 + (4 * cg4) * XSP + rr;
          xo[0] = acc.x / rden[hd]; xo[XSP] = acc.y / rden[hd]; xo[2 * XSP] = acc.z / rden[hd]; xo[3 * XSP] = acc.w / rden[hd];
        }
      }
    } else
    // ---- fill X0 with the input tile, transposed: consecutive threads read consecutive channels of one row
    {
      static_assert(NT % C == 0 || C % NT == 0, "threads and channels must nest");
      constexpr int RSTEP = NT >= C ? NT / C : 1, CSTEP = NT >= C ? C : NT;
      const int c0 = tid % CSTEP, r0 = tid / CSTEP;
#pragma unroll 4
      for (int r = r0; r < XS; r += RSTEP) {
        const float* xr = A.x + (size_t)min(row0 + r, A.M - 1) * A.ldx;
#pragma unroll
        for (int c = c0; c < C; c += CSTEP) X0[c * XSP + r] = xr[c];
      }
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t f4rs = frag_rsrc(A.wf4, 5u * CC4);
    if (KIND != 3) {
      // ---- stage 1: first Linear + bias + residual [+ LayerNorm] -> y1 and back into X0
      f32x16 acc[TPW][P];
      lds_gemm<KCH, XSP, P, TPW, true, TPW * P == 1>(frag_rsrc(A.w1, CC4), __builtin_amdgcn_readfirstlane(tile0 * KCH * 1024), KCH * 1024, X0, acc, lane);
      add_bias(acc, A.b1, 0);
      add_rows(acc, A.res, A.ldres, A.res_mod, row0);
      if (KIND == 0) layer_norm(acc, A.ln_g, A.ln_b, A.eps);
      to_global(acc, A.y1, A.ldy1, 0, row0);
      __syncthreads();                          // every wave is done reading the input tile
      to_lds(acc, X0);
      __syncthreads();
      if (KIND != 2) {
        if (A.n2 > 0) wide_linear(A.w2, A.b2, A.n2, X0, A.y2, A.ldy2, row0);
        continue;
      }
      // ---- kind D1.  reg_branch: u = relu(f Wreg0^T + b) (over f in X0: f is in HBM already), xyz' = xyz + u Wreg2^T + b
      {
        f32x16 u[TPW][P];
        lds_gemm<KCH, XSP, P, TPW, true, TPW * P == 1>(f4rs, __builtin_amdgcn_readfirstlane(tile0 * KCH * 1024), KCH * 1024, X0, u, lane);
        add_bias(u, A.bf4, 1);
        __syncthreads();                        // every wave is done reading f
        to_lds(u, X0);
      }
      __syncthreads();
      // one wave per row, lanes stride the channels: the fma chain and the reduction order of narrow_linear_kernel
      for (int r = wv; r < XS; r += NW) {
        const int row = row0 + r;
        float s3[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          float s = 0.f;
          for (int c = lane; c < C; c += 64) s = fmaf(X0[c * XSP + r], A.wreg2[n * C + c], s);
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
          s3[n] = s;
        }
        if (lane < 3 && row < A.M) {
          const float s = lane == 0 ? s3[0] : (lane == 1 ? s3[1] : s3[2]);
          A.xyz_out[(size_t)row * 3 + lane] = A.xyz_in[(size_t)row * 3 + lane] + (s + A.breg2[lane]);
        }
      }
      continue;
    }
    // ---- kind D2 (X0 = f)
    // ---- feed forward: o = sum_s gelu(f Wint_s^T + b_s) Wout[:, sC:(s+1)C]^T, slab by slab in k order
    f32x16 o[TPW][P];
    const __amdgpu_buffer_rsrc_t wors = frag_rsrc(A.wout, 4u * CC4);
#pragma unroll 1
    for (int sl = 0; sl < 4; ++sl) {
      f32x16 t[TPW][P];
      lds_gemm<KCH, XSP, P, TPW, true, TPW * P == 1>(f4rs, __builtin_amdgcn_readfirstlane(((1 + sl) * NTILE + tile0) * KCH * 1024), KCH * 1024, X0, t, lane);
      add_bias(t, A.bf4 + (1 + sl) * C, 2);
      __syncthreads();                        // readers of X1 (the previous slab's contraction)
      to_lds(t, X1);
      __syncthreads();
      // tile t of the (C x 4C) image spans 4 KCH chunks: slab sl starts sl * KCH chunks in
      const int wb = __builtin_amdgcn_readfirstlane(tile0 * 4 * KCH * 1024 + sl * KCH * 1024);
      if (sl == 0) lds_gemm<KCH, XSP, P, TPW, true, TPW * P == 1>(wors, wb, 4 * KCH * 1024, X1, o, lane);
      else lds_gemm<KCH, XSP, P, TPW, false, TPW * P == 1>(wors, wb, 4 * KCH * 1024, X1, o, lane);
    }
    add_bias(o, A.bout, 0);
    // residual f from X0 (lane = row, registers = channels: conflict-free reads)
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[tp][p][i] += X0[((tile0 + tp) * 32 + mfma_row(i, h)) * XSP + 32 * p + j];
    layer_norm(o, A.ln2_g, A.ln2_b, A.eps);
    to_global(o, A.y3, A.ldy3, 0, row0);
    if (A.n2 > 0) {
      __syncthreads();                        // every wave has read its residual from X0
      to_lds(o, X0);
      __syncthreads();
      wide_linear(A.w2, A.b2, A.n2, X0, A.y2, A.ldy2, row0);
    }
  }
}

template <int C, int P, int NW, int KIND>
static hipError_t launch_chain_k(const ChainArgs& a, int cus, hipStream_t s) {
  constexpr int XS = 32 * P, XSP = XS + 1;
  const size_t lds = ((size_t)(KIND == 3 ? 2 : 1) * C * XSP + (size_t)NW * XS) * sizeof(float);
  auto kern = chain_kernel<C, P, NW, KIND>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), lds, optin); e != hipSuccess) return e;
  const int items = (a.M + XS - 1) / XS;
  const int per_cu = std::max(1, std::min(32 / NW, (int)(160 * 1024 / lds)));
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min(items, cus * per_cu)), dim3(NW * 64), lds, s, a);
  return hipGetLastError();
}

template <int C, int P, int NW>
static hipError_t launch_chain_t(const ChainArgs& a, int cus, hipStream_t s) {
  switch (a.kind) {
    case 0: return launch_chain_k<C, P, NW, 0>(a, cus, s);
    case 1: return launch_chain_k<C, P, NW, 1>(a, cus, s);
    case 2: return launch_chain_k<C, P, NW, 2>(a, cus, s);
    case 3: return launch_chain_k<C, P, NW, 3>(a, cus, s);
    default: return hipErrorInvalidValue;
  }
}

// Widths the chain kernel serves (two activation tiles of kind D must fit the CU's 160 KB of LDS); the caller keeps the
// launch-per-operator path for the others (C = 32, 64: tiny test shapes; C = 1024).
extern "C" int poem_chain_supported(int C) { return C == 128 || C == 256 || C == 512; }
// kind A can take the cross attention's split-key partials as its input (ChainArgs::part_o) for 4 heads and <= 4 key chunks
extern "C" int poem_chain_combines(int C, int heads, int chunks) { return poem_chain_supported(C) && heads == 4 && chunks >= 1 && chunks <= 4; }

// Row-tile height.  Every tile height runs the same arithmetic per row (a row's fma chains, its LayerNorm sums and its
// residuals never see the other rows of the tile), so the choice is free per launch -- results stay bit-identical in any
// batch.  A CU's co-resident blocks share its matrix pipe, so a launch takes as long as the CU with the most rows:
// ceil(tiles / CUs) * rows per tile.  64-row tiles stream every weight fragment once per 64 rows (best when M is large);
// 32-row tiles put a small batch on twice as many CUs (M = 1598 at the reference's evaluation batch of 2: 50 CUs instead
// of 25) and quantise better for 16 < B < 32 (B = 24: 3 x 32 rows per CU instead of 2 x 64).
static int chain_tile_p(int M, int cus, int force) {
  if (force == 1 || force == 2) return force;
  const long r2 = ((long)(M + 63) / 64 + cus - 1) / cus * 64, r1 = ((long)(M + 31) / 32 + cus - 1) / cus * 32;
  return r1 < r2 ? 1 : 2;
}

extern "C" hipError_t poem_launch_chain16(const ChainArgs* a, int C, hipStream_t s);      // chain16.hip

// Which kernel takes a launch (tile_p = 0; measured per kind and batch, tools/run/gpu_chain_sweep.sh):
//   * kinds A / C / D1 (two co-resident blocks per CU): from 3 units of 16 rows per CU on, the 16x16x4 kernel with its
//     16-row granularity and a CU's share split over two co-resident blocks (chain16.hip); below that a tile would be 1-2
//     units, where every MFMA needs a fresh weight fragment from L2 -- the 32-row tiles of this file spread a small batch
//     almost as widely and stream half the weight bytes per row; round 4: ONE unit per CU (B <= 5 at 799 queries) goes to the
//     16x16x4 kernel again -- with its ring of eight weight fragments a one-unit tile is the shortest latency chain
//     (B = 1 / 2 / 5: -2.7 / -0.6 / -0.6 % of the forward; two units per CU: the 32-row tiles still win);
//   * kind D2 (two activation tiles: one block per CU at 64 rows): chain16 while a CU's share is one tile of <= 3 units,
//     the 32- / 64-row kernel with chain_tile_p's height above (its second tile of a CU starts without a new block).
// Any choice gives the same bits.
extern "C" hipError_t poem_launch_chain(const ChainArgs* a, int C, hipStream_t s) {
  const int cus = poem_device_cus();
  if (a->tile_p == 3) return poem_launch_chain16(a, C, s);
  if (a->tile_p == 0) {
    const int U = (a->M + 15) / 16, per_cu = (U + std::min(cus, U) - 1) / std::min(cus, U);
    if (a->kind == 3 ? per_cu <= 3 : (per_cu >= 3 || per_cu == 1)) return poem_launch_chain16(a, C, s);
  }
  const int p = chain_tile_p(a->M, cus, a->tile_p);
  switch (C) {
    case 128: return p == 1 ? launch_chain_t<128, 1, 4>(*a, cus, s) : launch_chain_t<128, 2, 4>(*a, cus, s);
    case 256: return p == 1 ? launch_chain_t<256, 1, 8>(*a, cus, s) : launch_chain_t<256, 2, 8>(*a, cus, s);
    case 512: return launch_chain_t<512, 1, 8>(*a, cus, s);
    default: return hipErrorInvalidValue;
  }
}

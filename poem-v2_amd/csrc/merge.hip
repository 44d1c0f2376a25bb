// Fused sampling front end (ptEmb_head.py:900-926 + merge_features_mv / _sv :745-771 upstream): bilinear sampling of the
// view feature planes, the Q1 re-interpretation and the whole merge MLP in two kernels, so that the sampled tensor
// g (sum N, C, S) -- 1.07 GB per 32-sample step at C = 256, written once and read once by the unfused sequence -- and the
// hidden layer of merge_net[0] never exist in HBM.
//
// Q1 (ptEmb_head.py:914-915): a sample's (N, C, S) block is *viewed* as (S, N, C), so Q1 row r of the sample holds the
// C consecutive floats f = r*C .. r*C + C-1 of the block: view n' = f / (C*S), channel c' = (f / S) % C, points
// s' = f % S.  With S % C == 0 a row is C consecutive basis points ("segment" seg = r % (S/C)) of ONE (view, channel)
// plane (plane index r / (S/C)), and merge_net[0]'s first Linear contracts over those C points.  Globally (all views of
// the batch back to back) row R = v*S + c'*(S/C) + seg, independent of the sample the view belongs to.
//
// sample_merge_kernel: a block owns the XS (64 or 32) rows {channels c0 .. c0+XS-1} x one (view, segment) and builds their
// activations X[k = point][row = channel] in LDS by sampling: the projection of a point is shared by all channels, so
// the per-point bilinear weights / tap offsets come from a table the projection kernel wrote (32 B per point), and with
// the feature planes stored channel-LAST a tap of 4 channels is one 16-byte load, 16 lanes covering the 64 channels of a
// point contiguously (256 B).  Then, exactly like the chain kernels (chain.hip): Linear(C, C) + ReLU with the packed
// weights streamed from L2, result back into LDS, Linear(C, C/2), h2 tile to HBM.  The rows with n == 0 (r % N == 0)
// are also written out (q1: the residual term), 1/N of the sampled values.
//   h2 is written tile-major: [(view * tiles_per_view + tile) * XS + row-in-tile][C/2] -- a block's 64 rows are one
// contiguous 32 KB instead of 64 rows 8 KB apart.
//
// merge_tail_kernel: a block owns 64 basis points of one sample: dot / weighted sum over the views (merge_features_mv;
// N = 1: the master row itself), merge_net[1] as two LDS-resident GEMMs, / N, + q1 -> bps_feat.
//
// Arithmetic: every sampled value, every Linear output element and every reduction is the same fma chain as in the
// unfused kernels (sample.hip, gemm.hip) except the order of the dot-product reduction over channels (16-lane groups
// here, a full-wave tree there); the two paths agree to fp32 round-off (tests/test_hip_parity.py).
#include "common.h"
#include <algorithm>

#include "lds_gemm.h"
#include "merge.h"

namespace {

// bias (+ ReLU) on a wave's accumulators: register i of tile `tile` is channel tile*32 + mfma_row(i, h)
template <int TPW, int P>
__device__ __forceinline__ void bias_act(f32x16 (&acc)[TPW][P], const float* __restrict__ bias, int tile0, int h, bool relu) {
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
          if (relu) v = relu_nan(v);
          acc[tp][p][4 * g + e] = v;
        }
    }
  }
}

template <int TPW, int P, int XSP>
__device__ __forceinline__ void acc_to_lds(const f32x16 (&acc)[TPW][P], float* __restrict__ X, int tile0, int col0, int j, int h) {
#pragma unroll
  for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int i = 0; i < 16; ++i) X[((tile0 + tp) * 32 + mfma_row(i, h)) * XSP + col0 + 32 * p + j] = acc[tp][p][i];
}

}  // namespace

// The samples sample_group_kernel takes (N divides its 8-segment units) when the batch is large enough to fill the chip with
// its units; sample_merge_kernel / merge_tail_kernel skip them.  Both forms produce the same bits (same fma chains, same
// reduction trees), so the choice may depend on the batch.
__device__ __forceinline__ bool poem_group_n(int N) { return N == 1 || N == 2 || N == 4 || N == 8; }
// ... and whether the grouped kernel runs at all in this forward: ONE expression for the three kernels (the batch's own view
// count, read from device memory and clamped to the capacity the launches were sized for, against the threshold) -- if they
// disagreed, samples would be sampled twice or not at all.
__device__ __forceinline__ int poem_view_count(const int* views_dev, int views_cap) { return views_dev ? min(*views_dev, views_cap) : views_cap; }
__device__ __forceinline__ bool poem_grouped_forward(const int* views_dev, int views_cap, int group_min_views) {
  return views_dev && poem_view_count(views_dev, views_cap) >= group_min_views;
}

template <int C, int P, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void sample_merge_kernel(SampleMergeArgs A) {
  constexpr int XS = 32 * P, XSP = XS + 1, NTILE = C / 32, TPW = NTILE / NW, KCH = C / 8, HALF = C / 2;
  constexpr int LPP = XS / 4, PPW = 64 / LPP, PPI = NW * PPW, NIT = C / PPI;   // lanes per point, points per wave / block pass
  static_assert(NTILE % NW == 0 && (C / 64) * P == NW && C % PPI == 0 && NIT % 2 == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X0 = smem;                       // C * XSP
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, h = lane >> 5;
  const int NSEG = A.S / C, TPV = (C / XS) * NSEG;
  // Work order: plain round-robin over (view, tile) -- the resident blocks work on a handful of views at a time, whose planes
  // and tables (384 KB per view) stay in every XCD's L2.  (An XCD-aware order -- the blocks of XCD x = blockIdx % 8 walking
  // the views x, x + 8, ... -- measured 1 % slower.)
  const int slot = (int)blockIdx.x, L = (int)gridDim.x;
  const int nv = poem_view_count(A.views_dev, A.views);
  const unsigned CC4 = (unsigned)(C * C * 4);
  const int q = lane / LPP, cg = lane % LPP;

  // The per-point table entries of a tile (NIT points per thread, 32 B each) are requested one tile ahead, right before the
  // short second GEMM of the previous tile: the fill then starts with its tap loads instead of two dependent L2 trips.
  float4 tw[NIT];
  uint2 to[NIT];                          // tap pixels, 16 bits each: (nw | ne << 16, sw | se << 16)
  auto load_table = [&](int w) {
    const int v = w / TPV, t = w % TPV;
    const size_t p0 = (size_t)v * A.S + (size_t)(t % NSEG) * C;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int k = i * PPI + wv * PPW + q;
      tw[i] = A.tab[p0 + k];
      to[i] = A.tabo[p0 + k];
    }
  };
  const bool grouped = poem_grouped_forward(A.views_dev, A.views, A.group_min_views);      // sample_group_kernel takes the samples with N | 8
  if (grouped) {                          // ... all of them? (a fixed-view batch of 1, 2, 4 or 8 views: nothing to do here)
    const int B = A.B;
    int mine = 0;
    for (int b = tid; b < B; b += NW * 64) mine |= !poem_group_n(A.offs[b + 1] - A.offs[b]);
    if (!__syncthreads_or(mine)) return;
  }
  int pref = -1;                          // the tile whose table entries the registers hold
  if (slot < nv * TPV && !grouped) { load_table(slot); pref = slot; }

  for (int w = slot; w < nv * TPV; w += L) {
    const int v = w / TPV, t = w % TPV;
    const int c0 = (t / NSEG) * XS, seg = t % NSEG;
    if (grouped) {
      const int bb = A.view_sample[v];
      if (poem_group_n(A.offs[bb + 1] - A.offs[bb])) continue;
    }
    if (pref != w) load_table(w);
    __syncthreads();                          // the previous tile's readers of X0
    // ---- fill: X0[k][row] = bilinear sample of plane (v, c0 + row) at point seg*C + k
    {
      const __amdgpu_buffer_rsrc_t xrs = frag_rsrc(A.xt + (size_t)v * A.hw * C, (unsigned)(A.hw * C * 4));
      const int loff = (c0 + 4 * cg) * 4;
      // software pipeline, DEPTH point-iterations of tap loads in flight (16 registers each; the table entries free theirs as
      // they are consumed): the scheduling barriers keep hipcc from hoisting all 4 * NIT loads and spilling
#ifndef POEM_SM_DEPTH
#define POEM_SM_DEPTH 3
#endif
      constexpr int DEPTH = POEM_SM_DEPTH;
      constexpr unsigned PIXB = C * 4;        // bytes per pixel of the channel-last planes
      float4 tap[DEPTH][4];
#define SM_TAPS(I)                                                                               \
      {                                                                                          \
        tap[(I) % DEPTH][0] = frag_load(xrs, (int)((to[I].x & 0xffffu) * PIXB) + loff, 0);       \
        tap[(I) % DEPTH][1] = frag_load(xrs, (int)((to[I].x >> 16) * PIXB) + loff, 0);           \
        tap[(I) % DEPTH][2] = frag_load(xrs, (int)((to[I].y & 0xffffu) * PIXB) + loff, 0);       \
        tap[(I) % DEPTH][3] = frag_load(xrs, (int)((to[I].y >> 16) * PIXB) + loff, 0);           \
      }
#pragma unroll
      for (int i = 0; i < DEPTH - 1; ++i) SM_TAPS(i)
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        if (i + DEPTH - 1 < NIT) SM_TAPS(i + DEPTH - 1)
        __builtin_amdgcn_sched_barrier(0);
        const int k = i * PPI + wv * PPW + q;
        float* xo = X0 + k * XSP + 4 * cg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // the accumulation order of ATen's grid_sampler_2d (nw, ne, sw, se), as sample.hip's grid_sample_kernel
          float a = (&tap[i % DEPTH][0].x)[e] * tw[i].x;
          a = fmaf((&tap[i % DEPTH][1].x)[e], tw[i].y, a);
          a = fmaf((&tap[i % DEPTH][2].x)[e], tw[i].z, a);
          a = fmaf((&tap[i % DEPTH][3].x)[e], tw[i].w, a);
          xo[e] = a;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef SM_TAPS
    }
    __syncthreads();
    // ---- q1: the rows with n == 0 leave as they are (residual of the merge)
    {
      const int b = A.view_sample[v];
      const int off = A.offs[b], N = A.offs[b + 1] - off;
      for (int jj = wv; jj < XS; jj += NW) {
        const int r = (v - off) * A.S + (c0 + jj) * NSEG + seg;
        if (r % N) continue;
        float* dst = A.q1 + ((size_t)b * A.S + r / N) * C;
#pragma unroll
        for (int k = lane; k < C; k += 64) dst[k] = X0[k * XSP + jj];
      }
    }
    // ---- merge_net[0].0 + ReLU
    {
      f32x16 acc[TPW][P];
      lds_gemm<KCH, XSP, P, TPW, true>(frag_rsrc(A.w0, CC4), __builtin_amdgcn_readfirstlane(wv * TPW * KCH * 1024), KCH * 1024, X0, acc, lane);
      bias_act<TPW, P>(acc, A.b0, wv * TPW, h, true);
      __syncthreads();                        // every wave is done reading the sampled tile
      acc_to_lds<TPW, P, XSP>(acc, X0, wv * TPW, 0, j, h);
    }
    __syncthreads();
    pref = w + L < nv * TPV ? w + L : w;
    load_table(pref);                         // unconditional: a conditional definition would keep the consumed entries live through the GEMMs
    // ---- merge_net[0].2: C/64 output tiles x P column tiles = one (tile, p) per wave
    {
      const int t2 = wv / P, p2 = wv % P;
      f32x16 acc[1][1];
      lds_gemm<KCH, XSP, 1, 1, true>(frag_rsrc(A.w1, CC4 / 2), __builtin_amdgcn_readfirstlane(t2 * KCH * 1024), KCH * 1024, X0 + 32 * p2, acc, lane);
      bias_act<1, 1>(acc, A.b1, t2, h, false);
      const int jj = 32 * p2 + j;
      const size_t row = A.h2_tiled ? ((size_t)v * TPV + t) * XS + jj : (size_t)v * A.S + (size_t)(c0 + jj) * NSEG + seg;
      float* yp = A.h2 + row * HALF + t2 * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(yp + 8 * g) = make_float4(acc[0][0][4 * g], acc[0][0][4 * g + 1], acc[0][0][4 * g + 2], acc[0][0][4 * g + 3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// merge_net[1] on a tile of reduced rows, shared by merge_tail_kernel and sample_group_kernel: X0 holds m[k = channel][col]
// (HALF x 32 P); Linear(HALF, HALF) + ReLU, Linear(HALF, C), / N, + q1 -> out.  Column col is row row_of(col) of q1 / out.
template <int C, int P, int NW, class RowFn>
__device__ __forceinline__ void tail_mlp(float* __restrict__ X0, const float4* __restrict__ w0, const float* __restrict__ b0,
                                         const float4* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ q1,
                                         float* __restrict__ out, int N, RowFn row_of, int lane, int wv) {
  constexpr int HALF = C / 2, XS = 32 * P, XSP = XS + 1, NTA = HALF / 32, NTB = C / 32, KCH = HALF / 8;
  constexpr bool WIDE = NTA >= NW;        // first GEMM: whole column range per wave, or one (tile, p) per wave
  constexpr int TPWA = WIDE ? NTA / NW : 1, PA = WIDE ? P : 1;
  constexpr int TPWB = NTB / NW;
  static_assert(WIDE || NTA * P == NW, "shape");
  static_assert(NTB % NW == 0, "shape");
  const int j = lane & 31, h = lane >> 5;
  // ---- merge_net[1].0 + ReLU (HALF -> HALF)
  {
    const int tile0 = WIDE ? wv * TPWA : wv / P, col0 = WIDE ? 0 : 32 * (wv % P);
    f32x16 acc[TPWA][PA];
    lds_gemm<KCH, XSP, PA, TPWA, true>(frag_rsrc(w0, (unsigned)(HALF * HALF * 4)), __builtin_amdgcn_readfirstlane(tile0 * KCH * 1024), KCH * 1024,
                                       X0 + col0, acc, lane);
    bias_act<TPWA, PA>(acc, b0, tile0, h, true);
    __syncthreads();
    acc_to_lds<TPWA, PA, XSP>(acc, X0, tile0, col0, j, h);
  }
  __syncthreads();
  // ---- merge_net[1].2 (HALF -> C), / N, + q1 -> bps_feat
  {
    const int tile0 = wv * TPWB;
    f32x16 acc[TPWB][P];
    lds_gemm<KCH, XSP, P, TPWB, true>(frag_rsrc(w1, (unsigned)(C * HALF * 4)), __builtin_amdgcn_readfirstlane(tile0 * KCH * 1024), KCH * 1024, X0, acc, lane);
    bias_act<TPWB, P>(acc, b1, tile0, h, false);
    const float fn = (float)N;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const size_t row = row_of(32 * p + j);
      const float* qp = q1 + row * C + 4 * h;
      float* yp = out + row * C + 4 * h;
#pragma unroll
      for (int tp = 0; tp < TPWB; ++tp)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 qq = *reinterpret_cast<const float4*>(qp + (tile0 + tp) * 32 + 8 * g);
          float4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float yv = acc[tp][p][4 * g + e];
            (&o.x)[e] = (&qq.x)[e] + (N == 1 ? yv : yv / fn);      // merge_finalize_kernel's expression
          }
          *reinterpret_cast<float4*>(yp + (tile0 + tp) * 32 + 8 * g) = o;
        }
    }
  }
}

// Reduction tree of the cross-view dot product <h_n, h_0> over the HALF channels (both kernels): per 4-channel chunk q a
// 4-term fma chain from 0; chunks of a 32-channel tile t2 = q / 8 -- q % 8 = 2 g + hh -- as ((g0 + g1) + (g2 + g3)) per hh,
// then hh = 0 + hh = 1; tiles pairwise: (t0 + t1) + (t2 + t3) ...  In the grouped kernel a lane holds the four g chunks of
// one (t2, hh) in its accumulator registers; here 16 lanes hold chunks q = l16 + 16 i.
template <int C, int NW>
__global__ __launch_bounds__(NW * 64, 2) void merge_tail_kernel(MergeTailArgs A) {
  constexpr int HALF = C / 2, XS = 64, XSP = XS + 1;
  constexpr int F4 = HALF / 64;           // float4 chunks per lane of a 16-lane row group
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X0 = smem;                       // HALF * XSP
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rg = lane >> 4, l16 = lane & 15;
  const int items = (A.B * A.S) / XS;
  const int NSEG = A.S / C;
  // tiles of the fused kernel (XS rows there = 32 * P' with P' = 2 for C <= 256, 1 for C = 512)
  constexpr int FXS = C == 512 ? 32 : 64;
  const int TPV = (C / FXS) * NSEG;
  const bool grouped = poem_grouped_forward(A.views_dev, A.views, A.group_min_views);
  if (grouped) {
    int mine = 0;
    for (int b = tid; b < A.B; b += NW * 64) mine |= !poem_group_n(A.offs[b + 1] - A.offs[b]);
    if (!__syncthreads_or(mine)) return;
  }

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int i0 = item * XS;               // first global basis-point row b * S + s
    const int b = i0 / A.S, s0 = i0 % A.S;
    const int off = A.offs[b], N = A.offs[b + 1] - off;
    if (grouped && poem_group_n(N)) continue;
    __syncthreads();                        // the previous item's readers of X0
    // ---- reduce over the views: 16 lanes per row, rows jj = 4 * (pass * NW + wv) + rg
    for (int pass = 0; pass < XS / (4 * NW); ++pass) {
      const int jj = 4 * (pass * NW + wv) + rg;
      const size_t r0 = (size_t)(s0 + jj) * N;                 // first Q1 row of this point within the sample
      auto rowp = [&](size_t r) -> const float4* {
        size_t row;
        if (A.h2_tiled) {
          const size_t vv = off + r / A.S, rem = r % A.S;
          const size_t cp = rem / NSEG, sg = rem % NSEG;
          row = (vv * TPV + (cp / FXS) * NSEG + sg) * FXS + cp % FXS;
        } else {
          row = (size_t)off * A.S + r;
        }
        return reinterpret_cast<const float4*>(A.h2 + row * HALF);
      };
      float4 mast[F4], acc[F4];
      {
        const float4* mp = rowp(r0);
#pragma unroll
        for (int i = 0; i < F4; ++i) { mast[i] = mp[l16 + 16 * i]; acc[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
      }
      if (N == 1) {
#pragma unroll
        for (int i = 0; i < F4; ++i) acc[i] = mast[i];
      }
      // the other views, four rows' loads in flight (a row's dot product needs the whole row: one load at a time would
      // serialise N - 1 HBM round trips per point); the sum runs in view order
      constexpr int UN = 4;
      for (int n0 = 1; n0 < N; n0 += UN) {
        float4 v[UN][F4];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const float4* hp = rowp(r0 + min(n0 + u, N - 1));
#pragma unroll
          for (int i = 0; i < F4; ++i) v[u][i] = hp[l16 + 16 * i];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          if (n0 + u >= N) break;
          float d[F4];
#pragma unroll
          for (int i = 0; i < F4; ++i) {
            float t = fmaf(v[u][i].x, mast[i].x, 0.f);
            t = fmaf(v[u][i].y, mast[i].y, t);
            t = fmaf(v[u][i].z, mast[i].z, t);
            t = fmaf(v[u][i].w, mast[i].w, t);
            t += __shfl_xor(t, 2, 64);          // g bit 0
            t += __shfl_xor(t, 4, 64);          // g bit 1
            t += __shfl_xor(t, 1, 64);          // hh
            t += __shfl_xor(t, 8, 64);          // the tile pair (t2 = 2 i, 2 i + 1)
            d[i] = t;
          }
          float dot;
          if constexpr (F4 == 1) dot = d[0];
          else if constexpr (F4 == 2) dot = d[0] + d[1];
          else dot = (d[0] + d[1]) + (d[2] + d[3]);
#pragma unroll
          for (int i = 0; i < F4; ++i) {
            acc[i].x = fmaf(dot, v[u][i].x, acc[i].x);
            acc[i].y = fmaf(dot, v[u][i].y, acc[i].y);
            acc[i].z = fmaf(dot, v[u][i].z, acc[i].z);
            acc[i].w = fmaf(dot, v[u][i].w, acc[i].w);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < F4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) X0[(4 * (l16 + 16 * i) + e) * XSP + jj] = (&acc[i].x)[e];
    }
    __syncthreads();
    tail_mlp<C, 2, NW>(X0, A.w0, A.b0, A.w1, A.b1, A.q1, A.out, N, [&](int col) { return (size_t)i0 + col; }, lane, wv);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// sample_group_kernel: the WHOLE sampling stage in one kernel for the samples whose view count N divides 8 -- merge_net[0]'s
// hidden rows (the (S, N, C/2) block: 537 MB written and read back per 32-sample step at C = 256) never leave the chip.
//   With N | NSEG the N Q1 rows of a basis point s -- rows s N .. s N + N - 1 -- are N consecutive segments of ONE (view,
// channel) plane (header: row r of a view = c' * NSEG + seg), so a block that walks the segments of its XS channels in order
// meets the master row (n = 0) of every point first and then its other views: unit = (view, XS channels, 8 consecutive
// segments) = 8 tiles of sample_merge_kernel's shape; per tile fill -> Linear + ReLU -> Linear -> h (16 accumulator registers
// per lane), then in registers: n = 0: h0 = h, m = 0 (N = 1: m = h); n > 0: m += <h, h0> h (merge_features_mv's weights; the
// dot product's partials meet through 1 KB of LDS); at n = N - 1 the XS finished rows m go through merge_net[1] (tail_mlp)
// and leave as bps_feat rows.  Only the master rows q1 (the residual) make a round trip: written at n = 0, read at n = N - 1.
//   Same fma chains and reduction trees as sample_merge_kernel + merge_tail_kernel: bit-identical results.
template <int C, int P, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void sample_group_kernel(SampleGroupArgs G) {
  const SampleMergeArgs& A = G.sm;
  constexpr int XS = 32 * P, XSP = XS + 1, NTILE = C / 32, TPW = NTILE / NW, KCH = C / 8, HALF = C / 2, NT2 = HALF / 32;
  constexpr int LPP = XS / 4, PPW = 64 / LPP, PPI = NW * PPW, NIT = C / PPI;   // lanes per point, points per wave / block pass
  constexpr int USEG = 8;                 // segments per unit
  static_assert(NTILE % NW == 0 && (C / 64) * P == NW && C % PPI == 0 && NIT % 2 == 0 && NT2 * P == NW, "shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X0 = smem;                       // C * XSP
  float* red = smem + C * XSP;            // NT2 * XS: per-tile partial dot products
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, h = lane >> 5;
  const int NSEG = A.S / C, UPC = NSEG / USEG, UPV = (C / XS) * UPC;
  const int slot = (int)blockIdx.x, L = (int)gridDim.x;
  const int nv = poem_view_count(A.views_dev, A.views);
  if (!poem_grouped_forward(A.views_dev, A.views, A.group_min_views)) return;
  const unsigned CC4 = (unsigned)(C * C * 4);
  const int q = lane / LPP, cg = lane % LPP;
  const int t2 = wv / P, p2 = wv % P;     // this wave's (32-channel tile, column tile) of h

  // The table entries of a tile's C points (weights 16 B, tap pixels 8 B) are copied into LDS by LDS-DMA (buffer_load ... lds:
  // memory pipeline -> LDS, no registers) a tile ahead, right before the short second GEMM of the previous tile: the fill
  // starts with LDS reads instead of two dependent L2 trips, and nothing of the table is held in registers across the GEMMs.
  float4* TW = reinterpret_cast<float4*>(red + NT2 * XS);     // C weights
  uint2* TO = reinterpret_cast<uint2*>(TW + C);               // C tap-pixel pairs
  auto stage_table = [&](int v, int seg) {
#if defined(__HIP_DEVICE_COMPILE__)
    const size_t p0 = (size_t)v * A.S + (size_t)seg * C;
    const __amdgpu_buffer_rsrc_t wr = frag_rsrc(A.tab + p0, (unsigned)(C * 16));
    const __amdgpu_buffer_rsrc_t orr = frag_rsrc(A.tabo + p0, (unsigned)(C * 8));
    for (int r = wv; r < C / 64; r += NW)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(TW + 64 * r), 16, lane * 16, r * 1024, 0, 0);
    for (int r = wv; r < C / 128; r += NW)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(orr, (__attribute__((address_space(3))) void*)(TO + 128 * r), 16, lane * 16, r * 1024, 0, 0);
#endif
  };
  int pref_v = -1, pref_seg = -1;

  // Unit order.  The UPV units of a view share its feature planes (each 64-channel slice is read by the units of both segment
  // halves) and its table (every 8-segment part by the units of all channel tiles); workgroups go round-robin over the eight
  // XCDs, each with its own L2.  XCD-aware order (A.xcd_order, grid a multiple of 8): the blocks of XCD x = blockIdx % 8 walk the
  // views x, x + 8, ... -- a view's planes and table are fetched from HBM by ONE L2 (381 -> 23x MB read per 32-sample step at
  // C = 256); plain order: u = blockIdx, blockIdx + grid, ...
  const bool xo = A.xcd_order && (L & 7) == 0;
  const int ustep = xo ? (L >> 3) : L, xcd = slot & 7;
  const int ulimit = xo ? ((nv - xcd + 7) >> 3) * UPV : nv * UPV;     // views xcd, xcd + 8, ... < nv
  for (int lu = xo ? (slot >> 3) : slot; lu < ulimit; lu += ustep) {
    const int v = xo ? xcd + 8 * (lu / UPV) : lu / UPV, rest = lu % UPV;
    const int c0 = (rest / UPC) * XS, seg0 = (rest % UPC) * USEG;
    const int b = A.view_sample[v];
    const int off = A.offs[b], N = A.offs[b + 1] - off;
    if (!poem_group_n(N)) continue;
    const int rbase = (v - off) * A.S + c0 * NSEG;      // Q1 row of (channel c0, segment 0) within the sample (< 8 S)
    const size_t obase = (size_t)b * A.S;
    const int nsh = __ffs(N) - 1;             // N is 1, 2, 4 or 8: a row's basis point is r >> nsh
    f32x16 h0, m;
    for (int i = 0; i <= USEG; ++i) {
      const int seg = seg0 + i;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of the staged table has landed
      __syncthreads();                        // the previous tile's readers of X0; the staged table
      if (i > 0) {
        // ---- tile i - 1 closed its group: the finished rows m go through merge_net[1]
        const int np = (i - 1) & (N - 1);
        if (np == N - 1) {
          {
            const f32x16 (&mm)[1][1] = reinterpret_cast<const f32x16 (&)[1][1]>(m);
            acc_to_lds<1, 1, XSP>(mm, X0, t2, 32 * p2, j, h);
          }
          h0 = zero16(); m = zero16();        // dead from here (the next tile is a group's first): not live across merge_net[1]
          __syncthreads();
          const int segl = seg - 1;           // the group's last segment
          tail_mlp<C, P, NW>(X0, G.w2, G.b2, G.w3, G.b3, A.q1, G.out, N,
                             [&](int col) { return obase + (size_t)((rbase + col * NSEG + segl) >> nsh); }, lane, wv);
          __syncthreads();                    // its readers of X0, before the next fill
        }
        if (i == USEG) break;
      }
      const int n = i & (N - 1);
      if (pref_v != v || pref_seg != seg) {   // (first unit of the block / after skipped units) the table is not staged yet
        stage_table(v, seg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      // ---- fill: X0[k][row] = bilinear sample of plane (v, c0 + row) at point seg*C + k
      {
        const __amdgpu_buffer_rsrc_t xrs = frag_rsrc(A.xt + (size_t)v * A.hw * C, (unsigned)(A.hw * C * 4));
        const int loff = (c0 + 4 * cg) * 4;
        constexpr int DEPTH = POEM_SM_DEPTH;
        constexpr unsigned PIXB = C * 4;        // bytes per pixel of the channel-last planes
        float4 tap[DEPTH][4], tww[2];
#define SG_TAPS(I)                                                                               \
        {                                                                                        \
          const uint2 o_ = TO[(I) * PPI + wv * PPW + q];                                         \
          tap[(I) % DEPTH][0] = frag_load(xrs, (int)((o_.x & 0xffffu) * PIXB) + loff, 0);        \
          tap[(I) % DEPTH][1] = frag_load(xrs, (int)((o_.x >> 16) * PIXB) + loff, 0);            \
          tap[(I) % DEPTH][2] = frag_load(xrs, (int)((o_.y & 0xffffu) * PIXB) + loff, 0);        \
          tap[(I) % DEPTH][3] = frag_load(xrs, (int)((o_.y >> 16) * PIXB) + loff, 0);            \
        }
#pragma unroll
        for (int ii = 0; ii < DEPTH - 1; ++ii) SG_TAPS(ii)
        tww[0] = TW[wv * PPW + q];
#pragma unroll
        for (int ii = 0; ii < NIT; ++ii) {
          if (ii + DEPTH - 1 < NIT) SG_TAPS(ii + DEPTH - 1)
          if (ii + 1 < NIT) tww[(ii + 1) & 1] = TW[(ii + 1) * PPI + wv * PPW + q];
          __builtin_amdgcn_sched_barrier(0);
          const int k = ii * PPI + wv * PPW + q;
          float* xo = X0 + k * XSP + 4 * cg;
          const float4 ww = tww[ii & 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // the accumulation order of ATen's grid_sampler_2d (nw, ne, sw, se), as sample.hip's grid_sample_kernel
            float a = (&tap[ii % DEPTH][0].x)[e] * ww.x;
            a = fmaf((&tap[ii % DEPTH][1].x)[e], ww.y, a);
            a = fmaf((&tap[ii % DEPTH][2].x)[e], ww.z, a);
            a = fmaf((&tap[ii % DEPTH][3].x)[e], ww.w, a);
            xo[e] = a;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#undef SG_TAPS
      }
      __syncthreads();
      // ---- q1: a group's first tile is its master rows (the residual of the merge)
      if (n == 0) {
        for (int jj = wv; jj < XS; jj += NW) {
          float* dst = A.q1 + (obase + (size_t)((rbase + jj * NSEG + seg) >> nsh)) * C;
#pragma unroll
          for (int k = lane; k < C; k += 64) dst[k] = X0[k * XSP + jj];
        }
      }
      // ---- merge_net[0].0 + ReLU
      {
        f32x16 acc[TPW][P];
        lds_gemm<KCH, XSP, P, TPW, true>(frag_rsrc(A.w0, CC4), __builtin_amdgcn_readfirstlane(wv * TPW * KCH * 1024), KCH * 1024, X0, acc, lane);
        bias_act<TPW, P>(acc, A.b0, wv * TPW, h, true);
        __syncthreads();                        // every wave is done reading the sampled tile
        acc_to_lds<TPW, P, XSP>(acc, X0, wv * TPW, 0, j, h);
      }
      __syncthreads();
      // the next tile's tap pixels: this unit's next segment, or the first tile of this block's next unit
      if (i + 1 < USEG) { pref_v = v; pref_seg = seg + 1; }
      else {
        const int un = lu + ustep < ulimit ? lu + ustep : lu;
        pref_v = xo ? xcd + 8 * (un / UPV) : un / UPV; pref_seg = ((un % UPV) % UPC) * USEG;
      }
      stage_table(pref_v, pref_seg);           // (every wave is past the fill: the staged table has no readers left)
      // ---- merge_net[0].2: one (32-channel tile, column tile) of h per wave
      {
        f32x16 acc[1][1];
        lds_gemm<KCH, XSP, 1, 1, true>(frag_rsrc(A.w1, CC4 / 2), __builtin_amdgcn_readfirstlane(t2 * KCH * 1024), KCH * 1024, X0 + 32 * p2, acc, lane);
        bias_act<1, 1>(acc, A.b1, t2, h, false);
        const f32x16& hn = acc[0][0];
        if (n == 0) {
          h0 = hn;
#pragma unroll
          for (int e = 0; e < 16; ++e) m[e] = N == 1 ? hn[e] : 0.f;
        } else {
          // <h, h0>: this lane's 16 channels as chunks g (registers 4 g .. 4 g + 3), ((g0 + g1) + (g2 + g3)), the two halves,
          // then the NT2 channel tiles through LDS -- merge_tail_kernel's tree; m += dot * h
          float pg[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float d = fmaf(hn[4 * g], h0[4 * g], 0.f);
            d = fmaf(hn[4 * g + 1], h0[4 * g + 1], d);
            d = fmaf(hn[4 * g + 2], h0[4 * g + 2], d);
            d = fmaf(hn[4 * g + 3], h0[4 * g + 3], d);
            pg[g] = d;
          }
          const float sd = half_sum((pg[0] + pg[1]) + (pg[2] + pg[3]));
          float dot = sd;
          if constexpr (NT2 > 1) {
            if (h == 0) red[t2 * XS + 32 * p2 + j] = sd;
            __syncthreads();
            const float* rp = red + 32 * p2 + j;
            if constexpr (NT2 == 2) dot = rp[0] + rp[XS];
            else if constexpr (NT2 == 4) dot = (rp[0] + rp[XS]) + (rp[2 * XS] + rp[3 * XS]);
            else dot = ((rp[0] + rp[XS]) + (rp[2 * XS] + rp[3 * XS])) + ((rp[4 * XS] + rp[5 * XS]) + (rp[6 * XS] + rp[7 * XS]));
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) m[e] = fmaf(dot, hn[e], m[e]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Projection of the basis points into every view (project_kernel of sample.hip, same fp32 chain) down to the bilinear
// weights and tap offsets of grid_sample(align_corners=False, zero padding): 24 bytes per (view, point): four weights | four tap pixels of 16 bits, in two arrays.
__global__ void project_table_kernel(const float* __restrict__ bps, const float* __restrict__ centre,
                                     const int* __restrict__ view_sample, const float* __restrict__ intr,
                                     const float* __restrict__ inv_extr, float4* __restrict__ tabw, uint2* __restrict__ tabo, float* __restrict__ uv,
                                     int views, int S, int fw, int fh, float inv_w, float inv_h, int C) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  if (s >= S) return;
  const float* c = centre + (size_t)view_sample[v] * 3;
  const float px = bps[s * 3 + 0] + c[0], py = bps[s * 3 + 1] + c[1], pz = bps[s * 3 + 2] + c[2];
  const float* T = inv_extr + (size_t)v * 16;
  const float cx = fmaf(T[2], pz, fmaf(T[1], py, T[0] * px)) + T[3];
  const float cy = fmaf(T[6], pz, fmaf(T[5], py, T[4] * px)) + T[7];
  const float cz = fmaf(T[10], pz, fmaf(T[9], py, T[8] * px)) + T[11];
  const float* K = intr + (size_t)v * 9;
  const float qx = fmaf(K[2], cz, fmaf(K[1], cy, K[0] * cx));
  const float qy = fmaf(K[5], cz, fmaf(K[4], cy, K[3] * cx));
  float qz = fmaf(K[8], cz, fmaf(K[7], cy, K[6] * cx));
  if (fabsf(qz) < 1e-7f) qz = 1e-7f;
  const float u = qx / qz, w = qy / qz;
  const float gx = u * inv_w * 2.0f - 1.0f, gy = w * inv_h * 2.0f - 1.0f;
  const float ix = ((gx + 1.0f) * (float)fw - 1.0f) / 2.0f;
  const float iy = ((gy + 1.0f) * (float)fh - 1.0f) / 2.0f;
  if (uv) reinterpret_cast<float2*>(uv)[(size_t)v * S + s] = make_float2(ix, iy);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - fx0, wy1 = iy - fy0;
  const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
  const bool vx0 = x0 >= 0 && x0 < fw, vx1 = x1 >= 0 && x1 < fw, vy0 = y0 >= 0 && y0 < fh, vy1 = y1 >= 0 && y1 < fh;
  const float w_nw = (vx0 && vy0) ? wx0 * wy0 : 0.f, w_ne = (vx1 && vy0) ? wx1 * wy0 : 0.f;
  const float w_sw = (vx0 && vy1) ? wx0 * wy1 : 0.f, w_se = (vx1 && vy1) ? wx1 * wy1 : 0.f;
  const int cx0 = min(max(x0, 0), fw - 1), cx1 = min(max(x1, 0), fw - 1);
  const int cy0 = min(max(y0, 0), fh - 1), cy1 = min(max(y1, 0), fh - 1);
  tabw[(size_t)v * S + s] = make_float4(w_nw, w_ne, w_sw, w_se);
  tabo[(size_t)v * S + s] = make_uint2((unsigned)(cy0 * fw + cx0) | ((unsigned)(cy0 * fw + cx1) << 16),
                                       (unsigned)(cy1 * fw + cx0) | ((unsigned)(cy1 * fw + cx1) << 16));
}

// The three small input kernels of a forward in ONE launch (round 4: each is a 5-7 us link of a small batch's latency chain):
// rows blockIdx.y < views are project_table_kernel with the view's inverse extrinsic computed by the block itself (invert4x4, the
// same fp64 elimination) and the sample's centre read from reference_joints; the rows behind them are prep_xyz_kernel's elements.
__global__ void input_tables_kernel(const float* __restrict__ bps, const float* __restrict__ ref_joints, const float* __restrict__ tmpl,
                                    const int* __restrict__ view_sample, const float* __restrict__ intr,
                                    const float* __restrict__ extr, float4* __restrict__ tabw, uint2* __restrict__ tabo, float* __restrict__ centre,
                                    float* __restrict__ pt_xyz, float* __restrict__ query_xyz, int views, int B, int S, int Q, int fw,
                                    int fh, float inv_w, float inv_h, float radius) {
  if ((int)blockIdx.y >= views) {
    const long i = ((long)(blockIdx.y - views) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    poem_prep_xyz_elem(i, ref_joints, bps, tmpl, centre, pt_xyz, query_xyz, B, S, Q, radius);
    return;
  }
  __shared__ float Tinv[16];
  __shared__ float ctr[3];
  if (threadIdx.x == 0) invert4x4(extr + (size_t)blockIdx.y * 16, Tinv);
  if (threadIdx.x >= 64 && threadIdx.x < 67) ctr[threadIdx.x - 64] = ref_joints[((size_t)view_sample[blockIdx.y] * 21 + 9) * 3 + (threadIdx.x - 64)];
  __syncthreads();
  float* const uv = nullptr;
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  if (s >= S) return;
  const float* c = ctr;
  const float px = bps[s * 3 + 0] + c[0], py = bps[s * 3 + 1] + c[1], pz = bps[s * 3 + 2] + c[2];
  const float* T = Tinv;
  const float cx = fmaf(T[2], pz, fmaf(T[1], py, T[0] * px)) + T[3];
  const float cy = fmaf(T[6], pz, fmaf(T[5], py, T[4] * px)) + T[7];
  const float cz = fmaf(T[10], pz, fmaf(T[9], py, T[8] * px)) + T[11];
  const float* K = intr + (size_t)v * 9;
  const float qx = fmaf(K[2], cz, fmaf(K[1], cy, K[0] * cx));
  const float qy = fmaf(K[5], cz, fmaf(K[4], cy, K[3] * cx));
  float qz = fmaf(K[8], cz, fmaf(K[7], cy, K[6] * cx));
  if (fabsf(qz) < 1e-7f) qz = 1e-7f;
  const float u = qx / qz, w = qy / qz;
  const float gx = u * inv_w * 2.0f - 1.0f, gy = w * inv_h * 2.0f - 1.0f;
  const float ix = ((gx + 1.0f) * (float)fw - 1.0f) / 2.0f;
  const float iy = ((gy + 1.0f) * (float)fh - 1.0f) / 2.0f;
  if (uv) reinterpret_cast<float2*>(uv)[(size_t)v * S + s] = make_float2(ix, iy);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - fx0, wy1 = iy - fy0;
  const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
  const bool vx0 = x0 >= 0 && x0 < fw, vx1 = x1 >= 0 && x1 < fw, vy0 = y0 >= 0 && y0 < fh, vy1 = y1 >= 0 && y1 < fh;
  const float w_nw = (vx0 && vy0) ? wx0 * wy0 : 0.f, w_ne = (vx1 && vy0) ? wx1 * wy0 : 0.f;
  const float w_sw = (vx0 && vy1) ? wx0 * wy1 : 0.f, w_se = (vx1 && vy1) ? wx1 * wy1 : 0.f;
  const int cx0 = min(max(x0, 0), fw - 1), cx1 = min(max(x1, 0), fw - 1);
  const int cy0 = min(max(y0, 0), fh - 1), cy1 = min(max(y1, 0), fh - 1);
  tabw[(size_t)v * S + s] = make_float4(w_nw, w_ne, w_sw, w_se);
  tabo[(size_t)v * S + s] = make_uint2((unsigned)(cy0 * fw + cx0) | ((unsigned)(cy0 * fw + cx1) << 16),
                                       (unsigned)(cy1 * fw + cx0) | ((unsigned)(cy1 * fw + cx1) << 16));
}


static int cu_count() { return poem_device_cus(); }

template <int C, int P, int NW>
static hipError_t launch_sample_merge_t(const SampleMergeArgs& a, hipStream_t s) {
  constexpr int XS = 32 * P, XSP = XS + 1;
  const size_t lds = (size_t)C * XSP * sizeof(float);
  auto kern = sample_merge_kernel<C, P, NW>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), lds, optin); e != hipSuccess) return e;
  const int tpv = (C / XS) * (a.S / C);
  const long items = (long)a.views * tpv;
  const long grid = std::min<long>(items, (long)cu_count() * 2);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, s, a);
  return hipGetLastError();
}

template <int C, int P, int NW>
static hipError_t launch_sample_group_t(const SampleGroupArgs& a, hipStream_t s) {
  constexpr int XS = 32 * P, XSP = XS + 1;
  const size_t lds = ((size_t)C * XSP + (size_t)(C / 64) * XS + (size_t)C * 6) * sizeof(float);   // X tile | dot partials | staged table
  auto kern = sample_group_kernel<C, P, NW>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), lds, optin); e != hipSuccess) return e;
  const long units = (long)a.sm.views * (C / XS) * (a.sm.S / C / 8);
  const long grid = std::min<long>(units, (long)cu_count() * 2);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, s, a);
  return hipGetLastError();
}

template <int C, int NW>
static hipError_t launch_merge_tail_t(const MergeTailArgs& a, hipStream_t s) {
  const size_t lds = (size_t)(C / 2) * 65 * sizeof(float);
  auto kern = merge_tail_kernel<C, NW>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), lds, optin); e != hipSuccess) return e;
  const long items = (long)a.B * a.S / 64;
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long>(items, (long)cu_count() * 2)), dim3(NW * 64), lds, s, a);
  return hipGetLastError();
}

// Shapes the fused front end serves: the chain widths, segments that tile the point axis, 64-point tail tiles.
extern "C" int poem_sample_merge_supported(int C, int S, int hw) {
  return (C == 128 || C == 256 || C == 512) && S % C == 0 && S % 64 == 0 && hw <= 65536 && (long)hw * C * 4 < (1l << 31);
}

extern "C" hipError_t poem_launch_project_table(const float* bps, const float* centre, const int* view_sample, const float* intr,
                                                const float* inv_extr, void* tabw, void* tabo, float* uv, int views, int C, int fh, int fw,
                                                int S, int img_w, int img_h, hipStream_t s) {
  hipLaunchKernelGGL(project_table_kernel, dim3((S + 255) / 256, views), dim3(256), 0, s, bps, centre, view_sample, intr, inv_extr,
                     (float4*)tabw, (uint2*)tabo, uv, views, S, fw, fh, 1.0f / (float)img_w, 1.0f / (float)img_h, C);
  return hipGetLastError();
}

extern "C" hipError_t poem_launch_input_tables(const float* bps, const float* ref_joints, const float* tmpl, const int* view_sample,
                                               const float* intr, const float* extr, void* tabw, void* tabo, float* centre, float* pt_xyz,
                                               float* query_xyz, int views, int B, int S, int Q, int fh, int fw, int img_w, int img_h,
                                               float radius, hipStream_t s) {
  const unsigned gx = (unsigned)((S + 255) / 256);
  const long total = (long)B * S * 3 + (long)B * Q * 3 + 3L * B;
  const unsigned prep_rows = (unsigned)((total + (long)gx * 256 - 1) / ((long)gx * 256));
  hipLaunchKernelGGL(input_tables_kernel, dim3(gx, (unsigned)views + prep_rows), dim3(256), 0, s, bps, ref_joints, tmpl, view_sample, intr,
                     extr, (float4*)tabw, (uint2*)tabo, centre, pt_xyz, query_xyz, views, B, S, Q, fw, fh, 1.0f / (float)img_w, 1.0f / (float)img_h, radius);
  return hipGetLastError();
}

extern "C" hipError_t poem_launch_sample_merge(const SampleMergeArgs* a, int C, hipStream_t s) {
  switch (C) {
    case 128: return launch_sample_merge_t<128, 2, 4>(*a, s);
    case 256: return launch_sample_merge_t<256, 2, 8>(*a, s);
    case 512: return launch_sample_merge_t<512, 1, 8>(*a, s);
    default: return hipErrorInvalidValue;
  }
}

extern "C" hipError_t poem_launch_sample_group(const SampleGroupArgs* a, int C, hipStream_t s) {
  if ((a->sm.S / C) % 8) return hipErrorInvalidValue;
  switch (C) {
    case 128: return launch_sample_group_t<128, 2, 4>(*a, s);
    case 256: return launch_sample_group_t<256, 2, 8>(*a, s);
    case 512: return launch_sample_group_t<512, 1, 8>(*a, s);
    default: return hipErrorInvalidValue;
  }
}

extern "C" hipError_t poem_launch_merge_tail(const MergeTailArgs* a, int C, hipStream_t s) {
  switch (C) {
    case 128: return launch_merge_tail_t<128, 4>(*a, s);
    case 256: return launch_merge_tail_t<256, 8>(*a, s);
    case 512: return launch_merge_tail_t<512, 8>(*a, s);
    default: return hipErrorInvalidValue;
  }
}

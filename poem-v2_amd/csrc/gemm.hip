// fp32 MFMA GEMM (Y = act(X W^T + b) + R) on packed weights, LayerNorm, small dense helpers.  gfx950 only.
// Two kernels: gemm_panel_kernel (the workhorse: W panel resident in LDS, persistent blocks streaming row groups,
// per-segment output layouts) and gemm2_kernel (operands straight from global/L2, for the shapes the panel kernel does
// not take; also the packed-activation layouts).  In both, operands of the next K-chunk are prefetched into registers
// while the current chunk's MFMAs issue.
#include "common.h"
#include <algorithm>
#include <cstdlib>

// ---- opt-in split precision (POEM_PRECISION_SPLIT_F16X3_ALL; see vecattn_split.hip for the scheme) -----------------------
// A Linear's weight as hi | lo f16 fragments for v_mfma_f32_32x32x16_f16, pre-multiplied PER 32-ROW TILE by a power of two
// (max |w'| of the tile in [8,16)):  image[((nt * K/16 + kc) * 2 + part) * 64 + lane] = half8(W'[32 nt + (lane & 31)]
// [16 kc + 8 (lane >> 5) + 0..7]).  A tile is K * 128 bytes -- exactly the size of the fp32 fragment image's tile, so a
// split arena mirrors the fp32 packed arena byte for byte and any (base + tiles) pointer maps by adding one offset.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#ifndef POEM_GS_WAVES
#define POEM_GS_WAVES 12           // waves per block of the split panel kernel (activation-free / ReLU instantiation)
#endif
#define POEM_GEMM_SX 16.0f            // activation pre-scale (power of two); scaled activations saturate at +-60000
__global__ __launch_bounds__(256) void pack_split_tiles_kernel(const float* __restrict__ W, int N, int K,
                                                              h8* __restrict__ img, float* __restrict__ scales,
                                                              int scale_stride) {
  __shared__ float red[256];
  const int nt = blockIdx.x, KC = K / 16;
  float m = 0.f;
  for (int i = threadIdx.x; i < 32 * K; i += 256) {
    const int row = 32 * nt + i / K;
    if (row < N) m = fmaxf(m, fabsf(W[(size_t)row * K + i % K]));
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  m = red[0];
  int e = 0;
  if (m > 0.f) (void)frexpf(m, &e);
  const float sc = m > 0.f ? ldexpf(1.0f, 4 - e) : 1.0f;
  if (threadIdx.x == 0) scales[(size_t)nt * scale_stride] = sc;
  for (int idx = threadIdx.x; idx < KC * 64; idx += 256) {
    const int lane = idx & 63, kc = idx >> 6, row = 32 * nt + (lane & 31);
    h8 hi, lo;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float v = row < N ? W[(size_t)row * K + 16 * kc + 8 * (lane >> 5) + t] * sc : 0.f;
      hi[t] = (_Float16)v;
      lo[t] = (_Float16)(v - (float)hi[t]);
    }
    img[((size_t)(nt * KC + kc) * 2 + 0) * 64 + lane] = hi;
    img[((size_t)(nt * KC + kc) * 2 + 1) * 64 + lane] = lo;
  }
}

// scales[nt * scale_stride]: one float per 32-row tile (scale_stride in floats; K/2 when the scales mirror a packed arena
// at one float per 256 bytes of image)
extern "C" hipError_t poem_launch_pack_split_tiles(const float* w, int N, int K, void* img, float* scales, int scale_stride,
                                                   hipStream_t s) {
  if (K % 16) return hipErrorInvalidValue;
  pack_split_tiles_kernel<<<(N + 31) / 32, 256, 0, s>>>(w, N, K, (h8*)img, scales, scale_stride);
  return hipGetLastError();
}

// two fp32 fragments -> the hi | lo f16 operands of one 16-slot chunk (x16, saturating): the K / V images of the split mode
__device__ __forceinline__ void split_pair(const float4 a, const float4 b, h8& hi, h8& lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float x = __builtin_amdgcn_fmed3f(v[t] * POEM_GEMM_SX, -60000.f, 60000.f);
    hi[t] = (_Float16)x;
    lo[t] = (_Float16)(x - (float)hi[t]);
  }
}

// The split arena of the handle whose forward is being enqueued (api.cpp sets / clears it around poem_head_forward):
// fp32 image pointers inside [packed, packed + bytes) are redirected to the split image at the same offset; the tile
// scales sit at one float per 256 bytes of image.  Per-THREAD host state: a forward is enqueued by one host thread from start
// to end, so the context a call installs is seen by exactly the launches that call makes -- forwards of other handles on
// other host threads (one thread per GPU, say) neither see nor clear it (include/poem_hip.h, threading contract).
static thread_local struct { const char* packed; size_t bytes; const char* split; const float* scales; } g_split_ctx = {nullptr, 0, nullptr, nullptr};
static thread_local struct { const void* img; const float* scales; } g_explicit_split = {nullptr, nullptr};
extern "C" void poem_gemm_split_explicit(const void* img, const float* scales) { g_explicit_split = {img, scales}; }
extern "C" void poem_gemm_split_context(const void* packed, size_t bytes, const void* split, const float* scales) {
  g_split_ctx = {(const char*)packed, bytes, (const char*)split, scales};
}

// Will a panel GEMM with this weight pointer run the split variant (and therefore write SPLIT K / V images)?  Same rule as
// the dispatch in launch_gemm_split_impl; api.cpp tells the cross attention which image format it gets.
static thread_local int g_split_images = 0;
extern "C" void poem_gemm_split_images(int on) { g_split_images = on; }
extern "C" int poem_gemm_split_applies(const void* Wp, int M, int ldx, int K) {
  const bool in_arena = g_split_ctx.packed && (const char*)Wp >= g_split_ctx.packed &&
                        (const char*)Wp < g_split_ctx.packed + g_split_ctx.bytes;
  return g_split_images && (g_explicit_split.img || in_arena) && K % 16 == 0 &&
         (unsigned long long)M * ldx * 4ull + 64ull < (1ull << 32);
}

__global__ void pack_linear_kernel(const float* __restrict__ w, int N, int K, float4* __restrict__ out, int total) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int lane = i & 63;
  int kc = (i >> 6) % (K / 8);
  int nt = (i >> 6) / (K / 8);
  int row = nt * 32 + (lane & 31);
  int col = kc * 8 + 4 * (lane >> 5);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < N) {
    const float* p = w + (size_t)row * K + col;
    v = make_float4(p[0], p[1], p[2], p[3]);
  }
  out[i] = v;
}

// PA -> row-major (inverse of pack_linear_kernel); one float4 fragment per thread
__global__ void unpack_rows_kernel(const float4* __restrict__ pa, int N, int K, float* __restrict__ out, int total) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int lane = i & 63;
  int kc = (i >> 6) % (K / 8);
  int nt = (i >> 6) / (K / 8);
  int row = nt * 32 + (lane & 31);
  int col = kc * 8 + 4 * (lane >> 5);
  if (row < N) *reinterpret_cast<float4*>(out + (size_t)row * K + col) = pa[i];
}

extern "C" hipError_t poem_launch_unpack_rows(const void* pa, int N, int K, float* out, hipStream_t s) {
  int total = ((N + 31) / 32) * (K / 8) * 64;
  hipLaunchKernelGGL(unpack_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, s, (const float4*)pa, N, K, out, total);
  return hipGetLastError();
}

extern "C" hipError_t poem_launch_pack_linear(const float* w, int N, int K, void* out, hipStream_t s) {
  int total = ((N + 31) / 32) * (K / 8) * 64;
  hipLaunchKernelGGL(pack_linear_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, N, K, (float4*)out, total);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// GEMM: activations may live in "packed-activation" (PA) order -- the same fragment order as the packed weights,
//   PA[(mt * K/8 + kc) * 64 + lane] = float4( X[32*mt + (lane&31)][8*kc + 4*(lane>>5) + 0..3] )
// so that BOTH operands are fetched with fully coalesced 1 KiB wave loads.  Writing PA output uses the transposed
// formulation D[n][m] (A = W fragment, B = X fragment): lane = row m, registers = 16 output channels, i.e. exactly
// four PA float4 fragments of the result -> coalesced 1 KiB stores, no re-layout between chained GEMMs.
// Row-major (RM) input/output remain available (plain formulation D[m][n]) for tensors that are gathered by row.
// The 4 waves of a block share one column group (identical W stream -> L1 hits) and own consecutive row groups.
template <int MT, int NT, bool IN_PA, bool OUT_PA>
__global__ __launch_bounds__(256, 2) void gemm2_kernel(const float* __restrict__ X, int ldx,
                                                       const float4* __restrict__ Wp, const float* __restrict__ bias,
                                                       const float* __restrict__ R, int ldr, float* __restrict__ Y,
                                                       int ldy, int M, int N, int K, int act) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int rg = blockIdx.x * 4 + (threadIdx.x >> 6);     // row group (MT tiles of 32 rows)
  const int cg = blockIdx.y;                              // column group (NT tiles of 32 columns)
  const int mt0 = rg * MT;
  if (mt0 * 32 >= M) return;
  const int KC = K >> 3;
  const float4* xp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (IN_PA) {
      xp[i] = reinterpret_cast<const float4*>(X) + (size_t)(mt0 + i) * KC * 64 + lane;
    } else {
      const int row = min((mt0 + i) * 32 + r, M - 1);
      xp[i] = reinterpret_cast<const float4*>(X + (size_t)row * ldx + 4 * h);
    }
  }
  constexpr int XSTEP = IN_PA ? 64 : 2;   // float4 stride per k-chunk
  const float4* wp = Wp + (size_t)(cg * NT) * KC * 64 + lane;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[i][n] = zero16();

  // software pipeline, two named register sets (no copies for the compiler to rotate away): the loads of chunk kc+1
  // are in flight while the 4*MT*NT MFMAs of chunk kc issue.
  float4 a0[MT], b0[NT], a1[MT], b1[NT];
#define POEM_LOAD(A, B, KCI)                                                        \
  {                                                                                 \
    const int kq_ = min((KCI), KC - 1);                                             \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) A[i] = xp[i][(size_t)kq_ * XSTEP]; \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) B[n] = wp[((size_t)n * KC + kq_) * 64]; \
  }
#define POEM_MMA(A, B)                                                              \
  _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                   \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                \
      const float bv = (&B[n].x)[t];                                                \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) {                              \
        const float av = (&A[i].x)[t];                                              \
        acc[i][n] = OUT_PA ? mfma32(bv, av, acc[i][n]) : mfma32(av, bv, acc[i][n]); \
      }                                                                             \
    }                                                                               \
  }
  // sched_barrier(0) pins the order: without it hipcc sinks every load down to its first use (no prefetch distance).
  POEM_LOAD(a0, b0, 0)
  int kc = 0;
  for (; kc + 1 < KC; kc += 2) {
    POEM_LOAD(a1, b1, kc + 1)
    __builtin_amdgcn_sched_barrier(0);
    POEM_MMA(a0, b0)
    __builtin_amdgcn_sched_barrier(0);
    POEM_LOAD(a0, b0, kc + 2)
    __builtin_amdgcn_sched_barrier(0);
    POEM_MMA(a1, b1)
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kc < KC) { POEM_MMA(a0, b0) }   // odd K/8: the last chunk is already in (a0, b0)
#undef POEM_LOAD
#undef POEM_MMA

  if (OUT_PA) {
    const int KCO = N >> 3;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if ((mt0 + i) * 32 >= M) break;
      float4* yp = reinterpret_cast<float4*>(Y) + (size_t)(mt0 + i) * KCO * 64 + lane;
      const float4* rp = reinterpret_cast<const float4*>(R) + (size_t)(mt0 + i) * KCO * 64 + lane;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = (cg * NT + n) * 32 + 8 * g + 4 * h;
          const int kco = (cg * NT + n) * 4 + g;
          float4 v = make_float4(acc[i][n][4 * g], acc[i][n][4 * g + 1], acc[i][n][4 * g + 2], acc[i][n][4 * g + 3]);
          if (bias) {
            const float4 bb = *reinterpret_cast<const float4*>(bias + c0);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
          }
          if (act == 1) { v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w); }
          if (act == 2) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
          if (R) {
            const float4 rr = rp[(size_t)kco * 64];
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
          }
          yp[(size_t)kco * 64] = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = (cg * NT + n) * 32 + r;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (mt0 + i) * 32 + mfma_row(e, h);
          if (row < M) {
            float v = acc[i][n][e] + bv;
            if (act == 1) v = relu_nan(v);
            if (act == 2) v = gelu_erf(v);
            if (R) v += R[(size_t)row * ldr + col];
            Y[(size_t)row * ldy + col] = v;
          }
        }
      }
    }
  }
}

template <int MT, int NT>
static hipError_t launch_gemm2_t(const float* X, int ldx, const void* Wp, const float* bias, const float* R, int ldr,
                                 float* Y, int ldy, int M, int N, int K, int act, int in_pa, int out_pa, hipStream_t s) {
  const int ntiles = (N + 31) / 32, mtiles = (M + 31) / 32;
  const int rgroups = (mtiles + MT - 1) / MT;
  dim3 grid((unsigned)((rgroups + 3) / 4), (unsigned)(ntiles / NT)), block(256);
#define POEM_G2(IP, OP)                                                                                              \
  hipLaunchKernelGGL((gemm2_kernel<MT, NT, IP, OP>), grid, block, 0, s, X, ldx, (const float4*)Wp, bias, R, ldr, Y, \
                     ldy, M, N, K, act)
  if (in_pa && out_pa) POEM_G2(true, true);
  else if (in_pa) POEM_G2(true, false);
  else if (out_pa) POEM_G2(false, true);
  else POEM_G2(false, false);
#undef POEM_G2
  return hipGetLastError();
}

// in_pa / out_pa: operand layouts (0 = row-major with ld*, 1 = packed-activation; PA buffers hold ceil(M/32)*32 rows
// and the residual shares the output's layout).  PA output needs N % 32 == 0.
extern "C" hipError_t poem_launch_gemm2(const float* X, int ldx, const void* Wp, const float* bias, const float* R,
                                        int ldr, float* Y, int ldy, int M, int N, int K, int act, int in_pa, int out_pa,
                                        hipStream_t s) {
  const int ntiles = (N + 31) / 32, mtiles = (M + 31) / 32;
  // 64x128 wave tiles (6 operand loads per 32 MFMAs instead of 3 per 8) once they fill at least 3/4 of the 1024 SIMDs
  // (the K = 4C feed-forward output Linear at M = 25568: 800 wave tiles, 214 -> 164 us)
  const bool big = (long)((mtiles + 1) / 2) * (ntiles / 4) >= 768;
  if (ntiles % 4 == 0) {
    if (big) return launch_gemm2_t<2, 4>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, in_pa, out_pa, s);
    return launch_gemm2_t<1, 2>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, in_pa, out_pa, s);
  }
  if (ntiles % 2 == 0) return launch_gemm2_t<1, 2>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, in_pa, out_pa, s);
  return launch_gemm2_t<1, 1>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, in_pa, out_pa, s);
}

// Per-segment outputs of a fused-N GEMM (panel and K-slab kernels; the output modes are described at the panel kernel)
struct PanelSegs {
  float* ptr[6];
  int mode[6];
  int seg_cols;
  int split_images;      // split variant only: write the K / V images as hi | lo f16 chunk operands (head dims 32 / 64)
  // XCD-aware block -> (rows, panel) map (launch_panel_t sets it when the grid divides evenly): the blocks of XCD x = blockIdx % 8
  // own the x-th eighth of the row groups and, among themselves, take every panel of those rows -- an X row tile is fetched
  // from HBM once, by the first panel-block of the XCD that reaches it, and served from that XCD's L2 to the others.  With the
  // plain map (panel = blockIdx % panels) the panel-blocks of a row range sit on all eight XCDs and each L2 fetches X for
  // itself: 9.6x the algorithmic X traffic on the F1 GEMM (profiles/r03_pmc.json).
  int xcd_map;
};

// ---------------------------------------------------------------------------------------------------------
// K-slab GEMM (round 4; the Linears of POEM-huge, K = 1024 / 4096, on 6-33 K rows): the whole-K W panel of the panel kernel
// below does not fit LDS there (one 32-column tile = 128 KB at K = 1024), and the operands-from-L2 kernel above ran those
// shapes at 0.27 of the matrix pipe -- 800 wave tiles for 1024 SIMDs, one chunk of prefetch against an L2 / MALL round trip.
// Here a block of 8 waves owns 8 * MT * 32 rows x NT * 32 columns and walks K in SLABS of SKC k-chunks: the slab's NT * SKC
// weight fragments are staged through LDS once per block (double-buffered: the next slab's loads are in flight during the
// current slab's MFMAs, one barrier per slab) and read by all eight waves as conflict-free ds_read_b128; only the X
// fragments come through the vector memory path, one chunk ahead.  Two blocks per CU (64 KB each) = four waves per SIMD.
// The k-order of every output element's fma chain is the panel kernel's and gemm2's: bit-identical results.
// (OMODE: 0 row-major, 1 attention K image -- operands swapped in the MFMA, lane = row --, 2 attention V image; see PanelSegs)
template <int MT, int NT, int SKC, int OMODE>
__device__ __forceinline__ void kslab_body(const float* __restrict__ X, int ldx, const float4* __restrict__ Wp,
                                           const float* __restrict__ bias, const float* __restrict__ R, int ldr,
                                           float* __restrict__ Y, int ldy, int M, int K, int pact, int cb, int rb, int ycol0) {
  constexpr int FRAGS = NT * SKC;                  // 1 KiB fragments per slab
  constexpr int PER = FRAGS / 8;                   // fragments each wave copies per slab
  static_assert(FRAGS % 8 == 0, "a slab is dealt to the eight waves");
  extern __shared__ __attribute__((aligned(16))) float4 ws[];      // 2 x FRAGS x 64 float4
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const int KC = K >> 3, nslab = KC / SKC;
  const int mt0 = (rb * 8 + wv) * MT;
  const float4* wsrc = Wp + (size_t)(cb * NT) * KC * 64 + lane;
  const __amdgpu_buffer_rsrc_t xrs = frag_rsrc(X, 0xffffffffu);
  unsigned xo[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) xo[i] = (unsigned)min((mt0 + i) * 32 + r, M - 1) * (unsigned)(ldx * 4) + 16u * h;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[i][n] = zero16();

  float4 stage[PER];
  // fragment f = wv * PER + j of a slab: column tile f / SKC, chunk f % SKC
#define KS_FETCH(SL)                                                                                     \
  _Pragma("unroll") for (int j = 0; j < PER; ++j) {                                                      \
    const int f_ = wv * PER + j;                                                                         \
    stage[j] = wsrc[((size_t)(f_ / SKC) * KC + (size_t)(SL) * SKC + f_ % SKC) * 64];                      \
  }
#define KS_COMMIT(BUF)                                                                                   \
  _Pragma("unroll") for (int j = 0; j < PER; ++j) ws[((BUF) * FRAGS + wv * PER + j) * 64 + lane] = stage[j];
  KS_FETCH(0)
  KS_COMMIT(0)
  __syncthreads();
  // X fragments: a ring of four chunks in flight (a chunk is only 4 * MT * NT MFMAs -- 512 cycles at MT = 1 -- against an
  // L2 / MALL round trip of a few thousand), named registers so that nothing is rotated through copies
  float4 a0[MT], a1[MT], a2[MT], a3[MT];
#define KS_LOADA(A, KCI) { const int kq_ = min((KCI), KC - 1); _Pragma("unroll") for (int i = 0; i < MT; ++i) A[i] = frag_load(xrs, (int)xo[i], kq_ * 32); }
#define KS_MMA(A, WB, KL)                                                                                \
  {                                                                                                      \
    float4 b_[NT];                                                                                       \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) b_[n] = (WB)[(n * SKC + (KL)) * 64 + lane];            \
    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                        \
      _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                     \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                   \
          acc[i][n] = OMODE == 1 ? mfma32((&b_[n].x)[t], (&A[i].x)[t], acc[i][n]) : mfma32((&A[i].x)[t], (&b_[n].x)[t], acc[i][n]); \
  }
  KS_LOADA(a0, 0) KS_LOADA(a1, 1) KS_LOADA(a2, 2)
  static_assert(SKC % 4 == 0, "the ring is unrolled by four");
  for (int sl = 0; sl < nslab; ++sl) {
    const float4* wb = ws + (sl & 1) * FRAGS * 64;
    KS_FETCH(min(sl + 1, nslab - 1))      // (unconditional: a conditional definition sends the staging registers to scratch)
    __builtin_amdgcn_sched_barrier(0);
    const int kc0 = sl * SKC;
    for (int kl = 0; kl < SKC; kl += 4) {
      KS_LOADA(a3, kc0 + kl + 3)
      __builtin_amdgcn_sched_barrier(0);
      KS_MMA(a0, wb, kl)
      __builtin_amdgcn_sched_barrier(0);
      KS_LOADA(a0, kc0 + kl + 4)
      __builtin_amdgcn_sched_barrier(0);
      KS_MMA(a1, wb, kl + 1)
      __builtin_amdgcn_sched_barrier(0);
      KS_LOADA(a1, kc0 + kl + 5)
      __builtin_amdgcn_sched_barrier(0);
      KS_MMA(a2, wb, kl + 2)
      __builtin_amdgcn_sched_barrier(0);
      KS_LOADA(a2, kc0 + kl + 6)
      __builtin_amdgcn_sched_barrier(0);
      KS_MMA(a3, wb, kl + 3)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (sl + 1 < nslab) { KS_COMMIT((sl + 1) & 1) }
    __syncthreads();
  }
#undef KS_FETCH
#undef KS_COMMIT
#undef KS_LOADA
#undef KS_MMA
  const int col0 = cb * NT * 32;
  if (OMODE == 1) {     // D[n][m]: lane = row, register e = channel 8 (e >> 2) + 4 h + (e & 3) of column tile n (panel kernel, mode 1)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if ((mt0 + i) * 32 >= M) break;
      float4* yp = reinterpret_cast<float4*>(Y) + ((size_t)(mt0 + i) * (ldy >> 3) + (ycol0 >> 3)) * 64 + lane;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = make_float4(acc[i][n][4 * g], acc[i][n][4 * g + 1], acc[i][n][4 * g + 2], acc[i][n][4 * g + 3]);
          if (bias) {
            const float4 bb = *reinterpret_cast<const float4*>(bias + col0 + n * 32 + 8 * g + 4 * h);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
          }
          if (pact == 1) { v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w); }
          yp[(size_t)(n * 4 + g) * 64] = v;
        }
    }
    return;
  }
  if (OMODE == 2) {     // registers 4g .. 4g+3 of a lane = four consecutive rows of one column = one float4 of the V image
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const float bv = bias ? bias[col0 + n * 32 + r] : 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if ((mt0 + i) * 32 >= M) break;
        float4* yp = reinterpret_cast<float4*>(Y) + ((size_t)(mt0 + i) * (ldy >> 5) + (ycol0 >> 5) + n) * 256 + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = make_float4(acc[i][n][4 * g] + bv, acc[i][n][4 * g + 1] + bv, acc[i][n][4 * g + 2] + bv, acc[i][n][4 * g + 3] + bv);
          if (pact == 1) { v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w); }
          yp[(size_t)g * 64] = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = col0 + n * 32 + r, ycol = ycol0 + n * 32 + r;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row0 = (mt0 + i) * 32 + 4 * h;
      if (row0 >= M) continue;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        v[e] = acc[i][n][e] + bv;
        if (pact == 1) v[e] = relu_nan(v[e]);
        if (pact == 2) v[e] = gelu_erf(v[e]);
      }
      if (row0 + 28 < M) {          // whole tile in range (rows row0 + {0..3} + 8 {0..3})
        float* yl = Y + (size_t)row0 * ldy + ycol;
        if (R) {
          const float* rl = R + (size_t)row0 * ldr + col;
          float rr[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) rr[e] = rl[(size_t)((e & 3) + 8 * (e >> 2)) * ldr];
#pragma unroll
          for (int e = 0; e < 16; ++e) yl[(size_t)((e & 3) + 8 * (e >> 2)) * ldy] = v[e] + rr[e];
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) yl[(size_t)((e & 3) + 8 * (e >> 2)) * ldy] = v[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (e & 3) + 8 * (e >> 2);
          if (row < M) Y[(size_t)row * ldy + ycol] = v[e] + (R ? R[(size_t)row * ldr + col] : 0.f);
        }
      }
    }
  }
}

template <int MT, int NT, int SKC>
__global__ __launch_bounds__(512, 2) void gemm_kslab_kernel(const float* __restrict__ X, int ldx, const float4* __restrict__ Wp,
                                                            const float* __restrict__ bias, const float* __restrict__ R, int ldr,
                                                            float* __restrict__ Y, int ldy, int M, int N, int K, int act,
                                                            int act_split, int act2, PanelSegs segs) {
  // column blocks fastest: the blocks that share an X row range are neighbours in the logical order.  Logical order = XCD-major:
  // workgroups go round-robin over the eight XCDs, so the blocks of XCD x take the x-th eighth of the logical range -- a
  // contiguous run of row blocks with all their column blocks: an X row block is fetched from HBM by ONE XCD (with the plain
  // order every XCD holds some column blocks of every row block: 8x the X traffic -- 8.6 GB for POEM-huge's front-end Linear)
  const int ncb = N / (32 * NT);
  const int nb = (int)gridDim.x;
  const int lb = (nb & 7) == 0 ? (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int cb = lb % ncb, rb = lb / ncb;
  const int col0 = cb * NT * 32;
  const int pact = (col0 >= act_split) ? act2 : act;
  if (segs.seg_cols == 0) {
    kslab_body<MT, NT, SKC, 0>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, K, pact, cb, rb, col0);
    return;
  }
  const int sidx = col0 / segs.seg_cols, ycol0 = col0 - sidx * segs.seg_cols, mode = segs.mode[sidx];      // (block-uniform)
  float* ys = segs.ptr[sidx];
  if (mode == 1) kslab_body<MT, NT, SKC, 1>(X, ldx, Wp, bias, nullptr, 0, ys, segs.seg_cols, M, K, pact, cb, rb, ycol0);
  else if (mode == 2) kslab_body<MT, NT, SKC, 2>(X, ldx, Wp, bias, nullptr, 0, ys, segs.seg_cols, M, K, pact, cb, rb, ycol0);
  else kslab_body<MT, NT, SKC, 0>(X, ldx, Wp, bias, nullptr, 0, ys, segs.seg_cols, M, K, pact, cb, rb, ycol0);
}

// shapes the K-slab kernel takes: 64-column blocks, 128-deep slabs, 16-byte aligned rows within the 4 GiB of a buffer descriptor
static bool kslab_applies(const float* X, int ldx, int M, int N, int K, int act_split) {
  return K >= 512 && K % 128 == 0 && N % 64 == 0 && (act_split >= N || act_split % 64 == 0) && !((uintptr_t)X & 15) && ldx % 4 == 0 &&
         (unsigned long long)M * ldx * 4ull < (1ull << 32) && M >= 512;
}

static hipError_t launch_gemm_kslab(const float* X, int ldx, const void* Wp, const float* bias, const float* R, int ldr, float* Y,
                                    int ldy, int M, int N, int K, int act, int act_split, int act2, const PanelSegs& segs, hipStream_t s) {
  constexpr int NT = 2, SKC = 16;
  const int mtiles = (M + 31) / 32, ncb = N / (32 * NT);
  // 256-row blocks (MT = 1) unless 512-row blocks (MT = 2: half the LDS reads per MFMA) still give every CU two blocks
  const bool mt2 = (long)((mtiles + 15) / 16) * ncb >= 2 * poem_device_cus();
  const size_t lds = (size_t)2 * NT * SKC * 1024;
  if (mt2) {
    auto kern = gemm_kslab_kernel<2, NT, SKC>;
    hipLaunchKernelGGL(kern, dim3((unsigned)(((mtiles + 15) / 16) * ncb)), dim3(512), lds, s, X, ldx, (const float4*)Wp, bias, R, ldr, Y, ldy,
                       M, N, K, act, act_split, act2, segs);
  } else {
    auto kern = gemm_kslab_kernel<1, NT, SKC>;
    hipLaunchKernelGGL(kern, dim3((unsigned)(((mtiles + 7) / 8) * ncb)), dim3(512), lds, s, X, ldx, (const float4*)Wp, bias, R, ldr, Y, ldy,
                       M, N, K, act, act_split, act2, segs);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Panel GEMM (the workhorse for K <= 512): one persistent block per CU keeps a W panel of NT column tiles x K resident
// in LDS (fragment order, conflict-free ds_read_b128) and its 8 waves (2 per SIMD) stream row groups of MT x 32 rows:
// the only global traffic of the main loop is the X operand (16-byte fragment loads, one chunk ahead), so L2 carries
// 1/3 of what the operands-from-L2 kernel above needs, and a fused wide N (several Linears that share the input,
// weights concatenated along N) reads X from HBM once per panel pass.
// act2 applies to columns >= act_split (two Linears with different activations fused along N).
//
// Output modes, per column segment of a fused N (PanelSegs; seg_cols == 0 -> one row-major output Y/ldy):
//   0  row-major: plain formulation D[m][n] (lane = output column), every store writes two 128-byte row segments
//   1  K image for attn.hip: transposed formulation D[n][m] (operands swapped in the MFMA, same registers) -- lane =
//      row, registers = 16 output channels = four float4 fragments of the packed-activation order -> 1 KiB stores
//   2  V image for attn.hip: plain formulation, registers 4g..4g+3 of a lane are four consecutive rows of one column
//      = one float4 of the image -> 1 KiB stores
// The image modes need M % 32 == 0 and carry no residual.

template <int NT, int MT, bool GELU, int OMODE, bool SPLIT = false>
__device__ __forceinline__ void panel_rows(const float* __restrict__ X, int ldx, const float4* __restrict__ wl,
                                           const float* __restrict__ bias, const float* __restrict__ R, int ldr,
                                           float* __restrict__ Y, int ldy, int M, int K, int pact, int col0, int ycol0,
                                           int bip, int blocks_in_panel, const float* __restrict__ tile_scales = nullptr,
                                           int scale_stride = 0, int split_images = 0, int part = 0, int nparts = 1) {
  constexpr int NWV = (SPLIT && !GELU) ? POEM_GS_WAVES : 8;      // split variant: 3 waves per SIMD (<= 170 VGPRs)
  const int KC = K >> 3;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const int mtiles = (M + 31) / 32;
  const int rgroups = (mtiles + MT - 1) / MT;
  const __amdgpu_buffer_rsrc_t xrs = frag_rsrc(X, 0xffffffffu);
  float bv[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) bv[n] = (bias && OMODE != 1) ? bias[col0 + n * 32 + r] : 0.f;
  const size_t lane_yo = (size_t)(4 * h) * ldy + r, lane_ro = (size_t)(4 * h) * ldr + r;
  // (part / nparts: this block's share of the row groups -- the XCD's range under PanelSegs::xcd_map)
  const int rg_lo = (int)((long)rgroups * part / nparts), rg_hi = (int)((long)rgroups * (part + 1) / nparts);
#ifndef POEM_PANEL_LAB
#define POEM_PANEL_LAB 0          // tools/lab/f1_lab: 1 = no image stores (timing only), 4 = store epilogue at priority 3
#endif
#ifdef POEM_PANEL_SKEW            // tools/lab/f1_lab: the second wave of every SIMD starts this many 100 MHz ticks late
  if (wv >= NWV / 2) { const long long t0_ = wall_clock64(); while (wall_clock64() - t0_ < POEM_PANEL_SKEW) __builtin_amdgcn_s_sleep(32); }
#endif
  for (int rg = rg_lo + bip * NWV + wv; rg < rg_hi; rg += blocks_in_panel * NWV) {
#if POEM_PANEL_LAB & 4
    __builtin_amdgcn_s_setprio(0);
#endif
    const int mt0 = rg * MT;
    // X fragments through the buffer descriptor: per-lane byte offset (row, half) computed once per row group, the
    // k-chunk offset is scalar -- no VALU address arithmetic inside the MFMA loop
    unsigned xo[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xo[i] = (unsigned)min((mt0 + i) * 32 + r, M - 1) * (unsigned)(ldx * 4) + 16u * h;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[i][n] = zero16();
    if constexpr (SPLIT) {
      // f16 hi | lo splits on v_mfma_f32_32x32x16_f16, fp32 accumulation: per 16-k chunk the X fragment (8 fp32 per lane,
      // two 16-byte loads, one chunk ahead) is scaled, split in registers and multiplied with the panel's hi | lo weight
      // fragments from LDS as x_lo w_hi + x_hi w_hi + x_hi w_lo; the scales are undone below, before the shared epilogue.
      const int KC16 = K >> 4;
      const h8* wl8 = reinterpret_cast<const h8*>(wl);
      unsigned xo8[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) xo8[i] = (unsigned)min((mt0 + i) * 32 + r, M - 1) * (unsigned)(ldx * 4) + 32u * h;
      float4 ra[2][MT][2];
#define POEM_LOADA8(S, KCI) { const int kq_ = min((KCI), KC16 - 1); _Pragma("unroll") for (int i = 0; i < MT; ++i) { \
        ra[S][i][0] = frag_load(xrs, (int)xo8[i], kq_ * 64); ra[S][i][1] = frag_load(xrs, (int)xo8[i], kq_ * 64 + 16); } }
      POEM_LOADA8(0, 0)
      h8 bh_next = wl8[lane], bl_next = wl8[64 + lane];
      for (int kc = 0; kc < KC16; ++kc) {
        const int cur = kc & 1;
        if (cur == 0) { POEM_LOADA8(1, kc + 1) } else { POEM_LOADA8(0, kc + 1) }
        h8 ah[MT], al[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float4 lo4 = cur == 0 ? ra[0][i][0] : ra[1][i][0], hi4 = cur == 0 ? ra[0][i][1] : ra[1][i][1];
          const float xv[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
          for (int t = 0; t < 8; ++t) {
#if defined(POEM_GS_LAB) && POEM_GS_LAB == 1        // lab: no conversion arithmetic (wrong numbers, timing only)
            ah[i][t] = __builtin_bit_cast(_Float16, (unsigned short)__float_as_uint(xv[t]));
            al[i][t] = ah[i][t];
#else
            const float v = __builtin_amdgcn_fmed3f(xv[t] * POEM_GEMM_SX, -60000.f, 60000.f);
            ah[i][t] = (_Float16)v;
            al[i][t] = (_Float16)(v - (float)ah[i][t]);
#endif
          }
        }
        // weight fragments one column tile ahead (the first tile of the next chunk after the last one): an LDS read issued
        // right in front of its MFMAs stalls the wave for the LDS latency every six MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int nn = n + 1 < NT ? n + 1 : 0, kn = n + 1 < NT ? kc : min(kc + 1, KC16 - 1);
          const h8 bh = bh_next, bl = bl_next;
          bh_next = wl8[((size_t)(nn * KC16 + kn) * 2 + 0) * 64 + lane];
          bl_next = wl8[((size_t)(nn * KC16 + kn) * 2 + 1) * 64 + lane];
          __builtin_amdgcn_sched_barrier(0);
          // part-major: MT independent accumulators between two MFMAs on the same one
#define POEM_MMA8(XA, WB)                                                                                      \
          _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                       \
            acc[i][n] = OMODE == 1 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(WB, XA[i], acc[i][n], 0, 0, 0)     \
                                   : __builtin_amdgcn_mfma_f32_32x32x16_f16(XA[i], WB, acc[i][n], 0, 0, 0);
#if defined(POEM_GS_LAB) && POEM_GS_LAB == 2        // lab: no MFMAs
#pragma unroll
          for (int i = 0; i < MT; ++i) acc[i][n][0] += (float)al[i][0] * (float)bh[0] + (float)ah[i][1] * (float)bl[1];
#else
          POEM_MMA8(al, bh) POEM_MMA8(ah, bh) POEM_MMA8(ah, bl)
#endif
#undef POEM_MMA8
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#undef POEM_LOADA8
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float inv = 1.0f / (tile_scales[(size_t)(col0 / 32 + n) * scale_stride] * POEM_GEMM_SX);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][n][e] *= inv;
      }
    } else {
    float4 a0[MT], a1[MT], b0[NT], b1[NT];
#define POEM_LOADA(A, KCI) { const int kq_ = min((KCI), KC - 1); _Pragma("unroll") for (int i = 0; i < MT; ++i) A[i] = frag_load(xrs, (int)xo[i], kq_ * 32); }
#define POEM_LOADB(B, KCI) { const int kq_ = min((KCI), KC - 1); _Pragma("unroll") for (int n = 0; n < NT; ++n) B[n] = wl[(n * KC + kq_) * 64 + lane]; }
#define POEM_MMA(A, B)                                                          \
  _Pragma("unroll") for (int t = 0; t < 4; ++t) {                               \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) {                            \
      const float bw = (&B[n].x)[t];                                            \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                            \
        acc[i][n] = OMODE == 1 ? mfma32(bw, (&A[i].x)[t], acc[i][n]) : mfma32((&A[i].x)[t], bw, acc[i][n]); \
    }                                                                           \
  }
    POEM_LOADA(a0, 0) POEM_LOADB(b0, 0)
    int kc = 0;
    for (; kc + 1 < KC; kc += 2) {
      POEM_LOADA(a1, kc + 1) POEM_LOADB(b1, kc + 1)
      __builtin_amdgcn_sched_barrier(0);
      POEM_MMA(a0, b0)
      __builtin_amdgcn_sched_barrier(0);
      POEM_LOADA(a0, kc + 2) POEM_LOADB(b0, kc + 2)
      __builtin_amdgcn_sched_barrier(0);
      POEM_MMA(a1, b1)
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kc < KC) { POEM_MMA(a0, b0) }
#if POEM_PANEL_LAB & 4
    __builtin_amdgcn_s_setprio(3);
#endif
#undef POEM_LOADA
#undef POEM_LOADB
#undef POEM_MMA
    }
    if (OMODE == 1) {
      // D[n][m]: lane = row of the tile, register e = channel 8(e>>2) + 4h + (e&3) of column tile n.
      // image float4 index ((mt * ldy/8 + kco) * 64 + lane), kco = (ycol0 + 32n)/8 + g      (ldy = segment width)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if ((mt0 + i) * 32 >= M) break;
        float4* yp = reinterpret_cast<float4*>(Y) + ((size_t)(mt0 + i) * (ldy >> 3) + (ycol0 >> 3)) * 64 + lane;
        float4 vprev = make_float4(0.f, 0.f, 0.f, 0.f);
        (void)vprev;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 v = make_float4(acc[i][n][4 * g], acc[i][n][4 * g + 1], acc[i][n][4 * g + 2], acc[i][n][4 * g + 3]);
            if (bias) {      // (from LDS: see gemm_panel_kernel)
              const float4 bb = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(wl + NT * KC * 64) + n * 32 + 8 * g + 4 * h);
              v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            }
            if (pact == 1) { v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w); }
            if (GELU && pact == 2) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
            if (SPLIT && split_images) {
              // split images for the split cross attention: fragments (2c, 2c+1) of the fp32 image become the hi | lo f16
              // operands of 16-slot chunk c at the same two addresses (attn.hip, xattn_split_kernel)
              if (g & 1) {
                h8 hi, lo;
                split_pair(vprev, v, hi, lo);
                yp[(size_t)(n * 4 + g - 1) * 64] = __builtin_bit_cast(float4, hi);
                yp[(size_t)(n * 4 + g) * 64] = __builtin_bit_cast(float4, lo);
              }
              vprev = v;
            } else {
#if POEM_PANEL_LAB & 1
              asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
#else
              yp[(size_t)(n * 4 + g) * 64] = v;
#endif
            }
          }
      }
      continue;
    }
    // Epilogue, one 32 x 32 tile at a time (bias, activation, residual, store) so that the live temporaries stay one
    // tile wide -- the erf of the GELU variant over all MT x NT tiles at once spilled hundreds of registers.  Bias lives
    // in registers for the whole block, the store address is a wave-uniform row base + one per-lane offset, in-bounds
    // tiles skip the row checks; every VALU instruction here comes out of the matrix pipe's time.
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int trow0 = (mt0 + i) * 32;
      if (trow0 >= M) break;
      float* yl = Y + (size_t)trow0 * ldy + (size_t)ycol0 + lane_yo;               // per-lane base, once
      const float* rl = R ? R + (size_t)trow0 * ldr + (size_t)col0 + lane_ro : nullptr;
      const bool full = trow0 + 32 <= M;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float v[16];
        if (GELU && pact == 2) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = gelu_erf(acc[i][n][e] + bv[n]);
        } else if (pact == 1) {
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const f32x2 t = f32x2{acc[i][n][e], acc[i][n][e + 1]} + f32x2{bv[n], bv[n]};
            v[e] = relu_nan(t[0]); v[e + 1] = relu_nan(t[1]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const f32x2 t = f32x2{acc[i][n][e], acc[i][n][e + 1]} + f32x2{bv[n], bv[n]};
            v[e] = t[0]; v[e + 1] = t[1];
          }
        }
        if (OMODE == 2) {
          // image float4 index (((mt * ldy/32 + vt) * 4 + g) * 64 + lane), vt = ycol0/32 + n
          float4* yp = reinterpret_cast<float4*>(Y) + ((size_t)(mt0 + i) * (ldy >> 5) + (ycol0 >> 5) + n) * 256 + lane;
          if (SPLIT && split_images) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              h8 hi, lo;
              split_pair(make_float4(v[8 * c], v[8 * c + 1], v[8 * c + 2], v[8 * c + 3]),
                         make_float4(v[8 * c + 4], v[8 * c + 5], v[8 * c + 6], v[8 * c + 7]), hi, lo);
              yp[(size_t)(2 * c) * 64] = __builtin_bit_cast(float4, hi);
              yp[(size_t)(2 * c + 1) * 64] = __builtin_bit_cast(float4, lo);
            }
          } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#if POEM_PANEL_LAB & 1
            asm volatile("" :: "v"(v[4 * g]), "v"(v[4 * g + 1]), "v"(v[4 * g + 2]), "v"(v[4 * g + 3]));
#else
            yp[(size_t)g * 64] = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
#endif
          }
          }
        } else if (full) {
          if (rl) {
            float rr[16];                                      // 16 residual loads in flight, then 16 stores
#pragma unroll
            for (int e = 0; e < 16; ++e) rr[e] = rl[(size_t)((e & 3) + 8 * (e >> 2)) * ldr + n * 32];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 16; ++e) yl[(size_t)((e & 3) + 8 * (e >> 2)) * ldy + n * 32] = v[e] + rr[e];
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) yl[(size_t)((e & 3) + 8 * (e >> 2)) * ldy + n * 32] = v[e];   // row = ro + 4h
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int ro = (e & 3) + 8 * (e >> 2);
            if (trow0 + ro + 4 * h < M) yl[(size_t)ro * ldy + n * 32] = v[e] + (rl ? rl[(size_t)ro * ldr + n * 32] : 0.f);
          }
        }
      }
    }
  }
}

template <int NT, int MT, bool GELU, bool SPLIT = false>
__global__ __launch_bounds__((SPLIT && !GELU) ? POEM_GS_WAVES * 64 : 512, 2) void gemm_panel_kernel(const float* __restrict__ X, int ldx,
                                                            const float4* __restrict__ Wp, const float* __restrict__ bias,
                                                            const float* __restrict__ R, int ldr, float* __restrict__ Y,
                                                            int ldy, int M, int N, int K, int act, int act_split, int act2,
                                                            PanelSegs segs, const float* __restrict__ tile_scales = nullptr,
                                                            int scale_stride = 0) {
  constexpr int NWV = (SPLIT && !GELU) ? POEM_GS_WAVES : 8;      // split variant: 3 waves per SIMD (<= 170 VGPRs)
  const int KC = K >> 3;
  extern __shared__ __attribute__((aligned(16))) float4 wl[];   // NT * KC * 64 float4
  const int tid = threadIdx.x;
  const int panels = N / (32 * NT);
  const int xm = segs.xcd_map;
  const int lb = xm ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;           // xcd_map: slot within the XCD
  const int panel = lb % panels, bip = lb / panels;
  const int blocks_in_panel = xm ? (int)(gridDim.x >> 3) / panels : ((int)gridDim.x - panel + panels - 1) / panels;
  const int part = xm ? (int)(blockIdx.x & 7) : 0, nparts = xm ? 8 : 1;
  {
    const float4* src = Wp + (size_t)panel * NT * KC * 64;
    for (int i = tid; i < NT * KC * 64; i += NWV * 64) wl[i] = src[i];
    // the panel's NT * 32 bias values behind it (round 6): the K-image epilogue read them from global memory between its stores,
    // and on gfx9 a load's `s_waitcnt vmcnt(0)` also waits for every older STORE to be acknowledged -- the epilogue was a chain
    // of 32 store round trips per wave and row group (13 % of the F1 GEMM, tools/lab/f1_lab).  LDS reads count on lgkmcnt.
    float* lb = reinterpret_cast<float*>(wl + NT * KC * 64);
    if (tid < NT * 32) lb[tid] = bias ? bias[panel * NT * 32 + tid] : 0.f;
  }
  __syncthreads();
  const int col0 = panel * NT * 32;
  const int pact = (col0 >= act_split) ? act2 : act;
  if (segs.seg_cols == 0) {
    panel_rows<NT, MT, GELU, 0, SPLIT>(X, ldx, wl, bias, R, ldr, Y, ldy, M, K, pact, col0, col0, bip, blocks_in_panel,
                                       tile_scales, scale_stride, 0, part, nparts);
    return;
  }
  const int sidx = col0 / segs.seg_cols;
  const int ycol0 = col0 - sidx * segs.seg_cols;
  float* ys = segs.ptr[sidx];
  const int mode = segs.mode[sidx];
  // (the image modes exist in the activation-free instantiation only: the launcher sends segmented GEMMs there)
  if (!GELU && mode == 1)
    panel_rows<NT, MT, false, 1, SPLIT>(X, ldx, wl, bias, nullptr, 0, ys, segs.seg_cols, M, K, pact, col0, ycol0, bip,
                                        blocks_in_panel, tile_scales, scale_stride, segs.split_images, part, nparts);
  else if (!GELU && mode == 2)
    panel_rows<NT, MT, false, 2, SPLIT>(X, ldx, wl, bias, nullptr, 0, ys, segs.seg_cols, M, K, pact, col0, ycol0, bip,
                                        blocks_in_panel, tile_scales, scale_stride, segs.split_images, part, nparts);
  else
    panel_rows<NT, MT, GELU, 0, SPLIT>(X, ldx, wl, bias, nullptr, 0, ys, segs.seg_cols, M, K, pact, col0, ycol0, bip,
                                       blocks_in_panel, tile_scales, scale_stride, 0, part, nparts);
}

static int poem_num_cus() { return poem_device_cus(); }
static std::atomic<int> g_kslab{1};                            // A/B switch (poem_set_option "gemm_kslab"; process-wide, scheduling only)
extern "C" void poem_gemm_kslab(int on) { g_kslab = on; }
static std::atomic<int> g_panel_narrow{1};                     // A/B switch (lab): narrower panels when the wide ones leave waves idle
extern "C" void poem_gemm_panel_narrow(int on) { g_panel_narrow = on; }
static std::atomic<int> g_panel_xcd_map{1};                    // A/B switch (poem_set_option "gemm_xcd_map"; process-wide, scheduling only)
extern "C" void poem_gemm_xcd_map(int on) { g_panel_xcd_map = on; }

template <int NT, int MT, bool GELU, bool SPLIT = false>
static hipError_t launch_panel_t(const float* X, int ldx, const void* Wp, const float* bias, const float* R, int ldr,
                                 float* Y, int ldy, int M, int N, int K, int act, int act_split, int act2,
                                 const PanelSegs& segs, hipStream_t s, const float* tile_scales = nullptr,
                                 int scale_stride = 0) {
  const size_t lds = (size_t)NT * (K / 8) * 64 * 16 + (size_t)NT * 32 * sizeof(float);      // W panel | its bias values
  auto kern = gemm_panel_kernel<NT, MT, GELU, SPLIT>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), 128 * 1024 + 512, optin); e != hipSuccess) return e;
  const int panels = N / (32 * NT);
  const int grid = std::max(poem_num_cus(), panels);
  PanelSegs sg = segs;
  // XCD-aware map: every XCD's blocks (grid / 8, one per CU) divide evenly over the panels and there is enough work for
  // eight row ranges (rows of a range are dealt to (blocks per panel) x 8 waves x MT tiles)
  const int mtiles = (M + 31) / 32;
  sg.xcd_map = g_panel_xcd_map && grid % 8 == 0 && (grid / 8) % panels == 0 && mtiles >= 8 * ((grid / 8) / panels) * 8 * MT;
  hipLaunchKernelGGL(kern, dim3(grid), dim3((SPLIT && !GELU) ? POEM_GS_WAVES * 64 : 512), lds, s, X, ldx, (const float4*)Wp, bias, R,
                     ldr, Y, ldy, M, N, K, act, act_split, act2, sg, tile_scales, scale_stride);
  return hipGetLastError();
}

static hipError_t launch_gemm_split_impl(const float* X, int ldx, const void* Wp, const float* bias, const float* R,
                                         int ldr, float* Y, int ldy, int M, int N, int K, int act, int act_split,
                                         int act2, const PanelSegs& segs, hipStream_t s) {
  const bool seg = segs.seg_cols > 0;
  // panel kernel: N a multiple of 32*NT, panel (NT*K*128 B) within 128 KiB of LDS, the split on a panel boundary
  // (a few row tiles -- a small batch's F1: M = 4096 B rows -- on wide panels leave most waves without a tile: the widest panel
  //  that still gives every wave of the chip one (row tile, panel) pair; same fma chain per output element whatever the width)
  const bool in_split_mode = g_explicit_split.img || (g_split_ctx.packed && (const char*)Wp >= g_split_ctx.packed &&
                                                      (const char*)Wp < g_split_ctx.packed + g_split_ctx.bytes);
  int NT = 0;
  const long wave_slots = 8L * poem_num_cus(), row_tiles = (M + 31) / 32;
  for (int c : {4, 2, 1})
    if (N % (32 * c) == 0 && (size_t)c * K * 128 <= 128 * 1024 && (act_split >= N || act_split % (32 * c) == 0) &&
        (!seg || segs.seg_cols % (32 * c) == 0)) {
      NT = c;
      if (row_tiles * (N / (32 * c)) >= wave_slots || !g_panel_narrow || K > 256 || in_split_mode) break;
    }
  // deep K leaves room for a single 32-column tile per panel, which re-reads X every 8 MFMAs: the K-slab kernel (weights
  // staged through LDS slab by slab) takes those shapes -- POEM-huge's Linears, the K = 4C feed-forward output
  const bool split_mode = in_split_mode;
  if (NT <= 1 && !split_mode && g_kslab && kslab_applies(X, ldx, M, N, K, act_split) && (!seg || (segs.seg_cols % 64 == 0 && M % 32 == 0)))
    return launch_gemm_kslab(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, act_split, act2, segs, s);
  if (!seg && NT == 1 && N >= 64 && K >= 512 && !(act_split < N && act2 != act)) NT = 0;
  if (NT == 0 || K % 8 || ((uintptr_t)X & 15) || ldx % 4 || (unsigned long long)M * ldx * 4ull >= (1ull << 32)) {
    if (seg || (act_split < N && act2 != act) || g_explicit_split.img) return hipErrorInvalidValue;
    return poem_launch_gemm2(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, 0, 0, s);
  }
  const int mtiles = (M + 31) / 32, panels = N / (32 * NT);
  const int wpp = std::max(1, std::max(poem_num_cus(), panels) / panels) * 8;       // waves per panel
  auto cost = [&](int mt) { return (long)(((mtiles + mt - 1) / mt + wpp - 1) / wpp) * mt; };
  const bool mt2_default = 5 * cost(2) <= 6 * cost(1);   // 64-row wave tiles unless the 32-row split balances >= 20 % better
  const bool mt2 = mt2_default;
  const bool gelu = act == 2 || (act_split < N && act2 == 2);
  if (seg && gelu) return hipErrorInvalidValue;   // image outputs exist in the GELU-free instantiation only
  // opt-in split precision: the weight image lies inside the registered arena (or explicit images were given)
  const float* tsc = nullptr;
  int tstride = 0;
  const void* Wsplit = nullptr;
  if (g_explicit_split.img) {
    Wsplit = g_explicit_split.img; tsc = g_explicit_split.scales; tstride = 1;
  } else if (g_split_ctx.packed && (const char*)Wp >= g_split_ctx.packed && (const char*)Wp < g_split_ctx.packed + g_split_ctx.bytes) {
    const size_t off = (const char*)Wp - g_split_ctx.packed;
    Wsplit = g_split_ctx.split + off; tsc = g_split_ctx.scales + off / 256; tstride = K / 2;
  }
  if (g_explicit_split.img && K % 16) return hipErrorInvalidValue;
  if (Wsplit && K % 16 == 0 && (unsigned long long)M * ldx * 4ull + 64ull < (1ull << 32)) {
    // 32-row wave tiles always: the fp32->f16 split of the X fragment is serial work in front of each chunk's MFMAs,
    // and a 64-row tile doubles it per wave (measured: 131072 x 1536 x 256 0.47 vs 0.64 ms, 1M x 256 x 384 1.33 vs 2.5 ms)
#ifdef POEM_LAB
    static const int force_mt = getenv("POEM_GS_MT") ? atoi(getenv("POEM_GS_MT")) : 1;       // lab A/B: 2 = 64-row tiles
#else
    constexpr int force_mt = 1;
#endif
    const bool mt2 = force_mt == 2;
    (void)mt2_default;
#define POEM_PANEL_S(NTV)                                                                                                \
    if (gelu)                                                                                                            \
      return mt2 ? launch_panel_t<NTV, 2, true, true>(X, ldx, Wsplit, bias, R, ldr, Y, ldy, M, N, K, act, act_split, act2, segs, s, tsc, tstride)   \
                 : launch_panel_t<NTV, 1, true, true>(X, ldx, Wsplit, bias, R, ldr, Y, ldy, M, N, K, act, act_split, act2, segs, s, tsc, tstride);  \
    return mt2 ? launch_panel_t<NTV, 2, false, true>(X, ldx, Wsplit, bias, R, ldr, Y, ldy, M, N, K, act, act_split, act2, segs, s, tsc, tstride)    \
               : launch_panel_t<NTV, 1, false, true>(X, ldx, Wsplit, bias, R, ldr, Y, ldy, M, N, K, act, act_split, act2, segs, s, tsc, tstride)
    if (NT == 4) { POEM_PANEL_S(4); }
    if (NT == 2) { POEM_PANEL_S(2); }
    POEM_PANEL_S(1);
#undef POEM_PANEL_S
  }
#define POEM_PANEL(NTV)                                                                                                  \
  if (gelu)                                                                                                              \
    return mt2 ? launch_panel_t<NTV, 2, true>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, act_split, act2, segs, s)   \
               : launch_panel_t<NTV, 1, true>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, act_split, act2, segs, s);  \
  return mt2 ? launch_panel_t<NTV, 2, false>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, act_split, act2, segs, s)    \
             : launch_panel_t<NTV, 1, false>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, act_split, act2, segs, s)
  if (NT == 4) { POEM_PANEL(4); }
  if (NT == 2) { POEM_PANEL(2); }
  POEM_PANEL(1);
#undef POEM_PANEL
}

// Fused-N aware GEMM entry: columns [0, act_split) use `act`, columns [act_split, N) use `act2`.
extern "C" hipError_t poem_launch_gemm_split(const float* X, int ldx, const void* Wp, const float* bias, const float* R,
                                             int ldr, float* Y, int ldy, int M, int N, int K, int act, int act_split,
                                             int act2, hipStream_t s) {
  PanelSegs none{};
  return launch_gemm_split_impl(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, act_split, act2, none, s);
}

// Fused-N GEMM whose N = nsegs * seg_cols columns go to separate outputs, each in its own layout:
// mode 0 row-major (ld = seg_cols), 1 attention K image, 2 attention V image (see attn.hip).  M % 32 == 0.
extern "C" hipError_t poem_launch_gemm_segs(const float* X, int ldx, const void* Wp, const float* bias, int M, int K,
                                            int act, int seg_cols, int nsegs, float* const* outs, const int* modes,
                                            hipStream_t s) {
  if (nsegs < 1 || nsegs > 6 || seg_cols % 32 || M % 32) return hipErrorInvalidValue;
  PanelSegs segs{};
  segs.seg_cols = seg_cols;
  segs.split_images = g_split_images;
  for (int i = 0; i < nsegs; ++i) { segs.ptr[i] = outs[i]; segs.mode[i] = modes[i]; }
  const int N = seg_cols * nsegs;
  return launch_gemm_split_impl(X, ldx, Wp, bias, nullptr, 0, nullptr, 0, M, N, K, act, N, act, segs, s);
}

extern "C" hipError_t poem_launch_gemm(const float* X, int ldx, const void* Wp, const float* bias, const float* R,
                                       int ldr, float* Y, int ldy, int M, int N, int K, int act, hipStream_t s) {
  return poem_launch_gemm_split(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, N, act, s);
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wave per row, two-pass (mean, then biased variance), eps inside the sqrt.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y, int rows,
                                                        int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * cols;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += xr[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)cols;
  float v = 0.f;
  for (int c = lane; c < cols; c += 64) { float d = xr[c] - mean; v += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const float rstd = 1.0f / sqrtf(v / (float)cols + eps);
  float* yr = y + (size_t)row * cols;
  for (int c = lane; c < cols; c += 64) yr[c] = (xr[c] - mean) * rstd * g[c] + b[c];
}

extern "C" hipError_t poem_launch_layernorm(const float* x, const float* g, const float* b, float* y, int rows,
                                            int cols, float eps, hipStream_t s) {
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y, rows, cols, eps);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Narrow Linear (N <= 8 outputs): out[row, n] = (base ? base[row, n] : 0) + x[row, :] . w[n, :] + b[n].
// One wave per row, lanes stride the K dim.  Used for reg_branch.2 (C -> 3) with the xyz residual.
__global__ __launch_bounds__(256) void narrow_linear_kernel(const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ base, float* __restrict__ out,
                                                            int rows, int K, int N, int npb) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  // blockIdx.y: this wave's npb output columns (the same fma chain and reduction tree per (row, column) whatever the split)
  for (int n = blockIdx.y * npb, ne = min(N, n + npb); n < ne; ++n) {
    float s = 0.f;
    for (int c = lane; c < K; c += 64) s = fmaf(xr[c], w[(size_t)n * K + c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[(size_t)row * N + n] = (base ? base[(size_t)row * N + n] : 0.f) + (s + (b ? b[n] : 0.f));
  }
}

extern "C" hipError_t poem_launch_narrow_linear(const float* x, int ldx, const float* w, const float* b,
                                                const float* base, float* out, int rows, int K, int N,
                                                hipStream_t s) {
  // a wave walks its output columns one after the other (a latency chain of dependent loads and shuffles per column): fine for
  // the 3 columns of reg_branch.2 over 25 k rows, 170 us for the 106 columns of mano_linear over 32 rows (round 6: one column per
  // wave there -- the chip has the waves)
  const int npb = (long)((rows + 3) / 4) * N <= 65535 && (rows + 3) / 4 < 2048 ? 1 : N;
  hipLaunchKernelGGL(narrow_linear_kernel, dim3((rows + 3) / 4, (N + npb - 1) / npb), dim3(256), 0, s, x, ldx, w, b, base, out, rows, K, N, npb);
  return hipGetLastError();
}

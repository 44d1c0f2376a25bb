// Development lab (not product): fp32-MFMA GEMM variants timed with HIP events on random data.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gemm_lab gemm_lab.hip
#include "../../poem-v2_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// W panel resident in LDS, persistent waves streaming X in packed-activation order.
// block = NWV waves, one block per CU.  Panel = NT column tiles (NT*32 columns) x K.
// MODE bit0: never store (runtime-opaque predicate), bit1: X from an L2-resident window, bit2: skip MFMAs of odd chunks
template <int K, int NT, int MT, int NWV, int MODE>
__global__ __launch_bounds__(NWV * 64, NWV / 4) void gemm_wres_kernel(const float4* __restrict__ X, const float4* __restrict__ Wp,
                                                          const float* __restrict__ bias, float4* __restrict__ Y,
                                                          int M, int N, int act, int blocks_per_panel, long long* dbg) {
  constexpr int KC = K / 8;
  extern __shared__ __attribute__((aligned(16))) float4 wl[];   // NT*KC*64 float4
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = lane >> 5;
  const int panels = N / (32 * NT);
  int panel = blockIdx.x % panels, bip = blockIdx.x / panels;
  blocks_per_panel = ((int)gridDim.x - panel + panels - 1) / panels;
  if (MODE & 16) {   // XCD-aware: the `panels` blocks that stream the same rows sit on one XCD (shared L2)
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8, per = (int)gridDim.x / 8;
    const int parts = per / panels;
    panel = slot % panels;
    const int part = slot / panels;
    if (part >= parts) return;
    bip = xcd * parts + part;
    blocks_per_panel = 8 * parts;
  }
  {
    const float4* src = Wp + (size_t)panel * NT * KC * 64;
    for (int i = tid; i < NT * KC * 64; i += NWV * 64) wl[i] = src[i];
  }
  __syncthreads();
  const int mtiles = (M + 31) / 32;
  const int rgroups = (mtiles + MT - 1) / MT;
  const int KCO = N >> 3;
  for (int rg = bip * NWV + wv; rg < rgroups; rg += blocks_per_panel * NWV) {
    const int mt0 = rg * MT;
    const float4* xp[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
      xp[i] = X + (size_t)((MODE & 2) ? (wv * MT + i) : min(mt0 + i, mtiles - 1)) * KC * 64 + lane;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[i][n] = zero16();
    float4 a0[MT], a1[MT], b0[NT], b1[NT];
#define LOADA(A, KCI) { const int kq_ = min((KCI), KC - 1); _Pragma("unroll") for (int i = 0; i < MT; ++i) A[i] = xp[i][(size_t)kq_ * 64]; }
#define LOADB(B, KCI) { const int kq_ = min((KCI), KC - 1); _Pragma("unroll") for (int n = 0; n < NT; ++n) B[n] = wl[(n * KC + kq_) * 64 + lane]; }
#define MMA(A, B) _Pragma("unroll") for (int t = 0; t < 4; ++t) { _Pragma("unroll") for (int n = 0; n < NT; ++n) { const float bv = (&B[n].x)[t]; \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[i][n] = mfma32(bv, (&A[i].x)[t], acc[i][n]); } }
    long long t0 = 0, t1 = 0, t2 = 0;
    if (MODE & 8) t0 = clock64();
    LOADA(a0, 0) LOADB(b0, 0)
    for (int kc = 0; kc < KC; kc += 2) {
      LOADA(a1, kc + 1) LOADB(b1, kc + 1)
      __builtin_amdgcn_sched_barrier(0);
      MMA(a0, b0)
      __builtin_amdgcn_sched_barrier(0);
      LOADA(a0, kc + 2) LOADB(b0, kc + 2)
      __builtin_amdgcn_sched_barrier(0);
      if (!(MODE & 4)) { MMA(a1, b1) }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef LOADA
#undef LOADB
#undef MMA
    if (MODE & 8) t1 = clock64();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (mt0 + i >= mtiles) break;
      float4* yp = Y + (size_t)(mt0 + i) * KCO * 64 + lane;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = (panel * NT + n) * 32 + 8 * g + 4 * h;
          const int kco = (panel * NT + n) * 4 + g;
          float4 v = make_float4(acc[i][n][4 * g], acc[i][n][4 * g + 1], acc[i][n][4 * g + 2], acc[i][n][4 * g + 3]);
          if (bias) { const float4 bb = *reinterpret_cast<const float4*>(bias + c0); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
          if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (!(MODE & 1) || v.x == 1.2345e30f) yp[(size_t)kco * 64] = v;
        }
      }
    }
    if (MODE & 8) { t2 = clock64(); if (blockIdx.x == 3 && lane == 0) { const int slot = ((rg - bip * NWV - wv) / (blocks_per_panel * NWV)) * NWV + wv; if (slot < 256) { dbg[slot * 3] = t0; dbg[slot * 3 + 1] = t1; dbg[slot * 3 + 2] = t2; } } }
  }
}


// ---- V4: W panel resident, waves own contiguous tile ranges, partner waves (w >= NWV/2) start with a single tile so the
// two waves of a SIMD run half a period apart: one wave's epilogue stores overlap the other's MFMAs.
template <int K, int NT, int MT>
__device__ __forceinline__ void panel_rowgroup(const float4* __restrict__ X, const float4* wl, const float* __restrict__ bias,
                                               float4* __restrict__ Y, int mt0, int N, int panel, int act, int lane) {
  constexpr int KC = K / 8;
  const int h = lane >> 5;
  const int KCO = N >> 3;
  const float4* xp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) xp[i] = X + (size_t)(mt0 + i) * KC * 64 + lane;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[i][n] = zero16();
  float4 a0[MT], a1[MT], b0[NT], b1[NT];
#define LOADA(A, KCI) { const int kq_ = min((KCI), KC - 1); _Pragma("unroll") for (int i = 0; i < MT; ++i) A[i] = xp[i][(size_t)kq_ * 64]; }
#define LOADB(B, KCI) { const int kq_ = min((KCI), KC - 1); _Pragma("unroll") for (int n = 0; n < NT; ++n) B[n] = wl[(n * KC + kq_) * 64 + lane]; }
#define MMA(A, B) _Pragma("unroll") for (int t = 0; t < 4; ++t) { _Pragma("unroll") for (int n = 0; n < NT; ++n) { const float bv = (&B[n].x)[t]; \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[i][n] = mfma32(bv, (&A[i].x)[t], acc[i][n]); } }
  LOADA(a0, 0) LOADB(b0, 0)
  for (int kc = 0; kc < KC; kc += 2) {
    LOADA(a1, kc + 1) LOADB(b1, kc + 1)
    __builtin_amdgcn_sched_barrier(0);
    MMA(a0, b0)
    __builtin_amdgcn_sched_barrier(0);
    LOADA(a0, kc + 2) LOADB(b0, kc + 2)
    __builtin_amdgcn_sched_barrier(0);
    MMA(a1, b1)
    __builtin_amdgcn_sched_barrier(0);
  }
#undef LOADA
#undef LOADB
#undef MMA
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    float4* yp = Y + (size_t)(mt0 + i) * KCO * 64 + lane;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = (panel * NT + n) * 32 + 8 * g + 4 * h;
        const int kco = (panel * NT + n) * 4 + g;
        float4 v = make_float4(acc[i][n][4 * g], acc[i][n][4 * g + 1], acc[i][n][4 * g + 2], acc[i][n][4 * g + 3]);
        if (bias) { const float4 bb = *reinterpret_cast<const float4*>(bias + c0); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
        if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        yp[(size_t)kco * 64] = v;
      }
    }
  }
}

template <int K, int NT, int NWV, int STAG>
__global__ __launch_bounds__(NWV * 64, NWV / 4) void gemm_panel_kernel(const float4* __restrict__ X, const float4* __restrict__ Wp,
                                                          const float* __restrict__ bias, float4* __restrict__ Y,
                                                          int M, int N, int act, int blocks_per_panel, long long* dbg) {
  constexpr int KC = K / 8;
  extern __shared__ __attribute__((aligned(16))) float4 wl[];   // NT*KC*64 float4
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int panel = blockIdx.x / blocks_per_panel, bip = blockIdx.x % blocks_per_panel;
  {
    const float4* src = Wp + (size_t)panel * NT * KC * 64;
    for (int i = tid; i < NT * KC * 64; i += NWV * 64) wl[i] = src[i];
  }
  __syncthreads();
  const int mtiles = (M + 31) / 32;
  const int nwaves = blocks_per_panel * NWV;
  const int widx = bip * NWV + wv;
  // contiguous, balanced tile ranges
  const int base = mtiles / nwaves, rem = mtiles % nwaves;
  int mt = widx * base + min(widx, rem);
  const int end = mt + base + (widx < rem ? 1 : 0);
  if (STAG && wv >= NWV / 2 && mt < end && ((end - mt) & 1) == 0) {   // even count: 1 + 2.. + 1 ; odd count: 2.. + 1 vs 1 + 2..
    panel_rowgroup<K, NT, 1>(X, wl, bias, Y, mt, N, panel, act, lane);
    mt += 1;
  }
  if (STAG && wv < NWV / 2 && mt < end && ((end - mt) & 1) == 1) {
    panel_rowgroup<K, NT, 1>(X, wl, bias, Y, mt, N, panel, act, lane);
    mt += 1;
  }
  for (; mt + 2 <= end; mt += 2) panel_rowgroup<K, NT, 2>(X, wl, bias, Y, mt, N, panel, act, lane);
  if (mt < end) panel_rowgroup<K, NT, 1>(X, wl, bias, Y, mt, N, panel, act, lane);
}


// ---- V5: W panel resident in LDS; X row-major straight to registers (fragment-shaped 16-B loads), Y row-major via the
// plain formulation D[m][n] (lane = column: 128-B row segments per store).
template <int K, int NT, int MT, int NWV>
__global__ __launch_bounds__(NWV * 64, NWV / 4) void gemm_wres_rm_kernel(const float* __restrict__ X, const float4* __restrict__ Wp,
                                                          const float* __restrict__ bias, float* __restrict__ Y,
                                                          int M, int N, int act, int blocks_per_panel, long long* dbg) {
  constexpr int KC = K / 8;
  extern __shared__ __attribute__((aligned(16))) float4 wl[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const int panels = N / (32 * NT);
  const int panel = blockIdx.x % panels, bip = blockIdx.x / panels;
  blocks_per_panel = ((int)gridDim.x - panel + panels - 1) / panels;
  {
    const float4* src = Wp + (size_t)panel * NT * KC * 64;
    for (int i = tid; i < NT * KC * 64; i += NWV * 64) wl[i] = src[i];
  }
  __syncthreads();
  const int mtiles = (M + 31) / 32;
  const int rgroups = (mtiles + MT - 1) / MT;
  for (int rg = bip * NWV + wv; rg < rgroups; rg += blocks_per_panel * NWV) {
    const int mt0 = rg * MT;
    const float4* xp[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xp[i] = reinterpret_cast<const float4*>(X + (size_t)min((mt0 + i) * 32 + r, M - 1) * K + 4 * h);
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[i][n] = zero16();
    float4 a0[MT], a1[MT], b0[NT], b1[NT];
#define LOADA(A, KCI) { const int kq_ = min((KCI), KC - 1); _Pragma("unroll") for (int i = 0; i < MT; ++i) A[i] = xp[i][(size_t)kq_ * 2]; }
#define LOADB(B, KCI) { const int kq_ = min((KCI), KC - 1); _Pragma("unroll") for (int n = 0; n < NT; ++n) B[n] = wl[(n * KC + kq_) * 64 + lane]; }
#define MMA(A, B) _Pragma("unroll") for (int t = 0; t < 4; ++t) { _Pragma("unroll") for (int n = 0; n < NT; ++n) { const float bv = (&B[n].x)[t]; \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[i][n] = mfma32((&A[i].x)[t], bv, acc[i][n]); } }
    LOADA(a0, 0) LOADB(b0, 0)
    for (int kc = 0; kc < KC; kc += 2) {
      LOADA(a1, kc + 1) LOADB(b1, kc + 1)
      __builtin_amdgcn_sched_barrier(0);
      MMA(a0, b0)
      __builtin_amdgcn_sched_barrier(0);
      LOADA(a0, kc + 2) LOADB(b0, kc + 2)
      __builtin_amdgcn_sched_barrier(0);
      MMA(a1, b1)
      __builtin_amdgcn_sched_barrier(0);
    }
#undef LOADA
#undef LOADB
#undef MMA
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = (panel * NT + n) * 32 + r;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (mt0 + i) * 32 + mfma_row(e, h);
          if (row < M) {
            float v = acc[i][n][e] + bv;
            if (act == 1) v = fmaxf(v, 0.f);
            Y[(size_t)row * N + col] = v;
          }
        }
      }
    }
  }
}


// ---- V6: as V5 but the store epilogue is software-pipelined: a unit = (row group of MT tiles, NTH = NT/2 column tiles);
// two accumulator sets alternate, the previous unit's results are stored 2*MT*NTH values at a time between chunk pairs
// of the current unit's K loop, so stores trickle out at a steady rate instead of chip-wide bursts.
template <int NT, int MT, int NWV>
__global__ __launch_bounds__(NWV * 64, NWV / 4) void gemm_pipe_kernel(const float* __restrict__ X, const float4* __restrict__ Wp,
                                                          const float* __restrict__ bias, float* __restrict__ Y,
                                                          int M, int N, int K, int act) {
  constexpr int NTH = NT / 2;
  const int KC = K >> 3;
  const int PP = KC >> 4;                 // chunk pairs per two accumulator registers (K % 128 == 0)
  extern __shared__ __attribute__((aligned(16))) float4 wl[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const int panels = N / (32 * NT);
  const int panel = blockIdx.x % panels, bip = blockIdx.x / panels;
  const int blocks_in_panel = ((int)gridDim.x - panel + panels - 1) / panels;
  {
    const float4* src = Wp + (size_t)panel * NT * KC * 64;
    for (int i = tid; i < NT * KC * 64; i += NWV * 64) wl[i] = src[i];
  }
  __syncthreads();
  const int mtiles = (M + 31) / 32;
  const int rgroups = (mtiles + MT - 1) / MT;
  const int rg0 = bip * NWV + wv, rstride = blocks_in_panel * NWV;
  if (rg0 >= rgroups) return;
  const int nunits = ((rgroups - rg0 + rstride - 1) / rstride) * 2;     // (row group, half) units of this wave

  f32x16 accA[MT][NTH], accB[MT][NTH];
#define ZERO(ACC) _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int n = 0; n < NTH; ++n) ACC[i][n] = zero16();
#define STORE_REG(ACC, U, E)                                                                     \
  {                                                                                              \
    const int rgp_ = rg0 + ((U) >> 1) * rstride, hf_ = (U) & 1;                                    \
    _Pragma("unroll") for (int n = 0; n < NTH; ++n) {                                            \
      const int col_ = (panel * NT + hf_ * NTH + n) * 32 + r;                                    \
      const float bv_ = bias ? bias[col_] : 0.f;                                                 \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                           \
        const int row_ = (rgp_ * MT + i) * 32 + mfma_row((E), h);                                \
        if (row_ < M) {                                                                          \
          float v_ = ACC[i][n][(E)] + bv_;                                                       \
          if (act == 1) v_ = fmaxf(v_, 0.f);                                                     \
          Y[(size_t)row_ * N + col_] = v_;                                                       \
        }                                                                                        \
      }                                                                                          \
    }                                                                                            \
  }
#define LOADA(A, KCI) { const int kq_ = min((KCI), KC - 1); _Pragma("unroll") for (int i = 0; i < MT; ++i) A[i] = xp[i][(size_t)kq_ * 2]; }
#define LOADB(B, KCI) { const int kq_ = min((KCI), KC - 1); _Pragma("unroll") for (int n = 0; n < NTH; ++n) B[n] = wb[(n * KC + kq_) * 64]; }
#define MMA(ACC, A, B) _Pragma("unroll") for (int t = 0; t < 4; ++t) { _Pragma("unroll") for (int n = 0; n < NTH; ++n) { const float bv = (&B[n].x)[t]; \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) ACC[i][n] = mfma32((&A[i].x)[t], bv, ACC[i][n]); } }
  // one unit: K loop into CUR while the previous unit's PREV registers are stored two at a time
#define UNIT(CUR, PREV, U)                                                                       \
  {                                                                                              \
    const int rg_ = rg0 + ((U) >> 1) * rstride, half_ = (U) & 1;                                 \
    const float4* xp[MT];                                                                        \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                               \
      xp[i] = reinterpret_cast<const float4*>(X + (size_t)min((rg_ * MT + i) * 32 + r, M - 1) * K + 4 * h); \
    const float4* wb = wl + (size_t)half_ * NTH * KC * 64 + lane;                                \
    ZERO(CUR)                                                                                    \
    float4 a0[MT], a1[MT], b0[NTH], b1[NTH];                                                     \
    LOADA(a0, 0) LOADB(b0, 0)                                                                    \
    int kc = 0;                                                                                  \
    _Pragma("unroll") for (int e2 = 0; e2 < 8; ++e2) {                                           \
      for (int p = 0; p < PP; ++p, kc += 2) {                                                    \
        LOADA(a1, kc + 1) LOADB(b1, kc + 1)                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        MMA(CUR, a0, b0)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        LOADA(a0, kc + 2) LOADB(b0, kc + 2)                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        MMA(CUR, a1, b1)                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                       \
      }                                                                                          \
      if ((U) > 0) { STORE_REG(PREV, (U) - 1, 2 * e2) STORE_REG(PREV, (U) - 1, 2 * e2 + 1) }     \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    }                                                                                            \
  }
  for (int u = 0; u < nunits; u += 2) {
    UNIT(accA, accB, u)
    UNIT(accB, accA, u + 1)
  }
  // drain: the last unit (always odd index -> accB)
#pragma unroll
  for (int e = 0; e < 16; ++e) STORE_REG(accB, nunits - 1, e)
#undef ZERO
#undef STORE_REG
#undef LOADA
#undef LOADB
#undef MMA
#undef UNIT
}

static float time_it(std::function<void()> fn, int n = 10) {
  hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
  for (int i = 0; i < 2; ++i) fn();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(s));
  for (int i = 0; i < n; ++i) fn();
  CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
  float ms; CK(hipEventElapsedTime(&ms, s, e));
  return ms / n;
}

int main() {
  const int K = 256;
  for (int M : {131072}) {
    for (int N : {256}) {
      if (M > 200000 && N > 256) continue;
      size_t xb = packed_linear_floats(M, K) * 4, yb = packed_linear_floats(M, N) * 4;
      float *x, *xpa, *w, *wp, *b, *y1, *y2;
      CK(hipMalloc(&x, (size_t)((M + 31) / 32 * 32) * K * 4)); CK(hipMalloc(&xpa, xb)); CK(hipMalloc(&w, (size_t)N * K * 4));
      CK(hipMalloc(&wp, packed_linear_floats(N, K) * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&y1, yb)); CK(hipMalloc(&y2, yb));
      {
        std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N);
        for (auto& v : hx) v = (float)rand() / (float)RAND_MAX - 0.5f;
        for (auto& v : hw) v = ((float)rand() / (float)RAND_MAX - 0.5f) / 8;
        for (auto& v : hb) v = (float)rand() / (float)RAND_MAX - 0.5f;
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
      }
      CK(poem_launch_pack_linear(x, M, K, xpa, 0));
      CK(poem_launch_pack_linear(w, N, K, wp, 0));
      CK(hipMemset(y1, 0, yb)); CK(hipMemset(y2, 0, yb));
      const double fl = 2.0 * M * N * K;
      float ms = time_it([&] { (void)poem_launch_gemm2(xpa, K, wp, b, nullptr, 0, y1, N, M, N, K, 0, 1, 1, 0); });
      printf("M=%8d N=%d  gemm2 PA->PA          %8.1f us %6.1f TF\n", M, N, ms * 1e3, fl / ms / 1e9);
      long long* dbg; CK(hipMalloc(&dbg, 256 * 3 * 8)); CK(hipMemset(dbg, 0, 256 * 3 * 8));
      auto run_wres = [&](auto kern, int NT, int NWV, const char* name) {
        const int panels = N / (32 * NT);
        const int bpp = 256 / panels;
        const size_t lds = (size_t)NT * (K / 8) * 64 * 16;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipMemset(y2, 0, yb));
        float t = time_it([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(NWV * 64), lds, 0, (const float4*)xpa, (const float4*)wp, b, (float4*)y2, M, N, 0, bpp, dbg); });
        CK(hipGetLastError());
        std::vector<float> h1(yb / 4), h2(yb / 4);
        CK(hipMemcpy(h1.data(), y1, yb, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), y2, yb, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < h1.size(); ++i) if (h1[i] != h2[i]) ++bad;
        printf("M=%8d N=%d  %-22s %8.1f us %6.1f TF  mismatches=%zu\n", M, N, name, t * 1e3, fl / t / 1e9, bad);
      };
      run_wres(gemm_wres_kernel<256, 4, 2, 8, 0>, 4, 8, "wres MT2 8w");
      run_wres(gemm_wres_kernel<256, 4, 2, 4, 0>, 4, 4, "wres MT2 4w");
      run_wres(gemm_wres_kernel<256, 4, 1, 8, 0>, 4, 8, "wres MT1 8w");
      for (int mode : {8, 11}) {
        CK(hipMemset(dbg, 0, 256 * 3 * 8));
        if (mode == 8) run_wres(gemm_wres_kernel<256, 4, 2, 8, 8>, 4, 8, "wres MT2 8w timed");
        else run_wres(gemm_wres_kernel<256, 4, 2, 8, 11>, 4, 8, "wres MT2 8w timed nost+L2X");
        std::vector<long long> hd(256 * 3); CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
        long long base = hd[0];
        for (int sl = 0; sl < 16; ++sl) if (hd[sl * 3]) printf("   slot %2d (wave %d): start %8lld  loop %7lld  epi %6lld\n", sl, sl % 8, hd[sl * 3] - base, hd[sl * 3 + 1] - hd[sl * 3], hd[sl * 3 + 2] - hd[sl * 3 + 1]);
      }
      {
        float* yr1; float* yr2; const size_t rb = (size_t)((M + 31) / 32 * 32) * N * 4;
        CK(hipMalloc(&yr1, rb)); CK(hipMalloc(&yr2, rb)); CK(hipMemset(yr2, 0, rb));
        float t0 = time_it([&] { (void)poem_launch_gemm2(x, K, wp, b, nullptr, 0, yr1, N, M, N, K, 0, 0, 0, 0); });
        printf("M=%8d N=%d  gemm2 RM->RM          %8.1f us %6.1f TF\n", M, N, t0 * 1e3, fl / t0 / 1e9);
        auto kern = gemm_wres_rm_kernel<256, 4, 2, 8>;
        const size_t lds = (size_t)4 * (K / 8) * 64 * 16;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        float t = time_it([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, (const float*)x, (const float4*)wp, b, yr2, M, N, 0, 0, dbg); });
        CK(hipGetLastError());
        std::vector<float> h1((size_t)M * N), h2((size_t)M * N);
        CK(hipMemcpy(h1.data(), yr1, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), yr2, h2.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < h1.size(); ++i) if (h1[i] != h2[i]) ++bad;
        printf("M=%8d N=%d  %-22s %8.1f us %6.1f TF  mismatches=%zu\n", M, N, "wres RM->RM MT2 8w", t * 1e3, fl / t / 1e9, bad);
        {
          auto k2 = gemm_pipe_kernel<4, 2, 8>;
          CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          CK(hipMemset(yr2, 0, rb));
          float t2 = time_it([&] { hipLaunchKernelGGL(k2, dim3(256), dim3(512), lds, 0, (const float*)x, (const float4*)wp, b, yr2, M, N, K, 0); });
          CK(hipGetLastError());
          CK(hipMemcpy(h2.data(), yr2, h2.size() * 4, hipMemcpyDeviceToHost));
          size_t bad2 = 0; for (size_t i = 0; i < h1.size(); ++i) if (h1[i] != h2[i]) ++bad2;
          printf("M=%8d N=%d  %-22s %8.1f us %6.1f TF  mismatches=%zu\n", M, N, "pipe RM->RM MT2 8w", t2 * 1e3, fl / t2 / 1e9, bad2);
          auto k3 = gemm_pipe_kernel<4, 1, 8>;
          CK(hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          CK(hipMemset(yr2, 0, rb));
          t2 = time_it([&] { hipLaunchKernelGGL(k3, dim3(256), dim3(512), lds, 0, (const float*)x, (const float4*)wp, b, yr2, M, N, K, 0); });
          CK(hipGetLastError());
          CK(hipMemcpy(h2.data(), yr2, h2.size() * 4, hipMemcpyDeviceToHost));
          bad2 = 0; for (size_t i = 0; i < h1.size(); ++i) if (h1[i] != h2[i]) ++bad2;
          printf("M=%8d N=%d  %-22s %8.1f us %6.1f TF  mismatches=%zu\n", M, N, "pipe RM->RM MT1 8w", t2 * 1e3, fl / t2 / 1e9, bad2);
        }
        CK(hipFree(yr1)); CK(hipFree(yr2));
      }
      CK(hipFree(x)); CK(hipFree(xpa)); CK(hipFree(w)); CK(hipFree(wp)); CK(hipFree(b)); CK(hipFree(y1)); CK(hipFree(y2));
    }
  }
  return 0;
}

// OPT-IN split-precision variant of the fused vector attention (vecattn.hip, composed form).  NOT the default path:
// poem_set_precision(h, POEM_PRECISION_SPLIT_F16X3) selects it; the default stays the exact-fp32 kernel.
//
// The three C x C per-neighbour GEMMs (51 % of a step on the fp32 matrix pipe at 157 TFLOP/s) run on the f16 matrix
// pipe (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate) as hi/lo splits with fp32 accumulation:
//   x = x_hi + x_lo (+ O(2^-22 x)),  x_hi = f16(x), x_lo = f16(x - x_hi);   w likewise (split once at handle creation)
//   sum_k w x  ~=  sum_k (w_hi x_hi + w_hi x_lo + w_lo x_hi)      -- three MFMAs per tile and 16-k chunk, one fp32 accumulator
// The dropped w_lo x_lo term is O(2^-22) relative per product, i.e. at the level of fp32 round-off of a K = 256
// contraction.  Both operands are pre-multiplied by powers of two (weights: per matrix so that max |w'| is in [8,16);
// activations: SX = 64) so that the lo parts stay in the f16 normal range; the scales are undone exactly in the
// epilogues (folded into the bias fma / the softmax scale).  Activations saturate at 60000 / 64 = 937.5 (never inf).
// Everything else (first layer 3 -> C, gathers, pos, softmax, weighted sum) is the fp32 arithmetic of vecattn.hip.
// Measured parity (tests/test_hip_parity.py::test_split_precision_*): MPVPE vs the reference fixtures at the fp32
// kernel's own level (~1e-5 mm, tolerance 1e-3 mm); tools/lab/split_precision_probe.py is the CPU pre-study.
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define POEM_SPLIT_SX 64.0f
#define POEM_SPLIT_MAX 60000.0f       // scaled activations saturate here (|x| = 937.5) instead of overflowing to inf

// tools/lab only (POEM_VS_LAB): 1 = MFMAs replaced by a register touch (non-MFMA floor), 2 = every weight fragment load
// hits the same 2 KiB (L1-resident: what the L2 stream costs), 3 = no v_j gathers in the epilogue
#if defined(POEM_VS_LAB) && POEM_VS_LAB == 1
#define VS_LAB_MMA(M, C_, W_, X_) lab_touch(C_, W_, X_)
#else
#define VS_LAB_MMA(M, C_, W_, X_) (M)
#endif
#if defined(POEM_VS_LAB) && POEM_VS_LAB == 2
#define VS_LAB_WOFF(o) ((o) & 1024)
#else
#define VS_LAB_WOFF(o) (o)
#endif

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ h8 as_h8(float4 v) { return __builtin_bit_cast(h8, v); }
__device__ __forceinline__ f32x16 lab_touch(f32x16 c, h8 w, h8 x) {
  c[0] += (float)w[0] * (float)x[0];
  return c;
}

// ---- weight image: [(nt * KC + kc) * 2 + part][lane] = half8(W'[32 nt + (lane & 31)][16 kc + 8 (lane >> 5) + 0..7]),
// KC = C / 16, part 0 = hi, 1 = lo, W' = W * scale (scale = power of two, written to *scale_out).
__global__ __launch_bounds__(1024) void pack_split_kernel(const float* __restrict__ W, int C, h8* __restrict__ img,
                                                         float* __restrict__ scale_out) {
  __shared__ float red[1024];
  float m = 0.f;
  for (int i = threadIdx.x; i < C * C; i += 1024) m = fmaxf(m, fabsf(W[i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  m = red[0];
  int e = 0;
  if (m > 0.f) (void)frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1)
  const float s = m > 0.f ? ldexpf(1.0f, 4 - e) : 1.0f;   // m * s in [8, 16)
  if (threadIdx.x == 0) *scale_out = s;
  const int KC = C / 16;
  for (int idx = threadIdx.x; idx < (C / 32) * KC * 64; idx += 1024) {
    const int lane = idx & 63, kc = (idx >> 6) % KC, nt = (idx >> 6) / KC;
    const float* src = W + (size_t)(32 * nt + (lane & 31)) * C + 16 * kc + 8 * (lane >> 5);
    h8 hi, lo;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float v = src[t] * s;
      hi[t] = (_Float16)v;
      lo[t] = (_Float16)(v - (float)hi[t]);
    }
    img[((size_t)(nt * KC + kc) * 2 + 0) * 64 + lane] = hi;
    img[((size_t)(nt * KC + kc) * 2 + 1) * 64 + lane] = lo;
  }
}

extern "C" hipError_t poem_launch_pack_split(const float* w, int C, void* img, float* scale_out, hipStream_t s) {
  pack_split_kernel<<<1, 1024, 0, s>>>(w, C, (h8*)img, scale_out);
  return hipGetLastError();
}

struct VecAttnSplitArgs {
  const float* query_xyz;
  const float* src_xyz;
  const float* anchor_xyz;
  const int* idx;
  int shared_idx;
  const float* q;           // qg = W_g1 q + (W_g1 b_d2 + b_g1)
  const float* k;           // kg = W_g1 k
  const float* v;
  int NS;
  const float* wd1;         // (C,3) fp32
  const float* bd1;
  const void* wd2;          // split images
  const float* bd2;
  const void* wg1;          // W_g1 W_d2
  const void* wg2;
  const float* scales;      // device: {s_d2, s_g1, s_g2}
  float* out;
  int B, Q;
  int ldq, ldk, ldv;
};

// One GEMM of the chain: acc[tile][p] (+)= W'[tile rows][:] . X'[:][p columns]   (FLIP: X' rows x W' columns).
// X: LDS bytes, row per neighbour column: [C/8 chunks][hi 8 halfs | lo 8 halfs], ROW = 4 C + 16 bytes (the pad keeps the
// 16-lane groups of a ds_read_b128 on distinct banks).
template <int C, int P, int NW, int TPW, bool FLIP, bool INIT0>
__device__ __forceinline__ void chain_gemm_split(const void* __restrict__ Wimg, const char* __restrict__ X,
                                                 f32x16 (&acc)[TPW][P], int wv, int lane) {
  constexpr int KC = C / 16;
  constexpr int ROW = C * 4 + 16;
  const int j = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t wrs = frag_rsrc(Wimg, (unsigned)(C * C * 4));
  const int wbase = __builtin_amdgcn_readfirstlane(wv) * TPW * KC * 2048;
  const int loff = lane * 16;
  const char* xc = X + j * ROW + h * 32;
  // Weight fragments: ring of WR slots, chunk kc+WR-1 requested before the MFMAs of chunk kc issue (the fragment stream
  // is L2-latency-bound at distance 1: tools/lab, POEM_VS_LAB).  LDS operands: ONE slot -- the three MFMA groups of a
  // chunk run (w_hi x_lo), (w_hi x_hi), (w_lo x_hi), so x_lo of the next chunk is requested right after the first group
  // and x_hi after the last; the next chunk's first group covers the x_hi latency.  Order pinned with sched_barrier.
  constexpr int WR = 2;
  static_assert(KC % WR == 0 && KC >= 2 * WR, "ring depth must divide the chunk count");
  h8 wh[WR][TPW], wl[WR][TPW], xh[P], xl[P];
#define VS_LOADW(S, WOFF)                                                                  \
  _Pragma("unroll") for (int tp = 0; tp < TPW; ++tp) {                                     \
    wh[S][tp] = as_h8(frag_load(wrs, loff, VS_LAB_WOFF((WOFF) + tp * KC * 2048)));         \
    wl[S][tp] = as_h8(frag_load(wrs, loff, VS_LAB_WOFF((WOFF) + tp * KC * 2048 + 1024)));  \
  }
#define VS_READXH(XP) _Pragma("unroll") for (int p = 0; p < P; ++p) xh[p] = *reinterpret_cast<const h8*>((XP) + p * 32 * ROW);
#define VS_READXL(XP) _Pragma("unroll") for (int p = 0; p < P; ++p) xl[p] = *reinterpret_cast<const h8*>((XP) + p * 32 * ROW + 16);
#define VS_MMA(WA, XA, FIRST)                                                              \
  _Pragma("unroll") for (int tp = 0; tp < TPW; ++tp)                                       \
    _Pragma("unroll") for (int p = 0; p < P; ++p) {                                        \
      const f32x16 c_ = (FIRST) ? zero16() : acc[tp][p];                                   \
      acc[tp][p] = VS_LAB_MMA(FLIP ? mfma16(XA[p], WA[tp], c_) : mfma16(WA[tp], XA[p], c_), c_, WA[tp], XA[p]); \
    }
#if defined(POEM_VS_LAB) && POEM_VS_LAB == 4      // lab: no GEMM loop at all (what everything around the GEMMs costs)
  if (INIT0) {
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int p = 0; p < P; ++p) acc[tp][p] = zero16();
  }
  return;
#endif
#if defined(POEM_VS_LAB) && POEM_VS_LAB == 5      // lab: MFMAs only (operands loaded once, outside the loop)
  {
    h8 w0[TPW], w1[TPW], x0[P], x1[P];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) { w0[tp] = as_h8(frag_load(wrs, loff, wbase + tp * 2048)); w1[tp] = as_h8(frag_load(wrs, loff, wbase + tp * 2048 + 1024)); }
#pragma unroll
    for (int p = 0; p < P; ++p) { x0[p] = *reinterpret_cast<const h8*>(xc + p * 32 * ROW); x1[p] = *reinterpret_cast<const h8*>(xc + p * 32 * ROW + 16); }
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int p = 0; p < P; ++p) if (INIT0) acc[tp][p] = zero16();
#pragma unroll 2
    for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[tp][p] = FLIP ? mfma16(x1[p], w0[tp], acc[tp][p]) : mfma16(w0[tp], x1[p], acc[tp][p]);
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[tp][p] = FLIP ? mfma16(x0[p], w0[tp], acc[tp][p]) : mfma16(w0[tp], x0[p], acc[tp][p]);
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int p = 0; p < P; ++p) acc[tp][p] = FLIP ? mfma16(x0[p], w1[tp], acc[tp][p]) : mfma16(w1[tp], x0[p], acc[tp][p]);
    }
    return;
  }
#endif
  int woff = wbase;
  const char* xp = xc;
  // one group = WR consecutive chunks with compile-time ring slots
  auto group = [&](auto first_c, auto last_c) __attribute__((always_inline)) {
    constexpr bool FIRSTG = decltype(first_c)::value, LASTG = decltype(last_c)::value;
#pragma unroll
    for (int u = 0; u < WR; ++u) {
      if (!LASTG || u == 0) { VS_LOADW((u + WR - 1) % WR, woff + (u + WR - 1) * 2048) }
      const bool have_next = !(LASTG && u == WR - 1);
      __builtin_amdgcn_sched_barrier(0);
      VS_MMA(wh[u], xl, FIRSTG && INIT0 && u == 0)
      __builtin_amdgcn_sched_barrier(0);
      if (have_next) { VS_READXL(xp + (u + 1) * 64) }
      __builtin_amdgcn_sched_barrier(0);
      VS_MMA(wh[u], xh, false) VS_MMA(wl[u], xh, false)
      __builtin_amdgcn_sched_barrier(0);
      if (have_next) { VS_READXH(xp + (u + 1) * 64) }
    }
    woff += WR * 2048;
    xp += WR * 64;
  };
#pragma unroll
  for (int sl = 0; sl < WR - 1; ++sl) { VS_LOADW(sl, woff + sl * 2048) }
  VS_READXL(xp)
  VS_READXH(xp)
  group(std::true_type{}, std::false_type{});         // peeled: the very first MFMA takes the inline-constant 0 as C
  for (int kc = WR; kc < KC - WR; kc += WR) group(std::false_type{}, std::false_type{});
  group(std::false_type{}, std::true_type{});          // nothing left to prefetch after the last chunk
#undef VS_READXH
#undef VS_READXL
#undef VS_LOADW
#undef VS_MMA
}

// v (4 consecutive channels of one column, fp32, already scaled) -> hi | lo halfs in the LDS row
__device__ __forceinline__ void store_split(char* dst, f32x4 v) {
  const h4 hi = __builtin_convertvector(v, h4);
  const f32x4 r = v - __builtin_convertvector(hi, f32x4);
  const h4 lo = __builtin_convertvector(r, h4);
  *reinterpret_cast<h4*>(dst) = hi;
  *reinterpret_cast<h4*>(dst + 16) = lo;
}

template <int C, int P, int NW, int MINW>
__global__ __launch_bounds__(NW * 64, MINW) void vecattn_split_kernel(VecAttnSplitArgs A) {
  constexpr int TPW = C / 32 / NW;
  constexpr int ROW = C * 4 + 16;
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem_split[];
  char* X = smem_split;                                              // 32 P rows of ROW bytes
  float* dl = reinterpret_cast<float*>(X + 32 * P * ROW);            // P*32*3 coordinate deltas
  int* sidx = reinterpret_cast<int*>(dl + P * 32 * 3);
  int* voffs = sidx + P * 32;
  float* qs = reinterpret_cast<float*>(voffs + P * 32);              // P*C query rows

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, h = lane >> 5;
  const int groups = (A.Q + P - 1) / P;
  const int total = A.B * groups;
  const float inv_sqrt_c = 1.0f / sqrtf((float)C);
  const float s_d2 = A.scales[0], s_g1 = A.scales[1], s_g2 = A.scales[2];
  const float inv1 = 1.0f / (s_d2 * POEM_SPLIT_SX);          // GEMM 1 result -> pos
  const float S2 = s_g1 * POEM_SPLIT_SX;                     // scale GEMM 2 accumulates in
  const float f2 = 1.0f / s_g1;                              // relu(acc) -> SX-scaled activations
  const float k2 = inv_sqrt_c * 1.44269504088896340736f / (s_g2 * POEM_SPLIT_SX);   // softmax scale of the scaled logits

  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    const int b = item / groups;
    const int i0 = (item % groups) * P;
    __syncthreads();
    // ---- stage 0 (as vecattn.hip): neighbour ids, coordinate deltas, query rows
    if (tid < 32 * P) {
      const int p = tid >> 5, jj = tid & 31;
      const int qi = min(i0 + p, A.Q - 1);
      const int id = A.shared_idx ? A.idx[jj] : A.idx[((size_t)b * A.Q + qi) * 32 + jj];
      const float* qx = A.query_xyz + ((size_t)b * A.Q + qi) * 3;
      const float* nx = A.anchor_xyz ? A.anchor_xyz + jj * 3 : A.src_xyz + ((size_t)b * A.NS + id) * 3;
      dl[tid * 3 + 0] = qx[0] - nx[0];
      dl[tid * 3 + 1] = qx[1] - nx[1];
      dl[tid * 3 + 2] = qx[2] - nx[2];
      sidx[tid] = id;
      voffs[tid] = (int)(((unsigned)b * (unsigned)A.NS + (unsigned)id) * (unsigned)(A.ldv * 4));
    }
    for (int f = tid; f < P * C / 4; f += NT) {
      const int p = f / (C / 4), c4 = f % (C / 4);
      const int qi = min(i0 + p, A.Q - 1);
      reinterpret_cast<float4*>(qs)[f] = *reinterpret_cast<const float4*>(A.q + ((size_t)b * A.Q + qi) * A.ldq + 4 * c4);
    }
    __syncthreads();
    // h = relu(W_d1 delta + b_d1): exact fp32 (two zero-padded 32x32x2 k-steps), stored SX-scaled as hi | lo halfs
    {
      float db0[P], db1[P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        db0[p] = dl[(p * 32 + j) * 3 + h];
        db1[p] = h == 0 ? dl[(p * 32 + j) * 3 + 2] : 0.f;
      }
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        const int tile = wv * TPW + tp;
        const int crow = tile * 32 + j;
        const float wa0 = A.wd1[crow * 3 + h];
        const float wa1 = h == 0 ? A.wd1[crow * 3 + 2] : 0.f;
        const int cbase = tile * 32 + 4 * h;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          f32x16 hh = zero16();
          hh = mfma32(wa0, db0[p], hh);
          hh = mfma32(wa1, db1[p], hh);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 bb = *reinterpret_cast<const float4*>(A.bd1 + cbase + 8 * g);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e)      // relu, x64, saturating below f16's largest finite value (one v_med3)
              v[e] = __builtin_amdgcn_fmed3f((hh[4 * g + e] + (&bb.x)[e]) * POEM_SPLIT_SX, 0.f, POEM_SPLIT_MAX);
            store_split(X + (32 * p + j) * ROW + (tile * 4 + g) * 32 + h * 8, v);
          }
        }
      }
    }
    __syncthreads();

    f32x16 acc[TPW][P], pos[TPW][P];
    // kg_j rows wait in acc (GEMM 1 accumulates into pos)
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      const int cbase = (wv * TPW + tp) * 32 + 4 * h;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float* krow = A.k + ((size_t)b * A.NS + sidx[p * 32 + j]) * A.ldk + cbase;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 kk = *reinterpret_cast<const float4*>(krow + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[tp][p][4 * g + e] = (&kk.x)[e];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- GEMM 1: pos = W_d2 h + b_d2
    chain_gemm_split<C, P, NW, TPW, false, true>(A.wd2, X, pos, wv, lane);
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      const int cbase = (wv * TPW + tp) * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bb = *reinterpret_cast<const float4*>(A.bd2 + cbase + 8 * g);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float4 qq = *reinterpret_cast<const float4*>(qs + p * C + cbase + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const int i = 4 * g + e;
            const f32x2 pv = __builtin_elementwise_fma(f32x2{pos[tp][p][i], pos[tp][p][i + 1]}, f32x2{inv1, inv1},
                                                       f32x2{(&bb.x)[e], (&bb.x)[e + 1]});
            const f32x2 tv = (f32x2{(&qq.x)[e], (&qq.x)[e + 1]} - f32x2{acc[tp][p][i], acc[tp][p][i + 1]}) * f32x2{S2, S2};
            pos[tp][p][i] = pv[0]; pos[tp][p][i + 1] = pv[1];
            acc[tp][p][i] = tv[0]; acc[tp][p][i + 1] = tv[1];
          }
        }
      }
    }
    // ---- GEMM 2 on the same activations: acc (= S2 (qg_i - kg_j)) += (W_g1 W_d2)' h'
    chain_gemm_split<C, P, NW, TPW, false, false>(A.wg1, X, acc, wv, lane);
    __syncthreads();   // every wave is done reading h
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(acc[tp][p][4 * g + e] * f2, 0.f, POEM_SPLIT_MAX);
          store_split(X + (32 * p + j) * ROW + ((wv * TPW + tp) * 4 + g) * 32 + h * 8, v);
        }
    __syncthreads();

    // ---- GEMM 3 (operands swapped): a[j][c'] = W_g2 g, lane = channel, registers = neighbours
    chain_gemm_split<C, P, NW, TPW, true, true>(A.wg2, X, acc, wv, lane);
    __syncthreads();   // X is dead: per-wave transpose scratch from here on

    float* scr = reinterpret_cast<float*>(X) + wv * (32 * 33);
    const __amdgpu_buffer_rsrc_t vrs = frag_rsrc(A.v, 0xffffffffu);
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      const int cch = (wv * TPW + tp) * 32 + j;
#pragma unroll
      for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int i = 0; i < 16; ++i) scr[j * 33 + mfma_row(i, h)] = pos[tp][p][i];
        float vg[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#if defined(POEM_VS_LAB) && POEM_VS_LAB == 3
          vg[i] = 0.f;
#else
          vg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vrs, voffs[p * 32 + mfma_row(i, h)] + cch * 4, 0, 0));
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float pt[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pt[i] = scr[mfma_row(i, h) * 33 + j];
        f32x16& a = acc[tp][p];
        float mx = fmaxf(fmaxf(a[0], a[1]), a[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, a[i]), a[i + 1]);
        mx = half_max(fmaxf(mx, a[15]));
        const float nb = -mx * k2;
        const f32x2 k2v = {k2, k2}, nbv = {nb, nb};
        f32x2 sum2 = {0.f, 0.f}, res2 = {0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          f32x2 e = __builtin_elementwise_fma(f32x2{a[i], a[i + 1]}, k2v, nbv);
          e[0] = __builtin_amdgcn_exp2f(e[0]);
          e[1] = __builtin_amdgcn_exp2f(e[1]);
          const f32x2 val = f32x2{vg[i], vg[i + 1]} + f32x2{pt[i], pt[i + 1]};
          sum2 += e;
          res2 = __builtin_elementwise_fma(e, val, res2);
        }
        const float sum = half_sum(sum2[0] + sum2[1]);
        const float res = half_sum(res2[0] + res2[1]);
        if (h == 0 && i0 + p < A.Q) A.out[((size_t)b * A.Q + i0 + p) * C + cch] = res * __builtin_amdgcn_rcpf(sum);
      }
    }
  }
}

template <int C, int P, int NW, int MINW>
static hipError_t launch_vs(const VecAttnSplitArgs& a, hipStream_t s) {
  const int groups = (a.Q + P - 1) / P;
  const size_t lds = (size_t)32 * P * (C * 4 + 16) + P * 32 * 3 * 4 + 2 * P * 32 * 4 + (size_t)P * C * 4;
  auto kern = vecattn_split_kernel<C, P, NW, MINW>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), lds, optin); e != hipSuccess) return e;
  // Persistent: as many blocks as the chip holds, items dealt round-robin (they all cost the same).  The compiler hoists
  // ~20 per-lane address values out of the item loop and spills them; with one item per block that was 280 MB of scratch
  // writes per launch (PMC WRITE_SIZE 308 MB against 26 MB of results).
  int slots = 0;
  {
    slots = poem_device_cus() * ((size_t)2 * lds <= 160 * 1024 ? 2 : 1);
#ifdef POEM_LAB
    if (const char* e = getenv("POEM_VS_PERSIST")) if (atoi(e) == 0) slots = 1 << 30;
#endif
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long long>((long long)a.B * groups, slots)), dim3(NW * 64), lds, s, a);
  return hipGetLastError();
}

// C >= 128 only (the caller keeps the exact kernel for smaller widths)
extern "C" hipError_t poem_launch_vector_attention_split(const float* query_xyz, const float* src_xyz, const float* anchor_xyz,
                                                         const int* idx, int shared_idx, const float* q, const float* k,
                                                         const float* v, int nsrc, const float* wd1, const float* bd1,
                                                         const void* wd2, const float* bd2, const void* wg1, const void* wg2,
                                                         const float* scales, float* out, int B, int Q, int C, int ldq,
                                                         int ldk, int ldv, hipStream_t s) {
  VecAttnSplitArgs a{query_xyz, src_xyz, anchor_xyz, idx, shared_idx, q, k, v, nsrc, wd1, bd1, wd2, bd2, wg1, wg2, scales,
                     out, B, Q, ldq, ldk, ldv};
  switch (C) {
    case 128: return launch_vs<128, 4, 4, 2>(a, s);
    case 256: {
#ifdef POEM_LAB
      static const int cfg = getenv("POEM_VS_CFG") ? atoi(getenv("POEM_VS_CFG")) : 0;      // lab A/B of the block shape
#else
      constexpr int cfg = 0;
#endif
      if (cfg == 1) return launch_vs<256, 4, 8, 2>(a, s);     // one 8-wave block per CU, 4 queries share a weight stream
      if (cfg == 2) return launch_vs<256, 2, 8, 4>(a, s);     // two 8-wave blocks per CU (4 waves per SIMD)
      return launch_vs<256, 2, 4, 2>(a, s);
    }
    case 512: return launch_vs<512, 1, 4, 2>(a, s);
    case 1024: return launch_vs<1024, 1, 8, 2>(a, s);
    default: return hipErrorInvalidValue;
  }
}

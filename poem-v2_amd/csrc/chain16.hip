// Row-tile chains on v_mfma_f32_16x16x4_f32: the kinds, arguments and arithmetic of chain.hip (see there for what a chain
// is) with row tiles of 1..4 units of 16 rows instead of 32 / 64 rows -- so that
//   * a CU's share of the M = B * 799 rows quantises in 16-row steps: at the headline batch (25568 rows = 1598 units over 256
//     CUs = 6.24 units per CU) the busiest CU runs 7 units = 112 rows instead of two 64-row tiles = 128;
//   * a small batch spreads over more CUs (M = 1598 at the reference's evaluation batch of 2: 100 units).
// Every result element is the SAME fma chain as in chain.hip: the 16x16x4 instruction sums its four k-slots in order, and
// the slots are given the channels (8kc, 8kc+4, 8kc+1, 8kc+5) then (8kc+2, 8kc+6, 8kc+3, 8kc+7) -- the order in which two
// 32x32x2 steps of chain.hip's chunk visit them (half-wave 0 holds k, half-wave 1 holds k+4 of the fragment image).  The
// LayerNorm sums are taken in chain.hip's order too (a running sum that alternates between the half-waves), so a row's result
// does not depend on which kernel or which tile height processed it: bit-identical, tested.
//
// Layout.  X[channel][row] in LDS, row stride 16 MAXRU + 4 = 68 (== 4 mod 8: the two channels a 32-lane group reads in one ds_read_b32 are
// 4 apart -> 16 banks apart, conflict-free; same for the write-back).  Weight fragments: the 32-row fragment images of
// common.h; lane (i = lane & 15, g = lane >> 4) of 16-channel tile t loads the float4 of image lane 16 (t & 1) + i + 32 (g & 1)
// and takes components (x, z) for g < 2, (y, w) for g >= 2.  MFMA result: lane (g, j) holds channels 4g..4g+3 of the tile for
// row j -- float4 stores / residual loads per lane.
//
// Schedule (static).  Workgroups are dispatched breadth-first (tools/lab/census_lab: blocks b and b + 256 share a CU), so
// block (layer, c) = layer * ncu + c runs on CU c: CU c is given n_c = U / ncu (+1) consecutive units, cut into
// ceil(n_c / 4) nearly equal tiles, one per block (layer) -- co-resident where the LDS holds two tiles, else one after the other.
#include "common.h"
#include <algorithm>
#include <type_traits>

#include "chain.h"

#ifndef POEM_C16_RING_MAX
#define POEM_C16_RING_MAX 4      // RU * T16 up to which a tile runs the cross-phase weight ring on native images (gemm16_ring)
#endif
#ifndef POEM_C16_DEEP
#define POEM_C16_DEEP 8
#endif
#ifndef POEM_C16_NATIVE_ALL
#define POEM_C16_NATIVE_ALL 1    // tall tiles (gemm16) read the native 16x16x4 weight images too (round 6)
#endif
#ifdef POEM_C16_STAMPS   // tools/lab/c16_lab only: 100 MHz ticks at the phase boundaries of block 0's wave 0
__device__ long long c16_stamps[64];
__device__ long long c16_blocks[1024 * 4];      // per block: start, end (100 MHz ticks), units, XCC id
// per-wave trace of the two blocks that share CU 0 (blocks 0 and 256): s_memtime (shader clock) of every wave at every stamp
__device__ long long c16_wtrace[2 * 8 * 40];
__device__ unsigned c16_whw[2 * 8];
#define C16_STAMP(k) do { if ((blockIdx.x == 0 || blockIdx.x == 256) && threadIdx.x == 0) c16_stamps[(k) + (blockIdx.x ? 32 : 0)] = wall_clock64(); \
    if ((blockIdx.x == 0 || blockIdx.x == 256) && (threadIdx.x & 63) == 0 && (k) < 40) c16_wtrace[((blockIdx.x ? 8 : 0) + (threadIdx.x >> 6)) * 40 + (k)] = clock64(); \
    if (threadIdx.x == 0 && blockIdx.x < 1024) { if ((k) == 0) c16_blocks[blockIdx.x * 4] = wall_clock64(); else c16_blocks[blockIdx.x * 4 + 1] = wall_clock64(); } } while (0)
#else
#define C16_STAMP(k) do { } while (0)
#endif

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// value of the lower / upper half-wave's lanes in both halves (one v_permlane32_swap)
__device__ __forceinline__ float from_lower(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]);
}
__device__ __forceinline__ float from_upper(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[1]);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 frag_load2(__amdgpu_buffer_rsrc_t rs, int lane_off_bytes, int scalar_off_bytes) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off_bytes, scalar_off_bytes, 0);
  return float2{__uint_as_float(v[0]), __uint_as_float(v[1])};
}
template <bool NAT>
__device__ __forceinline__ auto frag_load_nat(__amdgpu_buffer_rsrc_t rs, int lane_off_bytes, int scalar_off_bytes) {
  if constexpr (NAT) return frag_load2(rs, lane_off_bytes, scalar_off_bytes);
  else return frag_load(rs, lane_off_bytes, scalar_off_bytes);
}
#ifndef POEM_C16_MMA_PRIO
#define POEM_C16_MMA_PRIO 0      // (lab switch: measured in the step -0.2 % with two co-resident tiles per CU; it pays only for one block per CU)
#endif
#if POEM_C16_MMA_PRIO
#define C16_PRIO_UP __builtin_amdgcn_s_setprio(1);
#define C16_PRIO_DOWN __builtin_amdgcn_s_setprio(0);
#else
#define C16_PRIO_UP
#define C16_PRIO_DOWN
#endif

// D[c'][row] (+)= sum_k W[c'][k0 + k] X[k][row] for this wave's T16 channel tiles and RU row units.  wbase: byte offset of
// the wave's first 32-row tile at the contraction's start; tile_stride: bytes between consecutive 32-row tiles.
// NAT (round 6): the weights' NATIVE 16x16x4 images (see frag_load2 below) for the tall tiles too -- one 8-byte load per lane
// and 16-channel tile instead of a 16-byte load of which half is used, and no selects: tools/lab/c16_issue_lab measures the four
// v_cndmask of a chunk (+ the s_nop the VALU -> MFMA hazard adds) at 8 % of the loop next to its 16 MFMAs, more than the eight LDS
// reads (3 %) or the two fragment loads (1 %).  PRIO: the MFMA burst of a chunk at priority 1 -- the waves of the other
// co-resident tile then issue their VALU / LDS / memory instructions in the gaps between bursts instead of inside them
// (same lab: 0.88 -> 0.93 of the pipe with four waves per SIMD).
template <int KCH, int XSP16, int RU, int T16, bool INIT0, bool NAT = false>
__device__ __forceinline__ void gemm16(const __amdgpu_buffer_rsrc_t wrs, int wbase, int tile_stride, const float* __restrict__ X,
                                       f32x4 (&acc)[T16][RU], int lane) {
  static_assert(KCH % 4 == 0 && T16 % 2 == 0, "shape");
  const int j = lane & 15, g = lane >> 4;
  const bool hi = g >= 2;
  const int loff0 = NAT ? lane * 8 : (j + 32 * (g & 1)) * 16, loff1 = loff0 + (NAT ? 512 : 256);
  const float* xc = X + (4 * (g & 1) + (g >> 1)) * XSP16 + j;
  if (INIT0) {
#pragma unroll
    for (int t = 0; t < T16; ++t)
#pragma unroll
      for (int u = 0; u < RU; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float xa[2][RU], xb[2][RU];
#define C16_READX(XR, KCI)                                                                                    \
  {                                                                                                           \
    const int kq_ = min((KCI), KCH - 1);                                                                      \
    _Pragma("unroll") for (int u = 0; u < RU; ++u) {                                                          \
      XR[0][u] = xc[(kq_ * 8) * XSP16 + 16 * u];                                                              \
      XR[1][u] = xc[(kq_ * 8 + 2) * XSP16 + 16 * u];                                                          \
    }                                                                                                         \
  }
#define C16_MMA(A, XR)                                                                                        \
  {                                                                                                           \
    float w1_[T16], w2_[T16];                                                                                 \
    _Pragma("unroll") for (int t = 0; t < T16; ++t) {                                                         \
      if constexpr (NAT) { w1_[t] = A[t].x; w2_[t] = A[t].y; }                                                \
      else { w1_[t] = hi ? A[t].y : A[t].x; w2_[t] = hi ? A[t].w : A[t].z; }                                  \
    }                                                                                                         \
    C16_PRIO_UP                                                                                               \
    _Pragma("unroll") for (int u = 0; u < RU; ++u)                                                            \
      _Pragma("unroll") for (int t = 0; t < T16; ++t) acc[t][u] = mfma16(w1_[t], XR[0][u], acc[t][u]);        \
    _Pragma("unroll") for (int u = 0; u < RU; ++u)                                                            \
      _Pragma("unroll") for (int t = 0; t < T16; ++t) acc[t][u] = mfma16(w2_[t], XR[1][u], acc[t][u]);        \
    C16_PRIO_DOWN                                                                                             \
  }
  if constexpr (RU * T16 <= 2 && KCH % 8 == 0) {
    // small tiles (one or two units of 16 rows: a chunk is 2 RU T16 MFMAs of 32 cycles -- 128 to 256 cycles): weight fragments
    // seven chunks ahead, as lds_gemm.h does for its small tiles and for the same reason (a small batch's tile is a latency chain)
    float4 ar[8][T16];
#define C16_LOADR(D, KCI)                                                                                     \
    {                                                                                                         \
      const int kq_ = min((KCI), KCH - 1);                                                                    \
      _Pragma("unroll") for (int t = 0; t < T16; ++t)                                                         \
          ar[D][t] = frag_load(wrs, (t & 1) ? loff1 : loff0, wbase + (t >> 1) * tile_stride + kq_ * 1024);    \
    }
#pragma unroll
    for (int d = 0; d < 7; ++d) C16_LOADR(d, d)
    C16_READX(xa, 0)
#pragma unroll 1
    for (int kc = 0; kc < KCH; kc += 8) {
      C16_LOADR(7, kc + 7) C16_READX(xb, kc + 1) C16_MMA(ar[0], xa)
      C16_LOADR(0, kc + 8) C16_READX(xa, kc + 2) C16_MMA(ar[1], xb)
      C16_LOADR(1, kc + 9) C16_READX(xb, kc + 3) C16_MMA(ar[2], xa)
      C16_LOADR(2, kc + 10) C16_READX(xa, kc + 4) C16_MMA(ar[3], xb)
      C16_LOADR(3, kc + 11) C16_READX(xb, kc + 5) C16_MMA(ar[4], xa)
      C16_LOADR(4, kc + 12) C16_READX(xa, kc + 6) C16_MMA(ar[5], xb)
      C16_LOADR(5, kc + 13) C16_READX(xb, kc + 7) C16_MMA(ar[6], xa)
      C16_LOADR(6, kc + 14) C16_READX(xa, kc + 8) C16_MMA(ar[7], xb)
    }
#undef C16_LOADR
  } else {
  std::conditional_t<NAT, float2, float4> a0[T16], a1[T16], a2[T16], a3[T16];
#define C16_LOADW(A, KCI)                                                                                     \
  {                                                                                                           \
    const int kq_ = min((KCI), KCH - 1);                                                                      \
    _Pragma("unroll") for (int t = 0; t < T16; ++t) {                                                         \
      if constexpr (NAT) A[t] = frag_load_nat<NAT>(wrs, (t & 1) ? loff1 : loff0, wbase + (t >> 1) * tile_stride + kq_ * 1024); \
      else A[t] = frag_load_nat<NAT>(wrs, (t & 1) ? loff1 : loff0, wbase + (t >> 1) * tile_stride + kq_ * 1024); \
    }                                                                                                         \
  }
  C16_LOADW(a0, 0)
  C16_READX(xa, 0)
  C16_LOADW(a1, 1)
  C16_LOADW(a2, 2)
#pragma unroll 1
  for (int kc = 0; kc < KCH; kc += 4) {
    C16_LOADW(a3, kc + 3) C16_READX(xb, kc + 1) C16_MMA(a0, xa)
    C16_LOADW(a0, kc + 4) C16_READX(xa, kc + 2) C16_MMA(a1, xb)
    C16_LOADW(a1, kc + 5) C16_READX(xb, kc + 3) C16_MMA(a2, xa)
    C16_LOADW(a2, kc + 6) C16_READX(xa, kc + 4) C16_MMA(a3, xb)
  }
  }
#undef C16_LOADW
#undef C16_READX
#undef C16_MMA
}

// ---- one-unit tiles (RU * T16 <= 2), round 5: the weight ring runs ACROSS the GEMM phases of a chain.  A 16-row tile is a
// latency chain -- 3.4 us of MFMAs per C x C phase against 5.2-6.2 us measured (tools/lab/c16_lab) -- and every phase began by
// requesting its first seven weight chunks and waiting for them, and ended by re-requesting its last chunk seven times (the
// clamped prefetch).  Here the last seven ring slots of a phase fetch the NEXT phase's first chunks instead, behind the
// epilogue (bias / activation / LayerNorm / barriers / LDS write-back) of this one, and the first phase's chunks are requested
// before the tile's fill.  Same MFMAs on the same operands in the same order: bit-identical.
struct WSrc {
  __amdgpu_buffer_rsrc_t rs;
  int base, stride;      // byte offset of the wave's first 32-row tile at the contraction's start; bytes between 32-row tiles
  bool valid;
};

// A lane needs TWO floats of a fragment image's float4 -- (x, z) in lane groups 0 and 1, (y, w) in groups 2 and 3 -- and the
// texture addresser is paid per lane: as one 16-byte load per lane (what gemm16 does for its taller tiles, where a fragment feeds
// RU units) a 16-row tile's weight stream costs the addresser as long as its MFMAs take (8 waves x 32 chunks x 2 x 16 cycles =
// 3.4 us per C x C phase: measured 4.8-5.0 us per phase against 2.8-3.4 without the loads, tools/lab/c16_lab; two strided 4-byte
// loads per lane: 9.6 us).  The NATIVE image holds each lane's two floats side by side -- per 1 KiB block of the 32-row image
//   N[half][lane = 16 g + i] = float2( F[16 half + i + 32 (g & 1)].comp[g >> 1], .comp[(g >> 1) + 2] ),   F = the block's 64 float4
// -- so a 16-channel tile's chunk is one fully coalesced 8-byte load per lane (512 bytes): half the addresser time, half the ring
// registers, no selects: 3.4-4.2 us per phase.  Built once per weight at poem_create (native16_kernel), same offsets in a mirror.
#ifdef POEM_C16_NOLOADS      // tools/lab only: the loop without its weight stream (what the MFMA / LDS side alone costs)
#define C16R_LOAD(SRC, D, KQ) if ((KQ) < 0) { _Pragma("unroll") for (int t = 0; t < T16; ++t) ar[D][t] = frag_load2((SRC).rs, loff0, (SRC).base); }
#else
#define C16R_LOAD(SRC, D, KQ)                                                                                 \
  _Pragma("unroll") for (int t = 0; t < T16; ++t)                                                             \
    ar[D][t] = frag_load2((SRC).rs, (t & 1) ? loff1 : loff0, (SRC).base + (t >> 1) * (SRC).stride + (KQ) * 1024);
#endif

template <int T16, int DEPTH>
__device__ __forceinline__ void ring_preload(float2 (&ar)[DEPTH][T16], const WSrc& src, int lane) {
  const int loff0 = lane * 8, loff1 = loff0 + 512;      // native 16x16x4 image (see below)
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) { C16R_LOAD(src, d, d) }
}

// The ring holds chunks 0..DEPTH-2 of `cur` in slots 0..DEPTH-2 on entry, and those of `nxt` on exit (when nxt.valid).
template <int KCH, int XSP16, int RU, int T16, bool INIT0, int DEPTH>
__device__ __forceinline__ void gemm16_ring(const WSrc& cur, const WSrc& nxt, const float* __restrict__ X, f32x4 (&acc)[T16][RU],
                                            float2 (&ar)[DEPTH][T16], int lane) {
  static_assert(KCH % DEPTH == 0 && KCH >= 2 * DEPTH && DEPTH % 2 == 0 && T16 % 2 == 0, "shape");
  const int j = lane & 15, g = lane >> 4;
  const int loff0 = lane * 8, loff1 = loff0 + 512;      // native 16x16x4 image (see below)
  const float* xc = X + (4 * (g & 1) + (g >> 1)) * XSP16 + j;
  if (INIT0) {
#pragma unroll
    for (int t = 0; t < T16; ++t)
#pragma unroll
      for (int u = 0; u < RU; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float xr[2][2][RU];      // [chunk parity][k pair][unit]
#define C16R_READX(P, KCI)                                                                                    \
  {                                                                                                           \
    const int kq_ = min((KCI), KCH - 1);                                                                      \
    _Pragma("unroll") for (int u = 0; u < RU; ++u) {                                                          \
      xr[P][0][u] = xc[(kq_ * 8) * XSP16 + 16 * u];                                                           \
      xr[P][1][u] = xc[(kq_ * 8 + 2) * XSP16 + 16 * u];                                                       \
    }                                                                                                         \
  }
#define C16R_MMA(A, P)                                                                                        \
  {                                                                                                           \
    _Pragma("unroll") for (int u = 0; u < RU; ++u)                                                            \
      _Pragma("unroll") for (int t = 0; t < T16; ++t) acc[t][u] = mfma16(A[t].x, xr[P][0][u], acc[t][u]);     \
    _Pragma("unroll") for (int u = 0; u < RU; ++u)                                                            \
      _Pragma("unroll") for (int t = 0; t < T16; ++t) acc[t][u] = mfma16(A[t].y, xr[P][1][u], acc[t][u]);     \
  }
  C16R_READX(0, 0)
#pragma unroll 1
  for (int kc = 0; kc < KCH - DEPTH; kc += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {      // chunk kc + d sits in slot d; the slot freed by the previous step takes chunk kc + d + DEPTH - 1
      C16R_LOAD(cur, (d + DEPTH - 1) % DEPTH, kc + d + DEPTH - 1)
      C16R_READX((d + 1) & 1, kc + d + 1)
      C16R_MMA(ar[d], d & 1)
    }
  }
  // last group of DEPTH chunks: its prefetches are the next phase's first DEPTH - 1 chunks (or nothing)
  {
    constexpr int kc = KCH - DEPTH;
    const bool nx = nxt.valid;      // wave-uniform
    C16R_LOAD(cur, DEPTH - 1, KCH - 1)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (d > 0 && nx) { C16R_LOAD(nxt, d - 1, d - 1) }
      if (d + 1 < DEPTH) C16R_READX((d + 1) & 1, kc + d + 1)
      C16R_MMA(ar[d], d & 1)
    }
  }
#undef C16R_READX
#undef C16R_MMA
}

}  // namespace

// MAXRU: largest tile in units (4 = 64 rows; 2 at C = 512, where two activation tiles of kind D2 must fit 160 KB of LDS)
template <int C, int NW, int KIND, int MAXRU>
__global__ __launch_bounds__(NW * 64, (KIND == 3 || (C == 512 && MAXRU == 4)) ? NW / 4 : NW / 2) void chain16_kernel(ChainArgs A, int ncu, int layers) {
  constexpr int XSP = 16 * MAXRU + 4, XROWS = 16 * MAXRU, T16 = C / 16 / NW, KCH = C / 8, NT = NW * 64, NTILE = C / 32;
  static_assert((C / 16) % NW == 0 && T16 % 2 == 0, "waves must divide the 32-channel tiles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X0 = smem;                       // C * XSP
  float* X1 = X0 + C * XSP;               // C * XSP (kind D2 only)
  float* red = KIND == 3 ? X1 + C * XSP : X0 + C * XSP;   // 2 x NW * XROWS partial row sums (one buffer per LayerNorm pass)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, g = lane >> 4;
  const int cw0 = wv * T16 * 16;          // this wave's first channel within a C-wide pass
  const int tile0 = wv * (T16 / 2);       // ... its first 32-row tile of a packed image
  const unsigned CC4 = (unsigned)(C * C * 4);

  // ---- this block's tile: CU c's share of the units, cut into `layers` nearly equal tiles (one per block: a tile loop around
  // the four tile-height variants makes the compiler keep every variant's loop invariants live at once -- 200+ spilled VGPRs)
  const int U = (A.M + 15) >> 4;
  const int c = blockIdx.x % ncu, layer = blockIdx.x / ncu;
  const int ubase = U / ncu, urem = U % ncu;
  const int cu_lo = c * ubase + min(c, urem), cu_n = ubase + (c < urem ? 1 : 0);
  const int tb = cu_n / layers, te = cu_n % layers;
  const int ru = tb + (layer < te ? 1 : 0);
  if (ru <= 0) return;
  const int tile_row0 = (cu_lo + layer * tb + min(layer, te)) * 16;
#ifdef POEM_C16_SKEW      // tools/lab only: the second tile of a CU starts this many 100 MHz ticks late (do co-resident tiles in lockstep lose time?)
  if (layer == 1) { const long long t0_ = wall_clock64(); while (wall_clock64() - t0_ < POEM_C16_SKEW) __builtin_amdgcn_s_sleep(8); }
#endif
  C16_STAMP(0);
#ifdef POEM_C16_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    c16_blocks[blockIdx.x * 4 + 2] = (long long)(((xcc & 0xf) << 12) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xf));
    c16_blocks[blockIdx.x * 4 + 3] = ru;
  }
  if ((blockIdx.x == 0 || blockIdx.x == 256) && (threadIdx.x & 63) == 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    c16_whw[(blockIdx.x ? 8 : 0) + (threadIdx.x >> 6)] = hwid;
  }
#endif

  auto run_tile = [&](auto ru_tag, const int row0) {
    constexpr int RU = decltype(ru_tag)::value;
    constexpr int XS = 16 * RU;
    // one-unit tiles: the weight ring across the phases (gemm16_ring); taller tiles: gemm16's own prologue per phase
    constexpr bool RING = RU * T16 <= POEM_C16_RING_MAX && T16 == 2 && KCH % 8 == 0 && KCH >= 16;
    constexpr bool NAT = POEM_C16_NATIVE_ALL && (C == 128 || C == 256);      // (the widths whose handle carries the native mirror)
    constexpr int DEPTH = RING ? ((KIND == 3 && KCH >= 32) ? POEM_C16_DEEP : 8) : 1;      // D2 runs two waves per SIMD: 256 registers
    float2 ring[DEPTH][T16];
    const WSrc none{frag_rsrc(A.w1, 0u), 0, 0, false};
    // (RING reads the weights' native 16x16x4 images: the same offsets, A.native_delta bytes away -- native16_kernel below)
    auto img = [&](const float4* w) { return reinterpret_cast<const char*>(w) + ((RING || NAT) ? A.native_delta : 0); };
    auto src_w1 = [&]() { return WSrc{frag_rsrc(img(A.w1), CC4), __builtin_amdgcn_readfirstlane(tile0 * KCH * 1024), KCH * 1024, true}; };
    auto src_w2 = [&](int pass) {
      return WSrc{frag_rsrc(img(A.w2), (unsigned)A.n2 * CC4), __builtin_amdgcn_readfirstlane((pass * NTILE + tile0) * KCH * 1024), KCH * 1024, pass < A.n2};
    };
    auto src_f4 = [&](int slab) {      // 0 = reg_branch.0, 1 + s = slab s of intermediate.dense
      return WSrc{frag_rsrc(img(A.wf4), 5u * CC4), __builtin_amdgcn_readfirstlane((slab * NTILE + tile0) * KCH * 1024), KCH * 1024, true};
    };
    auto src_wout = [&](int slab) {
      return WSrc{frag_rsrc(img(A.wout), 4u * CC4), __builtin_amdgcn_readfirstlane(tile0 * 4 * KCH * 1024 + slab * KCH * 1024), 4 * KCH * 1024, true};
    };
    // one GEMM phase: D (+)= W_cur X; RING: the ring holds cur's first chunks and leaves with nxt's
    auto phase = [&](auto init_tag, const WSrc& cur, const WSrc& nxt, const float* X, f32x4 (&acc)[T16][RU]) {
      constexpr bool INIT0 = decltype(init_tag)::value;
#ifdef POEM_C16_PRIO
      // Two tiles share a CU (kinds A, C, D1) and the arbiter serves the oldest wave: next to the other tile's MFMA stream the
      // VALU / LDS / memory instructions of an epilogue (bias, residual, LayerNorm, write-back) wait for a free slot one by one --
      // the younger tile's LayerNorm took 35 us instead of 2 (tools/lab/c16_lab).  Everything outside the GEMM loops runs at
      // priority 3, the loops at 0: an epilogue is a few hundred instructions, the MFMAs fill every slot they leave.
      if (KIND != 3) __builtin_amdgcn_s_setprio(0);
#endif
      if constexpr (RING) gemm16_ring<KCH, XSP, RU, T16, INIT0, DEPTH>(cur, nxt, X, acc, ring, lane);
      else gemm16<KCH, XSP, RU, T16, INIT0, NAT>(cur.rs, cur.base, cur.stride, X, acc, lane);
#ifdef POEM_C16_PRIO
      if (KIND != 3) __builtin_amdgcn_s_setprio(3);
#endif
    };
#ifdef POEM_C16_PRIO
    if (KIND != 3) __builtin_amdgcn_s_setprio(3);
#endif
    using init_t = std::integral_constant<bool, true>;
    using accum_t = std::integral_constant<bool, false>;
    // the first phase's weights are requested before the tile's fill (behind it where the fill combines the attention's
    // partials: that code needs the registers)
    const bool fill_combines = KIND == 0 && A.x == nullptr;
    if constexpr (RING) { if (KIND == 3) ring_preload<T16, DEPTH>(ring, src_f4(1), lane); }
    // channel of acc[t][u][r]: cw0 + 16 t + 4 g + r (+ pass * C); row: row0 + 16 u + j
    auto add_bias = [&](f32x4 (&acc)[T16][RU], const float* bias, int act) {
#pragma unroll
      for (int t = 0; t < T16; ++t) {
        const float4 bb = *reinterpret_cast<const float4*>(bias + cw0 + 16 * t + 4 * g);
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[t][u][e] + (&bb.x)[e];
            if (act == 1) v = relu_nan(v);
            if (act == 2) v = gelu_erf(v);
            acc[t][u][e] = v;
          }
      }
    };
    auto add_rows = [&](f32x4 (&acc)[T16][RU], const float* R, int ld, int mod) {
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        int row = min(row0 + 16 * u + j, A.M - 1);
        if (mod > 0) row %= mod;
        const float* rp = R + (size_t)row * ld + cw0 + 4 * g;
#pragma unroll
        for (int t = 0; t < T16; ++t) {
          const float4 rr = *reinterpret_cast<const float4*>(rp + 16 * t);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][u][e] += (&rr.x)[e];
        }
      }
    };
    auto to_lds = [&](const f32x4 (&acc)[T16][RU], float* X) {
#pragma unroll
      for (int t = 0; t < T16; ++t)
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) X[(cw0 + 16 * t + 4 * g + e) * XSP + 16 * u + j] = acc[t][u][e];
    };
    auto to_global = [&](const f32x4 (&acc)[T16][RU], float* Y, int ld, int col0) {
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int row = row0 + 16 * u + j;
        if (row >= A.M) continue;
        float* yp = Y + (size_t)row * ld + col0 + cw0 + 4 * g;
#pragma unroll
        for (int t = 0; t < T16; ++t)
          *reinterpret_cast<float4*>(yp + 16 * t) = make_float4(acc[t][u][0], acc[t][u][1], acc[t][u][2], acc[t][u][3]);
      }
    };
    // LayerNorm over the C channels of every row, in place.  chain.hip sums, per wave and half-wave h, the 16 registers of
    // each 32-channel tile in order -- channels 4h.., 8+4h.., 16+4h.., 24+4h.. -- then adds the two halves, then the waves
    // in order.  Here those channel groups sit in lane groups (tile 2m, g = h), (2m, g = 2 + h), (2m+1, h), (2m+1, 2 + h):
    // the running sum hops between the half-waves (lanes l and l + 32 share row and h) and ends in the upper one.
    auto layer_norm = [&](f32x4 (&acc)[T16][RU], const float* gamma, const float* beta, float eps) {
      float mean[RU], rstd[RU];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        float s[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          float p = 0.f;
#pragma unroll
          for (int t = 0; t < T16; ++t) {
#pragma unroll
            for (int stage = 0; stage < 2; ++stage) {       // 0: lanes g < 2 hold the running sum, 1: lanes g >= 2
              if (t > 0 || stage > 0) p = stage == 0 ? from_upper(p) : from_lower(p);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float d = pass == 0 ? acc[t][u][e] : acc[t][u][e] - mean[u];
                p = pass == 0 ? p + d : __builtin_fmaf(d, d, p);
              }
            }
          }
          s[u] = p + __shfl_xor(p, 16, 64);                // lanes g = 2, 3: (half-wave 0's sum) + (half-wave 1's)
        }
        C16_STAMP(30 + 2 * pass);
        // (one buffer per pass, one LayerNorm per tile: nobody can still be reading what is written here -- round 6: the two
        //  "previous readers are done" barriers are gone; a barrier next to the other tile's MFMA stream costs microseconds)
        float* rp = red + pass * (NW * XROWS);
        if (g == 2)
#pragma unroll
          for (int u = 0; u < RU; ++u) rp[wv * XROWS + 16 * u + j] = s[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) t += rp[w * XROWS + 16 * u + j];
          if (pass == 0) mean[u] = t / (float)C;
          else rstd[u] = 1.0f / sqrtf(t / (float)C + eps);
        }
        C16_STAMP(31 + 2 * pass);
      }
#pragma unroll
      for (int t = 0; t < T16; ++t) {
        const float4 gg = *reinterpret_cast<const float4*>(gamma + cw0 + 16 * t + 4 * g);
        const float4 bb = *reinterpret_cast<const float4*>(beta + cw0 + 16 * t + 4 * g);
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][u][e] = __builtin_fmaf((acc[t][u][e] - mean[u]) * rstd[u], (&gg.x)[e], (&bb.x)[e]);
      }
    };
    // trailing wide Linear: n C-wide passes over X, results straight to global
    auto wide_linear = [&](const float* bias, int n, const float* X, float* Y, int ld) {      // (A.w2; RING: pass 0 is in the ring)
      for (int pass = 0; pass < n; ++pass) {
        f32x4 acc[T16][RU];
        phase(init_t{}, src_w2(pass), src_w2(pass + 1), X, acc);
        C16_STAMP(22 + 2 * pass);
        add_bias(acc, bias + pass * C, 0);
        to_global(acc, Y, ld, pass * C);
        C16_STAMP(23 + 2 * pass);
      }
    };

    if (fill_combines) {
      // ---- fill X0 with the cross attention's context, combined from the split-key partials (chain.hip; per (row, channel)
      // the arithmetic of attn_combine_kernel, in its order).  Threads are laid out over XM >= XS rows so that a thread owns
      // one row and G float4 groups per head.
      constexpr int DH4 = C / 4, HEADS = 4, XM = XS <= 32 ? (NT / 32 <= DH4 / HEADS ? 32 : 64) : 64;
      constexpr int RS = NT / XM, G = DH4 / (HEADS * RS), DT = C / HEADS / 32;
      static_assert(NT % XM == 0 && G >= 1 && DH4 % (HEADS * RS) == 0 && DT >= 1, "shape");
      const int rr = tid % XM, cgw = tid / XM;
      if (rr < XS) {
        const int i = min(row0 + rr, A.M - 1);
        const int b = i / A.pc_nq, q = i % A.pc_nq, qt = q >> 5, r = q & 31;
        const int nqt = (A.pc_nq + 31) >> 5, nch = A.pc_chunks;
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
#pragma unroll
        for (int hd = 0; hd < HEADS; ++hd) {
          float4 p[G][4];
#pragma unroll
          for (int gi = 0; gi < G; ++gi) {
            const int cg4 = hd * (DH4 / HEADS) + cgw + RS * gi;
            const int d = (cg4 >> 3) % DT, gq = (cg4 & 7) >> 1, hh = cg4 & 1;
#pragma unroll
            for (int s = 0; s < 4; ++s)
              p[gi][s] = A.part_o[((((size_t)(b * HEADS + hd) * nch + min(s, nch - 1)) * nqt + qt) * (DT * 4) + d * 4 + gq) * 64 + r + 32 * hh];
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
      }
      if constexpr (RING) ring_preload<T16, DEPTH>(ring, src_w1(), lane);
    } else {
      if constexpr (RING) { if (KIND != 3) ring_preload<T16, DEPTH>(ring, src_w1(), lane); }
      // ---- fill X0 with the input tile, transposed: consecutive threads read consecutive channels of one row
      static_assert(NT % C == 0 || C % NT == 0, "threads and channels must nest");
      constexpr int RSTEP = NT >= C ? NT / C : 1, CSTEP = NT >= C ? C : NT;
      const int c0 = tid % CSTEP, r0 = tid / CSTEP;
#pragma unroll 4
      for (int r = r0; r < XS; r += RSTEP) {
        const float* xr = A.x + (size_t)min(row0 + r, A.M - 1) * A.ldx;
#pragma unroll
        for (int cc = c0; cc < C; cc += CSTEP) X0[cc * XSP + r] = xr[cc];
      }
    }
    __syncthreads();
    C16_STAMP(1);
    if (KIND != 3) {
      // ---- stage 1: first Linear + bias + residual [+ LayerNorm] -> y1 and back into X0
      f32x4 acc[T16][RU];
      phase(init_t{}, src_w1(), KIND == 2 ? src_f4(0) : src_w2(0), X0, acc);
      C16_STAMP(2);
      add_bias(acc, A.b1, 0);
      add_rows(acc, A.res, A.ldres, A.res_mod);
      C16_STAMP(3);
      if (KIND == 0) layer_norm(acc, A.ln_g, A.ln_b, A.eps);
      C16_STAMP(4);
      to_global(acc, A.y1, A.ldy1, 0);
      __syncthreads();                          // every wave is done reading the input tile
      to_lds(acc, X0);
      __syncthreads();
      C16_STAMP(5);
      if (KIND != 2) {
        if (A.n2 > 0) wide_linear(A.b2, A.n2, X0, A.y2, A.ldy2);
        C16_STAMP(6);
        return;
      }
      // ---- kind D1.  reg_branch: u = relu(f Wreg0^T + b), xyz' = xyz + u Wreg2^T + b
      {
        f32x4 uu[T16][RU];
        phase(init_t{}, src_f4(0), none, X0, uu);
        add_bias(uu, A.bf4, 1);
        __syncthreads();                        // every wave is done reading f
        to_lds(uu, X0);
      }
      __syncthreads();
      // one wave per row, lanes stride the channels: the fma chain and the reduction order of narrow_linear_kernel
      for (int r = wv; r < XS; r += NW) {
        const int row = row0 + r;
        float s3[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          float s = 0.f;
          for (int cc = lane; cc < C; cc += 64) s = fmaf(X0[cc * XSP + r], A.wreg2[n * C + cc], s);
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
          s3[n] = s;
        }
        if (lane < 3 && row < A.M) {
          const float s = lane == 0 ? s3[0] : (lane == 1 ? s3[1] : s3[2]);
          A.xyz_out[(size_t)row * 3 + lane] = A.xyz_in[(size_t)row * 3 + lane] + (s + A.breg2[lane]);
        }
      }
      return;
    }
    // ---- kind D2 (X0 = f): o = sum_s gelu(f Wint_s^T + b_s) Wout[:, sC:(s+1)C]^T, slab by slab in k order
    f32x4 o[T16][RU];
#pragma unroll 1
    for (int sl = 0; sl < 4; ++sl) {
      f32x4 t[T16][RU];
      phase(init_t{}, src_f4(1 + sl), src_wout(sl), X0, t);
      C16_STAMP(2 + 4 * sl);
      add_bias(t, A.bf4 + (1 + sl) * C, 2);
      C16_STAMP(3 + 4 * sl);
      __syncthreads();                        // readers of X1 (the previous slab's contraction)
      to_lds(t, X1);
      __syncthreads();
      C16_STAMP(4 + 4 * sl);
      const WSrc after = sl < 3 ? src_f4(2 + sl) : src_w2(0);      // (src_w2(0).valid = there is a trailing Linear)
      if (sl == 0) phase(init_t{}, src_wout(sl), after, X1, o);
      else phase(accum_t{}, src_wout(sl), after, X1, o);
      C16_STAMP(5 + 4 * sl);
    }
    add_bias(o, A.bout, 0);
#pragma unroll
    for (int t = 0; t < T16; ++t)
#pragma unroll
      for (int u = 0; u < RU; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[t][u][e] += X0[(cw0 + 16 * t + 4 * g + e) * XSP + 16 * u + j];
    C16_STAMP(18);
    layer_norm(o, A.ln2_g, A.ln2_b, A.eps);
    C16_STAMP(19);
    to_global(o, A.y3, A.ldy3, 0);
    if (A.n2 > 0) {
      __syncthreads();                        // every wave has read its residual from X0
      to_lds(o, X0);
      __syncthreads();
      C16_STAMP(20);
      wide_linear(A.b2, A.n2, X0, A.y2, A.ldy2);
    }
    C16_STAMP(21);
  };

  if constexpr (MAXRU == 4) {
    switch (ru) {
      case 1: run_tile(std::integral_constant<int, 1>{}, tile_row0); break;
      case 2: run_tile(std::integral_constant<int, 2>{}, tile_row0); break;
      case 3: run_tile(std::integral_constant<int, 3>{}, tile_row0); break;
      default: run_tile(std::integral_constant<int, 4>{}, tile_row0); break;
    }
  } else {
    if (ru == 1) run_tile(std::integral_constant<int, 1>{}, tile_row0);
    else run_tile(std::integral_constant<int, 2>{}, tile_row0);
  }
}

template <int C, int NW, int KIND, int MAXRU>
static hipError_t launch_chain16_k(const ChainArgs& a, int cus, hipStream_t s) {
  const size_t lds = ((size_t)(KIND == 3 ? 2 : 1) * C * (16 * MAXRU + 4) + (size_t)2 * NW * 16 * MAXRU) * sizeof(float);
  auto kern = chain16_kernel<C, NW, KIND, MAXRU>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), lds, optin); e != hipSuccess) return e;
  const int U = (a.M + 15) / 16;
  const int ncu = std::min(cus, U);
  const int per_cu = (U + ncu - 1) / ncu;                         // units of the busiest CU
  // tiles per CU: two co-resident ones from 3 units on where the LDS holds two (one block's fill / LayerNorm / stores then
  // overlap the other's MFMAs: B = 16, 4 units per CU, 127 us as one tile vs 116 us as two), else as few as fit
  const bool pair = 2 * lds <= 160 * 1024 && per_cu >= 3;
  int layers = std::max(pair ? 2 : 1, (per_cu + MAXRU - 1) / MAXRU);
#ifdef POEM_C16_STAMPS   // tools/lab only: more, smaller tiles per CU (the later ones go to whichever CU frees a slot first)
  if (const char* e = getenv("POEM_C16_LAYERS")) layers = std::max(layers, atoi(e));
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)(ncu * layers)), dim3(NW * 64), lds, s, a, ncu, layers);
  return hipGetLastError();
}

template <int C, int NW, int MAXRU>
static hipError_t launch_chain16_t(const ChainArgs& a, int cus, hipStream_t s) {
  switch (a.kind) {
    case 0: return launch_chain16_k<C, NW, 0, MAXRU>(a, cus, s);
    case 1: return launch_chain16_k<C, NW, 1, MAXRU>(a, cus, s);
    case 2: return launch_chain16_k<C, NW, 2, MAXRU>(a, cus, s);
    case 3: return launch_chain16_k<C, NW, 3, MAXRU>(a, cus, s);
    default: return hipErrorInvalidValue;
  }
}

// ---- native 16x16x4 image of a packed (32-row fragment order) weight: a permutation inside every 1 KiB block (see C16R_LOAD)
__global__ void native16_kernel(const float4* __restrict__ src, float2* __restrict__ dst, size_t n2) {
  const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // output float2
  if (o >= n2) return;
  const size_t blk = o >> 7;
  const int w = (int)(o & 127), half = w >> 6, g = (w >> 4) & 3, i = w & 15;
  const float4 f = src[blk * 64 + 16 * half + i + 32 * (g & 1)];
  dst[o] = (g >> 1) ? float2{f.y, f.w} : float2{f.x, f.z};
}
extern "C" hipError_t poem_launch_native16(const void* packed, void* native, size_t bytes, hipStream_t s) {
  if (bytes % 1024) return hipErrorInvalidValue;
  const size_t n2 = bytes / 8;
  if (!n2) return hipSuccess;
  hipLaunchKernelGGL(native16_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, (const float4*)packed, (float2*)native, n2);
  return hipGetLastError();
}
extern "C" int poem_chain16_wants_native(int C) { return C == 128 || C == 256; }      // the widths whose one-unit tiles run the ring

extern "C" hipError_t poem_launch_chain16(const ChainArgs* a, int C, hipStream_t s) {
  if (poem_chain16_wants_native(C) && a->native_delta == 0) return hipErrorInvalidValue;      // (one-unit tiles read the native images)
  const int cus = poem_device_cus();
  switch (C) {
    case 128: return launch_chain16_t<128, 4, 4>(*a, cus, s);
    case 256: return launch_chain16_t<256, 8, 4>(*a, cus, s);
    case 512:
#ifdef POEM_C16_STAMPS   // tools/lab only: one tile of up to 4 units per CU for kinds A / C / D1 (139 KB of LDS; the weights cross the L2 once per CU)
      if (getenv("POEM_C16_RU512") && a->kind != 3) {
        switch (a->kind) {
          case 0: return launch_chain16_k<512, 8, 0, 4>(*a, cus, s);
          case 1: return launch_chain16_k<512, 8, 1, 4>(*a, cus, s);
          default: return launch_chain16_k<512, 8, 2, 4>(*a, cus, s);
        }
      }
#endif
      return launch_chain16_t<512, 8, 2>(*a, cus, s);
    default: return hipErrorInvalidValue;
  }
}

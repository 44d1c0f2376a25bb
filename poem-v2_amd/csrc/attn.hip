// BERT-style multi-head cross attention of the 799 queries over the 4096 basis points, flash-style, exact fp32 MFMA.
// softmax(Q K^T / sqrt(dh)) V without materialising the (B, heads, Q, S) score tensor.
//
// Keys and values arrive as MFMA *fragment images* (written directly by the projection GEMM's epilogue, gemm.hip
// output modes 1 and 2, or by the repack kernels below for the row-major op-level entry point):
//   K image  KI[(kt * C/8  + kco) * 64 + lane]       = float4( K[32kt + (lane&31)][8kco + 4(lane>>5) + 0..3] )
//   V image  VI[((kt * C/32 + vt) * 4 + g) * 64 + lane] = float4( V[32kt + 8g + 4(lane>>5) + 0..3][32vt + (lane&31)] )
// so every operand of both contractions is one coalesced 1 KiB wave load that lands in exactly the registers the MFMA
// reads -- no LDS, no barriers, every wave is independent:
//   S^T = K . Q^T      A = K fragment (row = key),   B = Q fragment held in registers (lane = query)
//                      -> D[key][query]: lane = query, registers = keys; softmax statistics are lane-local
//   O^T += V^T . P^T   A = V fragment (row = channel), B = P registers used *directly* as the operand:
//                      k-step i consumes key (i&3) + 8(i>>2) + 4*half, exactly the key that register i holds.
// The K fragments of tile t+1 are requested right after the QK^T MFMAs of tile t have consumed the registers, the V
// fragments right after the PV MFMAs: each load has a whole 32-MFMA phase (>= 2048 cycles) to arrive.
//
// fp32 MFMA and the VALU share the SIMD's fp32 lanes on gfx950 (tools/lab/dual_lab, phase_lab: a VALU instruction
// costs its full issue time whether it is interleaved with MFMAs, placed between them, or issued by another wave), so
// the softmax is written for instruction count: raw scores, the scale and log2(e) folded into one packed fma per pair
// (v_pk_fma_f32), v_exp_f32, packed adds for the row sums, v_max3 for the tile maximum, and a *lazy* running maximum
// -- the stabiliser only moves (and O is only rescaled) when a tile exceeds it by more than 2^LAZY_LOG2; any
// stabiliser gives the same softmax, the rescale branch is wave-uniform and rare.  Row sums stay per half-wave until
// the end of a chunk.
//
// Work decomposition: item = (batch, head, key chunk, 32-query tile); a chunk is a fixed number of key tiles that
// depends only on (NK, dh) -- never on the batch size or the chip -- and every item starts from an empty state and
// writes un-normalised (O, m, l) partials that `attn_combine_kernel` merges in fixed order: a sample's result is
// bit-identical whatever batch it travels in.  One persistent block per CU; items are dealt to the CU's four SIMDs in
// contiguous, equal shares (the matrix pipe is per SIMD) and round-robin to the waves of a SIMD, so co-resident waves
// read the same K/V chunk (L1/L2 hits) for neighbouring query tiles.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

#define POEM_ATTN_LAZY_LOG2 8.0f
#ifndef POEM_XA_VARIANT
#define POEM_XA_VARIANT 0
#endif

// streaming (non-temporal) 16-byte accesses for the partials: written once, read once -- they should not displace the
// step's reusable tensors (the vector attention's gather sources) from L2 / Infinity Cache
__device__ __forceinline__ void nt_store4(float4* p, float4 v) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(p));
}
__device__ __forceinline__ float4 nt_load4(const float4* p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// Softmax numerators of one 32-key tile, in place (s: raw scores -> p), with the lazy running stabiliser (see the
// header): 8 v_max3 + 8 v_pk_fma + 16 v_exp_f32 + 9 packed adds per tile; the rescale branch is wave-uniform and rare.
// (A macro, not a function: with the accumulator array passed by reference hipcc kept part of it in scratch -- 31 spilled
// VGPRs in the head-dim-64 kernel.)
#define POEM_SOFTMAX_TILE(DTV)                                                                                  \
  {                                                                                                             \
    float mx_ = max3f(s[0], s[1], s[2]);                                                                        \
    mx_ = max3f(mx_, s[3], s[4]);                                                                               \
    mx_ = max3f(mx_, s[5], s[6]);                                                                               \
    mx_ = max3f(mx_, s[7], s[8]);                                                                               \
    mx_ = max3f(mx_, s[9], s[10]);                                                                              \
    mx_ = max3f(mx_, s[11], s[12]);                                                                             \
    mx_ = max3f(mx_, s[13], s[14]);                                                                             \
    mx_ = fmaxf(mx_, s[15]);                                                                                    \
    if (__any(mx_ > m_ref + lazy_raw)) {          /* wave-uniform, rare after the first tile */                 \
      const float mf_ = half_max(mx_);            /* both halves of a query agree on the new stabiliser */      \
      const float m_new_ = (mf_ > m_ref + lazy_raw) ? mf_ : m_ref;                                              \
      const float alpha_ = __builtin_amdgcn_exp2f((m_ref - m_new_) * kc2);   /* 1 where unchanged, 0 at first */ \
      _Pragma("unroll") for (int d_ = 0; d_ < (DTV); ++d_)                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) o[d_][i_] *= alpha_;                                  \
      l_run *= alpha_;                                                                                          \
      m_ref = m_new_;                                                                                           \
      nbias = -m_new_ * kc2;                                                                                    \
    }                                                                                                           \
    const f32x2 kc2v_ = {kc2, kc2}, nbv_ = {nbias, nbias};                                                      \
    f32x2 ps_ = {0.f, 0.f};                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; i_ += 2) {                                                      \
      f32x2 tv_ = {s[i_], s[i_ + 1]};                                                                           \
      tv_ = __builtin_elementwise_fma(tv_, kc2v_, nbv_);                                                        \
      tv_[0] = __builtin_amdgcn_exp2f(tv_[0]);                                                                  \
      tv_[1] = __builtin_amdgcn_exp2f(tv_[1]);                                                                  \
      s[i_] = tv_[0];                                                                                           \
      s[i_ + 1] = tv_[1];                                                                                       \
      ps_ += tv_;                                                                                               \
    }                                                                                                           \
    l_run += ps_[0] + ps_[1];                                                                                   \
  }

#ifdef POEM_LAB   // tools/lab only: per-wave (shader cycles, 100 MHz ticks, items) of the last launch
__device__ long long xattn_dbg[4096 * 4];
#endif
#ifdef POEM_XA_STAMPS   // tools/lab only: cycles per phase of wave 0 of block 5
__device__ long long xattn_ph[8];
#define XA_STAMP(k) do { if (dbg_on) { const long long t_ = clock64(); if ((k) != 0) dbg_ph[(k)] += t_ - dbg_last; else if (dbg_last) dbg_ph[0] += t_ - dbg_last; dbg_last = t_; } } while (0)
#else
#define XA_STAMP(k) do { } while (0)
#endif

// ---- round 6: the REMAINDER items of a launch as channel-tile halves.  The static map deals a CU pair's n items to its 8 W waves
// round-robin: n = 100 at the headline batch, 24 waves -> four full rounds and a remainder of 4 items, which four of the pair's
// eight SIMDs run as a 13th item while the other four idle (12.5 items per SIMD on average, 13 on the busiest: the kernel's
// 0.96 quantisation; at 64 samples, 25.0 items per SIMD, the same kernel is 0.885 MFMA-busy instead of 0.845).  A remainder of
// r <= 4 items is dealt as 2 r HALF items instead, one per SIMD of the pair: the item's full score contraction and softmax but
// ONE 32-channel tile of P V (48 instead of 64 MFMAs per key tile, the HALF idea of the merged kernel), so the last round is
// 0.75 of an item on every SIMD instead of a whole one on half of them.  Each output element of the partial is the same fma
// chain over the keys whichever wave owns its channel tile, and (m, l) -- functions of the scores only -- are written by the
// tile-0 half: bit-identical partials, same layout.
template <int DH>
__device__ __forceinline__ void xattn_half_item(const float* __restrict__ q, int ldq, int qbr, const __amdgpu_buffer_rsrc_t krs,
                                                const __amdgpu_buffer_rsrc_t vrs, float4* __restrict__ part_o,
                                                float2* __restrict__ part_ml, int item, int d0, int nqt, int chunks, int heads, int NQ,
                                                int nkt, int C, int tpc, float kc2, float lazy_raw) {
  constexpr int KC = DH / 8, DTF = DH / 32, DT = 1;
  // (the lane id afresh from the hardware -- mbcnt -- instead of the kernel's `lane`: keeping that one alive across the main
  //  item loop for this tail was the 169th register of a kernel that fits three waves per SIMD in 168)
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int r = lane & 31, h = lane >> 5, loff = lane * 16;
  const int qt = item % nqt;
  int t = item / nqt;
  const int ch = t % chunks;
  t /= chunks;
  const int head = t % heads, b = t / heads;
  const int qrow = min(qt * 32 + r, NQ - 1);
  float4 qf[KC];
  {
    const float* qp = q + ((size_t)b * qbr + qrow) * ldq + head * DH + 4 * h;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) qf[kc] = *reinterpret_cast<const float4*>(qp + 8 * kc);
  }
  const int kt0 = ch * tpc;
  const int ktile_bytes = C * 128;
  int koff = __builtin_amdgcn_readfirstlane((b * nkt + kt0) * ktile_bytes + head * KC * 1024);
  int voff = __builtin_amdgcn_readfirstlane((b * nkt + kt0) * ktile_bytes + ((head * DH) / 32 + d0) * 4096);
  float4 kf[KC], vf[DT][4];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) kf[kc] = frag_load(krs, loff, koff + kc * 1024);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < 4; ++g) vf[0][g] = frag_load(vrs, loff, voff + g * 1024);
  __builtin_amdgcn_sched_barrier(0);
  f32x16 o[DT];
  o[0] = zero16();
  float m_ref = -INFINITY, nbias = 0.f, l_run = 0.f;
  for (int kt = 0; kt < tpc; ++kt) {
    if ((kt & 3) == 0 && kt) __builtin_amdgcn_s_barrier();      // (as the full items: the block's live waves stay on the same tiles)
    f32x16 s = zero16();
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      s = mfma32(kf[kc].x, qf[kc].x, s);
      s = mfma32(kf[kc].y, qf[kc].y, s);
      s = mfma32(kf[kc].z, qf[kc].z, s);
      s = mfma32(kf[kc].w, qf[kc].w, s);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int adv = (kt + 1 < tpc) ? ktile_bytes : 0;
    koff += adv;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) kf[kc] = frag_load(krs, loff, koff + kc * 1024);
    __builtin_amdgcn_sched_barrier(0);
    POEM_SOFTMAX_TILE(DT)
    __builtin_amdgcn_sched_barrier(0);
    voff += adv;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      o[0] = mfma32((&vf[0][i >> 2].x)[i & 3], s[i], o[0]);
      if ((i & 3) == 3) {
        __builtin_amdgcn_sched_barrier(0);
        vf[0][i >> 2] = frag_load(vrs, loff, voff + (i >> 2) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  l_run = half_sum(l_run);
  float4* po = part_o + (size_t)item * (DTF * 4) * 64 + lane;
#pragma unroll
  for (int g = 0; g < 4; ++g) nt_store4(po + (d0 * 4 + g) * 64, make_float4(o[0][4 * g], o[0][4 * g + 1], o[0][4 * g + 2], o[0][4 * g + 3]));
  if (h == 0 && d0 == 0) part_ml[(size_t)item * 32 + r] = make_float2(m_ref, l_run);
  __builtin_amdgcn_s_waitcnt(0x0F70);
}

// MERGE (round 3; four key chunks, the head path's shape): the four chunks of a query tile run on four waves of ONE block at
// the same time -- wave w = (group w / 4, chunk w % 4), a block's W groups take consecutive query tiles -- and the block
// merges their partials through LDS (attn_combine_kernel's arithmetic, chunk order) and writes the normalised context rows:
// the partials (4 x 26 MB written per launch at the headline batch and read back by the consumer) never reach HBM.  The waves
// of a group read different K/V chunks, the W waves with the same chunk the same one (the tile barrier keeps them together).
// HALF (MERGE only, round 4): an item is one 32-channel tile of a query tile's output instead of all DH / 32 of them -- twice the
// items, each with the full score contraction and softmax but half the P V products (48 instead of 64 MFMAs per key tile).  More
// MFMAs in all, so it only pays where the launch is one under-filled round anyway: a single sample (100 query-tile items on 256
// CUs -> 200).  Every output element is the same fma chain as without it.
template <int DH, int W, bool MERGE = false, bool HALF = false>
__global__ __launch_bounds__(256 * W, W) void xattn_kernel(const float* __restrict__ q, int ldq, int qbr,
                                                           const float4* __restrict__ kimg,
                                                           const float4* __restrict__ vimg,
                                                           float4* __restrict__ part_o, float2* __restrict__ part_ml,
                                                           int B, int NQ, int NK, int C, int heads, int tpc, float kc2,
                                                           float lazy_raw, int map, int prio_rot, float* __restrict__ ctx) {
  const bool split_tail = (map & 16) != 0;      // (bit 4 of `map`: the remainder items as halves)
  map &= 15;
  constexpr int KC = DH / 8;               // K fragments (float4) per key tile
  constexpr int DTF = (DH + 31) / 32;      // 32-channel tiles of the output
  constexpr int DT = HALF ? 1 : DTF;       // ... of an item
  static_assert(!HALF || MERGE, "channel-tile items exist in the merged form only");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int nqt = (NQ + 31) / 32, nkt = NK / 32, chunks = nkt / tpc;
  const int items = MERGE ? B * heads * nqt * (HALF ? DTF : 1) : B * heads * chunks * nqt;      // MERGE: one item = a query tile, all four chunks
  extern __shared__ __attribute__((aligned(16))) float xa_lds[];            // MERGE: 4W x (DT*4 x 64 float4 | 32 float2)
  // logical block id: blocks of one XCD (blockIdx % 8) take neighbouring item ranges -> one L2 serves a K/V chunk
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  // map 1: a CU owns a contiguous item range and deals it round-robin to all its waves (wave w sits on SIMD w % 4, so
  //        the SIMDs stay balanced to within one item) -- every wave of the CU streams the same K/V chunk at the same
  //        time: one HBM/L2 fetch serves them all.
  // map 0: each SIMD owns a contiguous range, dealt to its W waves.
  // map 2: TWO neighbouring blocks of an XCD (logical ids 2k, 2k + 1) share a contiguous item range and deal it round-robin to
  //        their 8 W waves: the 25 query tiles of a (sample, head, key chunk) group then meet its K / V chunk in one sweep of two
  //        CUs instead of 2.1 sweeps of one (each sweep beyond the first re-fetches the chunk: 16 MB of chunks per XCD do not
  //        stay in its 4 MB L2).
  const bool pair = !MERGE && map == 2 && nb % 16 == 0;
  const int sg = pair ? (lb >> 1) : ((map || MERGE) ? lb : lb * 4 + (wv & 3)), ng = pair ? (nb >> 1) : ((map || MERGE) ? nb : nb * 4);
  const int ibase = items / ng, irem = items % ng;
  const int lo = ibase * sg + min(sg, irem), hi = lo + ibase + (sg < irem ? 1 : 0);
  const int first = MERGE ? (wv >> 2) : (pair ? 2 * wv + (lb & 1) : (map ? wv : (wv >> 2)));      // (pair: the two blocks' waves alternate)
  const int stride = MERGE ? W : (pair ? 8 * W : (map ? 4 * W : W));
  const __amdgpu_buffer_rsrc_t krs = frag_rsrc(kimg, 0xffffffffu), vrs = frag_rsrc(vimg, 0xffffffffu);
  const int loff = lane * 16;
  const bool sync_tiles = MERGE || (map != 0 && prio_rot != 3);      // (prio_rot == 3: lab switch to turn the tile barrier off)
#ifdef POEM_LAB
  const long long dbg_c0 = clock64(), dbg_w0 = wall_clock64();
  int dbg_items = 0;
#endif
#ifdef POEM_XA_STAMPS
  const bool dbg_on = blockIdx.x == 5 && wv == 0;
  long long dbg_ph[4] = {0, 0, 0, 0}, dbg_last = 0;
#endif

  // (pair map, head dim 64: a remainder of <= 4 items behind the full rounds runs as halves -- xattn_half_item above)
  // (wave-uniform values the compiler cannot prove uniform -- they derive from threadIdx -- pinned to scalar registers: the
  //  kernel sits at its 168-register budget for three waves per SIMD)
  int hi_full = hi;
  if constexpr (!MERGE && DH == 64) {
    const int rem = (hi - lo) % stride;
    if (pair && split_tail && rem > 0 && 2 * rem <= 8 && W == 3) hi_full = hi - rem;
    hi_full = __builtin_amdgcn_readfirstlane(hi_full);
  }
  const int tail_items = __builtin_amdgcn_readfirstlane(hi - hi_full), tail_first = __builtin_amdgcn_readfirstlane(first);
  for (int item = lo + first; item < hi_full; item += stride) {
#ifdef POEM_LAB
    ++dbg_items;
#endif
    const int d0 = HALF ? item % DTF : 0;      // first output channel tile of this item
    const int qt = (HALF ? item / DTF : item) % nqt;
    int t = (HALF ? item / DTF : item) / nqt;
    const int ch = MERGE ? (wv & 3) : t % chunks;
    if (!MERGE) t /= chunks;
    const int head = t % heads, b = t / heads;
    const int qrow = min(qt * 32 + r, NQ - 1);

    // Q fragment: lane (query r, half h) holds Q[r][8kc + 4h + t]
    float4 qf[KC];
    {
      const float* qp = q + ((size_t)b * qbr + qrow) * ldq + head * DH + 4 * h;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) qf[kc] = *reinterpret_cast<const float4*>(qp + 8 * kc);
    }
    // scalar byte offsets of this item's first key tile in the two images
    const int kt0 = ch * tpc;
    const int ktile_bytes = C * 128;                                        // 32 keys x C floats, both images
    int koff = __builtin_amdgcn_readfirstlane((b * nkt + kt0) * ktile_bytes + head * KC * 1024);
    int voff = __builtin_amdgcn_readfirstlane((b * nkt + kt0) * ktile_bytes + ((head * DH) / 32) * 4096);

    float4 kf[KC], vf[DT][4];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) kf[kc] = frag_load(krs, loff, koff + kc * 1024);
    __builtin_amdgcn_sched_barrier(0);   // issue order Q, K, V as in the loop: the loop header then waits for K only
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) vf[d][g] = frag_load(vrs, loff, voff + ((d0 + d) * 4 + g) * 1024);
    __builtin_amdgcn_sched_barrier(0);

    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = zero16();
    float m_ref = -INFINITY, nbias = 0.f, l_run = 0.f;

    for (int kt = 0; kt < tpc; ++kt) {
      XA_STAMP(0);
      // A block barrier every 4 key tiles keeps all waves of the CU (they stream the same K/V chunk, map 1) within 4 tiles
      // of each other, so one fetch from HBM / Infinity Cache serves all 12 of them through L2: 3.4 -> 1.0 GB fetched per
      // launch, -3.6 % kernel time, and less cache pollution for the kernels that follow.  Left alone the waves drift by
      // whole items (the arbiter serves the oldest wave of a SIMD first).  Waves that ran out of items have exited; a
      // terminated wave no longer counts at the barrier.
      if (sync_tiles && (kt & 3) == 0 && kt) __builtin_amdgcn_s_barrier();
#ifdef POEM_LAB   // experiment kept for the lab build only (does not pay)
      if (W > 1 && prio_rot == 1) {
        const int pr = (kt + (wv >> 2)) % W;
        if (pr == 0) __builtin_amdgcn_s_setprio(0);
        else if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
      }
#endif
      // ---- S^T = K . Q^T (raw scores)
      f32x16 s = zero16();
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        s = mfma32(kf[kc].x, qf[kc].x, s);
        s = mfma32(kf[kc].y, qf[kc].y, s);
        s = mfma32(kf[kc].z, qf[kc].z, s);
        s = mfma32(kf[kc].w, qf[kc].w, s);
      }
      __builtin_amdgcn_sched_barrier(0);
      // next tile's K fragments into the registers the MFMAs above have just read (clamped: the last prefetch of an
      // item re-reads its own last tile)
      const int adv = (kt + 1 < tpc) ? ktile_bytes : 0;
      koff += adv;
#ifndef POEM_XA_NOLOADS
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) kf[kc] = frag_load(krs, loff, koff + kc * 1024);
#endif
      __builtin_amdgcn_sched_barrier(0);

      XA_STAMP(1);
      // ---- softmax numerators, lazy stabiliser
      POEM_SOFTMAX_TILE(DT)
      __builtin_amdgcn_sched_barrier(0);
      XA_STAMP(2);

      // ---- O^T += V^T . P^T; the V fragments of register group g are re-requested (next tile) as soon as the four
      //      k-steps that read them have been issued
      voff += adv;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int d = 0; d < DT; ++d) o[d] = mfma32((&vf[d][i >> 2].x)[i & 3], s[i], o[d]);
        if ((i & 3) == 3) {
          __builtin_amdgcn_sched_barrier(0);
#ifndef POEM_XA_NOLOADS
#pragma unroll
          for (int d = 0; d < DT; ++d) vf[d][i >> 2] = frag_load(vrs, loff, voff + ((d0 + d) * 4 + (i >> 2)) * 1024);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      XA_STAMP(3);
    }

    l_run = half_sum(l_run);
    if constexpr (MERGE) {
      // ---- the four chunk partials of this wave's group meet in LDS (fragment order, as the HBM partials)
      constexpr int WSTRIDE = DT * 4 * 64 * 4 + 64;          // floats per wave: O image | (m, l) of its 32 rows
      float* mine = xa_lds + wv * WSTRIDE;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          reinterpret_cast<float4*>(mine)[(d * 4 + g) * 64 + lane] = make_float4(o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]);
      if (h == 0) reinterpret_cast<float2*>(mine + DT * 4 * 64 * 4)[r] = make_float2(m_ref, l_run);
      __syncthreads();                        // every live wave is at the end of an item (all items have tpc tiles)
      // ---- ctx[b, q, head*DH + c] = sum_s w_s O_s[c] / sum_s w_s l_s (attn_combine_kernel, chunk order); chunk-wave c of the
      // group takes float4 groups c, c + 4, ... of the DT * 4
      const float* grp = xa_lds + (wv & ~3) * WSTRIDE;
      float w4[4], M = -INFINITY, den = 0.f;
#pragma unroll
      for (int sx = 0; sx < 4; ++sx) {
        w4[sx] = reinterpret_cast<const float2*>(grp + sx * WSTRIDE + DT * 4 * 64 * 4)[r].x;
        M = fmaxf(M, w4[sx]);
      }
#pragma unroll
      for (int sx = 0; sx < 4; ++sx) {
        const float lx = reinterpret_cast<const float2*>(grp + sx * WSTRIDE + DT * 4 * 64 * 4)[r].y;
        w4[sx] = (w4[sx] == M) ? 1.0f : __builtin_amdgcn_exp2f((w4[sx] - M) * kc2);
        den = fmaf(w4[sx], lx, den);
      }
      if (qt * 32 + r < NQ) {
        float* out = ctx + ((size_t)b * NQ + qt * 32 + r) * C + head * DH;
#pragma unroll
        for (int k = (wv & 3); k < DT * 4; k += 4) {
          float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int sx = 0; sx < 4; ++sx) {
            const float4 pp = reinterpret_cast<const float4*>(grp + sx * WSTRIDE)[k * 64 + lane];
            acc.x = fmaf(w4[sx], pp.x, acc.x); acc.y = fmaf(w4[sx], pp.y, acc.y);
            acc.z = fmaf(w4[sx], pp.z, acc.z); acc.w = fmaf(w4[sx], pp.w, acc.w);
          }
          const int d = d0 + (k >> 2), g = k & 3;
          *reinterpret_cast<float4*>(out + 32 * d + 8 * g + 4 * h) = make_float4(acc.x / den, acc.y / den, acc.z / den, acc.w / den);
        }
      }
      // (the next item's tile barriers -- tpc >= 8, checked at launch -- order these reads before its LDS writes)
    } else {
      // partial (O, m, l): fragment order, one coalesced 1 KiB store per (channel tile, register group)
      float4* po = part_o + (size_t)item * (DT * 4) * 64 + lane;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          nt_store4(po + (d * 4 + g) * 64, make_float4(o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]));
      if (h == 0) part_ml[(size_t)item * 32 + r] = make_float2(m_ref, l_run);
    }
    // drain the stores here: with stores possibly pending at the key loop's header hipcc cannot count on in-order
    // returns and waits for vmcnt(0) on every iteration, i.e. for the V prefetch it has just issued
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  if constexpr (!MERGE && DH == 64) {
    // halves of the remainder: pair-wave index `first` (even: first block of the pair, odd: second) -> SIMD first % 8 of the pair
    if (tail_first < 2 * tail_items)
      xattn_half_item<DH>(q, ldq, qbr, krs, vrs, part_o, part_ml, hi_full + (tail_first >> 1), tail_first & 1, nqt, chunks, heads, NQ, nkt, C,
                          tpc, kc2, lazy_raw);
  }
#ifdef POEM_LAB
  if (lane == 0) {
    long long* d = &xattn_dbg[((size_t)blockIdx.x * 4 * W + wv) % 4096 * 4];
    d[0] = clock64() - dbg_c0; d[1] = wall_clock64() - dbg_w0; d[2] = dbg_items; d[3] = blockIdx.x;
  }
#endif
#ifdef POEM_XA_STAMPS
  if (dbg_on && lane == 0) for (int i = 0; i < 4; ++i) xattn_ph[i] = dbg_ph[i];
#endif
}

// ---- opt-in split precision (POEM_PRECISION_SPLIT_F16X3_ALL; scheme: vecattn_split.hip) --------------------------------
// Same work decomposition, partials and combine as xattn_kernel; the two contractions run on v_mfma_f32_32x32x16_f16 as
// hi | lo f16 splits with fp32 accumulation.  It consumes the SAME fp32 fragment images: a 32x32x16 MFMA sums over 16
// k-slots and does not care which k they hold as long as both operands agree, so slot (half h, t) of chunk c takes
//   t < 4: element 4h + t of fragment 2c,   t >= 4: element 4h + t - 4 of fragment 2c + 1
// -- exactly the two float4 a lane already holds of K (channels), of V (keys) and, by the C/D layout, of P (registers
// 8c .. 8c+7 of the score tile).  Operands are scaled by powers of two (K, Q, V: 16; P: 16) and split in registers; the
// score scale is folded into the softmax constants, the output scale is undone once per item.
typedef _Float16 xh8 __attribute__((ext_vector_type(8)));
#define POEM_XS_SCALE 16.0f
__device__ __forceinline__ void xsplit8(const float4 a, const float4 b, xh8& hi, xh8& lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float x = __builtin_amdgcn_fmed3f(v[t] * POEM_XS_SCALE, -60000.f, 60000.f);
    hi[t] = (_Float16)x;
    lo[t] = (_Float16)(x - (float)hi[t]);
  }
}
__device__ __forceinline__ f32x16 xmfma16(xh8 a, xh8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// PRE: the images already hold hi | lo f16 operands (written by the split projection GEMM, gemm.hip output modes 1 / 2 in
// the split variant): fragment 2c = the hi halfs of chunk c, fragment 2c+1 = the lo halfs -- no conversion of K / V here.
template <int DH, int W, bool PRE>
__global__ __launch_bounds__(256 * W, W) void xattn_split_kernel(const float* __restrict__ q, int ldq, int qbr,
                                                                 const float4* __restrict__ kimg,
                                                                 const float4* __restrict__ vimg,
                                                                 float4* __restrict__ part_o, float2* __restrict__ part_ml,
                                                                 int B, int NQ, int NK, int C, int heads, int tpc, float kc2,
                                                                 float lazy_raw) {
  constexpr int KC = DH / 8;               // fp32 K fragments (float4) per key tile
  constexpr int KC16 = DH / 16;            // 16-channel chunks of the score contraction
  constexpr int DT = (DH + 31) / 32;
  constexpr float SS = POEM_XS_SCALE * POEM_XS_SCALE;      // scale the raw scores / the outputs carry
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int nqt = (NQ + 31) / 32, nkt = NK / 32, chunks = nkt / tpc;
  const int items = B * heads * chunks * nqt;
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int ibase = items / nb, irem = items % nb;
  const int lo_i = ibase * lb + min(lb, irem), hi_i = lo_i + ibase + (lb < irem ? 1 : 0);
  const __amdgpu_buffer_rsrc_t krs = frag_rsrc(kimg, 0xffffffffu), vrs = frag_rsrc(vimg, 0xffffffffu);
  const int loff = lane * 16;
  const float kc2s = kc2 / SS, lazy_s = lazy_raw * SS;      // the softmax macro below works on the scaled scores
  for (int item = lo_i + wv; item < hi_i; item += 4 * W) {
    const int qt = item % nqt;
    int t = item / nqt;
    const int ch = t % chunks;
    t /= chunks;
    const int head = t % heads, b = t / heads;
    const int qrow = min(qt * 32 + r, NQ - 1);
    xh8 qh[KC16], ql[KC16];
    {
      const float* qp = q + ((size_t)b * qbr + qrow) * ldq + head * DH + 4 * h;
#pragma unroll
      for (int c = 0; c < KC16; ++c)
        xsplit8(*reinterpret_cast<const float4*>(qp + 16 * c), *reinterpret_cast<const float4*>(qp + 16 * c + 8), qh[c], ql[c]);
    }
    const int kt0 = ch * tpc;
    const int ktile_bytes = C * 128;
    int koff = __builtin_amdgcn_readfirstlane((b * nkt + kt0) * ktile_bytes + head * KC * 1024);
    int voff = __builtin_amdgcn_readfirstlane((b * nkt + kt0) * ktile_bytes + ((head * DH) / 32) * 4096);
    float4 kf[KC], vf[DT][4];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) kf[kc] = frag_load(krs, loff, koff + kc * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) vf[d][g] = frag_load(vrs, loff, voff + (d * 4 + g) * 1024);
    __builtin_amdgcn_sched_barrier(0);

    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = zero16();
    float m_ref = -INFINITY, nbias = 0.f, l_run = 0.f;
    {
    const float kc2 = kc2s, lazy_raw = lazy_s;               // shadow: POEM_SOFTMAX_TILE reads these names
    for (int kt = 0; kt < tpc; ++kt) {
      if ((kt & 3) == 0 && kt) __builtin_amdgcn_s_barrier();   // keeps the CU's waves on the same K/V tiles (see above)
      // ---- S^T = K . Q^T on the scaled splits
      // two accumulators (even / odd chunks), part-major: consecutive MFMAs never share an accumulator (a dependent
      // 32x32x16 pair does not issue back to back)
      f32x16 s = zero16(), s1 = zero16();
      xh8 kh[KC16], kl[KC16];
#pragma unroll
      for (int c = 0; c < KC16; ++c) {
        if constexpr (PRE) { kh[c] = __builtin_bit_cast(xh8, kf[2 * c]); kl[c] = __builtin_bit_cast(xh8, kf[2 * c + 1]); }
        else xsplit8(kf[2 * c], kf[2 * c + 1], kh[c], kl[c]);
      }
#pragma unroll
      for (int c = 0; c < KC16; ++c) { if (c & 1) s1 = xmfma16(kh[c], ql[c], s1); else s = xmfma16(kh[c], ql[c], s); }
#pragma unroll
      for (int c = 0; c < KC16; ++c) { if (c & 1) s1 = xmfma16(kh[c], qh[c], s1); else s = xmfma16(kh[c], qh[c], s); }
#pragma unroll
      for (int c = 0; c < KC16; ++c) { if (c & 1) s1 = xmfma16(kl[c], qh[c], s1); else s = xmfma16(kl[c], qh[c], s); }
      if (KC16 > 1) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const f32x2 t2 = f32x2{s[i], s[i + 1]} + f32x2{s1[i], s1[i + 1]};
          s[i] = t2[0]; s[i + 1] = t2[1];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const int adv = (kt + 1 < tpc) ? ktile_bytes : 0;
      koff += adv;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) kf[kc] = frag_load(krs, loff, koff + kc * 1024);
      __builtin_amdgcn_sched_barrier(0);
      POEM_SOFTMAX_TILE(DT)
      __builtin_amdgcn_sched_barrier(0);
      // ---- O^T += V^T . P^T: registers 8c .. 8c+7 of the numerators are the B operand of key chunk c
      voff += adv;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        xh8 ph, pl;
        xsplit8(make_float4(s[8 * c], s[8 * c + 1], s[8 * c + 2], s[8 * c + 3]),
                make_float4(s[8 * c + 4], s[8 * c + 5], s[8 * c + 6], s[8 * c + 7]), ph, pl);
        xh8 vh[DT], vl[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          if constexpr (PRE) { vh[d] = __builtin_bit_cast(xh8, vf[d][2 * c]); vl[d] = __builtin_bit_cast(xh8, vf[d][2 * c + 1]); }
          else xsplit8(vf[d][2 * c], vf[d][2 * c + 1], vh[d], vl[d]);
        }
#pragma unroll
        for (int d = 0; d < DT; ++d) o[d] = xmfma16(vh[d], pl, o[d]);
#pragma unroll
        for (int d = 0; d < DT; ++d) o[d] = xmfma16(vh[d], ph, o[d]);
#pragma unroll
        for (int d = 0; d < DT; ++d) o[d] = xmfma16(vl[d], ph, o[d]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          vf[d][2 * c] = frag_load(vrs, loff, voff + (d * 4 + 2 * c) * 1024);
          vf[d][2 * c + 1] = frag_load(vrs, loff, voff + (d * 4 + 2 * c + 1) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
    // partial (O, m, l) in the exact kernel's units: O / (16 * 16), m in raw-score units
    l_run = half_sum(l_run);
    float4* po = part_o + (size_t)item * (DT * 4) * 64 + lane;
    const float un = 1.0f / SS;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        nt_store4(po + (d * 4 + g) * 64, make_float4(o[d][4 * g] * un, o[d][4 * g + 1] * un, o[d][4 * g + 2] * un, o[d][4 * g + 3] * un));
    if (h == 0) part_ml[(size_t)item * 32 + r] = make_float2(m_ref * un, l_run);
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
}

// Head dims 128 and 256 (POEM-large / -huge): the K and V fragments of a key tile no longer fit the register file next
// to Q and O, so they stream through a two-slot ring of 8 fragments (32 registers each): a tile is a fixed sequence of
// NG = DH/64 + DH/64 operand groups -- K channel groups of 64, then V channel-tile pairs -- and while the 32 MFMAs of
// group n issue, group n+1 (the next tile's first K group after the last V pair) is in flight into the other slot.
// NG is even, so the slot of every group is a compile-time constant.  Same items, partials and combine as above.
template <int DH, int W>
__global__ __launch_bounds__(256 * W, W) void xattn_stream_kernel(const float* __restrict__ q, int ldq, int qbr,
                                                                  const float4* __restrict__ kimg,
                                                                  const float4* __restrict__ vimg,
                                                                  float4* __restrict__ part_o,
                                                                  float2* __restrict__ part_ml, int B, int NQ, int NK,
                                                                  int C, int heads, int tpc, float kc2, float lazy_raw,
                                                                  int map) {
  constexpr int KC = DH / 8, DT = DH / 32, NGK = KC / 8, NGV = DT / 2;
  static_assert(NGK == NGV && NGK >= 1, "head dim must be a multiple of 64");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int nqt = (NQ + 31) / 32, nkt = NK / 32, chunks = nkt / tpc;
  const int items = B * heads * chunks * nqt;
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int sg = map ? lb : lb * 4 + (wv & 3), ng = map ? nb : nb * 4;
  const int ibase = items / ng, irem = items % ng;
  const int lo = ibase * sg + min(sg, irem), hi = lo + ibase + (sg < irem ? 1 : 0);
  const int first = map ? wv : (wv >> 2), stride = map ? 4 * W : W;
  const __amdgpu_buffer_rsrc_t krs = frag_rsrc(kimg, 0xffffffffu), vrs = frag_rsrc(vimg, 0xffffffffu);
  const int loff = lane * 16;

  for (int item = lo + first; item < hi; item += stride) {
    const int qt = item % nqt;
    int t = item / nqt;
    const int ch = t % chunks;
    t /= chunks;
    const int head = t % heads, b = t / heads;
    const int qrow = min(qt * 32 + r, NQ - 1);
    // The query fragment (KC float4 per lane: 128 registers at head dim 256): in registers for head dim 64; in LDS for the
    // wide heads (a wave's own KC KB, fragment order, conflict-free ds_read_b128 right in front of the MFMAs that use it) --
    // the registers it frees are what the four-group ring below is made of.
    constexpr bool QLDS = NGK >= 2;
    extern __shared__ __attribute__((aligned(16))) float4 xs_q[];      // QLDS: (waves of the block) x KC x 64
    float4* qs = xs_q + (size_t)wv * KC * 64 + lane;
    float4 qf[QLDS ? 1 : KC];
    {
      const float* qp = q + ((size_t)b * qbr + qrow) * ldq + head * DH + 4 * h;
      if constexpr (QLDS) {
        constexpr int QB = 8;
#pragma unroll
        for (int k0 = 0; k0 < KC; k0 += QB) {
          float4 t[QB];
#pragma unroll
          for (int u = 0; u < QB; ++u) t[u] = *reinterpret_cast<const float4*>(qp + 8 * (k0 + u));
#pragma unroll
          for (int u = 0; u < QB; ++u) qs[(k0 + u) * 64] = t[u];
        }
      } else {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) qf[kc] = *reinterpret_cast<const float4*>(qp + 8 * kc);
      }
    }
    const int kt0 = ch * tpc;
    const int ktile_bytes = C * 128;
    int koff = __builtin_amdgcn_readfirstlane((b * nkt + kt0) * ktile_bytes + head * KC * 1024);
    int voff = __builtin_amdgcn_readfirstlane((b * nkt + kt0) * ktile_bytes + ((head * DH) / 32) * 4096);
    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = zero16();
    float m_ref = -INFINITY, nbias = 0.f, l_run = 0.f;
    if constexpr (NGK >= 2) {
      // Head dims 128 / 256 (round 4): ONE wave per SIMD (the query fragment and the output tiles alone are 192 / 256 registers),
      // so nothing hides a load but the wave's own distance to it -- and one 8 KB group ahead is 32 MFMAs = 0.85 us, less than
      // an L2 / MALL round trip under load.  Ring of FOUR groups, three in flight: the
      // 2 NGK groups of a tile (K groups, then V pairs) are 4 or 8, so every group's slot is a compile-time constant.
      float4 ring[4][8];
      // group n of the current tile: K group n | V pair n - NGK | the NEXT tile's K group n - 2 NGK | its V pair n - 3 NGK
#define XS_LOAD(N, ADV)                                                                                        \
      _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                            \
        ring[(N) & 3][e] = (N) < NGK ? frag_load(krs, loff, koff + ((N) * 8 + e) * 1024)                       \
                         : (N) < 2 * NGK ? frag_load(vrs, loff, voff + (((N) - NGK) * 8 + e) * 1024)           \
                         : (N) < 3 * NGK ? frag_load(krs, loff, koff + (ADV) + (((N) - 2 * NGK) * 8 + e) * 1024) \
                                         : frag_load(vrs, loff, voff + (ADV) + (((N) - 3 * NGK) * 8 + e) * 1024);
      XS_LOAD(0, 0) XS_LOAD(1, 0) XS_LOAD(2, 0)
      __builtin_amdgcn_sched_barrier(0);
      for (int kt = 0; kt < tpc; ++kt) {
        if (map && (kt & 3) == 0 && kt) __builtin_amdgcn_s_barrier();   // keep the CU's waves on the same K/V tiles (see xattn_kernel)
        const int adv = (kt + 1 < tpc) ? ktile_bytes : 0;
        f32x16 s = zero16();
#pragma unroll
        for (int g = 0; g < NGK; ++g) {
          XS_LOAD(g + 3, adv)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float4 a = ring[g & 3][e];
            const float4 bq = qs[(g * 8 + e) * 64];
            s = mfma32(a.x, bq.x, s);
            s = mfma32(a.y, bq.y, s);
            s = mfma32(a.z, bq.z, s);
            s = mfma32(a.w, bq.w, s);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        POEM_SOFTMAX_TILE(DT)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < NGV; ++v) {
          XS_LOAD(NGK + v + 3, adv)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
#pragma unroll
            for (int dd = 0; dd < 2; ++dd)
              o[2 * v + dd] = mfma32((&ring[(NGK + v) & 3][dd * 4 + (i >> 2)].x)[i & 3], s[i], o[2 * v + dd]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        koff += adv;
        voff += adv;
      }
#undef XS_LOAD
    } else {
    float4 ring[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ring[0][e] = frag_load(krs, loff, koff + e * 1024);
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < tpc; ++kt) {
      if (map && (kt & 3) == 0 && kt) __builtin_amdgcn_s_barrier();   // keep the CU's waves on the same K/V tiles (see xattn_kernel)
      const int adv = (kt + 1 < tpc) ? ktile_bytes : 0;
      f32x16 s = zero16();
#pragma unroll
      for (int g = 0; g < NGK; ++g) {
        // next group: K group g+1 of this tile, or V pair 0 behind the last K group
#pragma unroll
        for (int e = 0; e < 8; ++e)
          ring[(g + 1) & 1][e] = (g + 1 < NGK) ? frag_load(krs, loff, koff + ((g + 1) * 8 + e) * 1024)
                                               : frag_load(vrs, loff, voff + e * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float4 a = ring[g & 1][e];
          const float4 bq = qf[g * 8 + e];
          s = mfma32(a.x, bq.x, s);
          s = mfma32(a.y, bq.y, s);
          s = mfma32(a.z, bq.z, s);
          s = mfma32(a.w, bq.w, s);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      POEM_SOFTMAX_TILE(DT)
      __builtin_amdgcn_sched_barrier(0);
      koff += adv;
#pragma unroll
      for (int v = 0; v < NGV; ++v) {
        // next group: V pair v+1 of this tile, or the next tile's K group 0 behind the last pair
#pragma unroll
        for (int e = 0; e < 8; ++e)
          ring[(NGK + v + 1) & 1][e] = (v + 1 < NGV) ? frag_load(vrs, loff, voff + ((v + 1) * 8 + e) * 1024)
                                                     : frag_load(krs, loff, koff + e * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
          for (int dd = 0; dd < 2; ++dd)
            o[2 * v + dd] = mfma32((&ring[(NGK + v) & 1][dd * 4 + (i >> 2)].x)[i & 3], s[i], o[2 * v + dd]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      voff += adv;
    }
    }
    l_run = half_sum(l_run);
    float4* po = part_o + (size_t)item * (DT * 4) * 64 + lane;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        nt_store4(po + (d * 4 + g) * 64, make_float4(o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]));
    if (h == 0) part_ml[(size_t)item * 32 + r] = make_float2(m_ref, l_run);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // see xattn_kernel
  }
}

// ctx[b, q, head*DH + c] = sum_s w_s O_s[c] / sum_s w_s l_s,   w_s = 2^{(m_s - M) kc2},  M = max_s m_s
template <int DH>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float4* __restrict__ part_o,
                                                           const float2* __restrict__ part_ml, float* __restrict__ ctx,
                                                           int NQ, int C, int heads, int chunks, int total_waves,
                                                           float kc2) {
  constexpr int DT = (DH + 31) / 32;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per (b, head, qtile)
  if (wid >= total_waves) return;
  const int nqt = (NQ + 31) / 32;
  const int qt = wid % nqt, bh = wid / nqt, head = bh % heads, b = bh / heads;
  const int qrow = qt * 32 + r;
  float w[16], M = -INFINITY;      // chunks <= 16
  for (int s = 0; s < chunks; ++s) {
    const float2 ml = part_ml[((size_t)(bh * chunks + s) * nqt + qt) * 32 + r];
    w[s] = ml.x;
    M = fmaxf(M, ml.x);
  }
  float den = 0.f;
  for (int s = 0; s < chunks; ++s) {
    const float2 ml = part_ml[((size_t)(bh * chunks + s) * nqt + qt) * 32 + r];
    w[s] = (w[s] == M) ? 1.0f : __builtin_amdgcn_exp2f((w[s] - M) * kc2);
    den = fmaf(w[s], ml.y, den);
  }
  if (qrow >= NQ) return;
  const int c0 = head * DH, vt0 = c0 / 32;
  float* out = ctx + ((size_t)b * NQ + qrow) * C;
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = 32 * (vt0 + d) + 8 * g + 4 * h;          // absolute channel of this float4
      if (ch < c0 || ch >= c0 + DH) continue;                  // head dims < 32 share a channel tile with other heads
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = 0; s < chunks; ++s) {
        const float4 p = nt_load4(part_o + (((size_t)(bh * chunks + s) * nqt + qt) * (DT * 4) + d * 4 + g) * 64 + lane);
        acc.x = fmaf(w[s], p.x, acc.x); acc.y = fmaf(w[s], p.y, acc.y);
        acc.z = fmaf(w[s], p.z, acc.z); acc.w = fmaf(w[s], p.w, acc.w);
      }
      *reinterpret_cast<float4*>(out + ch) = make_float4(acc.x / den, acc.y / den, acc.z / den, acc.w / den);
    }
}

// row-major (B*NK, ld) keys / values -> fragment images (op-level entry point and tests; the decoder's projection
// GEMM writes the images itself)
__global__ void attn_pack_k_kernel(const float* __restrict__ k, int ld, int C, float4* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const long f = i >> 6;
  const int kco = (int)(f % (C / 8));
  const long mt = f / (C / 8);
  const float* p = k + (size_t)(mt * 32 + (lane & 31)) * ld + 8 * kco + 4 * (lane >> 5);
  out[i] = make_float4(p[0], p[1], p[2], p[3]);
}

__global__ void attn_pack_v_kernel(const float* __restrict__ v, int ld, int C, float4* __restrict__ out, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const int g = (int)((i >> 6) & 3);
  const long f = i >> 8;
  const int vt = (int)(f % (C / 32));
  const long mt = f / (C / 32);
  const float* p = v + (size_t)(mt * 32 + 8 * g + 4 * (lane >> 5)) * ld + 32 * vt + (lane & 31);
  out[i] = make_float4(p[0], p[(size_t)ld], p[2 * (size_t)ld], p[3 * (size_t)ld]);
}

// key tiles per chunk: a function of the key count and head dim only (see the header)
static int attn_tiles_per_chunk(int NK, int dh) {
  const int nkt = NK / 32;
  (void)dh;
#ifdef POEM_LAB
  if (const char* e = getenv("POEM_ATTN_TPC")) { const int t = atoi(e); if (t > 0 && nkt % t == 0) return t; }
#endif
  if (nkt % 32 == 0 && nkt >= 64) return 32;
  return nkt;
}

// scratch floats: partial (O, m, l) of every item [+ the two fragment images when the caller passes row-major k, v]
extern "C" size_t poem_cross_attention_scratch_floats(int B, int NQ, int NK, int C, int heads, int with_images) {
  const int dh = C / heads;
  const int nqt = (NQ + 31) / 32;
  const int chunks = (NK / 32) / attn_tiles_per_chunk(NK, dh);
  const int DT = (dh + 31) / 32;
  const size_t items = (size_t)B * heads * chunks * nqt;
  size_t n = items * (size_t)DT * 4 * 64 * 4 + items * 32 * 2;
  n = (n + 63) / 64 * 64;
  if (with_images) n += 2 * (size_t)B * NK * C;
  return n;
}

static int poem_attn_cus() { return poem_device_cus(); }

// opt-in split precision for the calls enqueued while it is set (api.cpp: around poem_head_forward in SPLIT_F16X3_ALL mode,
// and by the operator-level entry point); head dims 32 and 64 only, the others keep the exact kernels
static std::atomic<int> g_xattn_half{1};      // A/B: channel-tile items of the merged kernel for a single sample (poem_set_option "xattn_half")
extern "C" void poem_cross_attention_half(int on) { g_xattn_half = on; }
static std::atomic<int> g_xattn_tail_halves{1};      // A/B: a launch's remainder items as channel-tile halves (poem_set_option "xattn_tail")
extern "C" void poem_cross_attention_tail_halves(int on) { g_xattn_tail_halves = on; }
static thread_local int g_xattn_split = 0;      // per host thread, like gemm.hip's split context.  1: split from fp32 images, 2: the images are already split (gemm.hip split output modes)
extern "C" void poem_cross_attention_split(int on) { g_xattn_split = on; }

// q (B, NQ, ldq) row-major; kimg / vimg: fragment images of the (B*NK, C) key / value matrices
// q_batch_rows: rows between consecutive samples' queries (NQ; 0 = every sample reads the same NQ query rows)
extern "C" hipError_t poem_launch_cross_attention_imgq(const float* q, int ldq, int q_batch_rows, const void* kimg,
                                                       const void* vimg, float* ctx, int B, int NQ, int NK, int C,
                                                       int heads, float* scratch, hipStream_t s);
// Where poem_launch_cross_attention_imgq leaves its split-key partials inside `scratch` (ctx == nullptr there skips the
// combine launch: the consumer -- chain.hip kind A -- combines them while it fills its activation tile).
extern "C" void poem_cross_attention_partials(int B, int NQ, int NK, int C, int heads, float* scratch, const void** part_o,
                                              const void** part_ml, int* chunks, float* kc2) {
  const int dh = C / heads, nqt = (NQ + 31) / 32, DT = (dh + 31) / 32;
  *chunks = (NK / 32) / attn_tiles_per_chunk(NK, dh);
  const size_t items = (size_t)B * heads * (size_t)*chunks * nqt;
  *part_o = scratch;
  *part_ml = scratch + items * (size_t)DT * 4 * 64 * 4;
  *kc2 = (float)(1.4426950408889634 / sqrt((double)dh));
}

extern "C" hipError_t poem_launch_cross_attention_img(const float* q, int ldq, const void* kimg, const void* vimg,
                                                      float* ctx, int B, int NQ, int NK, int C, int heads,
                                                      float* scratch, hipStream_t s) {
  return poem_launch_cross_attention_imgq(q, ldq, NQ, kimg, vimg, ctx, B, NQ, NK, C, heads, scratch, s);
}
extern "C" hipError_t poem_launch_cross_attention_imgq(const float* q, int ldq, int qbr, const void* kimg,
                                                       const void* vimg, float* ctx, int B, int NQ, int NK, int C,
                                                       int heads, float* scratch, hipStream_t s) {
  const int dh = C / heads;
  if (NK % 32 || C % 32 || (size_t)B * NK * C * 4 >= (1ull << 31)) return hipErrorInvalidValue;
  const int nqt = (NQ + 31) / 32;
  const int tpc = attn_tiles_per_chunk(NK, dh);
  const int chunks = (NK / 32) / tpc;
  if (chunks > 16) return hipErrorInvalidValue;
  const int DT = (dh + 31) / 32;
  const size_t items = (size_t)B * heads * chunks * nqt;
  float4* part_o = reinterpret_cast<float4*>(scratch);
  float2* part_ml = reinterpret_cast<float2*>(scratch + items * (size_t)DT * 4 * 64 * 4);
  const float kc2 = (float)(1.4426950408889634 / sqrt((double)dh));
  const float lazy_raw = POEM_ATTN_LAZY_LOG2 / kc2;
  const int cus = poem_attn_cus();
  const int grid = (int)std::min<size_t>((size_t)cus, (items + 3) / 4);
  const int waves = B * heads * nqt;
#define POEM_XATTN(D, WV)                                                                                         \
  hipLaunchKernelGGL((xattn_kernel<D, WV>), dim3(grid), dim3(256 * WV), 0, s, q, ldq, qbr, (const float4*)kimg,        \
                     (const float4*)vimg, part_o, part_ml, B, NQ, NK, C, heads, tpc, kc2, lazy_raw, map, prio_rot, (float*)nullptr); \
  if (ctx) hipLaunchKernelGGL((attn_combine_kernel<D>), dim3((waves + 3) / 4), dim3(256), 0, s, part_o, part_ml, ctx, NQ, C, \
                     heads, chunks, waves, kc2)
#define POEM_XSTREAM(D, WV)                                                                                        \
  {                                                                                                                \
    static std::atomic<unsigned long long> optin_{0};                                                              \
    if (hipError_t e_ = poem_optin_lds(reinterpret_cast<const void*>(xattn_stream_kernel<D, WV>), (size_t)4 * WV * (D / 8) * 1024, optin_); \
        e_ != hipSuccess) return e_;                                                                               \
  }                                                                                                                \
  hipLaunchKernelGGL((xattn_stream_kernel<D, WV>), dim3(grid), dim3(256 * WV), (size_t)4 * WV * (D / 8) * 1024, s, q, ldq, qbr, (const float4*)kimg,   \
                     (const float4*)vimg, part_o, part_ml, B, NQ, NK, C, heads, tpc, kc2, lazy_raw, map & 15);      \
  if (ctx) hipLaunchKernelGGL((attn_combine_kernel<D>), dim3((waves + 3) / 4), dim3(256), 0, s, part_o, part_ml, ctx, NQ, C, \
                     heads, chunks, waves, kc2)
  int wsel = 0, map = 2 | (g_xattn_tail_halves ? 16 : 0), prio_rot = 0;      // (map 2: two CUs of an XCD share an item range -- 27 % fewer HBM bytes, same time; bit 4: remainder items as halves)
  (void)wsel;
#ifdef POEM_LAB
  if (const char* e = getenv("POEM_ATTN_PRIO")) prio_rot = atoi(e);
  if (const char* e = getenv("POEM_ATTN_W")) wsel = atoi(e);
  if (const char* e = getenv("POEM_ATTN_MAP")) map = atoi(e);
#endif
#define POEM_XSPLIT(D, WV, PREV)                                                                                         \
  hipLaunchKernelGGL((xattn_split_kernel<D, WV, PREV>), dim3(grid), dim3(256 * WV), 0, s, q, ldq, qbr, (const float4*)kimg, \
                     (const float4*)vimg, part_o, part_ml, B, NQ, NK, C, heads, tpc, kc2, lazy_raw);                \
  if (ctx) hipLaunchKernelGGL((attn_combine_kernel<D>), dim3((waves + 3) / 4), dim3(256), 0, s, part_o, part_ml, ctx, NQ, C, \
                     heads, chunks, waves, kc2)
  if (g_xattn_split && (dh == 32 || dh == 64)) {
#ifdef POEM_LAB
    static const int w3 = getenv("POEM_XS_W") ? atoi(getenv("POEM_XS_W")) : 2;      // lab A/B: waves per SIMD, head dim 64
#else
    constexpr int w3 = 2;
#endif
    if (g_xattn_split == 2) { if (dh == 32) { POEM_XSPLIT(32, 3, true); } else if (w3 == 3) { POEM_XSPLIT(64, 3, true); } else { POEM_XSPLIT(64, 2, true); } }
    else { if (dh == 32) { POEM_XSPLIT(32, 3, false); } else { POEM_XSPLIT(64, 2, false); } }
    return hipGetLastError();
  }
#undef POEM_XSPLIT
  switch (dh) {
    case 8: POEM_XATTN(8, 4); break;
    case 16: POEM_XATTN(16, 4); break;
    case 32: POEM_XATTN(32, 4); break;
    case 64:
#ifdef POEM_LAB   // other waves-per-SIMD shapes for tools/lab only (W = 4 spills; W = 1, 2 are within 3 % of W = 3)
      if (wsel == 2) { POEM_XATTN(64, 2); break; }
      if (wsel == 1) { POEM_XATTN(64, 1); break; }
      if (wsel == 4) { POEM_XATTN(64, 4); break; }
#endif
      POEM_XATTN(64, 3);
      break;
    case 128: POEM_XSTREAM(128, 2); break;
    case 256: POEM_XSTREAM(256, 1); break;
    default:
      return hipErrorInvalidValue;
  }
#undef POEM_XATTN
#undef POEM_XSTREAM
  return hipGetLastError();
}

// The same attention with the split-key partials merged inside the kernel (xattn_kernel MERGE): ctx is written directly, no
// scratch.  hipErrorNotSupported for shapes other than head dim 64 with four key chunks of >= 8 tiles (the head path's
// 4096 keys): the caller uses poem_launch_cross_attention_imgq then.  Same bits as partials + attn_combine_kernel.
extern "C" int poem_cross_attention_merges(int NK, int C, int heads) {
  if (heads <= 0 || C % heads || NK % 32) return 0;
  const int dh = C / heads, tpc = attn_tiles_per_chunk(NK, dh);
  return dh == 64 && (NK / 32) / tpc == 4 && tpc >= 8;
}
extern "C" hipError_t poem_launch_cross_attention_merged(const float* q, int ldq, int qbr, const void* kimg, const void* vimg,
                                                         float* ctx, int B, int NQ, int NK, int C, int heads, hipStream_t s) {
  if (!poem_cross_attention_merges(NK, C, heads) || !ctx) return hipErrorNotSupported;
  if ((size_t)B * NK * C * 4 >= (1ull << 31)) return hipErrorInvalidValue;
  constexpr int DH = 64, WV = 3, DT = 2;
  const int tpc = attn_tiles_per_chunk(NK, DH), nqt = (NQ + 31) / 32;
  const float kc2 = (float)(1.4426950408889634 / sqrt((double)DH));
  const float lazy_raw = POEM_ATTN_LAZY_LOG2 / kc2;
  const size_t lds = (size_t)4 * WV * (DT * 4 * 64 * 4 + 64) * sizeof(float);
  const long items = (long)B * heads * nqt;
  if (g_xattn_half && items * DT <= poem_attn_cus()) {      // one under-filled round: channel-tile items (xattn_kernel HALF)
    auto kh = xattn_kernel<DH, WV, true, true>;
    static std::atomic<unsigned long long> optin_h{0};
    if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kh), lds, optin_h); e != hipSuccess) return e;
    hipLaunchKernelGGL(kh, dim3((unsigned)(items * DT)), dim3(256 * WV), lds, s, q, ldq, qbr, (const float4*)kimg, (const float4*)vimg,
                       (float4*)nullptr, (float2*)nullptr, B, NQ, NK, C, heads, tpc, kc2, lazy_raw, 1, 0, ctx);
    return hipGetLastError();
  }
  auto kern = xattn_kernel<DH, WV, true>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), lds, optin); e != hipSuccess) return e;
  const int grid = (int)std::min<long>(poem_attn_cus(), items);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * WV), lds, s, q, ldq, qbr, (const float4*)kimg, (const float4*)vimg,
                     (float4*)nullptr, (float2*)nullptr, B, NQ, NK, C, heads, tpc, kc2, lazy_raw, 1, 0, ctx);
  return hipGetLastError();
}

// row-major k, v (B*NK rows, ldkv) -> images in scratch -> kernel above
extern "C" hipError_t poem_launch_cross_attention(const float* q, const float* k, const float* v, float* ctx, int B,
                                                  int NQ, int NK, int C, int heads, int ldkv, float* scratch,
                                                  hipStream_t s) {
  if (NK % 32 || C % 32) return hipErrorInvalidValue;
  const size_t part = poem_cross_attention_scratch_floats(B, NQ, NK, C, heads, 0);
  float4* kimg = reinterpret_cast<float4*>(scratch + part);
  float4* vimg = kimg + (size_t)B * NK * C / 4;
  const long total = (long)B * NK * C / 4;
  hipLaunchKernelGGL(attn_pack_k_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k, ldkv, C, kimg, total);
  hipLaunchKernelGGL(attn_pack_v_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, v, ldkv, C, vimg, total);
  return poem_launch_cross_attention_img(q, C, kimg, vimg, ctx, B, NQ, NK, C, heads, scratch, s);
}

// row-major k, v -> images in scratch -> the merged kernel (operator-level entry point of MERGE, for the tests)
extern "C" hipError_t poem_launch_cross_attention_merged_rm(const float* q, const float* k, const float* v, float* ctx, int B,
                                                            int NQ, int NK, int C, int heads, float* scratch, hipStream_t s) {
  if (!poem_cross_attention_merges(NK, C, heads)) return hipErrorNotSupported;
  const size_t part = poem_cross_attention_scratch_floats(B, NQ, NK, C, heads, 0);
  float4* kimg = reinterpret_cast<float4*>(scratch + part);
  float4* vimg = kimg + (size_t)B * NK * C / 4;
  const long total = (long)B * NK * C / 4;
  hipLaunchKernelGGL(attn_pack_k_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k, C, C, kimg, total);
  hipLaunchKernelGGL(attn_pack_v_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, v, C, C, vimg, total);
  return poem_launch_cross_attention_merged(q, C, NQ, kimg, vimg, ctx, B, NQ, NK, C, heads, s);
}

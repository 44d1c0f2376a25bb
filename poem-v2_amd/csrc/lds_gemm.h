// LDS-resident row-tile GEMM shared by the chain kernels (chain.hip) and the fused sampling + merge kernels (merge.hip):
// activations X[channel][row] in LDS (row stride XSP), packed weight fragments (common.h "fragment order") as the A operand
// streamed from L2 through a buffer descriptor, an X row as the B operand.
#pragma once
#include "common.h"

namespace {

// D[c'][j] (+)= sum_k W[c'][k0 + k] X[k][j] for this wave's TPW tiles; KCH = K / 8 chunks, tile t of the image starts at
// t * tile_stride bytes, the contraction at byte k0.  Same two-stage software pipeline as vecattn.hip's chain_gemm.
// DEEP: weight fragments seven chunks ahead instead of three (small tiles of the chain kernels; see below).
template <int KCH, int XSP, int P, int TPW, bool INIT0, bool DEEP = false>
__device__ __forceinline__ void lds_gemm(const __amdgpu_buffer_rsrc_t wrs, int wbase, int tile_stride, const float* __restrict__ X,
                                         f32x16 (&acc)[TPW][P], int lane) {
  static_assert(KCH % 2 == 0, "K must be a multiple of 16");
  const int j = lane & 31, h = lane >> 5;
  const int loff = lane * 16;
  const float* xc = X + (4 * h) * XSP + j;
  float4 a0[TPW], a1[TPW];
  float xa[P], xb[P];
#define CH_LOADW(A, KCI)                                                                  \
  {                                                                                       \
    const int kq_ = min((KCI), KCH - 1);                                                  \
    _Pragma("unroll") for (int tp = 0; tp < TPW; ++tp) A[tp] = frag_load(wrs, loff, wbase + tp * tile_stride + kq_ * 1024); \
  }
#define CH_READX(XR, KCI, T)                                                              \
  {                                                                                       \
    const int kq_ = min((KCI), KCH - 1);                                                  \
    _Pragma("unroll") for (int p = 0; p < P; ++p) XR[p] = xc[(kq_ * 8 + (T)) * XSP + 32 * p]; \
  }
#define CH_MMA(A, T, XR, INIT)                                                            \
  _Pragma("unroll") for (int tp = 0; tp < TPW; ++tp) {                                    \
    const float av = (&A[tp].x)[T];                                                       \
    _Pragma("unroll") for (int p = 0; p < P; ++p) {                                       \
      const f32x16 c_ = (INIT) ? zero16() : acc[tp][p];                                   \
      acc[tp][p] = mfma32(av, XR[p], c_);                                                 \
    }                                                                                     \
  }
#define CH_CHUNK(A, KCI, INIT)                                                            \
  CH_READX(xb, KCI, 1) __builtin_amdgcn_sched_barrier(0); CH_MMA(A, 0, xa, INIT) __builtin_amdgcn_sched_barrier(0); \
  CH_READX(xa, KCI, 2) __builtin_amdgcn_sched_barrier(0); CH_MMA(A, 1, xb, false) __builtin_amdgcn_sched_barrier(0); \
  CH_READX(xb, KCI, 3) __builtin_amdgcn_sched_barrier(0); CH_MMA(A, 2, xa, false) __builtin_amdgcn_sched_barrier(0); \
  CH_READX(xa, (KCI) + 1, 0) __builtin_amdgcn_sched_barrier(0); CH_MMA(A, 3, xb, false) __builtin_amdgcn_sched_barrier(0);
  if constexpr (TPW * P <= 2) {
    // Few MFMAs per k-step (TPW * P <= 2: 128 cycles): an X operand requested one step ahead is not back when its MFMAs
    // are due (the ISA showed an s_waitcnt lgkmcnt in front of nearly every MFMA pair).  Four operand registers per column
    // tile, one per k-step of a chunk; each is re-requested for the NEXT chunk right behind the MFMAs that read it: a whole
    // chunk (4 k-steps) of MFMAs between an LDS read and its use.
    float x0[P], x1[P], x2[P], x3[P];
#define CH_CHUNK4(A, KCI, INIT)                                                             \
    CH_MMA(A, 0, x0, INIT) __builtin_amdgcn_sched_barrier(0); CH_READX(x0, (KCI) + 1, 0) __builtin_amdgcn_sched_barrier(0);  \
    CH_MMA(A, 1, x1, false) __builtin_amdgcn_sched_barrier(0); CH_READX(x1, (KCI) + 1, 1) __builtin_amdgcn_sched_barrier(0); \
    CH_MMA(A, 2, x2, false) __builtin_amdgcn_sched_barrier(0); CH_READX(x2, (KCI) + 1, 2) __builtin_amdgcn_sched_barrier(0); \
    CH_MMA(A, 3, x3, false) __builtin_amdgcn_sched_barrier(0); CH_READX(x3, (KCI) + 1, 3) __builtin_amdgcn_sched_barrier(0);
    if constexpr (DEEP && KCH % 8 == 0) {
      // one or two MFMAs per k-step: a chunk is 256 / 512 cycles of MFMAs, far less than an L2 round trip -- weight fragments
      // SEVEN chunks ahead (ring of eight; round 3: three ahead).  A chain tile on a small batch is a latency chain -- one tile
      // per CU, nothing else to switch to -- and with three chunks in flight every GEMM phase waited ~0.2 us per chunk for its
      // weights (chain kind D2 at a batch of 2: 97 us for 17 us of MFMAs).  32 VGPRs at TPW = 1.
      float4 ar[8][TPW];
#define CH_LOADR(D, KCI)                                                                  \
      {                                                                                   \
        const int kq_ = min((KCI), KCH - 1);                                              \
        _Pragma("unroll") for (int tp = 0; tp < TPW; ++tp) ar[D][tp] = frag_load(wrs, loff, wbase + tp * tile_stride + kq_ * 1024); \
      }
#pragma unroll
      for (int d = 0; d < 7; ++d) CH_LOADR(d, d)
      CH_READX(x0, 0, 0) CH_READX(x1, 0, 1) CH_READX(x2, 0, 2) CH_READX(x3, 0, 3)
      __builtin_amdgcn_sched_barrier(0);
#define CH_STEP8(D, KCI, INIT)                                                            \
      CH_LOADR(((D) + 7) & 7, (KCI) + 7) __builtin_amdgcn_sched_barrier(0); CH_CHUNK4(ar[D], (KCI), INIT)
      CH_STEP8(0, 0, INIT0) CH_STEP8(1, 1, false) CH_STEP8(2, 2, false) CH_STEP8(3, 3, false)
      CH_STEP8(4, 4, false) CH_STEP8(5, 5, false) CH_STEP8(6, 6, false) CH_STEP8(7, 7, false)
      for (int kc = 8; kc < KCH; kc += 8) {
        CH_STEP8(0, kc, false) CH_STEP8(1, kc + 1, false) CH_STEP8(2, kc + 2, false) CH_STEP8(3, kc + 3, false)
        CH_STEP8(4, kc + 4, false) CH_STEP8(5, kc + 5, false) CH_STEP8(6, kc + 6, false) CH_STEP8(7, kc + 7, false)
      }
#undef CH_STEP8
#undef CH_LOADR
      return;
    }
    if constexpr (KCH % 4 == 0) {
      // one or two MFMAs per k-step: a chunk is 256 / 512 cycles of MFMAs, less than an L2 round trip -- weight fragments
      // three chunks ahead (ring of four) instead of one; no extra registers at the kernels' peaks
      float4 a2[TPW], a3[TPW];
      CH_LOADW(a0, 0)
      CH_READX(x0, 0, 0) CH_READX(x1, 0, 1) CH_READX(x2, 0, 2) CH_READX(x3, 0, 3)
      CH_LOADW(a1, 1)
      CH_LOADW(a2, 2)
      __builtin_amdgcn_sched_barrier(0);
#define CH_GROUP(KC0, INIT)                                                                     \
      CH_LOADW(a3, (KC0) + 3) __builtin_amdgcn_sched_barrier(0); CH_CHUNK4(a0, (KC0), INIT)      \
      CH_LOADW(a0, (KC0) + 4) __builtin_amdgcn_sched_barrier(0); CH_CHUNK4(a1, (KC0) + 1, false) \
      CH_LOADW(a1, (KC0) + 5) __builtin_amdgcn_sched_barrier(0); CH_CHUNK4(a2, (KC0) + 2, false) \
      CH_LOADW(a2, (KC0) + 6) __builtin_amdgcn_sched_barrier(0); CH_CHUNK4(a3, (KC0) + 3, false)
      CH_GROUP(0, INIT0)
      for (int kc = 4; kc < KCH; kc += 4) { CH_GROUP(kc, false) }
#undef CH_GROUP
      return;
    }
    CH_LOADW(a0, 0)
    CH_READX(x0, 0, 0) CH_READX(x1, 0, 1) CH_READX(x2, 0, 2) CH_READX(x3, 0, 3)
    CH_LOADW(a1, 1)
    __builtin_amdgcn_sched_barrier(0);
    CH_CHUNK4(a0, 0, INIT0)
    CH_LOADW(a0, 2)
    __builtin_amdgcn_sched_barrier(0);
    CH_CHUNK4(a1, 1, false)
    for (int kc = 2; kc < KCH; kc += 2) {
      CH_LOADW(a1, kc + 1)
      __builtin_amdgcn_sched_barrier(0);
      CH_CHUNK4(a0, kc, false)
      CH_LOADW(a0, kc + 2)
      __builtin_amdgcn_sched_barrier(0);
      CH_CHUNK4(a1, kc + 1, false)
    }
#undef CH_CHUNK4
    return;
  }
  CH_LOADW(a0, 0)
  CH_READX(xa, 0, 0)
  CH_LOADW(a1, 1)
  __builtin_amdgcn_sched_barrier(0);
  CH_CHUNK(a0, 0, INIT0)
  CH_LOADW(a0, 2)
  __builtin_amdgcn_sched_barrier(0);
  CH_CHUNK(a1, 1, false)
  for (int kc = 2; kc < KCH; kc += 2) {
    CH_LOADW(a1, kc + 1)
    __builtin_amdgcn_sched_barrier(0);
    CH_CHUNK(a0, kc, false)
    CH_LOADW(a0, kc + 2)
    __builtin_amdgcn_sched_barrier(0);
    CH_CHUNK(a1, kc + 1, false)
  }
#undef CH_LOADW
#undef CH_READX
#undef CH_MMA
#undef CH_CHUNK
}

}  // namespace

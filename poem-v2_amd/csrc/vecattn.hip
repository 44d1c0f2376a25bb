// Fused Point-Transformer vector attention over K = 32 neighbours (the path's dominant cost):
//   pos_ij = W_d2 relu(W_d1 (xyz_i - nxyz_j) + b_d1) + b_d2
//   a_ij   = W_g2 relu(W_g1 (q_i - k_j + pos_ij) + b_g1) + b_g2
//   r_i    = sum_j softmax_j(a_ij / sqrt(C)) * (v_j + pos_ij)          (softmax per channel over the 32 neighbours)
// (point_transformers.py:88-95 / :144-151 upstream; fc1/w_k/w_v are applied to the source rows beforehand.)
//
// One block (NW waves) owns P queries = P*32 neighbour columns.  The three C x C per-neighbour GEMMs are chained
// through ONE LDS activation buffer X[channel][column] without ever touching HBM, in "transposed" orientation:
//   D[c'][j] = sum_c W[c'][c] * X[c][j]     A = packed weight fragment (global/L2), B = X row read (lane = column j)
// whose result layout (lane = column, registers = channels) is exactly what is written back to X for the next
// GEMM.  The last GEMM is issued with the operands swapped (A = X, B = W) so its result is D[j][c'] -- lane =
// channel, registers = neighbours -- which makes the softmax over the 32 neighbours register-local (+1 exchange
// between half-waves) and the v_j gathers coalesced.  Wave w owns output channel tiles [w*TPW, (w+1)*TPW).
#include "common.h"
#ifndef POEM_VA_XCD_SAMPLES
#define POEM_VA_XCD_SAMPLES 1
#endif
#include <algorithm>
#include <cstdlib>

#ifdef POEM_VA_DBG   // tools/lab only: per-phase cycle stamps of a few blocks
__device__ long long va_dbg[64 * 4 * 8];
#define VA_STAMP(k) do { if (item >= 6000 && item < 6064 && lane == 0) va_dbg[((item - 6000) * 4 + wv) * 8 + (k)] = clock64(); } while (0)
#else
#define VA_STAMP(k) do { } while (0)
#endif

struct VecAttnArgs {
  const float* query_xyz;   // (B,Q,3)
  const float* src_xyz;     // (B,NS,3) or null when anchor_xyz given
  const float* anchor_xyz;  // (32,3) or null
  const int* idx;           // (B,Q,32) or (32) when shared_idx
  int shared_idx;
  const float* q;           // (B,Q,C)
  const float* k;           // (B,NS,C)
  const float* v;           // (B,NS,C)
  int NS;
  const float* wd1;         // (C,3)
  const float* bd1;
  const float4* wd2;        // packed (C,C)
  const float* bd2;
  const float4* wg1;
  const float* bg1;
  const float4* wg2;
  const float* bg2;
  float* out;               // (B,Q,C)
  int B, Q;
  int ldq, ldk, ldv;        // row strides (floats) of q, k, v: they may be column blocks of a fused projection
  int stagger;              // > 0: persistent launch, second block of each CU starts `stagger` cycles late
  int composed;             // 1: q, k are W_g1 q + (W_g1 b_d2 + b_g1) and W_g1 k, `wg1` is W_g1 W_d2 (see vecattn_kernel)
  float4* tab_g;            // MODE 1 writes / MODE 2 reads: (W_g1 W_d2) h_ij per (query, anchor), C/D fragment images
  float4* tab_p;            // MODE 1 writes / MODE 2 reads: pos_ij = W_d2 h_ij + b_d2, transposed (lane = channel) images
  int kvalid;               // MODE 3: the first `kvalid` of the 32 neighbour columns count (N_NEIGHBOR / N_NEIGHBOR_QUERY < 32)
};

// arrival parity per CU (key: XCC id, HW_ID[15:8]); atomicInc wraps 0 -> 1 -> 0, so the table resets itself when
// every CU hosts two blocks.  Used for speed only: a wrong parity costs overlap, never correctness.
__device__ unsigned int va_cu_slot[8 * 256];

template <int C, int P, int NW, int TPW, bool FLIP, bool INIT0 = true>
__device__ __forceinline__ void chain_gemm(const float4* __restrict__ Wp, const float* __restrict__ X,
                                           f32x16 (&acc)[TPW][P], int wv, int lane) {
  constexpr int KC = C / 8;      // even for every supported C
  constexpr int XS = 32 * P;
  const int j = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t wrs = frag_rsrc(Wp, (unsigned)(C * C * 4));
  const int wbase = __builtin_amdgcn_readfirstlane(wv) * TPW * KC * 1024;      // bytes, wave-uniform
  const int loff = lane * 16;
  const float* xc = X + (4 * h) * XS + j;
  // Software pipeline with pinned order (sched_barrier): the weight fragments of chunk kc+1 are in flight while the MFMAs
  // of chunk kc issue, and the LDS operands run a whole chunk ahead (below).  Left to itself hipcc sinks every load to
  // its first use and the wave stalls on L2/LDS latency in front of each MFMA group.
  float4 a0[TPW], a1[TPW];
#define VA_LOADW(A, KCI)                                                                  \
  {                                                                                       \
    const int kq_ = min((KCI), KC - 1);                                                   \
    _Pragma("unroll") for (int tp = 0; tp < TPW; ++tp) A[tp] = frag_load(wrs, loff, wbase + (tp * KC + kq_) * 1024); \
  }
#define VA_READX(XR, KCI, T)                                                              \
  {                                                                                       \
    const int kq_ = min((KCI), KC - 1);                                                   \
    _Pragma("unroll") for (int p = 0; p < P; ++p) XR[p] = xc[(kq_ * 8 + (T)) * XS + 32 * p]; \
  }
  // INIT: the very first k-step takes its C operand from the inline constant 0 -- the accumulators are never zeroed
  // with v_mov (64 VALU instructions per GEMM that would come straight out of the matrix pipe's time)
#define VA_MMA(A, T, XR, INIT)                                                            \
  _Pragma("unroll") for (int tp = 0; tp < TPW; ++tp) {                                    \
    const float av = (&A[tp].x)[T];                                                       \
    _Pragma("unroll") for (int p = 0; p < P; ++p) {                                       \
      const f32x16 c_ = (INIT) ? zero16() : acc[tp][p];                                   \
      acc[tp][p] = FLIP ? mfma32(XR[p], av, c_) : mfma32(av, XR[p], c_);                  \
    }                                                                                     \
  }
  // X operands a whole chunk ahead (one register set per k-step of a chunk, re-requested right behind the MFMAs that read it)
  float x0[P], x1[P], x2[P], x3[P];
#define VA_CHUNK4(A, KCI, INIT)                                                             \
  VA_MMA(A, 0, x0, INIT) __builtin_amdgcn_sched_barrier(0); VA_READX(x0, (KCI) + 1, 0) __builtin_amdgcn_sched_barrier(0);  \
  VA_MMA(A, 1, x1, false) __builtin_amdgcn_sched_barrier(0); VA_READX(x1, (KCI) + 1, 1) __builtin_amdgcn_sched_barrier(0); \
  VA_MMA(A, 2, x2, false) __builtin_amdgcn_sched_barrier(0); VA_READX(x2, (KCI) + 1, 2) __builtin_amdgcn_sched_barrier(0); \
  VA_MMA(A, 3, x3, false) __builtin_amdgcn_sched_barrier(0); VA_READX(x3, (KCI) + 1, 3) __builtin_amdgcn_sched_barrier(0);
  if constexpr (TPW <= 2) {
  // weight fragments three chunks ahead (ring of four; 16 more VGPRs at TPW = 2 -- wider waves keep the ring of two)
  static_assert(KC % 4 == 0, "C must be a multiple of 32");
  float4 a2[TPW], a3[TPW];
  VA_LOADW(a0, 0)
  VA_READX(x0, 0, 0) VA_READX(x1, 0, 1) VA_READX(x2, 0, 2) VA_READX(x3, 0, 3)
  VA_LOADW(a1, 1)
  VA_LOADW(a2, 2)
  __builtin_amdgcn_sched_barrier(0);
#define VA_GROUP(KC0, INIT)                                                   \
  VA_LOADW(a3, (KC0) + 3) __builtin_amdgcn_sched_barrier(0); VA_CHUNK4(a0, (KC0), INIT)      \
  VA_LOADW(a0, (KC0) + 4) __builtin_amdgcn_sched_barrier(0); VA_CHUNK4(a1, (KC0) + 1, false) \
  VA_LOADW(a1, (KC0) + 5) __builtin_amdgcn_sched_barrier(0); VA_CHUNK4(a2, (KC0) + 2, false) \
  VA_LOADW(a2, (KC0) + 6) __builtin_amdgcn_sched_barrier(0); VA_CHUNK4(a3, (KC0) + 3, false)
  VA_GROUP(0, INIT0)
  for (int kc = 4; kc < KC; kc += 4) { VA_GROUP(kc, false) }
#undef VA_GROUP
  } else {
  VA_LOADW(a0, 0)
  VA_READX(x0, 0, 0) VA_READX(x1, 0, 1) VA_READX(x2, 0, 2) VA_READX(x3, 0, 3)
  VA_LOADW(a1, 1)
  __builtin_amdgcn_sched_barrier(0);
  VA_CHUNK4(a0, 0, INIT0)
  VA_LOADW(a0, 2)
  __builtin_amdgcn_sched_barrier(0);
  VA_CHUNK4(a1, 1, false)
  for (int kc = 2; kc < KC; kc += 2) {
    VA_LOADW(a1, kc + 1)
    __builtin_amdgcn_sched_barrier(0);
    VA_CHUNK4(a0, kc, false)
    VA_LOADW(a0, kc + 2)
    __builtin_amdgcn_sched_barrier(0);
    VA_CHUNK4(a1, kc + 1, false)
  }
  }
#undef VA_CHUNK4
#undef VA_LOADW
#undef VA_READX
#undef VA_MMA
}

// COMP (the decoder's form): W_g1 is linear, so W_g1 (q_i - k_j + pos_ij) + b_g1 = qg_i - kg_j + (W_g1 W_d2) h_ij with
// qg = W_g1 q + (W_g1 b_d2 + b_g1) and kg = W_g1 k -- both composed into the projections that produce q and k anyway
// (api.cpp), W_g1 W_d2 composed once at handle creation.  GEMM 2 then reads the SAME activations h as GEMM 1: the two
// run back to back with no LDS round trip and no barrier between them, and the gathered kg_j rows simply wait in GEMM 2's
// accumulator registers (acc = qg_i - kg_j before its first k-step).
//
// MODE (composed form only) -- the first decoder block's neighbours are the 32 fixed anchors for every query of every
// sample (quirk Q2, point_transformers.py:10-32 upstream) and its query coordinates are the hand template, so h_ij,
// pos_ij and (W_g1 W_d2) h_ij do not depend on the sample:
//   MODE 1  one pass over the Q queries (B = 1) that stops after GEMM 2 and writes both products as register-fragment
//           images (1 KiB coalesced stores): tab_g = (W_g1 W_d2) h in the C/D layout GEMM 2 leaves it in, tab_p = pos
//           already transposed to the softmax epilogue's layout (lane = channel, registers = neighbours);
//   MODE 2  per sample: g = relu(qg_i - kg_j + tab_g) -> X, GEMM 3, softmax-sum with pos from tab_p -- one C x C GEMM per
//           neighbour column instead of three, no coordinate stage, no transpose scratch.  Items are dealt so that the
//           blocks an XCD runs back to back share a query group (its table tiles stay in that XCD's L2).
//   MODE 3  = MODE 0 for neighbour counts below 32 (the reference's N_NEIGHBOR / N_NEIGHBOR_QUERY keys, ptEmb_transformer.py:30-31
//           upstream; every release config sets 32): the search returns its 32 nearest in ascending order, the tile keeps its 32
//           columns, and the softmax takes the columns from `kvalid` on as -inf -- weight exactly 0 in the sum and in the weighted
//           sum.  The 32-neighbour kernels are untouched by it.
template <int C, int P, int NW, int MINW, bool COMP, int MODE = 0>
__global__ __launch_bounds__(NW * 64, MINW) void vecattn_kernel(VecAttnArgs A) {
  static_assert(MODE == 0 || COMP, "table modes exist for the composed form only");
  constexpr bool MASK = MODE == 3;
  constexpr int NTILE = C / 32;
  constexpr int TPW = C / 32 / NW;
  constexpr int XS = 32 * P;
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X = smem;                                   // C * XS
  float* dl = smem + C * XS;                         // P*32*3 coordinate deltas
  int* sidx = reinterpret_cast<int*>(dl + P * 32 * 3);  // P*32 neighbour row ids
  int* voffs = sidx + P * 32;                           // P*32 byte offsets of the neighbours' v rows
  float* qs = reinterpret_cast<float*>(voffs + P * 32); // P*C query rows

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, h = lane >> 5;
  const int groups = (A.Q + P - 1) / P;
  const int total = A.B * groups;
  const float inv_sqrt_c = 1.0f / sqrtf((float)C);   // exact for C in {64, 256, 1024}; <= 1 ulp from the division otherwise

  if (A.stagger > 0) {
    // Two blocks share a CU (LDS-limited).  Left alone they run in lockstep -- both in their MFMA-free phases at the
    // same time -- so the second arrival on each CU starts half a period late: its prologue/epilogue phases then fall
    // inside the partner's GEMM phases and the matrix pipe never idles.
    if (tid == 0) {
      const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
      const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));    // HW_REG_XCC_ID[3:0]
      const unsigned key = ((xcc & 7u) << 8) | ((hw >> 8) & 0xffu);
      sidx[0] = (int)atomicInc(&va_cu_slot[key], 1u);
    }
    __syncthreads();
    if (sidx[0] & 1) {
      const long long t0 = clock64();
      while (clock64() - t0 < (long long)A.stagger) __builtin_amdgcn_s_sleep(64);
    }
  }

  for (int item = blockIdx.x; item < total; item += gridDim.x) {
  int b = item / groups, ig = item % groups;
  if ((MODE == 0 || MODE == 3) && POEM_VA_XCD_SAMPLES && gridDim.x == (unsigned)total && (total & 7) == 0) {
    // block ids go round-robin over the 8 XCDs: XCD x takes the x-th eighth of the (sample, query group) list, i.e. whole
    // samples -- a sample's key / value rows (8 MB at S = 4096, C = 256) are then gathered through ONE L2 instead of all eight
    const int it = (item & 7) * (total >> 3) + (item >> 3);
    b = it / groups; ig = it % groups;
  }
  if (MODE == 2) {
    if (groups % 8 == 0) {   // block ids go round-robin over the 8 XCDs: XCD x walks groups x, x + 8, ... sample by sample
      const int r = item >> 3;
      ig = (r / A.B) * 8 + (item & 7);
      b = r % A.B;
    } else {
      ig = item / A.B;
      b = item % A.B;
    }
  }
  const int i0 = ig * P;
  __syncthreads();   // previous item's epilogue still reads sidx / the scratch inside X

  VA_STAMP(0);
  // ---- stage 0: neighbour ids, coordinate deltas, first-layer activations h = relu(W_d1 delta + b_d1) -> X
  if (tid < 32 * P) {
    const int p = tid >> 5, jj = tid & 31;
    const int qi = min(i0 + p, A.Q - 1);
#ifdef POEM_VA_NOLOADS0   // tools/lab only: upper bound of what prefetching stage 0's dependent loads could buy
    const int id = (qi * 37 + jj * 101) % A.NS;
    dl[tid * 3 + 0] = 0.01f * (float)jj; dl[tid * 3 + 1] = 0.02f * (float)(qi & 7); dl[tid * 3 + 2] = 0.03f;
    (void)0;
#else
    const int id = A.shared_idx ? A.idx[jj] : A.idx[((size_t)b * A.Q + qi) * 32 + jj];
    if (MODE != 2) {
      const float* qx = A.query_xyz + ((size_t)b * A.Q + qi) * 3;
      const float* nx = A.anchor_xyz ? A.anchor_xyz + jj * 3 : A.src_xyz + ((size_t)b * A.NS + id) * 3;
      dl[tid * 3 + 0] = qx[0] - nx[0];
      dl[tid * 3 + 1] = qx[1] - nx[1];
      dl[tid * 3 + 2] = qx[2] - nx[2];
    }
#endif
    sidx[tid] = id;
    voffs[tid] = (int)(((unsigned)b * (unsigned)A.NS + (unsigned)id) * (unsigned)(A.ldv * 4));
  }
  if (MODE != 1)
  for (int f = tid; f < P * C / 4; f += NT) {           // query rows -> LDS (read back as broadcasts in epilogue 1)
    const int p = f / (C / 4), c4 = f % (C / 4);
    const int qi = min(i0 + p, A.Q - 1);
    reinterpret_cast<float4*>(qs)[f] = *reinterpret_cast<const float4*>(A.q + ((size_t)b * A.Q + qi) * A.ldq + 4 * c4);
  }
  __syncthreads();
  // h = relu(W_d1 delta + b_d1) as two MFMA k-steps per tile (K = 3 zero-padded to 4): the same k-ordered fma chain
  // fma(dz, w2, fma(dy, w1, dx * w0)) as a scalar loop, with the result already in the layout X wants
  // (lane = column, registers = channels; conflict-free stores).
  if (MODE != 2) {
    float db0[P], db1[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      db0[p] = dl[(p * 32 + j) * 3 + h];                    // k-step 0: (dx | dy) by half-wave
      db1[p] = h == 0 ? dl[(p * 32 + j) * 3 + 2] : 0.f;     // k-step 1: (dz | 0)
    }
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      const int crow = (wv * TPW + tp) * 32 + j;            // A operand: lane = channel row of the tile
      const float wa0 = A.wd1[crow * 3 + h];
      const float wa1 = h == 0 ? A.wd1[crow * 3 + 2] : 0.f;
      const int cbase = (wv * TPW + tp) * 32 + 4 * h;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        f32x16 hh = zero16();
        hh = mfma32(wa0, db0[p], hh);
        hh = mfma32(wa1, db1[p], hh);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bb = *reinterpret_cast<const float4*>(A.bd1 + cbase + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const int i = 4 * g + e;
            const f32x2 v = f32x2{hh[i], hh[i + 1]} + f32x2{(&bb.x)[e], (&bb.x)[e + 1]};
            X[((wv * TPW + tp) * 32 + mfma_row(i, h)) * XS + 32 * p + j] = fmaxf(v[0], 0.f);
            X[((wv * TPW + tp) * 32 + mfma_row(i + 1, h)) * XS + 32 * p + j] = fmaxf(v[1], 0.f);
          }
        }
      }
    }
    __syncthreads();
  }

  f32x16 acc[TPW][P], pos[TPW][P];   // initialised by the first k-step of a GEMM (chain_gemm, INIT0) unless stated

  // k_j (COMP: kg_j) for this lane's (channel tile, neighbour) cells: 16-byte row gathers issued BEFORE the first GEMM
  // so their L2/HBM latency hides under its MFMAs.  They wait in registers the first GEMM does not touch: `pos` in the
  // plain form (GEMM 1 accumulates into acc), `acc` in the composed form (GEMM 1 accumulates into pos).
  if (MODE != 1)
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
        for (int e = 0; e < 4; ++e) (COMP ? acc : pos)[tp][p][4 * g + e] = (&kk.x)[e];
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  VA_STAMP(1);
  if (MODE == 2) {
    // ---- table mode: g = relu((qg_i - kg_j) + tab_g) -> X ; pos fragments (already in the epilogue's layout) -> `pos`
    const float4* tg = A.tab_g + ((size_t)ig * NTILE + wv * TPW) * (P * 4 * 64) + lane;
    const float4* tq = A.tab_p + ((size_t)ig * NTILE + wv * TPW) * (P * 4 * 64) + lane;
    float4 gf[TPW][P][4];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int g = 0; g < 4; ++g) gf[tp][p][g] = tg[((tp * P + p) * 4 + g) * 64];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 t = tq[((tp * P + p) * 4 + g) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) pos[tp][p][4 * g + e] = (&t.x)[e];
        }
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      const int cbase = (wv * TPW + tp) * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float4 qq = *reinterpret_cast<const float4*>(qs + p * C + cbase + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const int i = 4 * g + e;
            const f32x2 tv = (f32x2{(&qq.x)[e], (&qq.x)[e + 1]} - f32x2{acc[tp][p][i], acc[tp][p][i + 1]}) +
                             f32x2{(&gf[tp][p][g].x)[e], (&gf[tp][p][g].x)[e + 1]};
            X[((wv * TPW + tp) * 32 + mfma_row(i, h)) * XS + 32 * p + j] = fmaxf(tv[0], 0.f);
            X[((wv * TPW + tp) * 32 + mfma_row(i + 1, h)) * XS + 32 * p + j] = fmaxf(tv[1], 0.f);
          }
        }
    }
    __syncthreads();
  } else if (COMP) {
    // ---- GEMM 1: pos = W_d2 h + b_d2, with the operands SWAPPED (as GEMM 3): the composed form needs pos only in the
    // softmax-sum at the end, whose layout is lane = channel, registers = neighbours -- exactly what the swapped product
    // leaves, so the [c'][j] -> [j][c'] transpose through LDS (32 LDS instructions per tile) disappears.  Same products, same
    // k order: the same values.
    chain_gemm<C, P, NW, TPW, true>(A.wd2, X, pos, wv, lane);
    VA_STAMP(2);
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      const float bch = A.bd2[(wv * TPW + tp) * 32 + j];       // this lane's channel
      const f32x2 bb2 = {bch, bch};
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const f32x2 pv = f32x2{pos[tp][p][i], pos[tp][p][i + 1]} + bb2;
          pos[tp][p][i] = pv[0]; pos[tp][p][i + 1] = pv[1];
        }
    }
    if (MODE != 1) {
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        const int cbase = (wv * TPW + tp) * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const float4 qq = *reinterpret_cast<const float4*>(qs + p * C + cbase + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {      // register pairs -> v_pk_add_f32
              const int i = 4 * g + e;
              const f32x2 tv = f32x2{(&qq.x)[e], (&qq.x)[e + 1]} - f32x2{acc[tp][p][i], acc[tp][p][i + 1]};
              acc[tp][p][i] = tv[0]; acc[tp][p][i + 1] = tv[1];
            }
          }
      }
    }
    VA_STAMP(3);
    // ---- GEMM 2 on the same activations: acc (= qg_i - kg_j) += (W_g1 W_d2) h ;  g = relu(acc)
    chain_gemm<C, P, NW, TPW, false, MODE == 1>(A.wg1, X, acc, wv, lane);   // MODE 1: from zero (no q, k)
    VA_STAMP(4);
    __syncthreads();   // every wave is done reading h
    if (MODE == 1) {
      // both products leave as fragment images: 16 registers per lane = four 1 KiB wave stores per (tile, query)
      float4* tg = A.tab_g + ((size_t)ig * NTILE + wv * TPW) * (P * 4 * 64) + lane;
      float4* tq = A.tab_p + ((size_t)ig * NTILE + wv * TPW) * (P * 4 * 64) + lane;
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            tg[((tp * P + p) * 4 + g) * 64] = float4{acc[tp][p][4 * g], acc[tp][p][4 * g + 1], acc[tp][p][4 * g + 2], acc[tp][p][4 * g + 3]};
            tq[((tp * P + p) * 4 + g) * 64] = float4{pos[tp][p][4 * g], pos[tp][p][4 * g + 1], pos[tp][p][4 * g + 2], pos[tp][p][4 * g + 3]};
          }
      continue;
    }
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          X[((wv * TPW + tp) * 32 + mfma_row(i, h)) * XS + 32 * p + j] = fmaxf(acc[tp][p][i], 0.f);
    __syncthreads();
  } else {
  // ---- GEMM 1: pos = W_d2 h + b_d2 ;  t = q_i - k_j + pos
  chain_gemm<C, P, NW, TPW, false>(A.wd2, X, acc, wv, lane);
  VA_STAMP(2);
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
        for (int e = 0; e < 4; e += 2) {      // register pairs -> v_pk_add_f32
          const int i = 4 * g + e;
          const f32x2 pv = f32x2{acc[tp][p][i], acc[tp][p][i + 1]} + f32x2{(&bb.x)[e], (&bb.x)[e + 1]};
          const f32x2 tv = (f32x2{(&qq.x)[e], (&qq.x)[e + 1]} - f32x2{pos[tp][p][i], pos[tp][p][i + 1]}) + pv;
          acc[tp][p][i] = tv[0]; acc[tp][p][i + 1] = tv[1];
          pos[tp][p][i] = pv[0]; pos[tp][p][i + 1] = pv[1];
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        X[((wv * TPW + tp) * 32 + mfma_row(i, h)) * XS + 32 * p + j] = acc[tp][p][i];
      }
  __syncthreads();

  VA_STAMP(3);
  // ---- GEMM 2: g = relu(W_g1 t + b_g1)
  chain_gemm<C, P, NW, TPW, false>(A.wg1, X, acc, wv, lane);
  VA_STAMP(4);
  __syncthreads();
#pragma unroll
  for (int tp = 0; tp < TPW; ++tp) {
    const int cbase = (wv * TPW + tp) * 32 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bb = *reinterpret_cast<const float4*>(A.bg1 + cbase + 8 * g);
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const int i = 4 * g + e;
          const f32x2 v = f32x2{acc[tp][p][i], acc[tp][p][i + 1]} + f32x2{(&bb.x)[e], (&bb.x)[e + 1]};
          X[((wv * TPW + tp) * 32 + mfma_row(i, h)) * XS + 32 * p + j] = fmaxf(v[0], 0.f);
          X[((wv * TPW + tp) * 32 + mfma_row(i + 1, h)) * XS + 32 * p + j] = fmaxf(v[1], 0.f);
        }
    }
  }
  __syncthreads();
  }

  // ---- GEMM 3 (operands swapped): a[j][c'] = W_g2 g + b_g2, lane = channel c', registers = neighbours
  VA_STAMP(5);
  chain_gemm<C, P, NW, TPW, true>(A.wg2, X, acc, wv, lane);
  VA_STAMP(6);
  if (!COMP) __syncthreads();   // plain form: X is dead from here on and serves as per-wave transpose scratch for pos

  // Softmax over the 32 neighbours (registers x 2 half-waves) and the weighted sum, written for instruction count --
  // every VALU instruction here comes out of the matrix pipe's time (tools/lab/phase_lab):
  //   * b_g2 is constant over the neighbours of a channel, so softmax_j((a_ij + b_g2)/sqrt(C)) == softmax_j(a_ij/sqrt(C)):
  //     the bias is not applied at all;
  //   * scale, log2(e) and the stabiliser fold into one packed fma per register pair, then v_exp_f32;
  //   * sums / weighted sums run on register pairs (v_pk_add_f32 / v_pk_fma_f32), the normalisation is one reciprocal
  //     at the end, the two half-waves meet through v_permlane32_swap;
  //   * the v_j gathers take a 32-bit row offset from LDS (voffs) through a buffer descriptor: one v_add per load.
  float* scr = X + wv * (32 * 33);
  const float k2 = inv_sqrt_c * 1.44269504088896340736f;
  const __amdgpu_buffer_rsrc_t vrs = frag_rsrc(A.v, 0xffffffffu);
  // The v_j gathers of tile q + 1 are issued before tile q's softmax: issued and consumed inside one tile, the L2 / HBM round
  // trip of every tile's sixteen 4-byte gathers stood in front of its weighted sum (vmcnt(14) ... vmcnt(0) in the ISA).
  float vgb[2][16];
#define VA_GATHER(Q)                                                                                          \
  {                                                                                                           \
    const int tp_ = (Q) / P, p_ = (Q) % P, cch_ = (wv * TPW + tp_) * 32 + j;                                  \
    _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                            \
      vgb[(Q) & 1][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vrs, voffs[p_ * 32 + mfma_row(i, h)] + cch_ * 4, 0, 0)); \
  }
  VA_GATHER(0)
#pragma unroll
  for (int q = 0; q < TPW * P; ++q) {
    const int tp = q / P, p = q % P;
    const int cch = (wv * TPW + tp) * 32 + j;   // this lane's output channel
    {
      // pos tile [c'][j] (lane = neighbour) -> [j][c'] (lane = channel) through the wave-private scratch
      if (!COMP) {
#pragma unroll
        for (int i = 0; i < 16; ++i) scr[j * 33 + mfma_row(i, h)] = pos[tp][p][i];
      }
      if (q + 1 < TPW * P) VA_GATHER(q + 1)
      __builtin_amdgcn_sched_barrier(0);
      const float (&vg)[16] = vgb[q & 1];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float pt[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) pt[i] = COMP ? pos[tp][p][i] : scr[mfma_row(i, h) * 33 + j];
      f32x16& a = acc[tp][p];
      if (MASK) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (mfma_row(i, h) >= A.kvalid) a[i] = -INFINITY;
      }
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
#undef VA_GATHER
  VA_STAMP(7);
  }   // item loop
}

template <int C, int P, int NW, int MINW, bool COMP, int MODE = 0>
static hipError_t launch_va_t(const VecAttnArgs& a, hipStream_t s) {
  const int groups = (a.Q + P - 1) / P;
  size_t lds = (size_t)C * 32 * P * 4 + P * 32 * 3 * 4 + 2 * P * 32 * 4 + (size_t)P * C * 4;
#ifdef POEM_VA_DBG
  if (const char* e = getenv("POEM_VA_LDSPAD")) lds += atoi(e);
#endif
  auto kern = vecattn_kernel<C, P, NW, MINW, COMP, MODE>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), lds, optin); e != hipSuccess) return e;
  unsigned grid = (unsigned)(a.B * groups);
  if (a.stagger > 0) {
    grid = std::min(grid, (unsigned)(2 * poem_device_cus()));
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, s, a);
  return hipGetLastError();
}

template <int C, int P, int NW, int MINW>
static hipError_t launch_va(const VecAttnArgs& a, hipStream_t s, int mode = 0) {
  if (mode == 1) return launch_va_t<C, P, NW, MINW, true, 1>(a, s);
  if (mode == 2) return launch_va_t<C, P, NW, MINW, true, 2>(a, s);
  if (mode == 3) return a.composed ? launch_va_t<C, P, NW, MINW, true, 3>(a, s) : hipErrorInvalidValue;
  return a.composed ? launch_va_t<C, P, NW, MINW, true>(a, s) : launch_va_t<C, P, NW, MINW, false>(a, s);
}

// Neighbour count of the next full-kernel launches from this thread (decoder.cpp sets it around the launches of blocks >= 1 when the
// handle's N_NEIGHBOR / N_NEIGHBOR_QUERY is below 32, and back): 32 = the unmasked kernels.
static thread_local int g_va_kvalid = 32;
extern "C" void poem_vecattn_valid_neighbours(int k) { g_va_kvalid = k; }

// One-query blocks for the full kernel (MODE 0) at embed 256 (A/B: poem_set_option "va_p1"; the table modes keep the query
// groups their images are laid out for): twice the items, half the work each -- a batch of two leaves the busiest CU with 7
// queries instead of 8; costs twice the weight traffic from L2 per query.
static thread_local int g_va_p1 = 0;
extern "C" void poem_vecattn_one_query_blocks(int on) { g_va_p1 = on; }

// queries per block of the instantiation that serves embed width C (the table images are laid out per query group)
static int va_group_size(int C) { return C == 128 ? 4 : (C >= 512 ? 1 : 2); }

static hipError_t dispatch_va(const VecAttnArgs& a, int C, hipStream_t s, int mode) {
  if (mode == 0 && a.kvalid != 32) {
    if (a.kvalid < 1 || a.kvalid > 32) return hipErrorInvalidValue;
    mode = 3;
  }
  switch (C) {
    case 32: return launch_va<32, 2, 1, 1>(a, s, mode);
    case 64: return launch_va<64, 2, 2, 1>(a, s, mode);
    case 128: return launch_va<128, 4, 4, 2>(a, s, mode);
    case 256: {
      // (-1: small batches only -- fewer than 16 queries per CU)
      const int g_va_p1 = ::g_va_p1 >= 0 ? ::g_va_p1 : ((long)a.B * a.Q <= 16L * poem_device_cus() ? 1 : 0);
      if (mode == 0 && g_va_p1 == 1 && a.composed) return launch_va_t<256, 1, 4, 3, true>(a, s);
      if (mode == 0 && g_va_p1 == 2 && a.composed) return launch_va_t<256, 1, 4, 2, true>(a, s);
      return launch_va<256, 2, 4, 2>(a, s, mode);
    }
    case 512: return launch_va<512, 1, 4, 2>(a, s, mode);
    case 1024: return launch_va<1024, 1, 8, 2>(a, s, mode);   // 8 waves x 4 channel tiles: 2 waves per SIMD, no spills (4 x 8 tiles spilled 158 VGPRs)
    default: return hipErrorInvalidValue;
  }
}

// Floats per table (tab_g or tab_p) for Q queries of width C: whole query groups x 32 anchors x C.
extern "C" size_t poem_vector_attention_table_floats(int Q, int C) {
  const int P = va_group_size(C);
  return (size_t)((Q + P - 1) / P) * P * 32 * C;
}

// MODE 1: the sample-independent products of an anchored (first-block) vector attention, from the Q query coordinates.
extern "C" hipError_t poem_launch_vector_attention_tables(const float* query_xyz, const float* anchor_xyz, const int* idx,
                                                          const float* wd1, const float* bd1, const void* wd2,
                                                          const float* bd2, const void* wg1d2, float* tab_g,
                                                          float* tab_p, int Q, int C, hipStream_t s) {
  VecAttnArgs a{query_xyz, nullptr, anchor_xyz, idx, 1, nullptr, nullptr, nullptr, 1, wd1, bd1, (const float4*)wd2, bd2,
                (const float4*)wg1d2, nullptr, nullptr, nullptr, nullptr, 1, Q, 0, 0, 0, 0, 1, (float4*)tab_g, (float4*)tab_p, 32};
  return dispatch_va(a, C, s, 1);
}

// MODE 2: the per-sample remainder on top of the tables (q, k are the composed qg, kg).
extern "C" hipError_t poem_launch_vector_attention_anchored(const int* idx, const float* qg, const float* kg,
                                                            const float* v, int nsrc, const void* wg2,
                                                            const float* tab_g, const float* tab_p, float* out, int B,
                                                            int Q, int C, int ldq, int ldk, int ldv, hipStream_t s) {
  VecAttnArgs a{nullptr, nullptr, nullptr, idx, 1, qg, kg, v, nsrc, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                (const float4*)wg2, nullptr, out, B, Q, ldq, ldk, ldv, 0, 1, (float4*)tab_g, (float4*)tab_p, 32};
  return dispatch_va(a, C, s, 2);
}

extern "C" hipError_t poem_launch_vector_attention(const float* query_xyz, const float* src_xyz, const float* anchor_xyz,
                                                   const int* idx, int shared_idx, const float* q, const float* k,
                                                   const float* v, int nsrc, const float* wd1, const float* bd1,
                                                   const void* wd2, const float* bd2, const void* wg1, const float* bg1,
                                                   const void* wg2, const float* bg2, float* out, int B, int Q, int C,
                                                   int ldq, int ldk, int ldv, int composed, hipStream_t s) {
  VecAttnArgs a{query_xyz, src_xyz, anchor_xyz, idx, shared_idx, q, k, v, nsrc, wd1, bd1, (const float4*)wd2, bd2,
                (const float4*)wg1, bg1, (const float4*)wg2, bg2, out, B, Q, ldq, ldk, ldv, 0, composed, nullptr, nullptr, g_va_kvalid};
#ifdef POEM_LAB
  static const int stagger_env = getenv("POEM_VA_STAGGER") ? atoi(getenv("POEM_VA_STAGGER")) : 0;   // lab only, read once
  a.stagger = stagger_env;
#endif
#ifdef POEM_VA_DBG
  if (C == 256)
    if (const char* e = getenv("POEM_VA_CFG")) {
      switch (atoi(e)) {
        case 1: return launch_va<256, 1, 4, 4>(a, s);
        case 2: return launch_va<256, 1, 4, 3>(a, s);
        case 3: return launch_va<256, 2, 8, 4>(a, s);
        case 4: return launch_va<256, 1, 8, 4>(a, s);
        case 5: return launch_va<256, 1, 2, 2>(a, s);
        default: break;
      }
    }
#endif
  return dispatch_va(a, C, s, 0);
}

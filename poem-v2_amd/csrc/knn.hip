// K = 32 nearest source points per query (pytorch3d.ops.knn_points semantics at the reference's call sites,
// lib/models/bricks/point_transformers.py:83,134 upstream): squared L2 evaluated as ((dx*dx + dy*dy) + dz*dz) in fp32 with
// every operation individually rounded (no fma contraction -- near-ties must order the way the CPU evaluation orders
// them; `fma` selects the CUDA kernel's contracted rounding instead), ascending, ties -> lower index.
//
// A block serves a contiguous query range of ONE sample: the sample's source coordinates are staged once in LDS as
// three planes (x | y | z, padded to a multiple of 128 with +inf: conflict-free 4-byte reads, no bounds test in the loops)
// and the block's waves pull queries from an LDS counter, one wave per query.
//
// Selection, two streaming passes over the candidates -- nothing but the running minimum is kept in registers, so the
// kernel needs ~40 VGPRs whatever NS is (round 2 kept NS/64 distances per lane: 128 VGPRs + spills at NS = 4096, 1.2 ms
// per launch next to the attention kernels):
//   pass 1  every lane's minimum over its NS/64 candidates (c = lane + 64 i, two per step on the packed fp32 instructions).  The 32nd smallest of the 64 lane minima
//           is a bound T >= the 32nd smallest distance overall (those 32 minima are 32 distinct candidates <= T).
//   pass 2  the same distances again (same instructions, same bits), every candidate with d <= T appended to a per-wave
//           LDS list with ballots (for scattered points ~44 of 4096), then ranked among themselves by (distance, index):
//           rank r < 32 writes output slot r.
// When more than SURV_CAP candidates tie below T, or fewer than 32 survive (NaN coordinates), or T is infinite, the wave
// falls back to 32 rounds of "smallest (distance, index) above the previous winner" over all its candidates -- slow, rare,
// and by construction the same strict order.
#include "common.h"
#include <algorithm>

#define POEM_KNN_SURV_CAP 128

namespace {

typedef float knn2 __attribute__((ext_vector_type(2)));

// two candidates at a time on the packed fp32 instructions (v_pk_add / v_pk_mul / v_pk_fma: per-component IEEE results)
template <bool FMA>
__device__ __forceinline__ knn2 knn_dist2(knn2 qx, knn2 qy, knn2 qz, knn2 sx, knn2 sy, knn2 sz) {
  // every product and sum individually rounded (HIP's __fmul_rn / __fadd_rn are plain operators and DO contract to
  // v_fma_f32 under hipcc's default -ffp-contract=fast: a 1-ulp difference that reorders near-tied candidates)
#pragma clang fp contract(off)
  const knn2 dx = qx - sx, dy = qy - sy, dz = qz - sz;
  const knn2 xx = dx * dx;
  // FMA: the rounding of pytorch3d's CUDA kernel instead (knn.cu: `dist += diff * diff` under nvcc's default -fmad=true
  // is fma(dz, dz, fma(dy, dy, dx * dx))) -- for evaluating against results produced on that path; the default is the
  // CPU path's (knn_cpu.cpp built without FMA), which BASELINE's parity bar is stated against
  if (FMA) return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, xx));
  const knn2 yy = dy * dy, zz = dz * dz;
  return (xx + yy) + zz;
}
template <bool FMA>
__device__ __forceinline__ float knn_dist(float qx, float qy, float qz, float sx, float sy, float sz) {
  const knn2 d = knn_dist2<FMA>(knn2{qx, qx}, knn2{qy, qy}, knn2{qz, qz}, knn2{sx, sx}, knn2{sy, sy}, knn2{sz, sz});
  return d[0];
}

// (distance, index) strictly below (distance, index)
__device__ __forceinline__ bool knn_less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }

// candidates of a lane: c = lane + 64 i, i < 2 * pairs, taken in pairs (i = 2j, 2j + 1: one ds_read2_b32 per plane)
template <bool FMA>
__device__ __forceinline__ void knn_query(const float* __restrict__ px, const float* __restrict__ py, const float* __restrict__ pz,
                                          float2* __restrict__ surv, float* __restrict__ lmin, float qx, float qy, float qz,
                                          int pairs, int NS, int lane, int* __restrict__ out) {
  const knn2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
  // ---- pass 1: lane minimum (fminf drops NaN distances)
  float lm = INFINITY;
#pragma unroll 4
  for (int j = 0; j < pairs; ++j) {
    const int c = lane + 128 * j;
    const knn2 d = knn_dist2<FMA>(qx2, qy2, qz2, knn2{px[c], px[c + 64]}, knn2{py[c], py[c + 64]}, knn2{pz[c], pz[c + 64]});
    lm = __builtin_fminf(__builtin_fminf(lm, d[0]), d[1]);
  }
  // ---- T = 32nd smallest lane minimum: the largest minimum with <= 31 minima strictly below it
  lmin[lane] = lm;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  int below = 0;
#pragma unroll
  for (int l = 0; l < 64; l += 4) {
    const float4 o = *reinterpret_cast<const float4*>(lmin + l);      // same address in every lane: an LDS broadcast
    below += (o.x < lm ? 1 : 0) + (o.y < lm ? 1 : 0) + (o.z < lm ? 1 : 0) + (o.w < lm ? 1 : 0);
  }
  float t = below <= 31 ? lm : -INFINITY;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t = fmaxf(t, __shfl_xor(t, o, 64));
  // ---- pass 2: survivors -> LDS list (order irrelevant: they are ranked below)
  int n = 0;                                    // wave-uniform
#pragma unroll 2
  for (int j = 0; j < pairs; ++j) {
    const int c = lane + 128 * j;
    const knn2 d = knn_dist2<FMA>(qx2, qy2, qz2, knn2{px[c], px[c + 64]}, knn2{py[c], py[c + 64]}, knn2{pz[c], pz[c + 64]});
    const bool k0 = d[0] <= t, k1 = d[1] <= t;
    const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
    if ((m0 | m1) == 0ull) continue;
    const int p0 = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u));
    if (k0 && p0 < POEM_KNN_SURV_CAP) surv[p0] = make_float2(d[0], __int_as_float(c));
    n += __popcll(m0);
    const int p1 = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u));
    if (k1 && p1 < POEM_KNN_SURV_CAP) surv[p1] = make_float2(d[1], __int_as_float(c + 64));
    n += __popcll(m1);
  }
  if (n >= 32 && n <= POEM_KNN_SURV_CAP && t < INFINITY) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- rank every survivor among the survivors by (distance, index); ranks 0..31 are the answer, in order
    for (int base = 0; base < n; base += 64) {
      const int me = base + lane;
      const float2 mine = surv[min(me, n - 1)];
      const int mi = __float_as_int(mine.y);
      int rank = 0;
#pragma unroll 4
      for (int j = 0; j < n; ++j) {
        const float2 o = surv[j];               // broadcast read
        rank += knn_less(o.x, __float_as_int(o.y), mine.x, mi) ? 1 : 0;
      }
      if (me < n && rank < 32) out[rank] = mi;
    }
    __builtin_amdgcn_wave_barrier();            // the list is reused by this wave's next query
    return;
  }
  const int per = 2 * pairs;
  // ---- fall-back: 32 rounds of "smallest (distance, index) strictly above the previous winner"
  float ld = -INFINITY;
  int li = -1, last_written = 0;
  for (int round = 0; round < 32; ++round) {
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int i = 0; i < per; ++i) {
      const int c = lane + 64 * i;
      if (c >= NS) break;
      const float d = knn_dist<FMA>(qx, qy, qz, px[c], py[c], pz[c]);
      const bool above = d > ld || (d == ld && c > li);
      if (above && knn_less(d, c, best, bi)) { best = d; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oc = __shfl_xor(bi, o, 64);
      if (knn_less(ob, oc, best, bi)) { best = ob; bi = oc; }
    }
    if (bi == 0x7fffffff) {                     // no comparable candidate left (NaN distances): repeat a valid index
      if (lane == 0) out[round] = last_written;
      continue;
    }
    if (lane == 0) out[round] = bi;
    last_written = bi;
    ld = best;
    li = bi;
  }
}

}  // namespace

// grid = B * G blocks; block (b, g) serves queries [g * NQ / G, (g + 1) * NQ / G) of sample b
template <int QPB, bool FMA>
__global__ __launch_bounds__(QPB * 64) void knn_kernel(const float* __restrict__ qxyz, const float* __restrict__ sxyz,
                                                       int* __restrict__ idx, int B, int NQ, int NS, int G) {
  extern __shared__ __attribute__((aligned(16))) float sp[];
  const int NSP = (NS + 127) & ~127, pairs = NSP >> 7;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x / G, g = blockIdx.x % G;
  float *px = sp, *py = sp + NSP, *pz = sp + 2 * NSP;
  float2* surv = reinterpret_cast<float2*>(sp + 3 * NSP) + wv * POEM_KNN_SURV_CAP;
  float* lmin = sp + 3 * NSP + QPB * POEM_KNN_SURV_CAP * 2 + wv * 64;
  int* counter = reinterpret_cast<int*>(sp + 3 * NSP + QPB * POEM_KNN_SURV_CAP * 2 + QPB * 64);
  {
    const float* src = sxyz + (size_t)b * NS * 3;
    for (int i = threadIdx.x; i < NS * 3; i += QPB * 64) {
      const int c = i / 3, k = i - 3 * c;
      sp[k * NSP + c] = src[i];
    }
    for (int i = NS + threadIdx.x; i < NSP; i += QPB * 64) px[i] = py[i] = pz[i] = INFINITY;
    if (threadIdx.x == 0) *counter = 0;
  }
  __syncthreads();
  const int q_lo = g * NQ / G, q_hi = (g + 1) * NQ / G;      // (launcher: G * NQ < 2^31)
  for (;;) {
    int qi = 0;
    if (lane == 0) qi = atomicAdd(counter, 1);
    qi = q_lo + __builtin_amdgcn_readfirstlane(qi);
    if (qi >= q_hi) break;
    const long wid = (long)b * NQ + qi;
    const float* qp = qxyz + wid * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    knn_query<FMA>(px, py, pz, surv, lmin, qx, qy, qz, pairs, NS, lane, idx + wid * 32);
  }
}

extern "C" hipError_t poem_launch_knn(const float* qxyz, const float* sxyz, int* idx, int B, int NQ, int NS, int fma,
                                      hipStream_t s) {
  constexpr int QPB = 16;
  if (B <= 0 || NQ <= 0 || NS <= 0) return hipErrorInvalidValue;
  const int NSP = (NS + 127) & ~127;
  const size_t lds = ((size_t)3 * NSP + (size_t)QPB * POEM_KNN_SURV_CAP * 2 + QPB * 64 + 4) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  auto kern = fma ? knn_kernel<QPB, true> : knn_kernel<QPB, false>;
  static std::atomic<unsigned long long> optin[2];
  // (opted in once per device to the CU's whole 160 KB: the need grows with NS, and the per-kernel "done" bit of poem_optin_lds
  //  does not remember the size it was set for)
  if (lds > 64 * 1024)
    if (hipError_t e = poem_optin_lds(reinterpret_cast<const void*>(kern), 160 * 1024, optin[fma ? 1 : 0]); e != hipSuccess) return e;
  // blocks per sample: about one block per CU over the batch (a block stages the sample's sources once: fewer, longer
  // blocks), never fewer than QPB queries per block
  const int cus = poem_device_cus();
  int G = (cus + B - 1) / B;
  G = std::max(1, std::min(G, (NQ + QPB - 1) / QPB));
  if ((long)G * NQ >= (1l << 31)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3((unsigned)(B * G)), dim3(QPB * 64), lds, s, qxyz, sxyz, idx, B, NQ, NS, G);
  return hipGetLastError();
}

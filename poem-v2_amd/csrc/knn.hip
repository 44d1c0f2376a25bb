// K = 32 nearest source points per query (pytorch3d.ops.knn_points semantics at the reference's call sites):
// squared L2 evaluated as ((dx*dx + dy*dy) + dz*dz) in fp32 with every operation individually rounded (no fma
// contraction -- near-ties must order the way the CPU evaluation orders them), ascending, ties -> lower index.
// One wave per query; each lane keeps ceil(NS/64) candidate distances in registers (strided so that the source
// coordinates are read coalesced) and the wave extracts the minimum 32 times.
#include "common.h"

template <int PER>
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ qxyz, const float* __restrict__ sxyz,
                                                  int* __restrict__ idx, int B, int NQ, int NS) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long)B * NQ) return;
  const int b = (int)(wid / NQ);
  const float* qp = qxyz + wid * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const float* sp = sxyz + (size_t)b * NS * 3;
  float d[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    if (c < NS) {
      const float dx = __fsub_rn(qx, sp[c * 3 + 0]), dy = __fsub_rn(qy, sp[c * 3 + 1]), dz = __fsub_rn(qz, sp[c * 3 + 2]);
      d[i] = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    } else {
      d[i] = INFINITY;
    }
  }
  int* out = idx + wid * 32;
  for (int round = 0; round < 32; ++round) {
    float best = d[0];
    int bi = 0;
#pragma unroll
    for (int i = 1; i < PER; ++i) {
      const bool lt = d[i] < best;      // strict: keeps the lower index on ties
      best = lt ? d[i] : best;
      bi = lt ? i : bi;
    }
    int bc = lane + 64 * bi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oc = __shfl_xor(bc, o, 64);
      const bool take = (ob < best) || (ob == best && oc < bc);
      best = take ? ob : best;
      bc = take ? oc : bc;
    }
    if (lane == 0) out[round] = bc;
#pragma unroll
    for (int i = 0; i < PER; ++i) d[i] = (lane + 64 * i == bc) ? INFINITY : d[i];
  }
}

extern "C" hipError_t poem_launch_knn(const float* qxyz, const float* sxyz, int* idx, int B, int NQ, int NS,
                                      hipStream_t s) {
  const long waves = (long)B * NQ;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  if (NS <= 64 * 16) hipLaunchKernelGGL((knn_kernel<16>), grid, block, 0, s, qxyz, sxyz, idx, B, NQ, NS);
  else if (NS <= 64 * 64) hipLaunchKernelGGL((knn_kernel<64>), grid, block, 0, s, qxyz, sxyz, idx, B, NQ, NS);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

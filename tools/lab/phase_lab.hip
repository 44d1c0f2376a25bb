// Development lab: what a VALU instruction costs next to fp32 MFMAs (v_mfma_f32_32x32x2_f32), by placement.
// Every wave runs the same loop: per iteration M = 32 MFMAs (4 accumulators) and V plain VALU ops (v_fma_f32 on
// independent registers).  ARR 0: phase-separated (all MFMAs, then all VALU - what a QK^T -> softmax -> PV loop looks
// like);  ARR 1: the same VALU ops spread evenly between the MFMAs of the SAME wave (software-pipelined form).
// W waves per SIMD.  Reports cycles per iteration per wave and the SIMD's matrix-pipe utilisation.
#include "../../poem-v2_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int ARR, int VPM, int NACC = 4>   // VPM = VALU ops per MFMA; NACC independent accumulators
__global__ __launch_bounds__(1024) void phase_kernel(const float* __restrict__ wsrc, float* __restrict__ out,
                                                     long long* __restrict__ cyc, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = zero16();
  float a[16];
  for (int i = 0; i < 16; ++i) a[i] = wsrc[i] + (float)lane;
  const float av = (float)(lane % 7) * 0.25f, bv = (float)(lane % 5) * 0.5f;
  const float x = 1.0f + 1e-6f * (float)lane, y = 1e-7f;
  // desynchronise the waves of one SIMD (they would otherwise run in lock step from the common start)
  for (int i = 0; i < (wv >> 2) * 9; ++i) acc[0] = mfma32(av, bv, acc[0]);
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (ARR == 0) {
#pragma unroll
      for (int m = 0; m < 32; ++m) acc[m % NACC] = mfma32(av, bv, acc[m % NACC]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < 32 * VPM; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[v & 15]) : "v"(x), "v"(y));
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        acc[m & 3] = mfma32(av, bv, acc[m & 3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < VPM; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(m * VPM + v) & 15]) : "v"(x), "v"(y));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = clock64();
  float res = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
  for (int i = 0; i < 16; ++i) res += a[i];
  out[(size_t)blockIdx.x * blockDim.x + tid] = res;
  if (lane == 0) cyc[(size_t)blockIdx.x * (blockDim.x >> 6) + wv] = t1 - t0;
}

template <int ARR, int VPM, int NACC = 4>
static void run(int W, float* w, float* out, long long* cyc) {
  const int blocks = 256, iters = 600, nw = 4 * W;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((phase_kernel<ARR, VPM, NACC>), dim3(blocks), dim3(nw * 64), 0, 0, w, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::vector<long long> h((size_t)blocks * nw);
  CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double s = 0, mx = 0;
  for (long long v : h) { s += v; if (v > mx) mx = v; }
  const double per_it = s / h.size() / iters;
  // SIMD-level: W waves x 32 MFMAs x 64 cycles of matrix work per (max wave span / iters) cycles
  const double util = (double)W * 32 * 64 * iters / mx;
  printf("%s NACC=%d VALU/MFMA=%d W=%d: %.0f cycles/iteration/wave (matrix-only floor %d), pipe utilisation %.1f %%, wall %.3f ms -> %.1f TFLOP/s\n",
         ARR ? "interleaved" : "phased     ", NACC, VPM, W, per_it, 2048 * W, 100.0 * util, ms,
         (double)blocks * nw * iters * 32 * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
  float* w; float* out; long long* cyc;
  CK(hipMalloc(&w, 4096)); CK(hipMemset(w, 0, 4096)); CK(hipMalloc(&out, (size_t)256 * 1024 * 4));
  CK(hipMalloc(&cyc, (size_t)256 * 16 * 8));
  for (int W : {1, 2, 3}) {
    run<0, 0, 1>(W, w, out, cyc); run<0, 0, 2>(W, w, out, cyc); run<0, 2, 1>(W, w, out, cyc); run<0, 2, 2>(W, w, out, cyc);
  }
  for (int W : {1, 2, 4}) {
    run<0, 0>(W, w, out, cyc);
    run<0, 2>(W, w, out, cyc); run<1, 2>(W, w, out, cyc);
    run<0, 5>(W, w, out, cyc); run<1, 5>(W, w, out, cyc);
    run<0, 10>(W, w, out, cyc); run<1, 10>(W, w, out, cyc);
  }
  return 0;
}

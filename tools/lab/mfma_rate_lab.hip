// Development lab: what does the fp32 matrix pipe deliver per instruction shape?  Whole chip, W waves per SIMD, each wave a chain
// of N independent accumulators; wall-clock TFLOP/s and shader cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(float* out, long long* cyc, int iters) {
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  long long t0, t1;
  float res = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    t0 = clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    t1 = clock64();
    for (int i = 0; i < NACC; ++i) for (int k = 0; k < 16; ++k) res += acc[i][k];
  } else {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;
    t0 = clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    t1 = clock64();
    for (int i = 0; i < NACC; ++i) for (int k = 0; k < 4; ++k) res += acc[i][k];
  }
  out[blockIdx.x * 256 + threadIdx.x] = res;
  if (blockIdx.x == 3 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int SHAPE, int NACC>
static void run(int waves_per_simd, int iters) {
  float* out; long long* cyc;
  const int blocks = 256 * waves_per_simd;       // 256-thread blocks = one wave per SIMD each
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&cyc, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((rate_kernel<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((rate_kernel<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double flop_per = SHAPE == 32 ? 32.0 * 32 * 2 * 2 : 16.0 * 16 * 4 * 2;
  const double n = (double)blocks * 4 * NACC * iters;
  printf("%dx%dx%d f32, %d accumulators, %d wave(s) per SIMD: %.1f TFLOP/s (wall), %.1f shader-clock ticks per MFMA of one wave\n",
         SHAPE, SHAPE, SHAPE == 32 ? 2 : 4, NACC, waves_per_simd, n * flop_per / (ms * 1e-3) / 1e12, (double)c / ((double)NACC * iters));
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  const int iters = 20000;
  for (int w : {1, 2, 4}) { run<32, 4>(w, iters); run<16, 4>(w, iters); run<16, 8>(w, iters); }
  run<32, 1>(1, iters); run<16, 1>(1, iters); run<16, 2>(1, iters);
  return 0;
}

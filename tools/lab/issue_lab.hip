// Development lab: cycles per fp32 MFMA for the vecattn-style inner loop (1 wave per SIMD), by instruction mix.
#include "../../poem-v2_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// MODE 0: MFMAs only (register operands)        MODE 1: + one ds_read2 per 4 MFMAs (one k-step ahead)
// MODE 2: + W float4 global loads per 16 MFMAs   MODE 3: both (the vecattn loop)
template <int MODE>
__global__ __launch_bounds__(256, 1) void loop_kernel(const float4* __restrict__ W, float* __restrict__ out, long long* cyc, int iters) {
  __shared__ float X[256 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, h = lane >> 5;
  for (int i = tid; i < 256 * 64; i += 256) X[i] = (float)(i % 17) * 0.01f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = zero16();
  const float4* wp = W + (size_t)(wv * 2) * 32 * 64 + lane;
  const float* xc = X + (4 * h) * 64 + j;
  float4 a0[2] = {wp[0], wp[32 * 64]}, a1[2];
  float xa[2] = {xc[0], xc[32]}, xb[2];
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    for (int kc = 0; kc < 32; kc += 2) {
#define LOADW(A, KCI) if (MODE & 2) { const int kq = min(KCI, 31); A[0] = wp[(size_t)kq * 64]; A[1] = wp[(size_t)(32 + kq) * 64]; }
#define READX(XR, KCI, T) if (MODE & 1) { const int kq = min(KCI, 31); XR[0] = xc[(kq * 8 + T) * 64]; XR[1] = xc[(kq * 8 + T) * 64 + 32]; }
#define MMA(A, T, XR) for (int tp = 0; tp < 2; ++tp) { const float av = (&A[tp].x)[T]; for (int p = 0; p < 2; ++p) acc[tp][p] = mfma32(av, XR[p], acc[tp][p]); }
#define CHUNK(A, KCI) \
  READX(xb, KCI, 1) __builtin_amdgcn_sched_barrier(0); MMA(A, 0, xa) __builtin_amdgcn_sched_barrier(0); \
  READX(xa, KCI, 2) __builtin_amdgcn_sched_barrier(0); MMA(A, 1, xb) __builtin_amdgcn_sched_barrier(0); \
  READX(xb, KCI, 3) __builtin_amdgcn_sched_barrier(0); MMA(A, 2, xa) __builtin_amdgcn_sched_barrier(0); \
  READX(xa, (KCI) + 1, 0) __builtin_amdgcn_sched_barrier(0); MMA(A, 3, xb) __builtin_amdgcn_sched_barrier(0);
      if (!(MODE & 2)) { a1[0] = a0[0]; a1[1] = a0[1]; }
      if (!(MODE & 1)) { xb[0] = xa[0]; xb[1] = xa[1]; }
      LOADW(a1, kc + 1)
      __builtin_amdgcn_sched_barrier(0);
      CHUNK(a0, kc)
      LOADW(a0, kc + 2)
      __builtin_amdgcn_sched_barrier(0);
      CHUNK(a1, kc + 1)
    }
  }
  const long long t1 = clock64();
  float sres = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) sres += acc[a][b][i];
  out[blockIdx.x * 256 + tid] = sres;
  if (blockIdx.x == 7 && lane == 0) cyc[wv] = t1 - t0;
}


template <int MODE>
__global__ __launch_bounds__(256, 1) void loop4_kernel(const float4* __restrict__ W, float* __restrict__ out, long long* cyc, int iters) {
  __shared__ float X[256 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, h = lane >> 5;
  for (int i = tid; i < 256 * 64; i += 256) X[i] = (float)(i % 17) * 0.01f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = zero16();
  const float4* wp = W + (size_t)(wv * 2) * 32 * 64 + lane;
  const float* xc = X + (4 * h) * 64 + j;
  float4 a0[2] = {wp[0], wp[32 * 64]}, a1[2] = {wp[64], wp[33 * 64]}, a2[2] = {wp[128], wp[34 * 64]}, a3[2];
  float xa[2] = {xc[0], xc[32]}, xb[2];
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    for (int kc = 0; kc < 32; kc += 4) {
      LOADW(a3, kc + 3)
      __builtin_amdgcn_sched_barrier(0);
      CHUNK(a0, kc)
      LOADW(a0, kc + 4)
      __builtin_amdgcn_sched_barrier(0);
      CHUNK(a1, kc + 1)
      LOADW(a1, kc + 5)
      __builtin_amdgcn_sched_barrier(0);
      CHUNK(a2, kc + 2)
      LOADW(a2, kc + 6)
      __builtin_amdgcn_sched_barrier(0);
      CHUNK(a3, kc + 3)
    }
  }
  const long long t1 = clock64();
  float sres = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) sres += acc[a][b][i];
  out[blockIdx.x * 256 + tid] = sres;
  if (blockIdx.x == 7 && lane == 0) cyc[wv] = t1 - t0;
}

// W fragments through a scalar base + running 32-bit offset (no 64-bit VALU address arithmetic per load)
template <int MODE>
__global__ __launch_bounds__(256, 1) void loop5_kernel(const float4* __restrict__ W, float* __restrict__ out, long long* cyc, int iters) {
  __shared__ float X[256 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 31, h = lane >> 5;
  for (int i = tid; i < 256 * 64; i += 256) X[i] = (float)(i % 17) * 0.01f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) acc[a][b] = zero16();
  const float4* wbase = W + (size_t)(wv * 2) * 32 * 64;       // wave-uniform
  const float* xc = X + (4 * h) * 64 + j;
  float4 a0[2] = {wbase[lane], wbase[32 * 64 + lane]}, a1[2];
  float xa[2] = {xc[0], xc[32]}, xb[2];
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kc = 0; kc < 32; kc += 2) {
#define LOADW5(A, KCI) { const int kq = (KCI) < 31 ? (KCI) : 31; A[0] = wbase[kq * 64 + lane]; A[1] = wbase[(32 + kq) * 64 + lane]; }
      LOADW5(a1, kc + 1)
      __builtin_amdgcn_sched_barrier(0);
      CHUNK(a0, kc)
      LOADW5(a0, kc + 2)
      __builtin_amdgcn_sched_barrier(0);
      CHUNK(a1, kc + 1)
    }
  }
  const long long t1 = clock64();
  float sres = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) sres += acc[a][b][i];
  out[blockIdx.x * 256 + tid] = sres;
  if (blockIdx.x == 7 && lane == 0) cyc[wv] = t1 - t0;
}

int main() {
  float4* W; float* out; long long* cyc;
  CK(hipMalloc(&W, 256 * 256 * 4)); CK(hipMemset(W, 0, 256 * 256 * 4)); CK(hipMalloc(&out, 4 * 256 * 1024)); CK(hipMalloc(&cyc, 64));
  const int iters = 40;
  for (int blocks : {256, 512}) {
#define RUN(M) { hipLaunchKernelGGL((loop_kernel<M>), dim3(blocks), dim3(256), 0, 0, W, out, cyc, iters); CK(hipDeviceSynchronize()); \
    hipLaunchKernelGGL((loop_kernel<M>), dim3(blocks), dim3(256), 0, 0, W, out, cyc, iters); CK(hipDeviceSynchronize()); \
    long long h[4]; CK(hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost)); \
    printf("blocks=%d mode %d: %.2f cycles per MFMA (wave 0), %.2f (wave 3)\n", blocks, M, (double)h[0] / (iters * 32 * 16), (double)h[3] / (iters * 32 * 16)); }
    RUN(0) RUN(1) RUN(2) RUN(3)
#define RUN4(K, M, NAME) { hipLaunchKernelGGL((K<M>), dim3(blocks), dim3(256), 0, 0, W, out, cyc, iters); CK(hipDeviceSynchronize()); \
    hipLaunchKernelGGL((K<M>), dim3(blocks), dim3(256), 0, 0, W, out, cyc, iters); CK(hipDeviceSynchronize()); \
    long long h[4]; CK(hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost)); \
    printf("blocks=%d %s mode %d: %.2f cycles per MFMA\n", blocks, NAME, M, (double)h[0] / (iters * 32 * 16)); }
    RUN4(loop4_kernel, 3, "ring4") RUN4(loop5_kernel, 3, "const-offset") RUN4(loop5_kernel, 2, "const-offset")
  }
  return 0;
}

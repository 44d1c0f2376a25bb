// Development lab (round 6): the F1 panel GEMM alone -- X (131072 x 256) x W (1024 x 256) -> four attention images (K, V, K, V),
// gemm_panel_kernel<4, 2>, the 4C-column launch of blocks >= 1 at the headline batch.  Built with -DPOEM_PANEL_LAB=<bits>
// (1 = no image stores, 4 = store epilogue at priority 3) and -DPOEM_PANEL_SKEW=<100 MHz ticks> (second wave of a SIMD late).
#include "../../poem-v2_amd/csrc/gemm.hip"
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  p[i] = ((float)(x & 0xffff) / 65536.0f - 0.5f);
}
static float* rnd(size_t n, unsigned seed) { float* p; CK(hipMalloc(&p, n * 4)); hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, seed); return p; }
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 131072, K = 256, C = 256, nsegs = argc > 2 ? atoi(argv[2]) : 4;
  float* X = rnd((size_t)M * K, 1);
  float* W = rnd((size_t)nsegs * C * K, 2);      // (used as a packed image: any values time alike)
  float* bias = rnd((size_t)nsegs * C, 3);
  float* outs[6]; int modes[6] = {1, 2, 1, 2, 0, 0};
  for (int i = 0; i < nsegs; ++i) outs[i] = rnd((size_t)M * C, 10 + i);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(poem_launch_gemm_segs(X, K, W, bias, M, K, 0, C, nsegs, outs, modes, 0));
  CK(hipDeviceSynchronize());
  const int n = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < n; ++i) CK(poem_launch_gemm_segs(X, K, W, bias, M, K, 0, C, nsegs, outs, modes, 0));
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double fl = 2.0 * M * (double)(nsegs * C) * K;
  printf("LAB=%d  M=%d N=%d K=%d: %.1f us per launch, %.1f TFLOP/s = %.3f of the fp32 matrix peak\n", POEM_PANEL_LAB, M, nsegs * C, K,
         ms / n * 1e3, fl / (ms / n * 1e-3) / 1e12, fl / (ms / n * 1e-3) / 1e12 / 157.3);
  return 0;
}

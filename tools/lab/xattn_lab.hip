// Development lab: the cross-attention kernel alone on random fragment images (B=32, 799 x 4096, C=256, 4 heads),
// HIP-event timing per variant (POEM_ATTN_W waves per SIMD, POEM_ATTN_MAP item mapping).
#define POEM_LAB 1
#ifdef STAMPS
#define POEM_XA_STAMPS 1
#endif
#include "../../poem-v2_amd/csrc/attn.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed;
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  p[i] = ((float)(x & 0xffff) / 65536.0f - 0.5f) * 2.0f;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, NQ = 799, NK = 4096, C = argc > 2 ? atoi(argv[2]) : 256, heads = 4;
  float *q, *ki, *vi, *ctx, *scr;
  const size_t nq = (size_t)B * NQ * C, nk = (size_t)B * NK * C;
  const size_t ns = poem_cross_attention_scratch_floats(B, NQ, NK, C, heads, 0);
  CK(hipMalloc(&q, nq * 4)); CK(hipMalloc(&ki, nk * 4)); CK(hipMalloc(&vi, nk * 4)); CK(hipMalloc(&ctx, nq * 4));
  CK(hipMalloc(&scr, ns * 4));
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, q, nq, 1u);
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, 0, ki, nk, 2u);
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, 0, vi, nk, 3u);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* ws[] = {"3", "2", "1"};
  const char* ms[] = {"1", "0"};
  if (argc > 5) setenv("POEM_ATTN_PRIO", argv[5], 1);
  const bool single = argc > 3;     // PMC runs: one variant only (W from argv[3], map from argv[4], prio rotation argv[5])
  if (single) { ws[0] = argv[3]; ms[0] = argc > 4 ? argv[4] : "0"; }
  const int nwv = single ? 1 : (C == 256 ? 3 : 1), nmp = single ? 1 : 2;
  for (int wi = 0; wi < nwv; ++wi) for (int mi = 0; mi < nmp; ++mi) {
    setenv("POEM_ATTN_W", ws[wi], 1); setenv("POEM_ATTN_MAP", ms[mi], 1);
    for (int i = 0; i < 3; ++i) CK(poem_launch_cross_attention_img(q, C, ki, vi, ctx, B, NQ, NK, C, heads, scr, 0));
    CK(hipDeviceSynchronize());
    const int n = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) CK(poem_launch_cross_attention_img(q, C, ki, vi, ctx, B, NQ, NK, C, heads, scr, 0));
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms_ = 0; CK(hipEventElapsedTime(&ms_, e0, e1));
    const double fl = 4.0 * B * NQ * (double)NK * C;
    printf("W=%s map=%s: %.1f us per attention (kernel + combine), %.1f TFLOP/s\n", ws[wi], ms[mi], ms_ / n * 1e3, fl / (ms_ / n * 1e-3) / 1e12);
    {
      std::vector<long long> d(4096 * 4);
      CK(hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(xattn_dbg), d.size() * 8));
      const int nw = 256 * 4 * atoi(ws[wi]);
      double cs = 0, wsum = 0, cmax = 0, cmin = 1e18; long long its = 0;
      for (int i = 0; i < nw && i < 4096; ++i) { cs += d[4 * i]; wsum += d[4 * i + 1]; its += d[4 * i + 2]; cmax = std::max(cmax, (double)d[4 * i]); cmin = std::min(cmin, (double)d[4 * i]); }
#ifdef STAMPS
      { long long ph[8]; CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(xattn_ph), 64));
        printf("    stamps (block 5 wave 0, cycles per tile-step): wrap/wait->QK start %.0f, QK issue %.0f, softmax %.0f, PV+V loads %.0f\n",
               ph[0] / 400.0, ph[1] / 400.0, ph[2] / 400.0, ph[3] / 400.0); }
#endif
      const double n_ = std::min(nw, 4096);
      printf("    per wave: %.0f shader cycles avg (min %.0f max %.0f), %.1f us by the 100 MHz clock -> %.3f GHz; items/wave %.2f; "
             "cycles per tile-step per SIMD %.0f (MFMA floor 4096)\n", cs / n_, cmin, cmax, wsum / n_ / 100.0, cs / wsum / 10.0, its / n_,
             cmax / ((double)its / n_ * atoi(ws[wi]) * 32));
    }
  }
  return 0;
}

// Development lab: the chain16 kernels alone on random data, every kind, M = 799 B rows; HIP-event time per launch and
// (STAMPS build) the 100 MHz ticks at the phase boundaries of block 0.
#define POEM_C16_STAMPS 1
#include "../../poem-v2_amd/csrc/chain16.hip"
#include <cstdio>
#include <vector>
#include <map>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed;
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
  p[i] = ((float)(x & 0xffff) / 65536.0f - 0.5f) * 2.0f * scale;
}
static float* rnd(size_t n, unsigned seed, float scale = 1.0f) {
  float* p; CK(hipMalloc(&p, n * 4));
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, seed, scale);
  return p;
}
// packed weights live in one arena with a native-image mirror at the same offsets (as handle.cpp lays them out)
static char *g_arena = nullptr, *g_mirror = nullptr;
static size_t g_used = 0;
static const float4* rnd_w(size_t n, unsigned seed) {
  const size_t cap = (size_t)64 << 20;
  if (!g_arena) { CK(hipMalloc(&g_arena, cap)); CK(hipMalloc(&g_mirror, cap)); }
  float* p = (float*)(g_arena + g_used);
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, seed, 0.05f);
  CK(poem_launch_native16(p, g_mirror + g_used, n * 4, 0));
  g_used += n * 4;
  return (const float4*)p;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 2, C = getenv("C16_LAB_C") ? atoi(getenv("C16_LAB_C")) : 256, M = 799 * B;
  const size_t MC = (size_t)M * C;
  ChainArgs a{};
  a.M = M; a.tile_p = 3; a.eps = 1e-12f;
  a.x = rnd(MC, 1); a.ldx = C;
  a.w1 = rnd_w((size_t)C * C, 2); a.b1 = rnd(C, 3);
  a.res = rnd(MC, 4); a.ldres = C; a.res_mod = 0;
  a.ln_g = rnd(C, 5); a.ln_b = rnd(C, 6);
  a.y1 = rnd(MC, 7); a.ldy1 = C;
  a.w2 = rnd_w((size_t)3 * C * C, 8); a.b2 = rnd(3 * C, 9); a.y2 = rnd(3 * MC, 10); a.ldy2 = 3 * C;
  a.wf4 = rnd_w((size_t)5 * C * C, 11); a.bf4 = rnd(5 * C, 12);
  a.wreg2 = rnd(3 * C, 13); a.breg2 = rnd(3, 14); a.xyz_in = rnd((size_t)M * 3, 15); a.xyz_out = rnd((size_t)M * 3, 16);
  a.wout = rnd_w((size_t)4 * C * C, 17); a.bout = rnd(C, 18);
  a.ln2_g = rnd(C, 19); a.ln2_b = rnd(C, 20); a.y3 = rnd(MC, 21); a.ldy3 = C;
  a.native_delta = g_mirror - g_arena;
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int kinds[4] = {0, 1, 2, 3}, n2s[4] = {3, 1, 0, 2};
  for (int ki = 0; ki < 4; ++ki) {
    a.kind = kinds[ki]; a.n2 = n2s[ki];
    for (int i = 0; i < 3; ++i) CK(poem_launch_chain16(&a, C, 0));
    CK(hipDeviceSynchronize());
    const int n = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) CK(poem_launch_chain16(&a, C, 0));
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    long long st[64]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(c16_stamps), sizeof(st)));
    printf("B=%d kind %d n2=%d: %.1f us per launch (back to back).  block 0 stamps, us since start:", B, a.kind, a.n2, ms / n * 1e3);
    const int last = a.kind == 3 ? 21 : 6;
    for (int k = 1; k <= last; ++k) if (st[k] > st[0]) printf(" [%d] %.2f", k, (st[k] - st[0]) / 100.0);
    printf("\n    block 256 (same CU), us since block 0's start:");
    for (int k = 0; k <= last; ++k) if (st[32 + k] > st[0]) printf(" [%d] %.2f", k, (st[32 + k] - st[0]) / 100.0);
    printf("\n");
    {
      std::vector<long long> bl(1024 * 4);
      CK(hipMemcpyFromSymbol(bl.data(), HIP_SYMBOL(c16_blocks), bl.size() * 8));
      long long t0 = 1LL << 62, t1 = 0; int nb = 0;
      for (int b = 0; b < 1024; ++b) if (bl[4 * b]) { t0 = std::min(t0, bl[4 * b]); t1 = std::max(t1, bl[4 * b + 1]); ++nb; }
      double s_late = 0, dur = 0, dmax = 0; long long smax = 0;
      for (int b = 0; b < 1024; ++b) if (bl[4 * b]) { s_late += bl[4 * b] - t0; smax = std::max(smax, bl[4 * b] - t0); dur += bl[4 * b + 1] - bl[4 * b]; dmax = std::max(dmax, (double)(bl[4 * b + 1] - bl[4 * b])); }
      printf("    %d blocks: first start -> last end %.1f us; block start after the first: avg %.1f max %.1f us; block duration avg %.1f max %.1f us\n",
             nb, (t1 - t0) / 100.0, s_late / nb / 100.0, smax / 100.0, dur / nb / 100.0, dmax / 100.0);
      {   // by physical CU: blocks, units, when the CU's last block ended
        std::map<long long, std::vector<int>> cu;
        for (int b = 0; b < 1024; ++b) if (bl[4 * b]) cu[bl[4 * b + 2]].push_back(b);
        std::map<std::pair<int, int>, std::pair<int, double>> h;      // (blocks, units) -> (CUs, sum of end times)
        for (auto& kv : cu) {
          int u = 0; long long e = 0;
          for (int b : kv.second) { u += (int)bl[4 * b + 3]; e = std::max(e, bl[4 * b + 1]); }
          auto& x = h[{(int)kv.second.size(), u}]; x.first++; x.second += (e - t0) / 100.0;
        }
        printf("    physical CUs %zu; (blocks, units) on a CU: CUs, avg end us:", cu.size());
        for (auto& kv : h) printf(" (%d,%d): %d, %.0f;", kv.first.first, kv.first.second, kv.second.first, kv.second.second / kv.second.first);
        printf("\n");
      }
      std::vector<long long> z(1024 * 4, 0); CK(hipMemcpyToSymbol(HIP_SYMBOL(c16_blocks), z.data(), z.size() * 8));
    }
    if (argc > 2) {   // per-wave trace of blocks 0 and 256 (CU 0): shader-clock cycles since the earliest stamp 0
      long long wt[2 * 8 * 40]; unsigned hw[16];
      CK(hipMemcpyFromSymbol(wt, HIP_SYMBOL(c16_wtrace), sizeof(wt))); CK(hipMemcpyFromSymbol(hw, HIP_SYMBOL(c16_whw), sizeof(hw)));
      long long t0 = 1LL << 62;
      for (int w = 0; w < 16; ++w) if (wt[w * 40]) t0 = std::min(t0, wt[w * 40]);
      const double cyc_per_tick = (st[last] > st[0] && wt[last] > wt[0]) ? (double)(wt[last] - wt[0]) / (double)(st[last] - st[0]) : 0.0;   // shader cycles per 10 ns
      printf("    wave trace (kcycles since the first wave's start; shader clock = %.0f MHz); columns = stamp ids\n", cyc_per_tick * 100.0);
      for (int simd = 0; simd < 4; ++simd)
        for (int w = 0; w < 16; ++w) {
          if (((hw[w] >> 4) & 3) != (unsigned)simd || !wt[w * 40]) continue;
          printf("      simd %d block %3d wave %d (slot %2u):", simd, w < 8 ? 0 : 256, w & 7, hw[w] & 15);
          for (int k = 0; k < 40; ++k) if (wt[w * 40 + k]) printf(" [%d] %.1f", k, (wt[w * 40 + k] - t0) / 1000.0);
          printf("\n");
        }
      { long long z[2 * 8 * 40] = {}; CK(hipMemcpyToSymbol(HIP_SYMBOL(c16_wtrace), z, sizeof(z))); }
    }
    { long long z[64] = {}; CK(hipMemcpyToSymbol(HIP_SYMBOL(c16_stamps), z, sizeof(z))); }
  }
  return 0;
}

// Development lab (not product): per-phase cycle stamps of the fused vector-attention kernel.
#define POEM_VA_DBG 1
#include "../../poem-v2_amd/csrc/vecattn.hip"
#include "../../poem-v2_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
template <typename T> T* dalloc(size_t n, bool rnd = true, float scale = 1.f) {
  T* p; CK(hipMalloc(&p, n * sizeof(T)));
  std::vector<T> h(n);
  for (auto& v : h) v = rnd ? (T)(((float)rand() / (float)RAND_MAX - 0.5f) * scale) : (T)0;
  CK(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
  return p;
}
int main() {
  const int B = 32, Q = 799, NS = 4096, C = 256;
  float* qxyz = dalloc<float>((size_t)B * Q * 3), *sxyz = dalloc<float>((size_t)B * NS * 3);
  float* q = dalloc<float>((size_t)B * Q * C), *k = dalloc<float>((size_t)B * NS * C), *v = dalloc<float>((size_t)B * NS * C);
  int* idx; CK(hipMalloc(&idx, (size_t)B * Q * 32 * 4));
  { std::vector<int> h((size_t)B * Q * 32); for (auto& x : h) x = rand() % NS; CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
  float* wd1 = dalloc<float>(C * 3), *bd1 = dalloc<float>(C), *bd2 = dalloc<float>(C), *bg1 = dalloc<float>(C), *bg2 = dalloc<float>(C);
  float* w = dalloc<float>((size_t)C * C, true, 0.125f);
  void* wp[3];
  for (int i = 0; i < 3; ++i) { CK(hipMalloc(&wp[i], packed_linear_floats(C, C) * 4)); CK(poem_launch_pack_linear(w, C, C, wp[i], 0)); }
  float* out = dalloc<float>((size_t)B * Q * C, false);
  hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
  for (int it = 0; it < 3; ++it) CK(poem_launch_vector_attention(qxyz, sxyz, nullptr, idx, 0, q, k, v, NS, wd1, bd1, wp[0], bd2, wp[1], bg1, wp[2], bg2, out, B, Q, C, C, C, C, 0, 0));
  CK(hipEventRecord(s));
  for (int it = 0; it < 5; ++it) CK(poem_launch_vector_attention(qxyz, sxyz, nullptr, idx, 0, q, k, v, NS, wd1, bd1, wp[0], bd2, wp[1], bg1, wp[2], bg2, out, B, Q, C, C, C, C, 0, 0));
  CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
  float ms; CK(hipEventElapsedTime(&ms, s, e)); ms /= 5;
  printf("vecattn %.1f us  %.1f TF\n", ms * 1e3, (double)B * Q * 32 * (6.0 * C * C + 6.0 * C) / ms / 1e9);
  std::vector<long long> d(64 * 4 * 8);
  CK(hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(va_dbg), d.size() * 8));
  const char* names[7] = {"stage0", "kpre", "gemm1", "epi1", "gemm2", "epi2+vpre", "gemm3"};
  for (int blk = 0; blk < 6; ++blk) {
    for (int wv = 0; wv < 0; wv += 3) {
      long long* t = &d[(blk * 4 + wv) * 8];
      printf("blk %d wave %d start %lld:", blk, wv, t[0] - d[0]);
      for (int i = 0; i < 7; ++i) printf(" %s %lld", names[i], t[i + 1] - t[i]);
      printf(" | epi3 total %lld\n", t[7] - t[0]);
    }
  }
  // aggregate over all stamped blocks
  double sum[7] = {0}; int n = 0;
  for (int blk = 0; blk < 64; ++blk) for (int wv = 0; wv < 4; ++wv) { long long* t = &d[(blk * 4 + wv) * 8]; if (!t[7]) continue; for (int i = 0; i < 7; ++i) sum[i] += t[i + 1] - t[i]; ++n; }
  printf("mean over %d waves:", n); double tot = 0; for (int i = 0; i < 7; ++i) { printf(" %s %.0f", names[i], sum[i] / n); tot += sum[i] / n; } printf(" total %.0f\n", tot);
  return 0;
}

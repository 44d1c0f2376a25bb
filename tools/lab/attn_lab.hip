// Development lab (not product): per-phase cycle stamps of the cross-attention kernel.
#define POEM_LAB 1
#define POEM_ATTN_DBG 1
#include "../../poem-v2_amd/csrc/attn.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static float* dalloc(size_t n, float scale = 1.f) {
  float* p; CK(hipMalloc(&p, n * 4));
  std::vector<float> h(n);
  for (auto& v : h) v = ((float)rand() / (float)RAND_MAX - 0.5f) * scale;
  CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
  return p;
}
int main() {
  const int B = 32, Q = 799, NS = 4096, C = 256, heads = 4;
  float* q = dalloc((size_t)B * Q * C, 4.f), *k = dalloc((size_t)B * NS * C, 4.f), *v = dalloc((size_t)B * NS * C);
  float* ctx = dalloc((size_t)B * Q * C);
  int ks = 1;
  size_t sf = poem_cross_attention_scratch_floats(B, Q, NS, C, heads, &ks);
  float* scratch; CK(hipMalloc(&scratch, sf * 4 + 64));
  hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
  for (int it = 0; it < 3; ++it) CK(poem_launch_cross_attention(q, k, v, ctx, B, Q, NS, C, heads, C, scratch, 0));
  CK(hipEventRecord(s));
  for (int it = 0; it < 5; ++it) CK(poem_launch_cross_attention(q, k, v, ctx, B, Q, NS, C, heads, C, scratch, 0));
  CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
  float ms; CK(hipEventElapsedTime(&ms, s, e)); ms /= 5;
  printf("attn %.1f us  %.1f TF\n", ms * 1e3, 4.0 * B * Q * NS * C / ms / 1e9);
  std::vector<long long> d(8 * 4 * 8);
  CK(hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(attn_dbg), d.size() * 8));
  for (int blk = 0; blk < 4; ++blk)
    for (int wv = 0; wv < 4; ++wv) {
      long long* t = &d[(blk * 4 + wv) * 8];
      printf("blk %d wave %d: tiles %lld  total %lld  per-tile %lld | qk+softmax %lld  pv-issue %lld  store+barrier %lld\n", blk, wv, t[0], t[1],
             t[0] ? t[1] / t[0] : 0, t[0] ? t[2] / t[0] : 0, t[0] ? t[3] / t[0] : 0, t[0] ? t[4] / t[0] : 0);
    }
  return 0;
}

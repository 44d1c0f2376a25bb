// Development lab (round 6): what the non-MFMA instructions of chain16's GEMM loop cost next to v_mfma_f32_16x16x4_f32.
// One chunk of the loop at a 4-unit tile (RU = 4, T16 = 2) is 16 MFMAs (512 cycles of the pipe) + 8 ds_read_b32 (the X
// operands) + 2 buffer_load_dwordx4 (the weight fragments) + 4 v_cndmask.  The kernel below runs that chunk with each
// ingredient switched on or off, 8 waves per block, one or two blocks per CU -- exactly the kernel's occupancy -- and
// reports time per chunk per SIMD against the MFMA-only loop.
//   bit 0: weight fragment loads      bit 1: X operands by 8 ds_read_b32      bit 2: X operands by 2 ds_read_b128 (interleaved rows)
//   bit 3: the four selects           bit 4: s_setprio(1) around the MFMA group (lab)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int VAR, int RU>
__global__ __launch_bounds__(512, 1) void loop_kernel(const float4* __restrict__ w, float* __restrict__ out, int nphase, long long* __restrict__ ticks) {
  constexpr int XSP = 68, KCH = 32, T16 = 2;
  extern __shared__ __attribute__((aligned(16))) float X[];      // 256 x 68
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = lane & 15, g = lane >> 4;
  for (int i = tid; i < 256 * XSP; i += 512) X[i] = (float)((i * 7 + blockIdx.x) & 15) * 0.0625f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(w), (short)0, 256 * 256 * 4, 0x00020000);
  const bool hi = g >= 2;
  const int loff0 = (j + 32 * (g & 1)) * 16, loff1 = loff0 + 256;
  const int wbase = __builtin_amdgcn_readfirstlane(wv * KCH * 1024);
  const float* xc = X + (4 * (g & 1) + (g >> 1)) * XSP + j;
  constexpr int XV = 64;      // interleaved rows: position 4 j + u of a 64-float row (no padding: the four 16-lane groups of a
                              // ds_read_b128 mix lanes of g = 0 / 1, whose channels are 4 apart = 256 floats = the same banks, on disjoint slots)
  const float* xv = X + (4 * (g & 1) + (g >> 1)) * XV + 4 * j;
  f32x4 acc[T16][RU];
  for (int t = 0; t < T16; ++t) for (int u = 0; u < RU; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 A0[T16], A1[T16];
  for (int t = 0; t < T16; ++t) A0[t] = A1[t] = make_float4(0.01f * (lane + t), 0.02f, 0.03f, 0.04f);
  float P0[4], P1[4], Q0[4], Q1[4];
  for (int u = 0; u < 4; ++u) { Q0[u] = P0[u] = 0.5f + u; Q1[u] = P1[u] = 0.25f * u; }
  // one chunk: request the NEXT chunk's operands into (AN, Y0, Y1), multiply this chunk's (AC, X0, X1)
#define LAB_CHUNK(KC, AC, X0, X1, AN, Y0, Y1)                                                                           \
  {                                                                                                                     \
    if (VAR & 1) {                                                                                                      \
      _Pragma("unroll") for (int t = 0; t < T16; ++t) {                                                                 \
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (t & 1) ? loff1 : loff0, wbase + (KC) * 1024, 0);     \
        AN[t] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])); \
      }                                                                                                                 \
    } else {                                                                                                            \
      _Pragma("unroll") for (int t = 0; t < T16; ++t) asm volatile("" : "+v"(AN[t].x), "+v"(AN[t].y), "+v"(AN[t].z), "+v"(AN[t].w)); \
    }                                                                                                                   \
    if (VAR & 2) {                                                                                                      \
      _Pragma("unroll") for (int u = 0; u < RU; ++u) { Y0[u] = xc[((KC) * 8) * XSP + 16 * u]; Y1[u] = xc[((KC) * 8 + 2) * XSP + 16 * u]; } \
    } else if (VAR & 4) {                                                                                               \
      const float4 v0 = *reinterpret_cast<const float4*>(xv + ((KC) * 8) * XV), v1 = *reinterpret_cast<const float4*>(xv + ((KC) * 8 + 2) * XV); \
      Y0[0] = v0.x; Y0[1] = v0.y; Y0[2] = v0.z; Y0[3] = v0.w; Y1[0] = v1.x; Y1[1] = v1.y; Y1[2] = v1.z; Y1[3] = v1.w;   \
    } else {                                                                                                            \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(Y0[u]), "+v"(Y1[u]));                        \
    }                                                                                                                   \
    float w1[T16], w2[T16];                                                                                             \
    _Pragma("unroll") for (int t = 0; t < T16; ++t) {                                                                   \
      if (VAR & 8) { w1[t] = hi ? AC[t].y : AC[t].x; w2[t] = hi ? AC[t].w : AC[t].z; }                                  \
      else { w1[t] = AC[t].x; w2[t] = AC[t].z; }                                                                        \
    }                                                                                                                   \
    if (VAR & 16) __builtin_amdgcn_s_setprio(1);                                                                        \
    _Pragma("unroll") for (int u = 0; u < RU; ++u)                                                                      \
      _Pragma("unroll") for (int t = 0; t < T16; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[t], X0[u], acc[t][u], 0, 0, 0); \
    _Pragma("unroll") for (int u = 0; u < RU; ++u)                                                                      \
      _Pragma("unroll") for (int t = 0; t < T16; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[t], X1[u], acc[t][u], 0, 0, 0); \
    if (VAR & 16) __builtin_amdgcn_s_setprio(0);                                                                        \
  }
  const long long t0 = clock64();
  for (int ph = 0; ph < nphase; ++ph) {
#pragma unroll 1
    for (int kc = 0; kc < KCH; kc += 2) {
      LAB_CHUNK(kc, A0, P0, P1, A1, Q0, Q1)
      LAB_CHUNK(kc + 1, A1, Q0, Q1, A0, P0, P1)
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int t = 0; t < T16; ++t) for (int u = 0; u < RU; ++u) s += acc[t][u][0] + acc[t][u][1] + acc[t][u][2] + acc[t][u][3];
  out[(size_t)blockIdx.x * 512 + tid] = s;
  if (lane == 0 && blockIdx.x < 512) ticks[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int VAR, int RU>
static void run(const char* name, const float4* w, float* out, long long* ticks, int blocks_per_cu) {
  const int nphase = 64, grid = 256 * blocks_per_cu;
  const size_t lds = 256 * 68 * 4;
  auto k = loop_kernel<VAR, RU>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, w, out, nphase, ticks);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int n = 5;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, w, out, nphase, ticks);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> tk(grid * 8); CK(hipMemcpy(tk.data(), ticks, tk.size() * 8, hipMemcpyDeviceToHost));
  double avg = 0, mx = 0; for (auto v : tk) { avg += (double)v; mx = std::max(mx, (double)v); } avg /= tk.size();
  // per SIMD: waves = 2 * blocks_per_cu, each 16 * RU/4... MFMAs of 32 cycles per chunk
  const double chunks = (double)nphase * 32, mfma_cyc = 2.0 * blocks_per_cu * (4.0 * RU) * 32.0;
  const double us = ms / n * 1e3;
  printf("%-44s RU %d blocks/CU %d: %8.1f us  %7.1f ns per chunk per SIMD  (MFMA floor %5.0f cycles = %6.1f ns at 2.4 GHz -> %.3f)   in-kernel clock64: avg %.0f max %.0f per chunk\n",
         name, RU, blocks_per_cu, us, us * 1e3 / chunks, mfma_cyc, mfma_cyc / 2.4, mfma_cyc / 2.4 / (us * 1e3 / chunks), avg / chunks, mx / chunks);
}

int main() {
  float4* w; float* out; long long* ticks;
  CK(hipMalloc(&w, 256 * 256 * 4)); CK(hipMemset(w, 0, 256 * 256 * 4));
  CK(hipMalloc(&out, 512 * 512 * 4)); CK(hipMalloc(&ticks, 512 * 8 * 8));
  for (int bpc = 1; bpc <= 2; ++bpc) {
    run<0, 4>("MFMA only", w, out, ticks, bpc);
    run<8, 4>("+ selects", w, out, ticks, bpc);
    run<1, 4>("+ weight loads", w, out, ticks, bpc);
    run<2, 4>("+ X by 8 ds_read_b32", w, out, ticks, bpc);
    run<4, 4>("+ X by 2 ds_read_b128", w, out, ticks, bpc);
    run<1 | 2 | 8, 4>("the kernel's chunk (loads + b32 + selects)", w, out, ticks, bpc);
    run<1 | 4 | 8, 4>("loads + b128 + selects", w, out, ticks, bpc);
    run<1 | 4, 4>("loads + b128", w, out, ticks, bpc);
    run<1 | 2 | 8 | 16, 4>("the kernel's chunk, setprio(1) on MFMAs", w, out, ticks, bpc);
    run<1 | 2 | 8, 3>("the kernel's chunk", w, out, ticks, bpc);
    run<1 | 4 | 8, 3>("loads + b128 + selects", w, out, ticks, bpc);
  }
  return 0;
}

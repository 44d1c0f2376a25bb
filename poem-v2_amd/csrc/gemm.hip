// fp32 MFMA GEMM (Y = act(X W^T + b) + R) on packed weights, LayerNorm, small dense helpers.
// gfx950 only.  Each wave owns a 64 x (32*NT) output tile and streams its A rows and the packed W fragments
// straight from global/L2 (fp32 MFMA needs only 8 B/lane per 64 cycles, so no LDS staging is required);
// operands for the next K-chunk are prefetched into registers while the current chunk's MFMAs issue.
#include "common.h"

__global__ void pack_linear_kernel(const float* __restrict__ w, int N, int K, float4* __restrict__ out, int total) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int lane = i & 63;
  int kc = (i >> 6) % (K / 8);
  int nt = (i >> 6) / (K / 8);
  int row = nt * 32 + (lane & 31);
  int col = kc * 8 + 4 * (lane >> 5);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < N) {
    const float* p = w + (size_t)row * K + col;
    v = make_float4(p[0], p[1], p[2], p[3]);
  }
  out[i] = v;
}

extern "C" hipError_t poem_launch_pack_linear(const float* w, int N, int K, void* out, hipStream_t s) {
  int total = ((N + 31) / 32) * (K / 8) * 64;
  hipLaunchKernelGGL(pack_linear_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, N, K, (float4*)out, total);
  return hipGetLastError();
}

template <int NT, int ACT, bool RES>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const float* __restrict__ X, int ldx, const float4* __restrict__ Wp,
                                                   const float* __restrict__ bias, const float* __restrict__ R,
                                                   int ldr, float* __restrict__ Y, int ldy, int M, int N, int K,
                                                   int col_groups) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int cg = wave % col_groups;
  const int m0 = (wave / col_groups) * 64;
  if (m0 >= M) return;
  const int KC = K >> 3;
  const int row0 = min(m0 + r, M - 1), row1 = min(m0 + 32 + r, M - 1);
  const float4* xa0 = reinterpret_cast<const float4*>(X + (size_t)row0 * ldx + 4 * h);
  const float4* xa1 = reinterpret_cast<const float4*>(X + (size_t)row1 * ldx + 4 * h);
  const float4* wp = Wp + (size_t)(cg * NT) * KC * 64 + lane;

  f32x16 acc0[NT], acc1[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { acc0[n] = zero16(); acc1[n] = zero16(); }

  float4 a0 = xa0[0], a1 = xa1[0];
  float4 b[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) b[n] = wp[(size_t)n * KC * 64];

  for (int kc = 0; kc < KC; ++kc) {
    const int kn = min(kc + 1, KC - 1);
    float4 na0 = xa0[kn * 2], na1 = xa1[kn * 2];
    float4 nb[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) nb[n] = wp[((size_t)n * KC + kn) * 64];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float av0 = (&a0.x)[t], av1 = (&a1.x)[t];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float bv = (&b[n].x)[t];
        acc0[n] = mfma32(av0, bv, acc0[n]);
        acc1[n] = mfma32(av1, bv, acc1[n]);
      }
    }
    a0 = na0; a1 = na1;
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = nb[n];
  }

#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = (cg * NT + n) * 32 + r;
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m0 + mi * 32 + mfma_row(i, h);
        if (row < M) {
          float v = (mi == 0 ? acc0[n][i] : acc1[n][i]) + bv;
          if (ACT == 1) v = fmaxf(v, 0.f);
          if (ACT == 2) v = gelu_erf(v);
          if (RES) v += R[(size_t)row * ldr + col];
          Y[(size_t)row * ldy + col] = v;
        }
      }
    }
  }
}

template <int NT>
static hipError_t launch_gemm_nt(const float* X, int ldx, const void* Wp, const float* bias, const float* R, int ldr,
                                 float* Y, int ldy, int M, int N, int K, int act, hipStream_t s) {
  const int ntiles = (N + 31) / 32;
  const int col_groups = ntiles / NT;
  const long waves = (long)((M + 63) / 64) * col_groups;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
#define POEM_GEMM_CASE(A, RS)                                                                                       \
  hipLaunchKernelGGL((gemm_kernel<NT, A, RS>), grid, block, 0, s, X, ldx, (const float4*)Wp, bias, R, ldr, Y, ldy, \
                     M, N, K, col_groups)
  if (R) {
    if (act == 0) POEM_GEMM_CASE(0, true);
    else if (act == 1) POEM_GEMM_CASE(1, true);
    else POEM_GEMM_CASE(2, true);
  } else {
    if (act == 0) POEM_GEMM_CASE(0, false);
    else if (act == 1) POEM_GEMM_CASE(1, false);
    else POEM_GEMM_CASE(2, false);
  }
#undef POEM_GEMM_CASE
  return hipGetLastError();
}

extern "C" hipError_t poem_launch_gemm(const float* X, int ldx, const void* Wp, const float* bias, const float* R,
                                       int ldr, float* Y, int ldy, int M, int N, int K, int act, hipStream_t s) {
  const int ntiles = (N + 31) / 32;
  if (ntiles % 4 == 0) return launch_gemm_nt<4>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, s);
  if (ntiles % 2 == 0) return launch_gemm_nt<2>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, s);
  return launch_gemm_nt<1>(X, ldx, Wp, bias, R, ldr, Y, ldy, M, N, K, act, s);
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wave per row, two-pass (mean, then biased variance), eps inside the sqrt.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y, int rows,
                                                        int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * cols;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += xr[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)cols;
  float v = 0.f;
  for (int c = lane; c < cols; c += 64) { float d = xr[c] - mean; v += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const float rstd = 1.0f / sqrtf(v / (float)cols + eps);
  float* yr = y + (size_t)row * cols;
  for (int c = lane; c < cols; c += 64) yr[c] = (xr[c] - mean) * rstd * g[c] + b[c];
}

extern "C" hipError_t poem_launch_layernorm(const float* x, const float* g, const float* b, float* y, int rows,
                                            int cols, float eps, hipStream_t s) {
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y, rows, cols, eps);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Narrow Linear (N <= 8 outputs): out[row, n] = (base ? base[row, n] : 0) + x[row, :] . w[n, :] + b[n].
// One wave per row, lanes stride the K dim.  Used for reg_branch.2 (C -> 3) with the xyz residual.
__global__ __launch_bounds__(256) void narrow_linear_kernel(const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ base, float* __restrict__ out,
                                                            int rows, int K, int N) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  for (int n = 0; n < N; ++n) {
    float s = 0.f;
    for (int c = lane; c < K; c += 64) s = fmaf(xr[c], w[(size_t)n * K + c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[(size_t)row * N + n] = (base ? base[(size_t)row * N + n] : 0.f) + (s + (b ? b[n] : 0.f));
  }
}

extern "C" hipError_t poem_launch_narrow_linear(const float* x, int ldx, const float* w, const float* b,
                                                const float* base, float* out, int rows, int K, int N,
                                                hipStream_t s) {
  hipLaunchKernelGGL(narrow_linear_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, w, b, base, out, rows, K, N);
  return hipGetLastError();
}

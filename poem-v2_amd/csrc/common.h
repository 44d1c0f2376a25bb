// Shared device helpers for the gfx950 kernels of libpoem_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

// ---- per-device launch state -------------------------------------------------------------------------------------------
// include/poem_hip.h lets handles on different host threads drive different GPUs of one process, so nothing a launcher
// caches may be process-wide: the CU count and the "> 64 KB of dynamic LDS" opt-in of a kernel are kept per device id.
static inline int poem_device_cus() {
  static std::atomic<int> cache[64];
  int dev = 0, c = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if ((c = cache[dev].load(std::memory_order_relaxed)) > 0) return c;
  if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
  cache[dev].store(c, std::memory_order_relaxed);
  return c;
}
// `done`: one function-local static bit mask per kernel instantiation (bit = device id)
static inline hipError_t poem_optin_lds(const void* kern, size_t bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  const bool idx = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
  if (idx && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess && idx) done.fetch_or(1ull << dev, std::memory_order_release);
  return e;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define POEM_WAVE 64

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32), exact fp32 (k-ordered fma chain).
// Operand layout (lane l): A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31];
// result layout: col j = l & 31, row i = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), reg in [0,16).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int mfma_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// exchange with the lane 32 positions away (the other half-wave)
__device__ __forceinline__ float xhalf(float v) { return __shfl_xor(v, 32, 64); }

// Reductions over the two half-waves (lane l and lane l ^ 32) with one v_permlane32_swap: after the swap of a value with
// itself, r[0] holds the lower half's values in both halves and r[1] the upper half's, so r[0] (+) r[1] is the
// two-half reduction in every lane (no LDS round trip, two VALU instructions).
__device__ __forceinline__ float half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float half_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// exp(x) for x <= 0 (softmax numerators) with ~1 ulp accuracy in 6 VALU ops: x*log2(e) is split into a rounded head
// (fed to the hardware exp2) and an fma-recovered tail applied as a first-order correction.
__device__ __forceinline__ float exp_neg(float x) {
  const float L2E_HI = 1.44269502162933349609375f;       // fp32(log2 e)
  const float L2E_LO = 1.925963033500011e-08f;           // log2 e - fp32(log2 e)
  const float LN2 = 0.693147180559945309417f;
  const float t = x * L2E_HI;
  float e = fmaf(x, L2E_HI, -t);
  e = fmaf(x, L2E_LO, e);
  const float r = __builtin_amdgcn_exp2f(t);
  return fmaf(r, e * LN2, r);
}

// Wave-wide 1 KiB fragment load through a buffer descriptor: address = SGPR base + SGPR byte offset + lane * 16.
// The per-chunk address update is then scalar arithmetic; a per-lane 64-bit pointer costs two VALU adds per load, and
// every VALU issue slot inside an MFMA loop costs matrix-pipe time (tools/lab/issue_lab: 72.6 -> 66 cycles per MFMA).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frag_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 frag_load(__amdgpu_buffer_rsrc_t rs, int lane_off_bytes, int scalar_off_bytes) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off_bytes, scalar_off_bytes, 0);
  return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

// ReLU as torch evaluates it: a NaN stays a NaN (v_max_f32 returns its other operand for a quiet NaN -- fmaxf(NaN, 0) = 0 -- and
// would turn a poisoned sample into finite garbage: the reference's `torch.nan_to_num(interm_ref_pts)` in front of the
// de-normalisation (ptEmb_head.py:944) relies on the NaN reaching it).  One compare + select instead of one max: used at the
// Linear epilogues between the sampled features and the coordinate update, not inside the vector attention (whose values carry
// the NaN past its own ReLUs).
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }

__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }

// Packed Linear weight image used by every MFMA kernel here ("fragment order"):
//   P[(nt * KC + kc) * 64 + lane] = float4( W[32*nt + (lane&31)][8*kc + 4*(lane>>5) + 0..3] ),  KC = K/8,
// zero-filled for rows >= N.  One wave-wide 1 KiB load yields the operand of four consecutive MFMA k-steps
// for a 32-row tile: k-step t of chunk kc multiplies channels (8kc+t | 8kc+4+t) held by the two half-waves.
static inline size_t packed_linear_floats(int N, int K) { return (size_t)((N + 31) / 32) * (size_t)(K / 8) * 64 * 4; }

// ---- shared by the input kernels of the whole path (misc.hip prep_xyz_kernel, sample.hip invert_extr_kernel, merge.hip
// input_tables_kernel) --------------------------------------------------------------------------------------------------------
// inverse of a camera->master 4x4: fp64 Gauss-Jordan with partial pivoting, rounded to fp32 (torch.linalg.inv's role, collation.py:56-61)
__device__ inline void invert4x4(const float* __restrict__ m, float* __restrict__ out) {
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { a[i][j] = (double)m[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    double best = fabs(a[c][c]);
    for (int i = c + 1; i < 4; ++i) if (fabs(a[i][c]) > best) { best = fabs(a[i][c]); piv = i; }
    if (piv != c) for (int j = 0; j < 8; ++j) { double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
    const double inv = 1.0 / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= inv;
    for (int i = 0; i < 4; ++i) if (i != c) {
      const double f = a[i][c];
      for (int j = 0; j < 8; ++j) a[i][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[i * 4 + j] = (float)a[i][4 + j];
}

// element i of the coordinate normalisation (see prep_xyz_kernel): centre = reference_joints[:, 9]; pt_xyz = ((bps + c) - c) / radius;
// query_xyz = ((c + template) - c) / radius, evaluated exactly the reference's way
__device__ inline void poem_prep_xyz_elem(long i, const float* __restrict__ ref_joints, const float* __restrict__ bps,
                                          const float* __restrict__ tmpl, float* __restrict__ centre, float* __restrict__ pt_xyz,
                                          float* __restrict__ query_xyz, int B, int S, int Q, float radius) {
  const long n_pt = (long)B * S * 3, n_q = (long)B * Q * 3;
  if (i < n_pt) {
    const int d = (int)(i % 3), s = (int)((i / 3) % S), b = (int)(i / (3L * S));
    const float c = ref_joints[((size_t)b * 21 + 9) * 3 + d];
    const float w = __fadd_rn(bps[s * 3 + d], c);
    pt_xyz[i] = __fdiv_rn(__fsub_rn(w, c), radius);
  } else if (i < n_pt + n_q) {
    const long k = i - n_pt;
    const int d = (int)(k % 3), qq = (int)((k / 3) % Q), b = (int)(k / (3L * Q));
    const float c = ref_joints[((size_t)b * 21 + 9) * 3 + d];
    const float w = __fadd_rn(c, tmpl[qq * 3 + d]);
    query_xyz[k] = __fdiv_rn(__fsub_rn(w, c), radius);
  } else if (i < n_pt + n_q + 3L * B) {
    const long k = i - n_pt - n_q;
    const int d = (int)(k % 3), b = (int)(k / 3);
    centre[k] = ref_joints[((size_t)b * 21 + 9) * 3 + d];
  }
}

// BERT-style multi-head cross attention of the 799 queries over the 4096 basis points, flash-style, exact fp32.
// softmax(Q K^T / sqrt(dh)) V without materialising the (B, heads, Q, S) score tensor.
//
// Block = NWV waves, one (batch, head, 32*NWV-query tile); every wave owns 32 queries.  Per 32-key tile:
//   S^T = K . Q^T      MFMA A = K tile from LDS (row = key), B = Q fragment held in registers (lane = query)
//                      -> D[key][query]: lane = query, registers = keys, so the per-query softmax statistics are
//                         lane-local (+1 exchange with the other half-wave)
//   O^T += V^T . P^T   MFMA A = V tile from LDS (lane = channel d), B = P registers used *directly* as the operand:
//                      k-step i consumes key row (i&3)+8(i>>2)+4*half, exactly the key that register i holds.
// K/V tiles are staged through double-buffered LDS, next tile prefetched into registers during the MFMAs.
#include "common.h"

template <int DH, int NWV>
__global__ __launch_bounds__(NWV * 64) void cross_attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, float* __restrict__ ctx,
                                                              int NQ, int NK, int C) {
  constexpr int DT = (DH + 31) / 32;       // 32-wide channel tiles of the output
  constexpr int KC = DH / 8;               // k-chunks of the QK^T contraction
  constexpr int KS = DH + 4;               // K tile row stride (floats): conflict-free ds_read_b128 by row
  constexpr int VS = 32 * DT;              // V tile row stride (zero padded to a whole channel tile)
  constexpr int NT = NWV * 64;
  constexpr int F4 = 8 * DH;               // float4s per 32-key tile
  constexpr int LD = (F4 + NT - 1) / NT;   // float4 loads per thread per tile
  __shared__ __attribute__((aligned(16))) float Ks[2][32 * KS];
  __shared__ __attribute__((aligned(16))) float Vs[2][32 * VS];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int qrow = blockIdx.x * (32 * NWV) + wv * 32 + r;
  const bool wave_live = blockIdx.x * (32 * NWV) + wv * 32 < NQ;
  const int qclamp = min(qrow, NQ - 1);
  const float* kb = k + (size_t)b * NK * C + head * DH;
  const float* vb = v + (size_t)b * NK * C + head * DH;

  // zero the padded V columns once (DH < 32 only)
  if (DH < VS) {
    for (int i = tid; i < 2 * 32 * VS; i += NT) (&Vs[0][0])[i] = 0.f;
    __syncthreads();
  }

  // Q fragment: lane (query r, half h) holds Q[r][8kc + 4h + t]
  float4 qf[KC];
  {
    const float* qp = q + ((size_t)b * NQ + qclamp) * C + head * DH + 4 * h;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) qf[kc] = *reinterpret_cast<const float4*>(qp + 8 * kc);
  }

  float4 kreg[LD], vreg[LD];
#define POEM_LOAD_TILE(KT)                                                                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < LD; ++i_) {                                                           \
    const int f_ = tid + NT * i_;                                                                               \
    if (F4 % NT == 0 || f_ < F4) {                                                                              \
      const int row_ = f_ / (DH / 4), c4_ = f_ % (DH / 4);                                                      \
      kreg[i_] = *reinterpret_cast<const float4*>(kb + (size_t)((KT) * 32 + row_) * C + 4 * c4_);               \
      vreg[i_] = *reinterpret_cast<const float4*>(vb + (size_t)((KT) * 32 + row_) * C + 4 * c4_);               \
    }                                                                                                           \
  }
#define POEM_STORE_TILE(BUF)                                                                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < LD; ++i_) {                                                           \
    const int f_ = tid + NT * i_;                                                                               \
    if (F4 % NT == 0 || f_ < F4) {                                                                              \
      const int row_ = f_ / (DH / 4), c4_ = f_ % (DH / 4);                                                      \
      *reinterpret_cast<float4*>(&Ks[BUF][row_ * KS + 4 * c4_]) = kreg[i_];                                     \
      *reinterpret_cast<float4*>(&Vs[BUF][row_ * VS + 4 * c4_]) = vreg[i_];                                     \
    }                                                                                                           \
  }

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) o[d] = zero16();
  float m_run = -INFINITY, l_run = 0.f;
  const float inv_sdh = 1.0f / sqrtf((float)DH);   // exact for dh in {16, 64, 256}; <= 1 ulp from the division otherwise

  const int ntiles = NK / 32;
  POEM_LOAD_TILE(0)
  POEM_STORE_TILE(0)
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int buf = kt & 1;
    { const int kn = min(kt + 1, ntiles - 1); POEM_LOAD_TILE(kn) }
    if (wave_live) {
      f32x16 s = zero16();
      const float* kr = &Ks[buf][r * KS + 4 * h];
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const float4 a = *reinterpret_cast<const float4*>(kr + 8 * kc);
        s = mfma32(a.x, qf[kc].x, s);
        s = mfma32(a.y, qf[kc].y, s);
        s = mfma32(a.z, qf[kc].z, s);
        s = mfma32(a.w, qf[kc].w, s);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = s[i] * inv_sdh; mx = fmaxf(mx, s[i]); }
      mx = fmaxf(mx, xhalf(mx));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = (m_new == m_run) ? 1.0f : (m_run == -INFINITY ? 0.0f : exp_neg(m_run - m_new));
      float ps = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = exp_neg(s[i] - m_new); ps += s[i]; }
      ps += xhalf(ps);
      l_run = l_run * alpha + ps;
      m_run = m_new;
      if (__any(alpha != 1.0f)) {   // exact skip: alpha == 1 whenever this lane's running max did not move
#pragma unroll
        for (int d = 0; d < DT; ++d) {
#pragma unroll
          for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float* vr = &Vs[buf][mfma_row(i, h) * VS + r];
#pragma unroll
        for (int d = 0; d < DT; ++d) o[d] = mfma32(vr[32 * d], s[i], o[d]);
      }
    }
    POEM_STORE_TILE(buf ^ 1)   // (the final, redundant store targets the buffer nobody reads again)
    __syncthreads();
  }

  if (wave_live && qrow < NQ) {
    float* out = ctx + ((size_t)b * NQ + qrow) * C + head * DH;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ch = 32 * d + mfma_row(i, h);
        if (ch < DH) out[ch] = o[d][i] / l_run;
      }
    }
  }
}

extern "C" hipError_t poem_launch_cross_attention(const float* q, const float* k, const float* v, float* ctx, int B,
                                                  int NQ, int NK, int C, int heads, hipStream_t s) {
  const int dh = C / heads;
  constexpr int NWV = 4;
  dim3 grid((NQ + 32 * NWV - 1) / (32 * NWV), heads, B), block(NWV * 64);
#define POEM_ATTN_CASE(D)                                                                                       \
  case D:                                                                                                       \
    hipLaunchKernelGGL((cross_attn_kernel<D, NWV>), grid, block, 0, s, q, k, v, ctx, NQ, NK, C);                \
    break
  switch (dh) {
    POEM_ATTN_CASE(8);
    POEM_ATTN_CASE(16);
    POEM_ATTN_CASE(32);
    POEM_ATTN_CASE(64);
    POEM_ATTN_CASE(128);
    default:
      return hipErrorInvalidValue;
  }
#undef POEM_ATTN_CASE
  return hipGetLastError();
}

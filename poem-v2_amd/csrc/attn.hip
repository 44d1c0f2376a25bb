// BERT-style multi-head cross attention of the 799 queries over the 4096 basis points, flash-style, exact fp32.
// softmax(Q K^T / sqrt(dh)) V without materialising the (B, heads, Q, S) score tensor.
//
// Block = NWV waves, one (batch, head, 32*NWV-query group, key split); every wave owns 32 queries.  Per 32-key tile:
//   S^T = K . Q^T      MFMA A = K tile from LDS (row = key), B = Q fragment held in registers (lane = query)
//                      -> D[key][query]: lane = query, registers = keys, so the per-query softmax statistics are
//                         lane-local (+1 exchange with the other half-wave)
//   O^T += V^T . P^T   MFMA A = V tile from LDS (lane = channel d), B = P registers used *directly* as the operand:
//                      k-step i consumes key row (i&3)+8(i>>2)+4*half, exactly the key that register i holds.
// K/V tiles are staged through double-buffered LDS, next tile prefetched into registers during the MFMAs.
//
// Work decomposition (the kernel is MFMA-bound; what matters is an even load on the 1024 SIMDs):
//   * 4 waves per block = one wave per SIMD (5-wave blocks covering 799 = 25 x 32 queries exactly measured 20 % slower:
//     two waves of one barrier-coupled block on one SIMD);
//   * the key axis is split KSPLIT ways when the unsplit grid would be a single ragged round over the 4-blocks-per-CU
//     residency; each split writes un-normalised partial (O, m, l) in fragment order and `attn_combine_kernel` merges
//     them with the usual log-sum-exp weights;
//   * <= 128 VGPRs -> 4 waves per SIMD, so one wave's softmax VALU work hides under the others' MFMAs
//     (measured: 86 % MFMA-busy while a round is full; the rest of the gap is round quantisation and the 31-query tail).
#include "common.h"
#include <cstdlib>

#ifdef POEM_ATTN_DBG   // tools/lab only
__device__ long long attn_dbg[8 * 4 * 8];
#endif

template <int DH, int NWV, int MINW>
__global__ __launch_bounds__(NWV * 64, MINW) void cross_attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, float* __restrict__ ctx,
                                                                 float4* __restrict__ part_o, float2* __restrict__ part_ml,
                                                                 int NQ, int NK, int C, int ldkv, int ksplit) {
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
  const int b = blockIdx.z / ksplit, ks = blockIdx.z % ksplit, head = blockIdx.y;
  const int heads = gridDim.y;
  const int qtile = blockIdx.x * NWV + wv;
  const int qrow = qtile * 32 + r;
  const bool wave_live = qtile * 32 < NQ;
  const int qclamp = min(qrow, NQ - 1);
  const float* kb = k + (size_t)b * NK * ldkv + head * DH;
  const float* vb = v + (size_t)b * NK * ldkv + head * DH;

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
      kreg[i_] = *reinterpret_cast<const float4*>(kb + (size_t)((KT) * 32 + row_) * ldkv + 4 * c4_);            \
      vreg[i_] = *reinterpret_cast<const float4*>(vb + (size_t)((KT) * 32 + row_) * ldkv + 4 * c4_);            \
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

  const int tps = NK / 32 / ksplit;                // key tiles of this split
  const int kt0 = ks * tps, kt1 = kt0 + tps;
  POEM_LOAD_TILE(kt0)
  POEM_STORE_TILE(0)
  __syncthreads();
#ifdef POEM_ATTN_DBG
  long long ta = 0, tb = 0, tc = 0, tstart = clock64();
#endif
  for (int kt = kt0; kt < kt1; ++kt) {
#ifdef POEM_ATTN_DBG
    const long long t0 = clock64();
#endif
    const int buf = (kt - kt0) & 1;
    { const int kn = min(kt + 1, kt1 - 1); POEM_LOAD_TILE(kn) }
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
#ifdef POEM_ATTN_DBG
      ta += clock64() - t0;
      const long long t1 = clock64();
#endif
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
#ifdef POEM_ATTN_DBG
      tb += clock64() - t1;
#endif
    }
#ifdef POEM_ATTN_DBG
    const long long t2 = clock64();
#endif
    POEM_STORE_TILE(buf ^ 1)   // (the final, redundant store targets the buffer nobody reads again)
    __syncthreads();
#ifdef POEM_ATTN_DBG
    tc += clock64() - t2;
#endif
  }
#ifdef POEM_ATTN_DBG
  if (blockIdx.x < 2 && blockIdx.y == 1 && blockIdx.z >= 20 && blockIdx.z < 22 && lane == 0 && wv < 4) {
    long long* d = &attn_dbg[(((blockIdx.z - 20) * 2 + blockIdx.x) * 4 + wv) * 8];
    d[0] = kt1 - kt0; d[1] = clock64() - tstart; d[2] = ta; d[3] = tb; d[4] = tc;
  }
#endif
#undef POEM_LOAD_TILE
#undef POEM_STORE_TILE

  if (!wave_live) return;
  if (ksplit == 1) {
    if (qrow < NQ) {
      float* out = ctx + ((size_t)b * NQ + qrow) * C + head * DH;
      const float inv_l = 1.0f / l_run;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = 32 * d + 8 * g + 4 * h;
          if (ch < DH)
            *reinterpret_cast<float4*>(out + ch) =
                make_float4(o[d][4 * g] / l_run, o[d][4 * g + 1] / l_run, o[d][4 * g + 2] / l_run, o[d][4 * g + 3] / l_run);
        }
      }
      (void)inv_l;
    }
  } else {
    // partial (O, m, l): fragment order, one coalesced 1 KiB store per (channel tile, register group)
    const int qtiles = (NQ + 31) / 32;
    const size_t slab = ((size_t)((b * heads + head) * ksplit + ks) * qtiles + qtile);
    float4* po = part_o + slab * (DT * 4) * 64 + lane;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        po[(d * 4 + g) * 64] = make_float4(o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]);
    if (h == 0) part_ml[slab * 32 + r] = make_float2(m_run, l_run);
  }
}

// ctx[b, q, head*DH + c] = sum_s e^{m_s - M} O_s[c] / sum_s e^{m_s - M} l_s      (M = max_s m_s)
template <int DH>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float4* __restrict__ part_o,
                                                           const float2* __restrict__ part_ml, float* __restrict__ ctx,
                                                           int NQ, int C, int heads, int ksplit, int total_waves) {
  constexpr int DT = (DH + 31) / 32;
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);       // one wave per (b, head, qtile)
  if (wid >= total_waves) return;
  const int qtiles = (NQ + 31) / 32;
  const int qtile = wid % qtiles, bh = wid / qtiles, head = bh % heads, b = bh / heads;
  const int qrow = qtile * 32 + r;
  float w[16], M = -INFINITY;      // ksplit <= 16
  for (int s = 0; s < ksplit; ++s) {
    const float2 ml = part_ml[((size_t)(bh * ksplit + s) * qtiles + qtile) * 32 + r];
    w[s] = ml.x;
    M = fmaxf(M, ml.x);
  }
  float den = 0.f;
  for (int s = 0; s < ksplit; ++s) {
    const float2 ml = part_ml[((size_t)(bh * ksplit + s) * qtiles + qtile) * 32 + r];
    w[s] = (w[s] == M) ? 1.0f : exp_neg(w[s] - M);
    den = fmaf(w[s], ml.y, den);
  }
  if (qrow >= NQ) return;
  float* out = ctx + ((size_t)b * NQ + qrow) * C + head * DH;
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = 32 * d + 8 * g + 4 * h;
      if (ch >= DH) continue;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s = 0; s < ksplit; ++s) {
        const float4 p = part_o[(((size_t)(bh * ksplit + s) * qtiles + qtile) * (DT * 4) + d * 4 + g) * 64 + lane];
        acc.x = fmaf(w[s], p.x, acc.x); acc.y = fmaf(w[s], p.y, acc.y);
        acc.z = fmaf(w[s], p.z, acc.z); acc.w = fmaf(w[s], p.w, acc.w);
      }
      *reinterpret_cast<float4*>(out + ch) = make_float4(acc.x / den, acc.y / den, acc.z / den, acc.w / den);
    }
}

// scratch floats needed by the split-key path (0 when the shape runs unsplit)
extern "C" size_t poem_cross_attention_scratch_floats(int B, int NQ, int NK, int C, int heads, int* ksplit_out) {
  const int dh = C / heads;
  const int qtiles = (NQ + 31) / 32;
  const int ktiles = NK / 32;
  // 4-wave blocks, 4 resident per CU (1024 slots).  Two key splits measured best on the headline shape (896 x 2 blocks);
  // the split count deliberately does NOT depend on the batch size: a sample's result is then bit-identical whatever
  // batch it travels in (the property data-parallel sharding relies on).
  (void)B; (void)heads; (void)qtiles;
  const int ks = (ktiles % 2 == 0 && ktiles / 2 >= 16) ? 2 : 1;
  if (ksplit_out) *ksplit_out = ks;
  if (ks == 1) return 0;
  const int DT = (dh + 31) / 32;
  const size_t slabs = (size_t)B * heads * ks * qtiles;
  return slabs * (size_t)DT * 4 * 64 * 4 + slabs * 32 * 2;
}

template <int NWV, int MINW>
static hipError_t launch_attn(const float* q, const float* k, const float* v, float* ctx, int B, int NQ, int NK, int C,
                              int heads, int ldkv, float* scratch, int ksplit, hipStream_t s) {
  const int dh = C / heads;
  const int qtiles = (NQ + 31) / 32;
  const int DT = (dh + 31) / 32;
  const size_t slabs = (size_t)B * heads * ksplit * qtiles;
  float4* part_o = reinterpret_cast<float4*>(scratch);
  float2* part_ml = reinterpret_cast<float2*>(scratch + slabs * (size_t)DT * 4 * 64 * 4);
  dim3 grid((qtiles + NWV - 1) / NWV, heads, B * ksplit), block(NWV * 64);
  const int waves = B * heads * qtiles;
#define POEM_ATTN_CASE(D)                                                                                       \
  case D:                                                                                                       \
    hipLaunchKernelGGL((cross_attn_kernel<D, NWV, MINW>), grid, block, 0, s, q, k, v, ctx, part_o, part_ml, NQ, NK, C, \
                       ldkv, ksplit);                                                                           \
    if (ksplit > 1)                                                                                             \
      hipLaunchKernelGGL((attn_combine_kernel<D>), dim3((waves + 3) / 4), dim3(256), 0, s, part_o, part_ml, ctx, NQ, \
                         C, heads, ksplit, waves);                                                              \
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

extern "C" hipError_t poem_launch_cross_attention(const float* q, const float* k, const float* v, float* ctx, int B,
                                                  int NQ, int NK, int C, int heads, int ldkv, float* scratch,
                                                  hipStream_t s) {
  int ksplit = 1;
  const size_t need = poem_cross_attention_scratch_floats(B, NQ, NK, C, heads, &ksplit);
  if (need && !scratch) ksplit = 1;
#ifdef POEM_LAB
  int cfg = 0;
  if (const char* e = getenv("POEM_ATTN_KS")) { const int kk = atoi(e); if (kk >= 1 && kk <= ksplit) ksplit = kk; }
  if (const char* e = getenv("POEM_ATTN_CFG")) cfg = atoi(e);
  switch (cfg) {
    case 1: return launch_attn<4, 3>(q, k, v, ctx, B, NQ, NK, C, heads, ldkv, scratch, ksplit, s);
    case 2: return launch_attn<4, 4>(q, k, v, ctx, B, NQ, NK, C, heads, ldkv, scratch, ksplit, s);
    case 3: return launch_attn<5, 3>(q, k, v, ctx, B, NQ, NK, C, heads, ldkv, scratch, ksplit, s);
    case 4: return launch_attn<8, 4>(q, k, v, ctx, B, NQ, NK, C, heads, ldkv, scratch, ksplit, s);
    case 5: return launch_attn<8, 3>(q, k, v, ctx, B, NQ, NK, C, heads, ldkv, scratch, ksplit, s);
    default: break;
  }
#endif
  // head dim 128 (POEM-large) needs ~200 VGPRs: two waves per SIMD without spills beat three with
  if (C / heads >= 128) return launch_attn<4, 2>(q, k, v, ctx, B, NQ, NK, C, heads, ldkv, scratch, ksplit, s);
  return launch_attn<4, 3>(q, k, v, ctx, B, NQ, NK, C, heads, ldkv, scratch, ksplit, s);
}

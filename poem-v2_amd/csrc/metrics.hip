// Evaluation metrics on the device (the step right after the path).  The reference moves every
// batch to the host (`.cpu().numpy()`, lib/metrics/pa_eval.py:48-49, pck.py:48-53) and loops over samples in Python /
// scipy; here one launch per batch accumulates into small device buffers that are read once at the end.
#include "common.h"

// ---- Procrustes-aligned end-point error --------------------------------------------------------------------------
// PAEval.align_w_scale (lib/metrics/pa_eval.py:104-124): centre both point sets, scale each to unit Frobenius norm
// (+1e-8), R, s = scipy.linalg.orthogonal_procrustes(gt_n, pred_n)  [R = U V^T, s = sum(w) for gt_n^T pred_n = U w V^T;
// no reflection handling -- exactly as upstream], aligned = pred_n R^T s * s_gt + mean_gt.
// One wave per sample; the 3x3 polar factor comes from a fp64 Jacobi eigen-decomposition of M^T M.
// out[b] = (mean_i |aligned_i - gt_i|, mean_i |pred_i - gt_i|)          (get_dist, pa_eval.py:40-43)
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void pa_epe_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                     float* __restrict__ out, int B, int P) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* p = pred + (size_t)b * P * 3;
  const float* g = gt + (size_t)b * P * 3;
  double sp[3] = {0, 0, 0}, sg[3] = {0, 0, 0};
  for (int i = lane; i < P; i += 64)
    for (int d = 0; d < 3; ++d) { sp[d] += p[i * 3 + d]; sg[d] += g[i * 3 + d]; }
  double mp[3], mg[3];
  for (int d = 0; d < 3; ++d) { mp[d] = wave_sum(sp[d]) / P; mg[d] = wave_sum(sg[d]) / P; }
  double np2 = 0, ng2 = 0, M[3][3] = {};            // M = gt_c^T pred_c (un-normalised; scales factor out below)
  for (int i = lane; i < P; i += 64) {
    double pc[3], gc[3];
    for (int d = 0; d < 3; ++d) { pc[d] = p[i * 3 + d] - mp[d]; gc[d] = g[i * 3 + d] - mg[d]; }
    for (int d = 0; d < 3; ++d) { np2 += pc[d] * pc[d]; ng2 += gc[d] * gc[d]; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r][c] += gc[r] * pc[c];
  }
  np2 = wave_sum(np2); ng2 = wave_sum(ng2);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r][c] = wave_sum(M[r][c]);
  const double s1 = sqrt(ng2) + 1e-8, s2 = sqrt(np2) + 1e-8;       // pa_eval.py:113-116
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r][c] /= (s1 * s2);
  // S = M^T M = V w^2 V^T  (Jacobi, fp64)
  double S[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { S[r][c] = 0; for (int k = 0; k < 3; ++k) S[r][c] += M[k][r] * M[k][c]; }
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = S[0][1] * S[0][1] + S[0][2] * S[0][2] + S[1][2] * S[1][2];
    const double diag = S[0][0] * S[0][0] + S[1][1] * S[1][1] + S[2][2] * S[2][2];
    if (off <= 1e-40 * diag) break;
    for (int pp = 0; pp < 2; ++pp)
      for (int q = pp + 1; q < 3; ++q) {
        if (S[pp][q] == 0.0) continue;
        const double theta = (S[q][q] - S[pp][pp]) / (2.0 * S[pp][q]);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
        for (int k = 0; k < 3; ++k) { const double a = S[k][pp], bq = S[k][q]; S[k][pp] = c * a - s * bq; S[k][q] = s * a + c * bq; }
        for (int k = 0; k < 3; ++k) { const double a = S[pp][k], bq = S[q][k]; S[pp][k] = c * a - s * bq; S[q][k] = s * a + c * bq; }
        for (int k = 0; k < 3; ++k) { const double a = V[k][pp], bq = V[k][q]; V[k][pp] = c * a - s * bq; V[k][q] = s * a + c * bq; }
      }
  }
  double w[3], scale = 0;
  for (int k = 0; k < 3; ++k) { w[k] = sqrt(fmax(S[k][k], 0.0)); scale += w[k]; }
  // R = U V^T = M V diag(1/w) V^T   (3x3); a vanishing singular value leaves its direction out (degenerate input)
  double MV[3][3], R[3][3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { MV[r][c] = 0; for (int k = 0; k < 3; ++k) MV[r][c] += M[r][k] * V[k][c]; }
  const double wmax = fmax(w[0], fmax(w[1], w[2]));
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      R[r][c] = 0;
      for (int k = 0; k < 3; ++k) if (w[k] > 1e-14 * wmax) R[r][c] += MV[r][k] / w[k] * V[c][k];
    }
  // aligned_i = (pred_c_i / s2) R^T * scale * s1 + mean_gt
  double dpa = 0, draw = 0;
  for (int i = lane; i < P; i += 64) {
    double pc[3], e2 = 0, r2 = 0;
    for (int d = 0; d < 3; ++d) pc[d] = (p[i * 3 + d] - mp[d]) / s2;
    for (int d = 0; d < 3; ++d) {
      const double al = (R[d][0] * pc[0] + R[d][1] * pc[1] + R[d][2] * pc[2]) * scale * s1 + mg[d];
      const double e = al - g[i * 3 + d], rr = (double)p[i * 3 + d] - (double)g[i * 3 + d];
      e2 += e * e; r2 += rr * rr;
    }
    dpa += sqrt(e2); draw += sqrt(r2);
  }
  dpa = wave_sum(dpa); draw = wave_sum(draw);
  if (lane == 0) { out[b * 2] = (float)(dpa / P); out[b * 2 + 1] = (float)(draw / P); }
}

extern "C" hipError_t poem_launch_pa_epe(const float* pred, const float* gt, float* out, int B, int P, hipStream_t s) {
  hipLaunchKernelGGL(pa_epe_kernel, dim3((B + 3) / 4), dim3(256), 0, s, pred, gt, out, B, P);
  return hipGetLastError();
}

// ---- PCK accumulators ------------------------------------------------------------------------------------------------
// _PCKMetric.feed / _get_pck (lib/metrics/pck.py:36-96): per key point the Euclidean distances of all fed samples;
// pck(t) = mean(dist <= t).  Instead of keeping the lists, counts[k][t] (#dist <= thr_t), sum[k] and n[k] accumulate on
// the device; thr_t = linspace(vmin, vmax, steps)[t] evaluated in fp64, the distance in fp32 with numpy's operation
// order ((dx^2 + dy^2) + dz^2, sqrt), the comparison in fp64 -- the same outcomes as the reference's numpy code.
__global__ void pck_accumulate_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int B, int P,
                                      double vmin, double vmax, int steps, unsigned int* __restrict__ counts,
                                      double* __restrict__ sum, unsigned int* __restrict__ n, float* __restrict__ dist_out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;      // key point
  if (k >= P) return;
  double acc = 0;
  for (int b = 0; b < B; ++b) {
    const float* p = pred + ((size_t)b * P + k) * 3;
    const float* g = gt + ((size_t)b * P + k) * 3;
    float d;
    {
      // numpy's operation order, every product and sum individually rounded (no v_fma contraction: the `<=` below must
      // agree with the reference's for a distance within an ulp of a threshold)
#pragma clang fp contract(off)
      const float dx = p[0] - g[0], dy = p[1] - g[1], dz = p[2] - g[2];
      const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
      d = __fsqrt_rn((xx + yy) + zz);
    }
    acc += (double)d;
    if (dist_out) dist_out[(size_t)b * P + k] = d;
    // thresholds ascend: first index whose threshold admits d, every later one does too
    const double step = steps > 1 ? (vmax - vmin) / (double)(steps - 1) : 0.0;
    for (int t = 0; t < steps; ++t) {
      const double thr = (t == steps - 1 && steps > 1) ? vmax : vmin + (double)t * step;   // numpy.linspace
      if ((double)d <= thr) counts[(size_t)k * steps + t] += 1u;
    }
  }
  sum[k] += acc;
  n[k] += (unsigned)B;
}

extern "C" hipError_t poem_launch_pck_accumulate(const float* pred, const float* gt, int B, int P, double vmin,
                                                 double vmax, int steps, unsigned int* counts, double* sum,
                                                 unsigned int* n, float* dist_out, hipStream_t s) {
  hipLaunchKernelGGL(pck_accumulate_kernel, dim3((P + 63) / 64), dim3(64), 0, s, pred, gt, B, P, vmin, vmax, steps, counts,
                     sum, n, dist_out);
  return hipGetLastError();
}

// ---- MANO vertices -> OpenPose-ordered joints ----------------------------------------------------------------------
// mano_to_openpose (lib/utils/transform.py:836-872): 16 joints = J_regressor (16,778) . verts, 5 finger tips = vertices
// {744, 320, 443, 555, 672} (CONST.MANO_KPID_2_VERTICES, lib/utils/misc.py:76-82), concatenated and re-ordered to the
// OpenPose numbering.  testing_step applies it to predicted AND ground-truth vertices of every batch (POEM.py:602-603).
// One wave per (sample, output joint); lane-strided partial sums in a fixed order -> batch-independent results.
__constant__ int kOpenposeFromMano[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};
__constant__ int kTipVertex[5] = {744, 320, 443, 555, 672};

__global__ __launch_bounds__(64) void mano_to_openpose_kernel(const float* __restrict__ jreg, const float* __restrict__ verts,
                                                              float* __restrict__ joints, int nverts) {
  const int b = blockIdx.x / 21, o = blockIdx.x % 21, lane = threadIdx.x;
  const int src = kOpenposeFromMano[o];
  const float* v = verts + (size_t)b * nverts * 3;
  float* out = joints + ((size_t)b * 21 + o) * 3;
  if (src >= 16) {
    if (lane < 3) out[lane] = v[kTipVertex[src - 16] * 3 + lane];
    return;
  }
  const float* w = jreg + (size_t)src * nverts;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int i = lane; i < nverts; i += 64) {
    const float wi = w[i];
    a0 = fmaf(wi, v[i * 3 + 0], a0); a1 = fmaf(wi, v[i * 3 + 1], a1); a2 = fmaf(wi, v[i * 3 + 2], a2);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64);
  }
  if (lane == 0) { out[0] = a0; out[1] = a1; out[2] = a2; }
}

extern "C" hipError_t poem_launch_mano_to_openpose(const float* jreg, const float* verts, float* joints, int B, int nverts,
                                                   hipStream_t s) {
  hipLaunchKernelGGL(mano_to_openpose_kernel, dim3(B * 21), dim3(64), 0, s, jreg, verts, joints, nverts);
  return hipGetLastError();
}

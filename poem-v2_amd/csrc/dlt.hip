// Ragged batched DLT triangulation of the 21 hand joints (the producer of `reference_joints`).
// Replaces lib/utils/triangulation.py:5-45 (batch_triangulate_dlt_torch) and the per-sample Python loop around it
// (lib/models/POEM.py:284-299 upstream): for every (sample, joint)
//   M_n = K_n . T_n[:3, :]            (T_n = inv(cam_extr_n) = master -> camera; fp32 like the reference's matmul)
//   A   = [u_n M_n[2] - M_n[0] ; v_n M_n[2] - M_n[1]]_n          (2 N_i x 4)
//   x   = right singular vector of A for the smallest singular value;  X = x[:3] / (x[3] + 1e-7)
// The reference calls torch.linalg.svd on every (2N x 4) matrix; here the 4x4 normal matrix A^T A is accumulated in
// fp64 and its smallest eigenvector found with cyclic Jacobi rotations in fp64 (same subspace; measured closer to the
// fp64 SVD than the reference's fp32 SVD is).  One thread per (sample, joint): the whole stage is ~700 threads.
#include "common.h"

__device__ inline void dlt_invert4x4(const float* __restrict__ m, double (&o)[4][4]) {
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { a[i][j] = (double)m[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    double best = fabs(a[c][c]);
    for (int i = c + 1; i < 4; ++i) if (fabs(a[i][c]) > best) { best = fabs(a[i][c]); piv = i; }
    if (piv != c) for (int j = 0; j < 8; ++j) { const double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
    const double inv = 1.0 / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= inv;
    for (int i = 0; i < 4; ++i) if (i != c) {
      const double f = a[i][c];
      for (int j = 0; j < 8; ++j) a[i][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) o[i][j] = a[i][4 + j];
}

__global__ __launch_bounds__(64) void dlt_kernel(const float* __restrict__ uv, const float* __restrict__ intr,
                                                 const float* __restrict__ mat, const int* __restrict__ offs,
                                                 float* __restrict__ out, int B, int J, int invert) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * J) return;
  const int b = t / J, j = t % J;
  const int v0 = offs[b], v1 = offs[b + 1];
  double G[4][4] = {};
  for (int v = v0; v < v1; ++v) {
    float T[3][4];
    if (invert) {
      double Ti[4][4];
      dlt_invert4x4(mat + (size_t)v * 16, Ti);
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) T[r][c] = (float)Ti[r][c];
    } else {
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) T[r][c] = mat[(size_t)v * 16 + r * 4 + c];
    }
    const float* K = intr + (size_t)v * 9;
    float M[3][4];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) M[r][c] = fmaf(K[r * 3 + 2], T[2][c], fmaf(K[r * 3 + 1], T[1][c], K[r * 3] * T[0][c]));
    const float u = uv[((size_t)v * J + j) * 2], w = uv[((size_t)v * J + j) * 2 + 1];
    float a0[4], a1[4];
    for (int c = 0; c < 4; ++c) { a0[c] = u * M[2][c] - M[0][c]; a1[c] = w * M[2][c] - M[1][c]; }
    for (int r = 0; r < 4; ++r)
      for (int c = r; c < 4; ++c) G[r][c] += (double)a0[r] * (double)a0[c] + (double)a1[r] * (double)a1[c];
  }
  for (int r = 1; r < 4; ++r) for (int c = 0; c < r; ++c) G[r][c] = G[c][r];
  // cyclic Jacobi: G <- R^T G R, V <- V R
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int r = 0; r < 4; ++r) { diag += G[r][r] * G[r][r]; for (int c = r + 1; c < 4; ++c) off += G[r][c] * G[r][c]; }
    if (off <= 1e-40 * diag) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (G[p][q] == 0.0) continue;
        const double theta = (G[q][q] - G[p][p]) / (2.0 * G[p][q]);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
        for (int k = 0; k < 4; ++k) { const double gkp = G[k][p], gkq = G[k][q]; G[k][p] = c * gkp - s * gkq; G[k][q] = s * gkp + c * gkq; }
        for (int k = 0; k < 4; ++k) { const double gpk = G[p][k], gqk = G[q][k]; G[p][k] = c * gpk - s * gqk; G[q][k] = s * gpk + c * gqk; }
        for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
      }
  }
  int m = 0;
  for (int k = 1; k < 4; ++k) if (G[k][k] < G[m][m]) m = k;
  double x[4] = {V[0][m], V[1][m], V[2][m], V[3][m]};
  if (x[3] < 0) { x[0] = -x[0]; x[1] = -x[1]; x[2] = -x[2]; x[3] = -x[3]; }      // the sign of a singular vector is free
  const double den = x[3] + 1e-7;                                                // triangulation.py:43
  out[(size_t)t * 3 + 0] = (float)(x[0] / den);
  out[(size_t)t * 3 + 1] = (float)(x[1] / den);
  out[(size_t)t * 3 + 2] = (float)(x[2] / den);
}

extern "C" hipError_t poem_launch_dlt(const float* uv, const float* intr, const float* mat, const int* offs, float* out,
                                      int B, int J, int invert, hipStream_t s) {
  const int total = B * J;
  hipLaunchKernelGGL(dlt_kernel, dim3((total + 63) / 64), dim3(64), 0, s, uv, intr, mat, offs, out, B, J, invert);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Heat-map read-out in front of the triangulation (tail of heatmap_stage, lib/models/POEM.py:213-222 upstream, with
// integral_heatmap2d, lib/models/integal_pose.py:194-218): per (view, joint)
//   pdf = hmap / (sum(hmap) + 1e-6);  u = sum_x (x / W_h) * sum_y pdf[y][x];  v = sum_y (y / H_h) * sum_x pdf[y][x]
//   uv_im = (u * W_img, v * H_img)
// One wave per (view, joint); fp32 like the reference, fp64 only for the three running sums.
__global__ __launch_bounds__(256) void heatmap_uv_kernel(const float* __restrict__ hmap, float* __restrict__ uv,
                                                         int maps, int hh, int hw, float img_w, float img_h) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= maps) return;
  const float* h = hmap + (size_t)m * hh * hw;
  double s = 0, su = 0, sv = 0;
  for (int i = lane; i < hh * hw; i += 64) {
    const float v = h[i];
    const int y = i / hw, x = i - y * hw;
    s += v;
    su += (double)v * ((float)x / (float)hw);
    sv += (double)v * ((float)y / (float)hh);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); su += __shfl_xor(su, o, 64); sv += __shfl_xor(sv, o, 64); }
  if (lane == 0) {
    const double den = (double)((float)s + 1e-6f);
    uv[(size_t)m * 2 + 0] = (float)(su / den) * img_w;
    uv[(size_t)m * 2 + 1] = (float)(sv / den) * img_h;
  }
}

extern "C" hipError_t poem_launch_heatmap_uv(const float* hmap, float* uv, int maps, int hh, int hw, float img_w,
                                             float img_h, hipStream_t s) {
  hipLaunchKernelGGL(heatmap_uv_kernel, dim3((maps + 3) / 4), dim3(256), 0, s, hmap, uv, maps, hh, hw, img_w, img_h);
  return hipGetLastError();
}

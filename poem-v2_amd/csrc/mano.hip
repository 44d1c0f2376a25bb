// MANO linear blend skinning on the device: the `ManoLayer(pose_aa, betas)` call of the medium_MANO parametric tail
// (lib/models/bricks/pt_metro_transformer.py:120-124,147-148 upstream: manotorch ManoLayer(joint_rot_mode="axisang",
// use_pca=False, flat_hand_mean=True, center_idx=9)) and the zero-pose template of the head (ptEmb_head.py:732-736,886-892).
//
// manotorch (docs/installation.md:41-45 pins @v0.0.2) is a third-party dependency that is absent from the reference tree
// and from this image, and the MANO assets are licence-gated: PARITY UNPINNED.  What is restated is the published MANO
// model (Romero et al. 2017) in the evaluation order of the manopth / manotorch layer:
//   R_j          = Rodrigues(pose_j) through the unit quaternion of (|pose_j + 1e-8|, pose_j / |.|), j = 0..15
//   v_shaped     = v_template + shapedirs . betas                        (778,3)
//   J            = J_regressor . v_shaped                                (16,3)
//   v_posed      = v_shaped + posedirs . vec(R_1 - I, ..., R_15 - I)     (135 pose-corrective coefficients)
//   G_0 = [R_0 | J_0];  G_j = G_parent(j) . [R_j | J_j - J_parent(j)]    (parents: 0 for joints 1,4,7,10,13, else j-1)
//   A_j = [G_j^R | G_j^t - G_j^R J_j]                                    (rest pose removed)
//   verts_v      = (sum_j w_vj A_j) . [v_posed_v; 1]
//   joints       = [G_j^t (16) | verts[745, 317, 444, 556, 673]] re-ordered to the 21-joint hand order, then both joints
//                  and verts minus joint `center_idx`.
// All assets are caller-owned device buffers (nothing is read from disk here).  One block per sample: the work is ~0.3 MFLOP.
#include "common.h"

namespace {
constexpr int NV = 778, NJ = 16, NPOSE = 135, NBETA = 10;
__constant__ int kTips[5] = {745, 317, 444, 556, 673};
__constant__ int kOrder[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};

__device__ __forceinline__ void rodrigues(const float* aa, float* R) {
  const float x = aa[0] + 1e-8f, y = aa[1] + 1e-8f, z = aa[2] + 1e-8f;
  const float angle = sqrtf(x * x + y * y + z * z);
  const float nx = aa[0] / angle, ny = aa[1] / angle, nz = aa[2] / angle;
  const float hs = sinf(0.5f * angle), w = cosf(0.5f * angle);
  float qx = hs * nx, qy = hs * ny, qz = hs * nz;
  const float qn = sqrtf(w * w + qx * qx + qy * qy + qz * qz);
  const float qw = w / qn;
  qx /= qn; qy /= qn; qz /= qn;
  const float w2 = qw * qw, x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
  const float wx = qw * qx, wy = qw * qy, wz = qw * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz;
  R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;    R[2] = 2 * wy + 2 * xz;
  R[3] = 2 * wz + 2 * xy;   R[4] = w2 - x2 + y2 - z2;  R[5] = 2 * yz - 2 * wx;
  R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;    R[8] = w2 - x2 - y2 + z2;
}

// G = P . [R | t]   (P, G: 3x4 as 12 floats, row-major)
__device__ __forceinline__ void chain(const float* P, const float* R, const float* t, float* G) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) G[r * 4 + c] = P[r * 4 + 0] * R[c] + P[r * 4 + 1] * R[3 + c] + P[r * 4 + 2] * R[6 + c];
    G[r * 4 + 3] = P[r * 4 + 0] * t[0] + P[r * 4 + 1] * t[1] + P[r * 4 + 2] * t[2] + P[r * 4 + 3];
  }
}
}  // namespace

__global__ __launch_bounds__(256) void mano_lbs_kernel(const float* __restrict__ pose, const float* __restrict__ betas,
                                                       const float* __restrict__ v_template,
                                                       const float* __restrict__ shapedirs,
                                                       const float* __restrict__ posedirs,
                                                       const float* __restrict__ j_regressor,
                                                       const float* __restrict__ weights, float* __restrict__ verts,
                                                       float* __restrict__ joints, int center_idx) {
  __shared__ float vs[NV * 3], vo[NV * 3], R[NJ * 9], pm[NPOSE], bet[NBETA], J[NJ * 3], G[NJ * 12], A[NJ * 12], jt[21 * 3];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < NJ) rodrigues(pose + (size_t)b * 48 + t * 3, R + t * 9);
  if (t >= 32 && t < 32 + NBETA) bet[t - 32] = betas[(size_t)b * NBETA + t - 32];
  __syncthreads();
  if (t < NPOSE) pm[t] = R[9 + t] - ((t % 9) % 4 == 0 ? 1.f : 0.f);
  for (int i = t; i < NV * 3; i += 256) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NBETA; ++k) acc = fmaf(shapedirs[(size_t)i * NBETA + k], bet[k], acc);
    vs[i] = v_template[i] + acc;
  }
  __syncthreads();
  if (t < NJ * 3) {
    const int j = t / 3, c = t % 3;
    float acc = 0.f;
    for (int v = 0; v < NV; ++v) acc = fmaf(j_regressor[j * NV + v], vs[v * 3 + c], acc);
    J[t] = acc;
  }
  __syncthreads();
  if (t < 5) {                                   // one thread per finger walks its three joints down from the root
    float root[12];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) root[r * 4 + c] = R[r * 3 + c];
      root[r * 4 + 3] = J[r];
    }
    if (t == 0) for (int e = 0; e < 12; ++e) G[e] = root[e];
    float par[12];
    for (int e = 0; e < 12; ++e) par[e] = root[e];
    int pj = 0;
    for (int l = 0; l < 3; ++l) {
      const int j = 1 + 3 * t + l;
      const float rel[3] = {J[j * 3] - J[pj * 3], J[j * 3 + 1] - J[pj * 3 + 1], J[j * 3 + 2] - J[pj * 3 + 2]};
      float g[12];
      chain(par, R + j * 9, rel, g);
      for (int e = 0; e < 12; ++e) { G[j * 12 + e] = g[e]; par[e] = g[e]; }
      pj = j;
    }
  }
  __syncthreads();
  if (t < NJ) {
    const float* g = G + t * 12;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) A[t * 12 + r * 4 + c] = g[r * 4 + c];
      A[t * 12 + r * 4 + 3] = g[r * 4 + 3] - (g[r * 4] * J[t * 3] + g[r * 4 + 1] * J[t * 3 + 1] + g[r * 4 + 2] * J[t * 3 + 2]);
    }
  }
  __syncthreads();
  for (int v = t; v < NV; v += 256) {
    float vp[3];
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;
      const float* pd = posedirs + (size_t)(v * 3 + c) * NPOSE;
      for (int k = 0; k < NPOSE; ++k) acc = fmaf(pd[k], pm[k], acc);
      vp[c] = vs[v * 3 + c] + acc;
    }
    float T[12];
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    for (int j = 0; j < NJ; ++j) {
      const float w = weights[v * NJ + j];
      for (int e = 0; e < 12; ++e) T[e] = fmaf(w, A[j * 12 + e], T[e]);
    }
    for (int r = 0; r < 3; ++r) vo[v * 3 + r] = T[r * 4] * vp[0] + T[r * 4 + 1] * vp[1] + T[r * 4 + 2] * vp[2] + T[r * 4 + 3];
  }
  __syncthreads();
  if (t < 21 * 3) {
    const int src = kOrder[t / 3], c = t % 3;
    jt[t] = src < NJ ? G[src * 12 + c * 4 + 3] : vo[kTips[src - NJ] * 3 + c];
  }
  __syncthreads();
  float ctr[3] = {0.f, 0.f, 0.f};
  if (center_idx >= 0) for (int c = 0; c < 3; ++c) ctr[c] = jt[center_idx * 3 + c];
  if (t < 21 * 3) joints[(size_t)b * 63 + t] = jt[t] - ctr[t % 3];
  for (int i = t; i < NV * 3; i += 256) verts[(size_t)b * NV * 3 + i] = vo[i] - ctr[i % 3];
}

extern "C" hipError_t poem_launch_mano_lbs(const float* pose, const float* betas, const float* v_template,
                                           const float* shapedirs, const float* posedirs, const float* j_regressor,
                                           const float* weights, float* verts, float* joints, int B, int center_idx,
                                           hipStream_t s) {
  hipLaunchKernelGGL(mano_lbs_kernel, dim3(B), dim3(256), 0, s, pose, betas, v_template, shapedirs, posedirs, j_regressor,
                     weights, verts, joints, center_idx);
  return hipGetLastError();
}

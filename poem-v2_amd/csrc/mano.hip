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
//
// Round 6 -- laid out for the chip instead of one 256-thread block per sample (268 us at batch 32: 12 % of the CUs, one
// thread per finger, posedirs read with a 135-float stride between neighbouring threads, the joint regression as 48 threads
// x 778 serial fmas).  The asset arrays are re-laid ONCE (`poem_mano_prepare`, at ManoLayer creation) into a table:
//   * the two blend-shape bases as ONE coefficient-major array D[k / 4][c][v][k % 4], k = 0..9 shape | 10..144 pose (zero padded
//     to 148): v_posed[c][v] = v_template[c][v] + sum_k coef[k] D[k][c][v] with coef = (betas | R - I); neighbouring lanes =
//     neighbouring vertices read neighbouring float4 -- 1 KiB per wave load;
//   * the joint regression composed with the template and the shape basis (fp64 products rounded once, like the decoder's
//     composed Linears): J = J_t + J_sd . betas -- 48 x 10 fmas instead of 48 x 778;
//   * the skinning weights joint-major W[j][v].
// `mano_lbs_kernel`: grid (13 vertex tiles of 64, B), 256 threads.  Every block repeats the sample's small prologue (16
// Rodrigues, J, the kinematic chain, A: ~1 us of latency, cheaper than a launch to share it); wave w sums its quarter of the
// 148 blend coefficients for the tile's 64 vertices, the four partial sums meet in LDS, and thread (vertex, row r < 3)
// blends its row of the 16 joint transforms and applies it.  Fixed summation order per output: a sample's result does not
// depend on the batch (tested bit-exact).
#include "common.h"

namespace {
constexpr int NV = 778, NJ = 16, NPOSE = 135, NBETA = 10;
constexpr int VP = 832;                    // vertices padded to 13 tiles of 64
constexpr int K4 = 37;                     // 148 / 4 coefficient groups (10 shape + 135 pose + 3 zero)
constexpr int OFF_VT = 0;                  // [3][VP]
constexpr int OFF_D = OFF_VT + 3 * VP;     // [K4][3][VP][4]
constexpr int OFF_W = OFF_D + K4 * 3 * VP * 4;   // [NJ][VP]
constexpr int OFF_JT = OFF_W + NJ * VP;    // [48]
constexpr int OFF_JSD = OFF_JT + 48;       // [48][10]
constexpr int TABLE_FLOATS = OFF_JSD + 48 * NBETA;
__constant__ int kTips[5] = {745, 317, 444, 556, 673};
__constant__ int kOrder[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};
const int kOrderHost[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};

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

// ---- the table (see the header): one thread per table float; the two composed joint regressions with fp64 accumulation
__global__ void mano_prepare_kernel(const float* __restrict__ v_template, const float* __restrict__ shapedirs,
                                    const float* __restrict__ posedirs, const float* __restrict__ j_regressor,
                                    const float* __restrict__ weights, float* __restrict__ table) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= TABLE_FLOATS) return;
  float out = 0.f;
  if (i < OFF_D) {
    const int c = i / VP, v = i % VP;
    if (v < NV) out = v_template[v * 3 + c];
  } else if (i < OFF_W) {
    const int e = i - OFF_D, kk = e & 3, v = (e >> 2) % VP, c = ((e >> 2) / VP) % 3, k = 4 * ((e >> 2) / (3 * VP)) + kk;
    if (v < NV && k < NBETA) out = shapedirs[(size_t)(v * 3 + c) * NBETA + k];
    else if (v < NV && k < NBETA + NPOSE) out = posedirs[(size_t)(v * 3 + c) * NPOSE + (k - NBETA)];
  } else if (i < OFF_JT) {
    const int e = i - OFF_W, j = e / VP, v = e % VP;
    if (v < NV) out = weights[v * NJ + j];
  } else if (i < OFF_JSD) {
    const int t = i - OFF_JT, j = t / 3, c = t % 3;
    double acc = 0.0;
    for (int v = 0; v < NV; ++v) acc += (double)j_regressor[j * NV + v] * (double)v_template[v * 3 + c];
    out = (float)acc;
  } else {
    const int e = i - OFF_JSD, t = e / NBETA, k = e % NBETA, j = t / 3, c = t % 3;
    double acc = 0.0;
    for (int v = 0; v < NV; ++v) acc += (double)j_regressor[j * NV + v] * (double)shapedirs[(size_t)(v * 3 + c) * NBETA + k];
    out = (float)acc;
  }
  table[i] = out;
}

__global__ __launch_bounds__(256) void mano_lbs_kernel(const float* __restrict__ pose, const float* __restrict__ betas,
                                                       const float* __restrict__ table, float* __restrict__ verts,
                                                       float* __restrict__ joints, int center_idx) {
  __shared__ float R[NJ * 9], coef[K4 * 4], J[NJ * 3], G[NJ * 12];
  __shared__ __attribute__((aligned(16))) float A[NJ * 12];
  __shared__ float part[4][3][64];
  const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int v = blockIdx.x * 64 + lane;              // < VP: padded columns of the table are zeros
  const int kbeg = wv == 0 ? 0 : 10 + 9 * (wv - 1), kend = wv == 0 ? 10 : kbeg + 9;      // of the 37 coefficient groups: wave 0 takes 10, waves 1..3 nine each
  const float4* D = reinterpret_cast<const float4*>(table + OFF_D);
  // ---- everything that does not depend on the sample is requested first: the wave's blend-shape fragments (<= 30 x 1 KiB), the
  // vertex's template coordinates and skinning weights -- their L2 / HBM round trip overlaps the prologue's latency chain
  float4 d[10][3];
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) d[i][c] = D[((size_t)min(kbeg + i, K4 - 1) * 3 + c) * VP + v];
  float wj[NJ], vt[3];
#pragma unroll
  for (int j = 0; j < NJ; ++j) wj[j] = table[OFF_W + j * VP + v];
#pragma unroll
  for (int c = 0; c < 3; ++c) vt[c] = table[OFF_VT + c * VP + v];
  // ---- prologue: the sample's 16 rotations, coefficient vector, joints, kinematic chain
  if (t < NJ) rodrigues(pose + (size_t)b * 48 + t * 3, R + t * 9);
  if (t >= 64 && t < 64 + NBETA) coef[t - 64] = betas[(size_t)b * NBETA + t - 64];
  if (t >= 128 && t < 128 + 3) coef[NBETA + NPOSE + t - 128] = 0.f;
  __syncthreads();
  if (t < NPOSE) coef[NBETA + t] = R[9 + t] - ((t % 9) % 4 == 0 ? 1.f : 0.f);
  if (t >= 192 && t < 192 + NJ * 3) {
    const int q = t - 192;
    float acc = table[OFF_JT + q];
#pragma unroll
    for (int k = 0; k < NBETA; ++k) acc = fmaf(table[OFF_JSD + q * NBETA + k], coef[k], acc);
    J[q] = acc;
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
  // ---- blend shapes: wave w's share of the 37 coefficient groups for the tile's 64 vertices (coalesced 1 KiB loads)
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    if (kbeg + i < kend) {                       // (wave-uniform)
      const int k4 = kbeg + i;
      const float c0 = coef[4 * k4], c1 = coef[4 * k4 + 1], c2 = coef[4 * k4 + 2], c3 = coef[4 * k4 + 3];
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = fmaf(d[i][c].w, c3, fmaf(d[i][c].z, c2, fmaf(d[i][c].y, c1, fmaf(d[i][c].x, c0, acc[c]))));
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) part[wv][c][lane] = acc[c];
  __syncthreads();                               // G complete, partial sums in place
  if (t < NJ) {
    const float* g = G + t * 12;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) A[t * 12 + r * 4 + c] = g[r * 4 + c];
      A[t * 12 + r * 4 + 3] = g[r * 4 + 3] - (g[r * 4] * J[t * 3] + g[r * 4 + 1] * J[t * 3 + 1] + g[r * 4 + 2] * J[t * 3 + 2]);
    }
  }
  __syncthreads();
  // centre: joint `center_idx` of the 21-joint order (a kinematic joint here: finger-tip centres are re-centred by a second
  // launch -- poem_launch_mano_lbs)
  float ctr = 0.f;
  const int r = wv;                                // thread (vertex, output row r): waves 0..2
  if (center_idx >= 0 && r < 3) ctr = G[kOrder[center_idx] * 12 + r * 4 + 3];
  if (r < 3 && v < NV) {
    float vp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) vp[c] = vt[c] + (((part[0][c][lane] + part[1][c][lane]) + part[2][c][lane]) + part[3][c][lane]);
    float4 T = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float w = wj[j];
      const float4 a = *reinterpret_cast<const float4*>(A + j * 12 + r * 4);
      T.x = fmaf(w, a.x, T.x); T.y = fmaf(w, a.y, T.y); T.z = fmaf(w, a.z, T.z); T.w = fmaf(w, a.w, T.w);
    }
    const float vo = T.x * vp[0] + T.y * vp[1] + T.z * vp[2] + T.w;
    verts[((size_t)b * NV + v) * 3 + r] = vo - ctr;
#pragma unroll
    for (int f = 0; f < 5; ++f)
      if (v == kTips[f]) {
        int pos = 0;
        for (int q = 0; q < 21; ++q) if (kOrder[q] == NJ + f) pos = q;
        joints[(size_t)b * 63 + pos * 3 + r] = vo - ctr;
      }
  }
  if (blockIdx.x == 0 && t < 21 * 3) {             // the 16 kinematic joints, by the sample's first tile
    const int src = kOrder[t / 3], c = t % 3;
    const float cc = center_idx >= 0 ? G[kOrder[center_idx] * 12 + c * 4 + 3] : 0.f;
    if (src < NJ) joints[(size_t)b * 63 + t] = G[src * 12 + c * 4 + 3] - cc;
  }
}

// centre = a finger-tip joint (a skinned vertex): subtract it after the uncentred launch -- the same fp32 subtraction per element
__global__ void mano_recentre_kernel(float* __restrict__ verts, float* __restrict__ joints, int centre_pos) {
  const int b = blockIdx.x, t = threadIdx.x;
  __shared__ float ctr[3];
  if (t < 3) ctr[t] = joints[(size_t)b * 63 + centre_pos * 3 + t];
  __syncthreads();
  for (int i = t; i < NV * 3; i += blockDim.x) verts[(size_t)b * NV * 3 + i] -= ctr[i % 3];
  __syncthreads();
  if (t < 63) joints[(size_t)b * 63 + t] -= ctr[t % 3];
}

extern "C" size_t poem_mano_table_floats_impl() { return (size_t)TABLE_FLOATS; }

extern "C" hipError_t poem_launch_mano_prepare(const float* v_template, const float* shapedirs, const float* posedirs,
                                               const float* j_regressor, const float* weights, float* table, hipStream_t s) {
  hipLaunchKernelGGL(mano_prepare_kernel, dim3((TABLE_FLOATS + 255) / 256), dim3(256), 0, s, v_template, shapedirs, posedirs,
                     j_regressor, weights, table);
  return hipGetLastError();
}

extern "C" hipError_t poem_launch_mano_lbs(const float* pose, const float* betas, const float* table, float* verts, float* joints,
                                           int B, int center_idx, hipStream_t s) {
  const bool tip_centre = center_idx >= 0 && kOrderHost[center_idx] >= NJ;
  hipLaunchKernelGGL(mano_lbs_kernel, dim3(VP / 64, B), dim3(256), 0, s, pose, betas, table, verts, joints, tip_centre ? -1 : center_idx);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  if (tip_centre) {
    hipLaunchKernelGGL(mano_recentre_kernel, dim3(B), dim3(256), 0, s, verts, joints, center_idx);
    return hipGetLastError();
  }
  return hipSuccess;
}

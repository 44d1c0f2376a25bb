// Small glue kernels of the path: coordinate normalisation, query-embedding broadcast, output de-normalisation,
// and the medium_MANO parametric tail (Q3 re-interpretation + rot6d -> axis-angle).
#include "common.h"
#include <algorithm>
#include "merge.h"

// centre = reference_joints[:, 9];  pt_xyz = ((bps + c) - c) / radius;  query_xyz = ((c + template) - c) / radius
// evaluated exactly the reference's way (ptEmb_head.py:873-874,893-894,934-935 upstream): (bps + c) - c is NOT bps in fp32.
__global__ void prep_xyz_kernel(const float* __restrict__ ref_joints, const float* __restrict__ bps,
                                const float* __restrict__ tmpl, float* __restrict__ centre, float* __restrict__ pt_xyz,
                                float* __restrict__ query_xyz, int B, int S, int Q, float radius) {
  poem_prep_xyz_elem((long)blockIdx.x * blockDim.x + threadIdx.x, ref_joints, bps, tmpl, centre, pt_xyz, query_xyz, B, S, Q, radius);
}

extern "C" hipError_t poem_launch_prep_xyz(const float* ref_joints, const float* bps, const float* tmpl, float* centre,
                                           float* pt_xyz, float* query_xyz, int B, int S, int Q, float radius,
                                           hipStream_t s) {
  const long total = (long)B * S * 3 + (long)B * Q * 3 + 3L * B;
  hipLaunchKernelGGL(prep_xyz_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ref_joints, bps, tmpl,
                     centre, pt_xyz, query_xyz, B, S, Q, radius);
  return hipGetLastError();
}

// t / radius: what ((c + t) - c) / radius is for every sample up to the rounding of c + t (block-0 anchor tables, api.cpp)
__global__ void canon_xyz_kernel(const float* __restrict__ tmpl, float* __restrict__ out, int n, float radius) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = __fdiv_rn(tmpl[i], radius);
}

extern "C" hipError_t poem_launch_canon_xyz(const float* tmpl, float* out, int n, float radius, hipStream_t s) {
  hipLaunchKernelGGL(canon_xyz_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tmpl, out, n, radius);
  return hipGetLastError();
}

// Block 0 touches only the 32 anchor rows of each sample's key / value sources (Q2): dst (B*32, C) <- src rows
// b*NS + idx[j]; `ident` (32 ints, optional) <- 0..31, the neighbour ids into the compacted rows.
__global__ void gather_anchor_rows_kernel(const float* __restrict__ src, int ld, const int* __restrict__ idx, int NS,
                                          float* __restrict__ dst, int B, int C, int* __restrict__ ident) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C / 4;
  if (i < (long)B * 32 * c4) {
    const int c = (int)(i % c4), r = (int)(i / c4), b = r >> 5, j = r & 31;
    reinterpret_cast<float4*>(dst)[i] = *reinterpret_cast<const float4*>(src + ((size_t)b * NS + idx[j]) * ld + 4 * c);
  }
  if (ident && i < 32) ident[i] = (int)i;
}

extern "C" hipError_t poem_launch_gather_anchor_rows(const float* src, int ld, const int* idx, int NS, float* dst, int B,
                                                     int C, int* ident, hipStream_t s) {
  const long total = (long)B * 32 * (C / 4);
  hipLaunchKernelGGL(gather_anchor_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, ld, idx, NS,
                     dst, B, C, ident);
  return hipGetLastError();
}

__global__ void broadcast_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long per, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) dst[i] = src[i % per];
}

extern "C" hipError_t poem_launch_broadcast(const float* src, float* dst, long per, int copies, hipStream_t s) {
  const long total = per * copies;
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, per, total);
  return hipGetLastError();
}

// out[l, b, q, :] = nan_to_num(xyz[l, b, q, :]) * radius + centre[b]      (ptEmb_head.py:944-951 upstream)
// mano_verts / mano_joints (a parametric head whose MANO layer runs inside the forward, poem_attach_mano): the LAST layer is the
// MANO layer's output, not scaled: out[L-1, b, :21] = nan_to_num(joints) + c, out[L-1, b, 21:] = nan_to_num(verts) + c
// (pt_metro_transformer.py:149-150, ptEmb_head.py:944,953-958 upstream) -- the arithmetic of finalize_param_kernel below.
__global__ void finalize_kernel(const float* __restrict__ xyz, const float* __restrict__ centre, float* __restrict__ out,
                                int L, int B, int Q, float radius, const float* __restrict__ mano_verts,
                                const float* __restrict__ mano_joints) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)L * B * Q * 3) return;
  const int d = (int)(i % 3);
  const int b = (int)((i / (3L * Q)) % B);
  if (mano_verts && i >= (long)(L - 1) * B * Q * 3) {
    const int qq = (int)((i / 3) % Q);
    float v = qq < 21 ? mano_joints[((size_t)b * 21 + qq) * 3 + d] : mano_verts[((size_t)b * (Q - 21) + (qq - 21)) * 3 + d];
    if (isnan(v)) v = 0.f;
    else if (isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    out[i] = v + centre[b * 3 + d];
    return;
  }
  float v = xyz[i];
  if (isnan(v)) v = 0.f;
  else if (isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  out[i] = v * radius + centre[b * 3 + d];
}

extern "C" hipError_t poem_launch_finalize(const float* xyz, const float* centre, float* out, int L, int B, int Q,
                                           float radius, const float* mano_verts, const float* mano_joints, hipStream_t s) {
  const long total = (long)L * B * Q * 3;
  hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, xyz, centre, out, L, B, Q,
                     radius, mano_verts, mano_joints);
  return hipGetLastError();
}

// PtEmbedTRv4.forward with the MANO layer inside the forward: the last layer's rows REPLACED by the layer's output, as is
// (get_parametric_output, pt_metro_transformer.py:149-150: verts[:, 21:] = mano_verts; verts[:, :21] = mano_joints)
__global__ void param_rows_kernel(const float* __restrict__ verts, const float* __restrict__ joints, float* __restrict__ out_last,
                                  int B, int Q) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Q * 3) return;
  const int d = (int)(i % 3), qq = (int)((i / 3) % Q), b = (int)(i / (3L * Q));
  out_last[i] = qq < 21 ? joints[((size_t)b * 21 + qq) * 3 + d] : verts[((size_t)b * (Q - 21) + (qq - 21)) * 3 + d];
}

extern "C" hipError_t poem_launch_param_rows(const float* verts, const float* joints, float* out_last, int B, int Q, hipStream_t s) {
  const long total = (long)B * Q * 3;
  hipLaunchKernelGGL(param_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, verts, joints, out_last, B, Q);
  return hipGetLastError();
}

// last layer of a parametric head: out[b, :21] = joints + c, out[b, 21:] = verts + c   (ptEmb_head.py:953-958)
__global__ void finalize_param_kernel(const float* __restrict__ verts, const float* __restrict__ joints,
                                      const float* __restrict__ ref_joints, float* __restrict__ out_last, int B, int Q) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Q * 3) return;
  const int d = (int)(i % 3), qq = (int)((i / 3) % Q), b = (int)(i / (3L * Q));
  const float c = ref_joints[((size_t)b * 21 + 9) * 3 + d];
  float v = qq < 21 ? joints[((size_t)b * 21 + qq) * 3 + d] : verts[((size_t)b * (Q - 21) + (qq - 21)) * 3 + d];
  if (isnan(v)) v = 0.f;
  else if (isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  out_last[i] = v + c;
}

extern "C" hipError_t poem_launch_finalize_param(const float* verts, const float* joints, const float* ref_joints,
                                                 float* out_last, int B, int Q, hipStream_t s) {
  const long total = (long)B * Q * 3;
  hipLaunchKernelGGL(finalize_param_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, verts, joints,
                     ref_joints, out_last, B, Q);
  return hipGetLastError();
}

// Q3 (pt_metro_transformer.py:139-143 upstream): feats (B,Q,C) re-interpreted as rows of Q floats:
//   t[b, r] = <flat_w, flat(feats[b])[r*Q : (r+1)*Q]> + flat_b,  r in [0, C)
// one wave per (b, r).
__global__ __launch_bounds__(256) void q3_flatten_kernel(const float* __restrict__ feats, const float* __restrict__ fw,
                                                         const float* __restrict__ fb, float* __restrict__ t, int B,
                                                         int Q, int C) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long)B * C) return;
  const float* row = feats + wid * Q;    // (b*C + r)*Q == b*Q*C + r*Q
  float s = 0.f;
  for (int m = lane; m < Q; m += 64) s = fmaf(row[m], fw[m], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) t[wid] = s + fb[0];
}

extern "C" hipError_t poem_launch_q3_flatten(const float* feats, const float* fw, const float* fb, float* t, int B, int Q,
                                             int C, hipStream_t s) {
  const long waves = (long)B * C;
  hipLaunchKernelGGL(q3_flatten_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, feats, fw, fb, t, B, Q, C);
  return hipGetLastError();
}

// params (B,106): [:96] sixteen 6-D rotations -> axis-angle (pytorch3d rotation_6d_to_matrix -> matrix_to_quaternion
// -> quaternion_to_axis_angle, published algorithms; upstream call: lib/utils/transform.py:448-466), [96:] betas.
__global__ void rot6d_to_aa_kernel(const float* __restrict__ par, float* __restrict__ pose_aa, float* __restrict__ betas,
                                   int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * 10) betas[i] = par[(i / 10) * 106 + 96 + i % 10];
  if (i >= B * 16) return;
  const float* d6 = par + (size_t)(i / 16) * 106 + (i % 16) * 6;
  const float a1x = d6[0], a1y = d6[1], a1z = d6[2], a2x = d6[3], a2y = d6[4], a2z = d6[5];
  const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
  const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
  const float dt = b1x * a2x + b1y * a2y + b1z * a2z;
  float b2x = a2x - dt * b1x, b2y = a2y - dt * b1y, b2z = a2z - dt * b1z;
  const float n2 = fmaxf(sqrtf(b2x * b2x + b2y * b2y + b2z * b2z), 1e-12f);
  b2x /= n2; b2y /= n2; b2z /= n2;
  const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
  // rows of R are b1, b2, b3
  const float m00 = b1x, m01 = b1y, m02 = b1z, m10 = b2x, m11 = b2y, m12 = b2z, m20 = b3x, m21 = b3y, m22 = b3z;
  float qa[4] = {sqrtf(fmaxf(1.f + m00 + m11 + m22, 0.f)), sqrtf(fmaxf(1.f + m00 - m11 - m22, 0.f)),
                 sqrtf(fmaxf(1.f - m00 + m11 - m22, 0.f)), sqrtf(fmaxf(1.f - m00 - m11 + m22, 0.f))};
  int best = 0;
  for (int k = 1; k < 4; ++k) if (qa[k] > qa[best]) best = k;
  float cw, cx, cy, cz;
  if (best == 0) { cw = qa[0] * qa[0]; cx = m21 - m12; cy = m02 - m20; cz = m10 - m01; }
  else if (best == 1) { cw = m21 - m12; cx = qa[1] * qa[1]; cy = m10 + m01; cz = m02 + m20; }
  else if (best == 2) { cw = m02 - m20; cx = m10 + m01; cy = qa[2] * qa[2]; cz = m12 + m21; }
  else { cw = m10 - m01; cx = m20 + m02; cy = m21 + m12; cz = qa[3] * qa[3]; }
  const float den = 2.0f * fmaxf(qa[best], 0.1f);
  cw /= den; cx /= den; cy /= den; cz /= den;
  const float nrm = sqrtf(cx * cx + cy * cy + cz * cz);
  const float half = atan2f(nrm, cw);
  const float ang = 2.0f * half;
  const float sc = fabsf(ang) < 1e-6f ? 0.5f - ang * ang / 48.0f : sinf(half) / ang;
  pose_aa[(size_t)i * 3 + 0] = cx / sc;
  pose_aa[(size_t)i * 3 + 1] = cy / sc;
  pose_aa[(size_t)i * 3 + 2] = cz / sc;
}

extern "C" hipError_t poem_launch_rot6d_to_aa(const float* par, float* pose_aa, float* betas, int B, hipStream_t s) {
  const int total = B * 16;
  hipLaunchKernelGGL(rot6d_to_aa_kernel, dim3((total + 63) / 64), dim3(64), 0, s, par, pose_aa, betas, B);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Composite Linears, built once at handle creation: two Linears in sequence without a non-linearity between them are one
// Linear, W = A B and b = A b1 + b2 (fp64 accumulation, rounded once to fp32).  The decoder uses them for the
// embedding -> key/value projections of every block (input: the basis-point features, which are the same for all
// blocks) and for embedding -> query, fc1 -> w_qs|w_ks|w_vs on the query side: fewer GEMMs, fewer HBM round trips.
// out[n][k] = sum_c A[n][c] * B[c][k]       A (N x Cm), B (Cm x K)
__global__ void compose_weight_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ out,
                                      int N, int Cm, int K) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * K) return;
  const int k = (int)(i % K), n = (int)(i / K);
  double acc = 0.0;
  for (int c = 0; c < Cm; ++c) acc += (double)A[(size_t)n * Cm + c] * (double)Bm[(size_t)c * K + k];
  out[i] = (float)acc;
}

// out[n] = sum_c A[n][c] * b1[c] + (b2 ? b2[n] : 0)
__global__ void compose_bias_kernel(const float* __restrict__ A, const float* __restrict__ b1, const float* __restrict__ b2,
                                    float* __restrict__ out, int N, int Cm) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double acc = b2 ? (double)b2[n] : 0.0;
  for (int c = 0; c < Cm; ++c) acc += (double)A[(size_t)n * Cm + c] * (double)b1[c];
  out[n] = (float)acc;
}

extern "C" hipError_t poem_launch_compose_weight(const float* A, const float* Bm, float* out, int N, int Cm, int K,
                                                 hipStream_t s) {
  const long total = (long)N * K;
  hipLaunchKernelGGL(compose_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, A, Bm, out, N, Cm, K);
  return hipGetLastError();
}

extern "C" hipError_t poem_launch_compose_bias(const float* A, const float* b1, const float* b2, float* out, int N, int Cm,
                                               hipStream_t s) {
  hipLaunchKernelGGL(compose_bias_kernel, dim3((N + 63) / 64), dim3(64), 0, s, A, b1, b2, out, N, Cm);
  return hipGetLastError();
}

// view_sample / pe_index / view_offsets of a ragged batch from the offsets held in the kernel arguments (merge.h ViewLayoutArgs;
// lib/utils/collation.py:7-25 upstream is where the per-sample view counts come from): thread v finds its sample by bisection.
__global__ void view_layout_kernel(ViewLayoutArgs A) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int BN = A.off16[A.B];
  if (v <= A.B) A.offs[v] = A.off16[v];
  if (v >= BN) return;
  int lo = 0, hi = A.B;                  // off16[lo] <= v < off16[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)A.off16[mid] <= v) lo = mid; else hi = mid;
  }
  const int n = (int)A.off16[lo + 1] - (int)A.off16[lo];
  A.view_sample[v] = lo;
  A.pe_index[v] = n * (n - 1) / 2 + (v - (int)A.off16[lo]);
}

extern "C" hipError_t poem_launch_view_layout(const ViewLayoutArgs* a, hipStream_t s) {
  const int n = std::max<int>(a->off16[a->B], a->B + 1);
  hipLaunchKernelGGL(view_layout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *a);
  return hipGetLastError();
}

extern "C" int poem_device_cu_count(void) { return poem_device_cus(); }

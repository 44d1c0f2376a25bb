// Operator-level entry points of libpoem_hip.so (include/poem_hip.h names the reference call each one replaces): argument
// checks around the launchers of launchers.h.  Stream-ordered, no allocation.
#include <cstring>
#include "engine.h"

extern "C" {

// ---- individual operators -------------------------------------------------------------------------------------
int poem_gemm(const float* x, int ldx, const void* w_packed, const float* bias, const float* residual, int ldr, float* y,
              int ldy, int M, int N, int K, int act, void* stream) {
  if (!x || !w_packed || !y || M <= 0 || N <= 0 || K <= 0 || K % 8 || ldx % 4 || ((uintptr_t)x & 15)) return POEM_E_ARG;
  if (act < 0 || act > 2) return POEM_E_ARG;
  HIPCHK(poem_launch_gemm(x, ldx, w_packed, bias, residual, ldr, y, ldy, M, N, K, act, (hipStream_t)stream));
  return POEM_OK;
}

int poem_gemm_ex(const float* x, int ldx, const void* w_packed, const float* bias, const float* residual, int ldr,
                 float* y, int ldy, int M, int N, int K, int act, int in_layout, int out_layout, void* stream) {
  if (!x || !w_packed || !y || M <= 0 || N <= 0 || K <= 0 || K % 8 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))
    return POEM_E_ARG;
  if (!in_layout && ldx % 4) return POEM_E_ARG;
  if (out_layout && N % 32) return POEM_E_ARG;
  if (act < 0 || act > 2 || (in_layout & ~1) || (out_layout & ~1)) return POEM_E_ARG;
  HIPCHK(poem_launch_gemm2(x, ldx, w_packed, bias, residual, ldr, y, ldy, M, N, K, act, in_layout, out_layout,
                           (hipStream_t)stream));
  return POEM_OK;
}

int poem_pack_rows(const float* x, int rows, int cols, void* packed, void* stream) {
  return poem_pack_linear(x, rows, cols, packed, stream);
}

int poem_unpack_rows(const void* packed, int rows, int cols, float* x, void* stream) {
  if (!packed || !x || rows <= 0 || cols % 8) return POEM_E_ARG;
  HIPCHK(poem_launch_unpack_rows(packed, rows, cols, x, (hipStream_t)stream));
  return POEM_OK;
}

int poem_layernorm(const float* x, const float* gamma, const float* beta, float* y, int rows, int cols, float eps,
                   void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || cols <= 0) return POEM_E_ARG;
  HIPCHK(poem_launch_layernorm(x, gamma, beta, y, rows, cols, eps, (hipStream_t)stream));
  return POEM_OK;
}

int poem_pe_table_ex(const void* adapt_w_packed, const float* adapt_b, int embed, int fh, int fw, int max_views, int normalize,
                     float* scratch_sine, float* table, void* stream) {
  if (!adapt_w_packed || !adapt_b || !scratch_sine || !table || embed % 2 || (fh * fw) % 32) return POEM_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(poem_launch_sine_pe(scratch_sine, embed / 2, fh, fw, max_views, normalize != 0, s));
  HIPCHK(poem_launch_conv1x1(scratch_sine, adapt_w_packed, adapt_b, nullptr, nullptr, table, nullptr, (int)pe_views(max_views),
                             3 * embed / 2, embed, fh * fw, s));
  return POEM_OK;
}

int poem_pe_table(const void* adapt_w_packed, const float* adapt_b, int embed, int fh, int fw, int max_views,
                  float* scratch_sine, float* table, void* stream) {
  return poem_pe_table_ex(adapt_w_packed, adapt_b, embed, fh, fw, max_views, 1, scratch_sine, table, stream);
}

int poem_frustum_features(const poem_config_t* cfg, const float* cam_intr, const float* cam_extr, int views, int img0, int img1,
                          float* out, void* stream) {
  if (!cfg || !cam_intr || !cam_extr || !out || views <= 0 || cfg->depth_num <= 0 || cfg->feat_h <= 0 || cfg->feat_w <= 0) return POEM_E_ARG;
  HIPCHK(poem_launch_frustum_features(cam_intr, cam_extr, out, views, cfg->feat_h, cfg->feat_w, cfg->depth_num, cfg->lid != 0,
                                      cfg->depth_start, cfg->depth_end, cfg->position_range, img0, img1, (hipStream_t)stream));
  return POEM_OK;
}

int poem_input_proj(const float* feat, const void* w_packed, const float* bias, const float* table,
                    const int32_t* pe_index, float* x, int views, int in_channels, int embed, int hw, void* stream) {
  if (!feat || !w_packed || !x || views <= 0 || in_channels % 8 || hw % 32) return POEM_E_ARG;
  if (table && !pe_index) return POEM_E_ARG;
  HIPCHK(poem_launch_conv1x1(feat, w_packed, bias, table, pe_index, x, nullptr, views, in_channels, embed, hw, (hipStream_t)stream));
  return POEM_OK;
}

int poem_project_sample(const float* x, const float* bps, const float* centre, const int32_t* view_sample,
                        const float* cam_intr, const float* cam_extr, float* uv_scratch, float* g, int views, int embed,
                        int fh, int fw, int nsample, int img_w, int img_h, void* stream) {
  if (!x || !bps || !centre || !view_sample || !cam_intr || !cam_extr || !uv_scratch || !g || views <= 0) return POEM_E_ARG;
  // the inverse extrinsics live in the tail of the uv scratch: caller provides (views*S*2 + views*16) floats
  float* inv = uv_scratch + (size_t)views * nsample * 2;
  HIPCHK(poem_launch_project_sample(x, bps, centre, view_sample, cam_intr, cam_extr, inv, uv_scratch, g, views, embed, fh,
                                    fw, nsample, img_w, img_h, (hipStream_t)stream));
  return POEM_OK;
}

int poem_merge_reduce(const float* h2, const int32_t* view_offsets, float* m, int batch, int nsample, int half,
                      void* stream) {
  if (!h2 || !view_offsets || !m || half > 512 || batch <= 0) return POEM_E_ARG;
  HIPCHK(poem_launch_merge_reduce(h2, view_offsets, m, batch, nsample, half, (hipStream_t)stream));
  return POEM_OK;
}

int poem_merge_finalize(const float* g, const float* y, const int32_t* view_offsets, float* out, int batch, int nsample,
                        int embed, void* stream) {
  if (!g || !y || !view_offsets || !out || batch <= 0) return POEM_E_ARG;
  HIPCHK(poem_launch_merge_finalize(g, y, view_offsets, out, batch, nsample, embed, (hipStream_t)stream));
  return POEM_OK;
}

size_t poem_cross_attention_scratch_bytes(int batch, int nq, int nk, int embed, int heads) {
  if (batch <= 0 || nq <= 0 || nk % 32 || heads <= 0 || embed % heads) return 0;
  return poem_cross_attention_scratch_floats(batch, nq, nk, embed, heads, 1) * sizeof(float);
}

int poem_cross_attention(const float* q, const float* k, const float* v, float* ctx, int batch, int nq, int nk, int embed,
                         int heads, void* scratch, size_t scratch_bytes, void* stream) {
  if (!q || !k || !v || !ctx || batch <= 0 || nq <= 0 || nk % 32 || heads <= 0 || embed % heads) return POEM_E_ARG;
  const size_t need = poem_cross_attention_scratch_bytes(batch, nq, nk, embed, heads);
  if (need && (!scratch || scratch_bytes < need || ((uintptr_t)scratch & 15))) return POEM_E_WORKSPACE;
  HIPCHK(poem_launch_cross_attention(q, k, v, ctx, batch, nq, nk, embed, heads, embed, (float*)scratch,
                                     (hipStream_t)stream));
  return POEM_OK;
}

int poem_cross_attention_merged(const float* q, const float* k, const float* v, float* ctx, int batch, int nq, int nk, int embed,
                                int heads, void* scratch, size_t scratch_bytes, void* stream) {
  if (!q || !k || !v || !ctx || batch <= 0 || nq <= 0 || nk % 32 || heads <= 0 || embed % heads) return POEM_E_ARG;
  const size_t need = poem_cross_attention_scratch_bytes(batch, nq, nk, embed, heads);
  if (need && (!scratch || scratch_bytes < need || ((uintptr_t)scratch & 15))) return POEM_E_WORKSPACE;
  const hipError_t e = poem_launch_cross_attention_merged_rm(q, k, v, ctx, batch, nq, nk, embed, heads, (float*)scratch, (hipStream_t)stream);
  if (e == hipErrorNotSupported) return POEM_E_UNSUPPORTED;
  HIPCHK(e);
  return POEM_OK;
}

int poem_pack_split_gemm(const float* w, int out_features, int in_features, void* image, float* scales, void* stream) {
  if (!w || !image || !scales || out_features <= 0 || in_features <= 0 || in_features % 16) return POEM_E_ARG;
  HIPCHK(poem_launch_pack_split_tiles(w, out_features, in_features, image, scales, 1, (hipStream_t)stream));
  return POEM_OK;
}

int poem_gemm_split(const float* x, int ldx, const void* image, const float* scales, const float* bias, const float* residual,
                    int ldr, float* y, int ldy, int m, int n, int k, int act, void* stream) {
  if (!x || !image || !scales || !y || m <= 0 || n <= 0 || k <= 0 || k % 16 || n % 32) return POEM_E_ARG;
  if (act < POEM_ACT_NONE || act > POEM_ACT_GELU) return POEM_E_ARG;
  poem_gemm_split_explicit(image, scales);
  const hipError_t e = poem_launch_gemm(x, ldx, image, bias, residual, ldr, y, ldy, m, n, k, act, (hipStream_t)stream);
  poem_gemm_split_explicit(nullptr, nullptr);
  if (e == hipErrorInvalidValue) return POEM_E_UNSUPPORTED;          // shape outside the panel kernel's range
  HIPCHK(e);
  return POEM_OK;
}

int poem_pack_split_linear(const float* w, int embed, void* image, float* scale, void* stream) {
  if (!w || !image || !scale || embed < 128 || embed > 1024 || (embed & (embed - 1))) return POEM_E_ARG;
  HIPCHK(poem_launch_pack_split(w, embed, image, scale, (hipStream_t)stream));
  return POEM_OK;
}

int poem_vector_attention_split(const float* query_xyz, const float* src_xyz, const float* anchor_xyz, const int32_t* idx,
                                int shared_idx, const float* qg, const float* kg, const float* v, int nsrc, const float* wd1,
                                const float* bd1, const void* wd2_image, const float* bd2, const void* wg1d2_image,
                                const void* wg2_image, const float* scales, float* out, int batch, int nq, int embed,
                                void* stream) {
  if (!query_xyz || (!src_xyz && !anchor_xyz) || !idx || !qg || !kg || !v || !wd1 || !bd1 || !wd2_image || !bd2 ||
      !wg1d2_image || !wg2_image || !scales || !out || batch <= 0 || nq <= 0 || nsrc <= 0)
    return POEM_E_ARG;
  if (embed < 128 || embed > 1024 || (embed & (embed - 1))) return POEM_E_UNSUPPORTED;
  HIPCHK(poem_launch_vector_attention_split(query_xyz, src_xyz, anchor_xyz, idx, shared_idx, qg, kg, v, nsrc, wd1, bd1,
                                            wd2_image, bd2, wg1d2_image, wg2_image, scales, out, batch, nq, embed, embed,
                                            embed, embed, (hipStream_t)stream));
  return POEM_OK;
}

int poem_cross_attention_split_f16x3(const float* q, const float* k, const float* v, float* ctx, int batch, int nq, int nk,
                                     int embed, int heads, void* scratch, size_t scratch_bytes, void* stream) {
  if (heads <= 0 || embed % heads || (embed / heads != 32 && embed / heads != 64)) return POEM_E_UNSUPPORTED;
  poem_cross_attention_split(1);
  const int rc = poem_cross_attention(q, k, v, ctx, batch, nq, nk, embed, heads, scratch, scratch_bytes, stream);
  poem_cross_attention_split(0);
  return rc;
}

int poem_knn(const float* query_xyz, const float* src_xyz, int32_t* idx, int batch, int nq, int nsrc, void* stream) {
  if (!query_xyz || !src_xyz || !idx || batch <= 0 || nq <= 0 || nsrc < 32 || nsrc > 4096) return POEM_E_ARG;
  HIPCHK(poem_launch_knn(query_xyz, src_xyz, idx, batch, nq, nsrc, 0, (hipStream_t)stream));
  return POEM_OK;
}

int poem_knn_ex(const float* query_xyz, const float* src_xyz, int32_t* idx, int batch, int nq, int nsrc, int fma_contract,
                void* stream) {
  if (!query_xyz || !src_xyz || !idx || batch <= 0 || nq <= 0 || nsrc < 32 || nsrc > 4096 || (fma_contract & ~1)) return POEM_E_ARG;
  HIPCHK(poem_launch_knn(query_xyz, src_xyz, idx, batch, nq, nsrc, fma_contract, (hipStream_t)stream));
  return POEM_OK;
}

int poem_vector_attention(const float* query_xyz, const float* src_xyz, const float* anchor_xyz, const int32_t* idx,
                          int shared_idx, const float* q, const float* k, const float* v, int nsrc, const float* wd1,
                          const float* bd1, const void* wd2_packed, const float* bd2, const void* wg1_packed,
                          const float* bg1, const void* wg2_packed, const float* bg2, float* out, int batch, int nq,
                          int embed, void* stream) {
  if (!query_xyz || (!src_xyz && !anchor_xyz) || !idx || !q || !k || !v || !wd1 || !bd1 || !wd2_packed || !bd2 ||
      !wg1_packed || !bg1 || !wg2_packed || !bg2 || !out || batch <= 0 || nq <= 0)
    return POEM_E_ARG;
  HIPCHK(poem_launch_vector_attention(query_xyz, src_xyz, anchor_xyz, idx, shared_idx, q, k, v, nsrc, wd1, bd1, wd2_packed,
                                      bd2, wg1_packed, bg1, wg2_packed, bg2, out, batch, nq, embed, embed, embed, embed, 0,
                                      (hipStream_t)stream));
  return POEM_OK;
}

int poem_reg_update(const float* r, const float* w, const float* b, const float* xyz_in, float* xyz_out, int rows,
                    int embed, void* stream) {
  if (!r || !w || !b || !xyz_in || !xyz_out || rows <= 0) return POEM_E_ARG;
  HIPCHK(poem_launch_narrow_linear(r, embed, w, b, xyz_in, xyz_out, rows, embed, 3, (hipStream_t)stream));
  return POEM_OK;
}

int poem_triangulate_dlt(const float* uv, const float* cam_intr, const float* cam_mat, const int32_t* view_offsets,
                         int batch, int njoints, int invert, float* out_xyz, void* stream) {
  if (!uv || !cam_intr || !cam_mat || !view_offsets || !out_xyz || batch <= 0 || njoints <= 0 || (invert & ~1))
    return POEM_E_ARG;
  HIPCHK(poem_launch_dlt(uv, cam_intr, cam_mat, view_offsets, out_xyz, batch, njoints, invert, (hipStream_t)stream));
  return POEM_OK;
}

int poem_heatmap_uv(const float* heatmaps, float* uv, int views, int njoints, int hm_h, int hm_w, float img_w, float img_h,
                    void* stream) {
  if (!heatmaps || !uv || views <= 0 || njoints <= 0 || hm_h <= 0 || hm_w <= 0) return POEM_E_ARG;
  HIPCHK(poem_launch_heatmap_uv(heatmaps, uv, views * njoints, hm_h, hm_w, img_w, img_h, (hipStream_t)stream));
  return POEM_OK;
}

size_t poem_conv3x3_packed_bytes(int cout, int cin) {
  if (cout <= 0 || cin <= 0 || cin % 8) return 0;
  return poem_conv3x3_packed_floats(cout, cin) * sizeof(float);
}

int poem_pack_conv3x3(const float* w_oihw, int cout, int cin, void* packed, void* stream) {
  if (!w_oihw || !packed || cout <= 0 || cin <= 0 || cin % 8) return POEM_E_ARG;
  HIPCHK(poem_launch_pack_conv3x3(w_oihw, cout, cin, packed, (hipStream_t)stream));
  return POEM_OK;
}

int poem_conv3x3(const float* in_padded, const void* w_packed, const float* scale, const float* shift,
                 const float* residual, float* out, int views, int cin, int cout, int h, int w, int stride, int relu,
                 int64_t out_view_stride, int out_ch_stride, int out_row_stride, int out_offset, void* stream) {
  if (!in_padded || !w_packed || !scale || !shift || !out || views <= 0 || cin <= 0 || cin % 8 || cout <= 0) return POEM_E_ARG;
  if ((stride != 1 && stride != 2) || h <= 0 || w <= 0 || h % stride || w % stride || ((h / stride) * (w / stride)) % 32)
    return POEM_E_UNSUPPORTED;
  HIPCHK(poem_launch_conv3x3(in_padded, w_packed, scale, shift, residual, out, views, cin, cout, h, w, stride, relu,
                             (long)out_view_stride, out_ch_stride, out_row_stride, out_offset, (hipStream_t)stream));
  return POEM_OK;
}

int poem_conv3x3_down2(const float* in, const void* w_packed, const float* scale, const float* shift, const float* residual,
                       float* out, int views, int cin, int cout, int h, int w, int relu, int64_t out_view_stride,
                       int out_ch_stride, int out_row_stride, int out_offset, void* stream) {
  if (!in || !w_packed || !scale || !shift || !out || views <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return POEM_E_ARG;
  const hipError_t e = poem_launch_conv3x3_down2(in, w_packed, scale, shift, residual, out, views, cin, cout, h, w, relu,
                                                 (long)out_view_stride, out_ch_stride, out_row_stride, out_offset, (hipStream_t)stream);
  if (e == hipErrorNotSupported) return POEM_E_UNSUPPORTED;
  HIPCHK(e);
  return POEM_OK;
}

int poem_upcat_conv3x3(const float* a_half, int ca, const float* b_full, int cb, const void* w_packed, const float* scale,
                       const float* shift, float* out, int views, int cout, int h, int w, int relu, int64_t out_view_stride,
                       int out_ch_stride, int out_row_stride, int out_offset, void* stream) {
  if ((!a_half && ca) || (!b_full && cb) || !w_packed || !scale || !shift || !out || views <= 0 || cout <= 0 || h <= 0 || w <= 0)
    return POEM_E_ARG;
  const hipError_t e = poem_launch_upcat_conv3x3(a_half, ca, b_full, cb, w_packed, scale, shift, out, views, cout, h, w, relu,
                                                 (long)out_view_stride, out_ch_stride, out_row_stride, out_offset, (hipStream_t)stream);
  if (e == hipErrorNotSupported) return POEM_E_UNSUPPORTED;
  HIPCHK(e);
  return POEM_OK;
}

int poem_upcat_conv3x3_pool_head(const float* a_half, int ca, const float* b_full, int cb, const void* w_packed, const float* scale,
                                 const float* shift, const float* head_w, const float* head_b, float* heatmaps, int views, int cout,
                                 int j, int h, int w, int relu, void* stream) {
  if (!a_half || !b_full || !w_packed || !scale || !shift || !head_w || !head_b || !heatmaps || views <= 0 || cout <= 0 || j <= 0 ||
      h <= 0 || w <= 0 || ca <= 0 || cb <= 0)
    return POEM_E_ARG;
  const hipError_t e = poem_launch_upcat_conv3x3_pool_head(a_half, ca, b_full, cb, w_packed, scale, shift, head_w, head_b, heatmaps, views,
                                                           cout, j, h, w, relu, (hipStream_t)stream);
  if (e == hipErrorNotSupported) return POEM_E_UNSUPPORTED;
  HIPCHK(e);
  return POEM_OK;
}

int poem_set_decode_option(const char* name, int value) {
  if (!name) return POEM_E_ARG;
  if (!strcmp(name, "s2_staging_wave")) { poem_decode_s2_staging_wave(value); return POEM_OK; }
  if (!strcmp(name, "row_stager")) { if (value < 0 || value > 3) return POEM_E_ARG; poem_decode_row_stager(value); return POEM_OK; }
  if (!strcmp(name, "pin32")) { poem_decode_pin32(value); return POEM_OK; }
  if (!strcmp(name, "pool_fused")) { poem_decode_pool_fused(value); return POEM_OK; }
  return POEM_E_ARG;
}

int poem_conv1x1_upsample2(const float* in, const void* w_packed, const float* bias, float* out, int views, int cin, int cout,
                           int h, int w, void* stream) {
  if (!in || !w_packed || !out || views <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return POEM_E_ARG;
  const hipError_t e = poem_launch_conv1x1_up2(in, w_packed, bias, out, views, cin, cout, h, w, (hipStream_t)stream);
  if (e == hipErrorNotSupported) return POEM_E_UNSUPPORTED;
  HIPCHK(e);
  return POEM_OK;
}

int poem_upsample2_concat_pad(const float* a, int ca, const float* b, int cb, float* out, int views, int h, int w, int pad,
                              void* stream) {
  if (!out || views <= 0 || ca < 0 || cb < 0 || ca + cb <= 0 || (ca && !a) || (cb && !b) || h <= 0 || w <= 0) return POEM_E_ARG;
  if (pad < 0 || pad > 1 || (ca && (h % 2 || w % 2))) return POEM_E_UNSUPPORTED;
  HIPCHK(poem_launch_upcat_pad(a, ca, b, cb, out, views, h, w, pad, (hipStream_t)stream));
  return POEM_OK;
}

int poem_pool_conv1x1_sigmoid(const float* x, const float* w, const float* bias, float* heatmaps, int views, int c, int j,
                              int h, int w_, void* stream) {
  if (!x || !w || !bias || !heatmaps || views <= 0 || c <= 0 || j <= 0) return POEM_E_ARG;
  if (c > 64 || j > 32 || h % 2 || w_ % 2) return POEM_E_UNSUPPORTED;
  HIPCHK(poem_launch_pool_head(x, w, bias, heatmaps, views, c, j, h, w_, (hipStream_t)stream));
  return POEM_OK;
}

int poem_pa_epe(const float* pred, const float* gt, float* out, int batch, int npoints, void* stream) {
  if (!pred || !gt || !out || batch <= 0 || npoints < 3) return POEM_E_ARG;
  HIPCHK(poem_launch_pa_epe(pred, gt, out, batch, npoints, (hipStream_t)stream));
  return POEM_OK;
}

int poem_pck_accumulate(const float* pred, const float* gt, int batch, int npoints, double val_min, double val_max,
                        int steps, uint32_t* counts, double* dist_sum, uint32_t* n, float* dist_out, void* stream) {
  if (!pred || !gt || !counts || !dist_sum || !n || batch <= 0 || npoints <= 0 || steps <= 0) return POEM_E_ARG;
  HIPCHK(poem_launch_pck_accumulate(pred, gt, batch, npoints, val_min, val_max, steps, counts, dist_sum, n, dist_out,
                                    (hipStream_t)stream));
  return POEM_OK;
}

int poem_mano_to_openpose(const float* j_regressor, const float* verts, float* joints, int batch, int nverts, void* stream) {
  if (!j_regressor || !verts || !joints || batch <= 0) return POEM_E_ARG;
  if (nverts != 778) return POEM_E_UNSUPPORTED;          // the tip vertex ids are MANO's
  HIPCHK(poem_launch_mano_to_openpose(j_regressor, verts, joints, batch, nverts, (hipStream_t)stream));
  return POEM_OK;
}

int poem_rot6d_to_axis_angle(const float* params, float* pose_aa, float* betas, int batch, void* stream) {
  if (!params || !pose_aa || !betas || batch <= 0) return POEM_E_ARG;
  HIPCHK(poem_launch_rot6d_to_aa(params, pose_aa, betas, batch, (hipStream_t)stream));
  return POEM_OK;
}

size_t poem_mano_table_bytes(void) { return poem_mano_table_floats_impl() * sizeof(float); }

int poem_mano_prepare(const float* v_template, const float* shapedirs, const float* posedirs, const float* j_regressor,
                      const float* weights, void* table, void* stream) {
  if (!v_template || !shapedirs || !posedirs || !j_regressor || !weights || !table) return POEM_E_ARG;
  HIPCHK(poem_launch_mano_prepare(v_template, shapedirs, posedirs, j_regressor, weights, (float*)table, (hipStream_t)stream));
  return POEM_OK;
}

int poem_mano_lbs(const float* pose_aa, const float* betas, const void* table, int batch, int center_idx, float* verts,
                  float* joints, void* stream) {
  if (!pose_aa || !betas || !table || !verts || !joints || batch <= 0 || batch > 65535 || center_idx < -1 || center_idx > 20)
    return POEM_E_ARG;
  HIPCHK(poem_launch_mano_lbs(pose_aa, betas, (const float*)table, verts, joints, batch, center_idx, (hipStream_t)stream));
  return POEM_OK;
}

int poem_warp_affine(const uint8_t* src, const int64_t* src_offsets, const int32_t* src_hw, const double* m_inv,
                     const double* gain, float* out_f32, uint8_t* out_u8, int views, int out_h, int out_w, void* stream) {
  if (!src || !src_offsets || !src_hw || !m_inv || (!out_f32 && !out_u8) || views <= 0 || out_h <= 0 || out_w <= 0)
    return POEM_E_ARG;
  if (views > 65535 || out_h > 4 * 65535) return POEM_E_UNSUPPORTED;      // grid y / z limits
  HIPCHK(poem_launch_warp_affine(src, (const long long*)src_offsets, src_hw, m_inv, gain, out_f32, out_u8, views, out_h,
                                 out_w, (hipStream_t)stream));
  return POEM_OK;
}
}  // extern "C"

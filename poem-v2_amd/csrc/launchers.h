// Kernel launchers of libpoem_hip.so: one per kernel family, defined in the .hip translation units, called by the host code
// (handle.cpp, decoder.cpp, forward.cpp, ops.cpp).  Plain C linkage so that tools/check_abi_decls.py can hold every
// declaration here against its definition.
#pragma once
#include <hip/hip_runtime.h>

#include "chain.h"
#include "merge.h"

extern "C" {
hipError_t poem_launch_pack_linear(const float* w, int N, int K, void* out, hipStream_t s);
hipError_t poem_launch_gemm(const float* X, int ldx, const void* Wp, const float* bias, const float* R, int ldr,
                            float* Y, int ldy, int M, int N, int K, int act, hipStream_t s);
hipError_t poem_launch_gemm2(const float* X, int ldx, const void* Wp, const float* bias, const float* R, int ldr,
                             float* Y, int ldy, int M, int N, int K, int act, int in_pa, int out_pa, hipStream_t s);
hipError_t poem_launch_unpack_rows(const void* pa, int N, int K, float* out, hipStream_t s);
hipError_t poem_launch_layernorm(const float* x, const float* g, const float* b, float* y, int rows, int cols,
                                 float eps, hipStream_t s);
hipError_t poem_launch_narrow_linear(const float* x, int ldx, const float* w, const float* b, const float* base,
                                     float* out, int rows, int K, int N, hipStream_t s);
hipError_t poem_launch_sine_pe(float* out, int F, int H, int W, int max_views, int normalize, hipStream_t s);
hipError_t poem_launch_frustum_features(const float* intr, const float* extr, float* out, int views, int H, int W, int D, int lid,
                                        double depth_start, double depth_end, const double* position_range, int img0, int img1,
                                        hipStream_t s);
hipError_t poem_launch_conv1x1_ex(const float* feat, const void* Wp, const float* bias, const float* table, const int* pe_index,
                                  float* x, float* xt, int views, int K, int C, int hw, int relu, hipStream_t s);
int poem_sample_merge_supported(int C, int S, int hw);
hipError_t poem_launch_project_table(const float* bps, const float* centre, const int* view_sample, const float* intr,
                                     const float* inv_extr, void* tabw, void* tabo, float* uv, int views, int C, int fh, int fw, int S,
                                     int img_w, int img_h, hipStream_t s);
hipError_t poem_launch_view_layout(const ViewLayoutArgs* a, hipStream_t s);
hipError_t poem_launch_input_tables(const float* bps, const float* ref_joints, const float* tmpl, const int* view_sample,
                                    const float* intr, const float* extr, void* tabw, void* tabo, float* centre, float* pt_xyz, float* query_xyz,
                                    int views, int B, int S, int Q, int fh, int fw, int img_w, int img_h, float radius, hipStream_t s);
hipError_t poem_launch_sample_merge(const SampleMergeArgs* a, int C, hipStream_t s);
hipError_t poem_launch_sample_group(const SampleGroupArgs* a, int C, hipStream_t s);
int poem_device_cu_count(void);
hipError_t poem_launch_merge_tail(const MergeTailArgs* a, int C, hipStream_t s);
hipError_t poem_launch_invert_extr(const float* extr, float* inv, int views, hipStream_t s);
hipError_t poem_launch_conv1x1(const float* feat, const void* Wp, const float* bias, const float* table,
                               const int* pe_index, float* x, float* xt, int views, int K, int C, int hw, hipStream_t s);
hipError_t poem_launch_project_sample(const float* x, const float* bps, const float* centre, const int* view_sample,
                                      const float* intr, const float* extr, float* inv_scratch, float* uv, float* g,
                                      int views, int C, int fh, int fw, int S, int img_w, int img_h, hipStream_t s);
hipError_t poem_launch_project_uv(const float* bps, const float* centre, const int* view_sample, const float* intr,
                                  const float* extr, float* inv_scratch, float* uv, int views, int fh, int fw, int S, int img_w,
                                  int img_h, hipStream_t s);
hipError_t poem_launch_grid_sample(const float* x, const float* uv, float* g, int views, int C, int fh, int fw, int S, hipStream_t s);
hipError_t poem_launch_merge_reduce(const float* h2, const int* offs, float* m, int B, int S, int HALF, hipStream_t s);
hipError_t poem_launch_merge_finalize(const float* g, const float* y, const int* offs, float* out, int B, int S, int C,
                                      hipStream_t s);
hipError_t poem_launch_cross_attention(const float* q, const float* k, const float* v, float* ctx, int B, int NQ, int NK,
                                       int C, int heads, int ldkv, float* scratch, hipStream_t s);
size_t poem_cross_attention_scratch_floats(int B, int NQ, int NK, int C, int heads, int with_images);
hipError_t poem_launch_cross_attention_img(const float* q, int ldq, const void* kimg, const void* vimg, float* ctx, int B,
                                           int NQ, int NK, int C, int heads, float* scratch, hipStream_t s);
hipError_t poem_launch_gemm_segs(const float* X, int ldx, const void* Wp, const float* bias, int M, int K, int act,
                                 int seg_cols, int nsegs, float* const* outs, const int* modes, hipStream_t s);
hipError_t poem_launch_cross_attention_imgq(const float* q, int ldq, int q_batch_rows, const void* kimg, const void* vimg,
                                            float* ctx, int B, int NQ, int NK, int C, int heads, float* scratch, hipStream_t s);
int poem_chain_supported(int C);
int poem_chain_combines(int C, int heads, int chunks);
void poem_cross_attention_partials(int B, int NQ, int NK, int C, int heads, float* scratch, const void** part_o,
                                   const void** part_ml, int* chunks, float* kc2);
hipError_t poem_launch_chain(const ChainArgs* a, int C, hipStream_t s);
// chain16.hip: native 16x16x4 image of a packed weight (`bytes` of 1 KiB fragment blocks), for the one-unit tiles' weight ring
hipError_t poem_launch_native16(const void* packed, void* native, size_t bytes, hipStream_t s);
int poem_chain16_wants_native(int C);
hipError_t poem_launch_knn(const float* qxyz, const float* sxyz, int* idx, int B, int NQ, int NS, int fma, hipStream_t s);
hipError_t poem_launch_vector_attention(const float* query_xyz, const float* src_xyz, const float* anchor_xyz,
                                        const int* idx, int shared_idx, const float* q, const float* k, const float* v,
                                        int nsrc, const float* wd1, const float* bd1, const void* wd2, const float* bd2,
                                        const void* wg1, const float* bg1, const void* wg2, const float* bg2, float* out,
                                        int B, int Q, int C, int ldq, int ldk, int ldv, int composed, hipStream_t s);
size_t poem_vector_attention_table_floats(int Q, int C);
hipError_t poem_launch_vector_attention_tables(const float* query_xyz, const float* anchor_xyz, const int* idx,
                                               const float* wd1, const float* bd1, const void* wd2, const float* bd2,
                                               const void* wg1d2, float* tab_g, float* tab_p, int Q, int C, hipStream_t s);
hipError_t poem_launch_vector_attention_anchored(const int* idx, const float* qg, const float* kg, const float* v, int nsrc,
                                                 const void* wg2, const float* tab_g, const float* tab_p, float* out, int B,
                                                 int Q, int C, int ldq, int ldk, int ldv, hipStream_t s);
hipError_t poem_launch_gather_anchor_rows(const float* src, int ld, const int* idx, int NS, float* dst, int B, int C,
                                          int* ident, hipStream_t s);
hipError_t poem_launch_canon_xyz(const float* tmpl, float* out, int n, float radius, hipStream_t s);
hipError_t poem_launch_pack_split(const float* w, int C, void* img, float* scale_out, hipStream_t s);
hipError_t poem_launch_pack_split_tiles(const float* w, int N, int K, void* img, float* scales, int scale_stride, hipStream_t s);
void poem_gemm_split_context(const void* packed, size_t bytes, const void* split, const float* scales);
void poem_gemm_split_explicit(const void* img, const float* scales);
void poem_cross_attention_split(int on);
void poem_cross_attention_half(int on);
void poem_cross_attention_tail_halves(int on);
int poem_cross_attention_merges(int NK, int C, int heads);
hipError_t poem_launch_cross_attention_merged_rm(const float* q, const float* k, const float* v, float* ctx, int B, int NQ, int NK,
                                                 int C, int heads, float* scratch, hipStream_t s);
hipError_t poem_launch_cross_attention_merged(const float* q, int ldq, int q_batch_rows, const void* kimg, const void* vimg, float* ctx,
                                              int B, int NQ, int NK, int C, int heads, hipStream_t s);
int poem_gemm_split_applies(const void* Wp, int M, int ldx, int K);
void poem_gemm_split_images(int on);
void poem_gemm_xcd_map(int on);
void poem_gemm_kslab(int on);
void poem_vecattn_one_query_blocks(int on);
void poem_vecattn_valid_neighbours(int k);
hipError_t poem_launch_vector_attention_split(const float* query_xyz, const float* src_xyz, const float* anchor_xyz,
                                              const int* idx, int shared_idx, const float* q, const float* k,
                                              const float* v, int nsrc, const float* wd1, const float* bd1,
                                              const void* wd2, const float* bd2, const void* wg1, const void* wg2,
                                              const float* scales, float* out, int B, int Q, int C, int ldq,
                                              int ldk, int ldv, hipStream_t s);
hipError_t poem_launch_dlt(const float* uv, const float* intr, const float* mat, const int* offs, float* out, int B, int J,
                           int invert, hipStream_t s);
hipError_t poem_launch_pa_epe(const float* pred, const float* gt, float* out, int B, int P, hipStream_t s);
hipError_t poem_launch_pck_accumulate(const float* pred, const float* gt, int B, int P, double vmin, double vmax, int steps,
                                      unsigned int* counts, double* sum, unsigned int* n, float* dist_out, hipStream_t s);
hipError_t poem_launch_mano_to_openpose(const float* jreg, const float* verts, float* joints, int B, int nverts,
                                        hipStream_t s);
hipError_t poem_launch_warp_affine(const unsigned char* src, const long long* src_off, const int* src_hw,
                                   const double* minv, const double* gain, float* out_f32, unsigned char* out_u8, int views,
                                   int OH, int OW, hipStream_t s);
hipError_t poem_launch_heatmap_uv(const float* hmap, float* uv, int maps, int hh, int hw, float img_w, float img_h,
                                  hipStream_t s);
size_t poem_conv3x3_packed_floats(int Cout, int Cin);
hipError_t poem_launch_upcat_conv3x3(const float* a_half, int Ca, const float* b_full, int Cb, const void* wp, const float* scale,
                                     const float* shift, float* out, int views, int Cout, int H, int W, int relu, long out_ns,
                                     int out_cs, int out_rs, int out_off, hipStream_t s);
hipError_t poem_launch_pack_conv3x3(const float* w, int Cout, int Cin, void* out, hipStream_t s);
hipError_t poem_launch_conv3x3(const float* in, const void* wp, const float* scale, const float* shift, const float* res,
                               float* out, int views, int Cin, int Cout, int H, int W, int stride, int relu, long out_ns,
                               int out_cs, int out_rs, int out_off, hipStream_t s);
hipError_t poem_launch_conv3x3_down2(const float* in, const void* wp, const float* scale, const float* shift, const float* res,
                                     float* out, int views, int Cin, int Cout, int H, int W, int relu, long out_ns, int out_cs,
                                     int out_rs, int out_off, hipStream_t s);
void poem_decode_s2_staging_wave(int on);
void poem_decode_row_stager(int on);
void poem_decode_pin32(int on);
void poem_decode_pool_fused(int on);
hipError_t poem_launch_upcat_conv3x3_pool_head(const float* a_half, int Ca, const float* b_full, int Cb, const void* wp,
                                               const float* scale, const float* shift, const float* head_w, const float* head_b,
                                               float* hmap, int views, int Cout, int J, int H, int W, int relu, hipStream_t s);
hipError_t poem_launch_conv1x1_up2(const float* in, const void* wp, const float* bias, float* out, int views, int K, int C, int h,
                                   int w, hipStream_t s);
hipError_t poem_launch_upcat_pad(const float* a, int Ca, const float* b, int Cb, float* out, int views, int H, int W,
                                 int pad, hipStream_t s);
hipError_t poem_launch_pool_head(const float* x, const float* w, const float* bias, float* hmap, int views, int C, int J,
                                 int H, int W, hipStream_t s);
hipError_t poem_launch_gemm_split(const float* X, int ldx, const void* Wp, const float* bias, const float* R, int ldr,
                                  float* Y, int ldy, int M, int N, int K, int act, int act_split, int act2, hipStream_t s);
hipError_t poem_launch_prep_xyz(const float* ref_joints, const float* bps, const float* tmpl, float* centre,
                                float* pt_xyz, float* query_xyz, int B, int S, int Q, float radius, hipStream_t s);
hipError_t poem_launch_broadcast(const float* src, float* dst, long per, int copies, hipStream_t s);
hipError_t poem_launch_finalize(const float* xyz, const float* centre, float* out, int L, int B, int Q, float radius,
                                const float* mano_verts, const float* mano_joints, hipStream_t s);
hipError_t poem_launch_finalize_param(const float* verts, const float* joints, const float* ref_joints, float* out_last,
                                      int B, int Q, hipStream_t s);
hipError_t poem_launch_q3_flatten(const float* feats, const float* fw, const float* fb, float* t, int B, int Q, int C,
                                  hipStream_t s);
hipError_t poem_launch_rot6d_to_aa(const float* par, float* pose_aa, float* betas, int B, hipStream_t s);
hipError_t poem_launch_mano_prepare(const float* v_template, const float* shapedirs, const float* posedirs,
                                    const float* j_regressor, const float* weights, float* table, hipStream_t s);
hipError_t poem_launch_mano_lbs(const float* pose, const float* betas, const float* table, float* verts, float* joints, int B,
                                int center_idx, hipStream_t s);
size_t poem_mano_table_floats_impl();
hipError_t poem_launch_param_rows(const float* verts, const float* joints, float* out_last, int B, int Q, hipStream_t s);
hipError_t poem_launch_compose_weight(const float* A, const float* Bm, float* out, int N, int Cm, int K, hipStream_t s);
hipError_t poem_launch_compose_bias(const float* A, const float* b1, const float* b2, float* out, int N, int Cm,
                                    hipStream_t s);
}

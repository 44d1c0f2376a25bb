// C ABI of libpoem_hip.so (see include/poem_hip.h): handle, weight packing, workspace plan and the launch sequence
// of the whole POEM_Generalized_Head + PtEmbedTRv4 path.  Host code only; all kernels live in the .hip files.
#include "../../include/poem_hip.h"
#include "chain.h"
#include "merge.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

// ---- kernel launchers (defined in the .hip translation units) -----------------------------------------------
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
hipError_t poem_launch_sine_pe(float* out, int F, int H, int W, int max_views, hipStream_t s);
int poem_sample_merge_supported(int C, int S, int hw);
hipError_t poem_launch_project_table(const float* bps, const float* centre, const int* view_sample, const float* intr,
                                     const float* inv_extr, void* tab, float* uv, int views, int C, int fh, int fw, int S,
                                     int img_w, int img_h, hipStream_t s);
hipError_t poem_launch_sample_merge(const SampleMergeArgs* a, int C, hipStream_t s);
hipError_t poem_launch_merge_tail(const MergeTailArgs* a, int C, hipStream_t s);
hipError_t poem_launch_invert_extr(const float* extr, float* inv, int views, hipStream_t s);
hipError_t poem_launch_conv1x1(const float* feat, const void* Wp, const float* bias, const float* table,
                               const int* pe_index, float* x, float* xt, int views, int K, int C, int hw, hipStream_t s);
hipError_t poem_launch_project_sample(const float* x, const float* bps, const float* centre, const int* view_sample,
                                      const float* intr, const float* extr, float* inv_scratch, float* uv, float* g,
                                      int views, int C, int fh, int fw, int S, int img_w, int img_h, hipStream_t s);
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
int poem_gemm_split_applies(const void* Wp, int M, int ldx, int K);
void poem_gemm_split_images(int on);
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
                                hipStream_t s);
hipError_t poem_launch_finalize_param(const float* verts, const float* joints, const float* ref_joints, float* out_last,
                                      int B, int Q, hipStream_t s);
hipError_t poem_launch_q3_flatten(const float* feats, const float* fw, const float* fb, float* t, int B, int Q, int C,
                                  hipStream_t s);
hipError_t poem_launch_rot6d_to_aa(const float* par, float* pose_aa, float* betas, int B, hipStream_t s);
hipError_t poem_launch_mano_lbs(const float* pose, const float* betas, const float* v_template, const float* shapedirs,
                                const float* posedirs, const float* j_regressor, const float* weights, float* verts,
                                float* joints, int B, int center_idx, hipStream_t s);
hipError_t poem_launch_compose_weight(const float* A, const float* Bm, float* out, int N, int Cm, int K, hipStream_t s);
hipError_t poem_launch_compose_bias(const float* A, const float* b1, const float* b2, float* out, int N, int Cm,
                                    hipStream_t s);
}

static thread_local int g_last_hip_error = 0;

// (the sticky per-thread HIP error is cleared first: a stale error left behind by another library on this thread --
//  PyTorch's allocator probing, for one -- would otherwise be read by the launcher's hipGetLastError() and blamed on us)
#define HIPCHK(expr)                                 \
  do {                                               \
    (void)hipGetLastError();                         \
    hipError_t e_ = (expr);                          \
    if (e_ != hipSuccess) {                          \
      g_last_hip_error = (int)e_;                    \
      return POEM_E_LAUNCH;                          \
    }                                                \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline size_t packed_bytes_linear(int N, int K) { return (size_t)((N + 31) / 32) * (size_t)(K / 8) * 64 * 16; }

// ---- canonical tensor table (== poem_v2_amd.weights.live_key_shapes order) -----------------------------------
struct TensorSpec {
  int rows, cols;  // Linear (out, in); 1-D tensors: rows = n, cols = 1
  bool pack;
};

enum {  // head-level slots
  T_INPROJ_W = 0, T_INPROJ_B, T_ADAPT_W, T_ADAPT_B, T_M00_W, T_M00_B, T_M02_W, T_M02_B, T_M10_W, T_M10_B, T_M12_W,
  T_M12_B, T_QEMB, T_HEAD_COUNT
};
enum {  // per-block slots
  B_EMB_W = 0, B_EMB_B,
  B_A1 = 2,    // attn:       +0 q.w +1 q.b +2 k.w +3 k.b +4 v.w +5 v.b +6 o.w +7 o.b +8 ln.w +9 ln.b
  B_A2 = 12,   // cross_attn: same
  B_VS = 22,   // query_self_attn: +0 fc1.w +1 fc1.b +2 fc2.w +3 fc2.b +4 d0.w +5 d0.b +6 d2.w +7 d2.b +8 g0.w +9 g0.b
               //                  +10 g2.w +11 g2.b +12 wq +13 wk +14 wv
  B_VC = 37,   // query_cross_attn: same
  B_REG0_W = 52, B_REG0_B, B_REG2_W, B_REG2_B, B_INT_W, B_INT_B, B_OUT_W, B_OUT_B, B_LN_W, B_LN_B,
  B_COUNT = 62,
  B_FLAT_W = 62, B_FLAT_B, B_MANO_W, B_MANO_B, B_COUNT_PARAM = 66
};

static std::vector<TensorSpec> tensor_table(const poem_config_t& c) {
  const int C = c.embed, Q = c.nquery;
  std::vector<TensorSpec> t;
  auto lin = [&](int o, int i, bool pack, bool bias = true) {
    t.push_back({o, i, pack});
    if (bias) t.push_back({o, 1, false});
  };
  lin(C, c.in_channels, true);
  lin(C, 3 * C / 2, true);
  lin(C, C, true);
  lin(C / 2, C, true);
  lin(C / 2, C / 2, true);
  lin(C, C / 2, true);
  t.push_back({Q, C, false});
  for (int b = 0; b < c.nblocks; ++b) {
    lin(C, C, true);
    for (int a = 0; a < 2; ++a) {
      lin(C, C, true); lin(C, C, true); lin(C, C, true); lin(C, C, true);
      t.push_back({C, 1, false}); t.push_back({C, 1, false});
    }
    for (int a = 0; a < 2; ++a) {
      lin(C, C, true); lin(C, C, true);
      lin(C, 3, false);
      lin(C, C, true); lin(C, C, true); lin(C, C, true);
      lin(C, C, true, false); lin(C, C, true, false); lin(C, C, true, false);
    }
    lin(C, C, true);
    lin(3, C, false);
    lin(4 * C, C, true);
    lin(C, 4 * C, true);
    t.push_back({C, 1, false}); t.push_back({C, 1, false});
    if (c.parametric) {
      lin(1, Q, false);
      lin(106, C, false);
    }
  }
  return t;
}

struct poem_handle_s {
  poem_config_t cfg;
  std::vector<TensorSpec> specs;
  std::vector<const float*> raw;     // caller-owned raw tensors (biases, narrow weights, embedding)
  std::vector<const void*> packed;   // fragment-order image per tensor (nullptr when not packed)
  const float* bps = nullptr;
  const float* anchor = nullptr;
  const int32_t* anchor_idx = nullptr;
  const float* tmpl = nullptr;
  float* pe_table = nullptr;         // (sum N, C, HW)
  // Linears that share their input are fused along N (packed images concatenate tile-wise; results are bit-identical
  // to the separate GEMMs -- every output column is its own fma chain):
  //   F4 (5C x C): reg_branch.0 (relu) | intermediate.dense (gelu)                                       input f_cross
  // and Linears that follow each other WITHOUT a non-linearity are composed into one (W = A B, b = A b1 + b2, fp64
  // products rounded once -- misc.hip compose_*; results agree with the sequential form to fp32 round-off):
  //   F1 (6C x C): (attn.key | attn.value | cross_attn.key | cross_attn.value) o embedding
  //                | (query_cross_attn.w_ks | w_vs) o query_cross_attn.fc1 o embedding          input pt_feats (all blocks)
  //   F2 (2C x C): embedding | attn.query o embedding                                            input query feats
  //   F3 (3C x C): (W_g1 w_qs | W_g1 w_ks | w_vs) o query_self_attn.fc1                          input h_cross
  //   [4] (C x C): W_g1 w_qs of the vector cross attention (bias W_g1 b_d2 + b_g1)               input f_self
  //   [5], [6]   : W_g1 W_d2 of the vector self / cross attention (vecattn.hip, composed form: fc_gamma.0 is linear, so
  //                it is applied to q and k where they are produced and to pos through W_g1 W_d2 -- GEMM 2 of the fused
  //                kernel then reads the same activations as GEMM 1)
  // so `ke`, `xk` and `xs` are never materialised and three GEMMs per block disappear.
  struct Fused { const void* w[7]; const float* b[7]; };   // [4] cross-attn query, [5]/[6] W_g1 W_d2 of self / cross
  std::vector<Fused> fused;
  // Opt-in split-precision vector attention (vecattn_split.hip): hi | lo f16 images of W_d2, W_g1 W_d2, W_g2 per block
  // and attention (self, cross) + their power-of-two scales, in handle-owned device memory (built at creation).
  struct SplitW { const void* w[3]; const float* scales; };
  std::vector<SplitW> split;         // [2 * block + (0 self | 1 cross)]
  void* split_mem = nullptr;
  int precision = 0;                 // POEM_PRECISION_FP32 | POEM_PRECISION_SPLIT_F16X3 | POEM_PRECISION_SPLIT_F16X3_ALL
  // SPLIT_F16X3_ALL: a byte-for-byte mirror of the packed arena holding the hi | lo f16 image of every packed Linear
  // (gemm.hip: same tile size as the fp32 fragment image) + one scale slot per 256 bytes of image
  bool kv_presplit[8] = {};
  const char* packed_base = nullptr;
  size_t packed_size = 0;
  char* gemm_split = nullptr;
  float* gemm_scales = nullptr;
  bool taps = false;
  struct Tap { const void* p; int64_t elems; };
  std::map<std::string, Tap> tapmap;
  // optional HIP-event timing of the dominant kernel (vector attention) on the launch stream
  std::vector<hipEvent_t> prof_ev;   // pairs (start, stop)
  int prof_used = 0;
  std::vector<char> prof_kind;       // per pair: 0 = the full fused kernel, 1 = the anchored (table) form of block 0
  bool prof_on = false;
  // Side streams: the basis-point-side projections of every block (they depend only on bps_feat and the weights) and
  // the neighbour searches run beside the query-side chain; events order them against the caller's stream.
  hipStream_t bps_stream = nullptr, knn_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join_bps = nullptr, ev_join_knn = nullptr, ev_tab = nullptr, ev_fork0 = nullptr;
  // Block 0 of the head path: every sample's query coordinates are the hand template ((c + t) - c)/r -- t/r up to the
  // rounding of c + t -- and the neighbours are the 32 fixed anchors (Q2), so the positional products of both vector
  // attentions are computed ONCE per forward from t/r (vecattn.hip MODE 1) and the per-sample kernels run one C x C GEMM
  // per neighbour column instead of three (MODE 2).  Not bit-identical to the per-sample form (inputs differ by <= 1 ulp
  // of the coordinate); same distance from the reference (tools/lab/hoist_probe.py).  poem_decoder_forward, whose
  // query coordinates are the caller's, never uses it.
  bool anchor_tables = true;
  // Query-side row-tile chains (chain.hip): the Linears / residuals / LayerNorms between the attention kernels of a block
  // run as four chain launches with the activations in LDS instead of ~14 operator launches (fp32 mode, C in {128,256,512}).
  bool chains = true;
  // Fused sampling front end (merge.hip): sampling + Q1 + merge MLP in two kernels, g / h1 never in HBM (fp32 mode, C in
  // {128,256,512}); 0 = the operator sequence of sample.hip + gemm.hip.
  bool fused_sampling = true;
  bool tables_first = true;    // the fused sampling kernel starts behind the anchor-table build (see poem_head_forward)
  bool chain_combine = true;   // chain kind A combines the cross attention's split-key partials itself (no attn_combine launch)
  bool knn_early = true;     // chain mode: issue block i+1's neighbour searches right behind block i's coordinate update
  // hipGraph replay of the step's launch list (everything between the four kernels that read the caller's inputs and the
  // one that writes the caller's output touches workspace / handle memory only): captured once per (batch, view layout,
  // workspace, option set) on an internal stream -- side-stream forks and joins become graph edges -- and replayed with one
  // hipGraphLaunch per forward instead of ~45 launches + ~25 event calls (host enqueue 0.35 -> ~0.1 ms per forward, which is
  // what a small batch's latency sees).  Not used while the HIP-event profile or the per-forward table build is on.
  bool graphs = true;
  bool graph_broken = false;         // a capture failed once on this handle: stay on plain launches
  hipStream_t cap_stream = nullptr;
  struct GraphEntry { std::vector<int64_t> key; hipGraphExec_t exec; uint64_t stamp; };
  std::vector<GraphEntry> graph_cache;
  uint64_t graph_clock = 0;
  static constexpr size_t GRAPH_CAP = 12;
  int knn_fma = 0;           // neighbour distances with the fma contraction of pytorch3d's CUDA kernel (knn.hip); default: the CPU path's rounding
  int chain_tile = 0;        // chain row-tile height: 0 = per launch (chain.hip chain_tile_p), 1 = 32 rows, 2 = 64 rows (A/B)
  // The block-0 anchor tables are functions of the handle's constants only (template, anchors, weights): like the folded
  // positional table they are built ONCE, at poem_create, into handle-owned memory ("block-0
  // fc_delta outputs ... a fixed (799,32,C) table per attention").  tables_cached = false rebuilds them on every forward
  // in the workspace (the round-1/2 behaviour; same kernel, same inputs: bit-identical, tested).
  bool tables_cached = true;
  bool tables_pending = false;       // this forward built the tables on the side stream: consumers wait for ev_tab
  float* tab_mem = nullptr;
  float *c_canon_xyz = nullptr, *c_tab_g[2] = {}, *c_tab_p[2] = {};
  hipEvent_t ev_bps[8] = {}, ev_xyz[8] = {}, ev_knn[8] = {};
  bool overlap = true;
  // Per-view index arrays (view_offsets | view_sample | pe_index) live in handle-owned device memory and are re-uploaded
  // only when the batch's view layout changes: a pageable H2D copy blocks the host until the stream reaches it, i.e.
  // until the PREVIOUS step has finished -- uploading per call kept the host in lock step with the GPU (0.6 ms of idle
  // GPU per step between the last kernel of one forward and the first of the next).
  static constexpr int IDX_CAP = 32768;
  int32_t* idx_dev = nullptr;
  std::vector<int32_t> idx_host;
  int block_base(int b) const { return T_HEAD_COUNT + b * (cfg.parametric ? B_COUNT_PARAM : B_COUNT); }
  const float* R(int idx) const { return raw[idx]; }
  const void* P(int idx) const { return packed[idx]; }
};

static int check_config(const poem_config_t* c) {
  if (!c) return POEM_E_ARG;
  const int C = c->embed;
  if (C < 32 || C > 1024 || (C & (C - 1))) return POEM_E_UNSUPPORTED;       // 32,64,...,1024
  if (c->knn != 32) return POEM_E_UNSUPPORTED;
  if (c->in_channels % 8 || c->nsample % 32 || c->nsample % C) return POEM_E_UNSUPPORTED;
  if (c->heads <= 0 || C % c->heads) return POEM_E_UNSUPPORTED;
  const int dh = C / c->heads;
  if (!(dh == 8 || dh == 16 || dh == 32 || dh == 64 || dh == 128 || dh == 256)) return POEM_E_UNSUPPORTED;
  if (c->nsample > 4096 || c->nquery > 4096 || c->nquery < 33) return POEM_E_UNSUPPORTED;
  if ((c->feat_h * c->feat_w) % 32) return POEM_E_UNSUPPORTED;
  if (c->max_views < 1 || c->max_views > 64 || c->nblocks < 1) return POEM_E_ARG;
  return POEM_OK;
}

// ---- workspace plan -------------------------------------------------------------------------------------------------
namespace {
struct Arena {
  char* base;
  size_t off = 0;
  explicit Arena(void* b) : base((char*)b) {}
  template <typename T>
  T* take(size_t n) {
    off = align_up(off, 256);
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

struct Plan {
  // ints
  int32_t *offs, *view_sample, *pe_index, *idx_self[8], *idx_cross[8];
  // sampling stage
  float *x, *uv, *g, *h1, *h2, *mm, *mh, *y, *bps_feat, *centre, *pt_xyz, *xyz[9];
  float *xt, *ptab, *q1;   // fused sampling: channel-last planes, projection table, residual rows
  // decoder (per call scratch)
  float *feats0, *qp, *ctx, *att, *h_attn, *y3, *rs, *qc, *rc, *y4, *ffo;
  // basis-point side, one set per block (produced ahead of time on the side stream):
  // y1 = 6 x (BS, C): K image 1 | V image 1 | K image 2 | V image 2 (MFMA fragment order, attn.hip) | kc | vc (row-major
  // keys / values of the vector cross attention)
  float *y1[8];
  float *qeqp;     // (BQ, 2C): [qe | first attention's query projection]
  // per block kept tensors (taps)
  float *h_cross[8], *f_self[8], *f_cross[8], *feats[8];
  float *q3t, *par, *attn_scratch, *g_pose, *g_betas;
  float *canon_xyz, *tab_g[2], *tab_p[2];   // block-0 anchor tables (self, cross) of the head path
  float *anch_x[2], *anch_kv[2], *qeqp0;    // block 0: anchor rows of the key/value sources, their (k | v) rows; F2 on Q rows
  int32_t* ident;
  size_t bytes;
};

Plan make_plan(const poem_config_t& c, int B, int BN, void* base) {
  Plan p{};
  Arena a(base);
  const size_t C = c.embed, S = c.nsample, Q = c.nquery, HW = (size_t)c.feat_h * c.feat_w;
  const size_t BS = (size_t)B * S, BQ = (size_t)B * Q, VS = (size_t)BN * S;
  p.offs = a.take<int32_t>(B + 1);
  p.view_sample = a.take<int32_t>(BN);
  p.pe_index = a.take<int32_t>(BN);
  p.x = a.take<float>((size_t)BN * C * HW);
  p.uv = a.take<float>(VS * 2 + (size_t)BN * 16);
  p.g = a.take<float>(VS * C);
  p.h1 = a.take<float>(VS * C);
  p.h2 = a.take<float>(VS * C / 2);
  p.mm = a.take<float>(BS * C / 2);
  p.mh = a.take<float>(BS * C / 2);
  p.y = a.take<float>(BS * C);
  p.bps_feat = a.take<float>(BS * C);
  p.xt = a.take<float>((size_t)BN * C * HW);
  p.ptab = a.take<float>(VS * 8);
  p.q1 = a.take<float>(BS * C);
  p.centre = a.take<float>((size_t)B * 3);
  p.pt_xyz = a.take<float>(BS * 3);
  for (int i = 0; i <= c.nblocks; ++i) p.xyz[i] = nullptr;
  float* xyz_all = a.take<float>((size_t)(c.nblocks + 1) * BQ * 3);   // [0] = initial, [1..] = per-block outputs (contiguous)
  for (int i = 0; i <= c.nblocks; ++i) p.xyz[i] = xyz_all ? xyz_all + (size_t)i * BQ * 3 : nullptr;
  p.feats0 = a.take<float>(BQ * C);
  p.qeqp = a.take<float>(BQ * C * 2);
  p.qp = a.take<float>(BQ * C);
  p.ctx = a.take<float>(BQ * C);
  p.att = a.take<float>(BQ * C);
  p.h_attn = a.take<float>(BQ * C);
  p.y3 = a.take<float>(BQ * C * 3);
  p.rs = a.take<float>(BQ * C);
  p.qc = a.take<float>(BQ * C);
  p.rc = a.take<float>(BQ * C);
  p.y4 = a.take<float>(BQ * C * 5);
  p.ffo = a.take<float>(BQ * C);
  for (int i = 0; i < c.nblocks; ++i) {
    p.h_cross[i] = a.take<float>(BQ * C);
    p.f_self[i] = a.take<float>(BQ * C);
    p.f_cross[i] = a.take<float>(BQ * C);
    p.feats[i] = a.take<float>(BQ * C);
    p.idx_self[i] = a.take<int32_t>(BQ * 32);
    p.idx_cross[i] = a.take<int32_t>(BQ * 32);
    p.y1[i] = a.take<float>(BS * C * 6);
  }
  p.q3t = a.take<float>((size_t)B * C);
  p.par = a.take<float>((size_t)B * 106);
  p.g_pose = a.take<float>((size_t)B * 48);
  p.g_betas = a.take<float>((size_t)B * 10);
  p.attn_scratch = a.take<float>(poem_cross_attention_scratch_floats(B, (int)Q, (int)S, (int)C, c.heads, 0) + 4);
  p.canon_xyz = a.take<float>(Q * 3);
  for (int k = 0; k < 2; ++k) {
    p.anch_x[k] = a.take<float>((size_t)B * 32 * C);
    p.anch_kv[k] = a.take<float>((size_t)B * 32 * C * 2);
  }
  p.qeqp0 = a.take<float>(Q * C * 2);
  p.ident = a.take<int32_t>(32);
  for (int k = 0; k < 2; ++k) {
    p.tab_g[k] = a.take<float>(poem_vector_attention_table_floats((int)Q, (int)C));
    p.tab_p[k] = a.take<float>(poem_vector_attention_table_floats((int)Q, (int)C));
  }
  p.bytes = align_up(a.off, 256);
  return p;
}
}  // namespace

// Block-0 anchor tables (poem_handle_s::anchor_tables): built once per forward on the neighbour-search stream, forked at
// the very top of poem_head_forward -- they depend on the handle's constants only, so they overlap the HBM-bound sampling
// front end instead of competing with the first basis-point GEMM the query side waits for (0.07 ms each when they get
// the chip; 0.7 ms and a later start of block 0 when issued next to that GEMM).
static int build_anchor_tables(poem_handle_t h, Plan& p, hipStream_t s, bool at_create = false) {
  const poem_config_t& c = h->cfg;
  const int C = c.embed, Q = c.nquery;
  const bool ov = !at_create && h->overlap && h->bps_stream && h->knn_stream;
  hipStream_t sk = ov ? h->knn_stream : s;
  if (ov) {
    HIPCHK(hipEventRecord(h->ev_fork0, s));      // the previous forward's readers of the tables are behind this point
    HIPCHK(hipStreamWaitEvent(sk, h->ev_fork0, 0));
  }
  const int bb0 = h->block_base(0);
  HIPCHK(poem_launch_canon_xyz(h->tmpl, p.canon_xyz, Q * 3, c.radius, sk));
  for (int k = 0; k < 2; ++k) {
    const int vb = bb0 + (k == 0 ? B_VS : B_VC);
    HIPCHK(poem_launch_vector_attention_tables(p.canon_xyz, h->anchor, h->anchor_idx, h->R(vb + 4), h->R(vb + 5),
                                               h->P(vb + 6), h->R(vb + 7), h->fused[0].w[5 + k], p.tab_g[k], p.tab_p[k],
                                               Q, C, sk));
  }
  if (ov) HIPCHK(hipEventRecord(h->ev_tab, sk));
  h->tables_pending = ov;
  return POEM_OK;
}

// Decoder (PtEmbedTRv4.forward): p.xyz[0] holds the initial normalised query coordinates; writes p.xyz[1..nblocks].
//
// Three HIP streams.  The caller's stream `s` carries the query-side chain (the critical path).  Everything that
// depends only on the basis-point features -- ke = embedding(pt_feats) and the key/value projections of both BERT
// cross attentions and of the vector cross attention, for EVERY block -- is issued up front on `bps_stream`; the
// neighbour searches of block i (they need only xyz_i) go to `knn_stream`.  The small, latency-bound query-side
// kernels thereby share the chip with the large basis-point GEMMs instead of leaving it half empty.
static int run_decoder(poem_handle_t h, Plan& p, const float* feats_in, const float* pt_xyz, const float* pt_feats, int B,
                       float* pose_aa, float* betas, hipStream_t s, bool template_queries = false) {
  const poem_config_t& c = h->cfg;
  const int C = c.embed, S = c.nsample, Q = c.nquery;
  const int BS = B * S, BQ = B * Q;
  const bool ov = h->overlap && h->bps_stream && h->knn_stream;
  hipStream_t sb = ov ? h->bps_stream : s;     // basis-point side
  hipStream_t sk = ov ? h->knn_stream : s;     // neighbour searches
#define GEMM_ON(ST, X, LDX, WI, BI, RES, LDR, Y, LDY, M, N, K, ACT) \
  HIPCHK(poem_launch_gemm(X, LDX, h->P(WI), (BI) >= 0 ? h->R(BI) : nullptr, RES, LDR, Y, LDY, M, N, K, ACT, ST))
#define GEMM(...) GEMM_ON(s, __VA_ARGS__)
#define PROF_START()                                                                              \
  const bool prof_ = h->prof_on && (size_t)(2 * h->prof_used + 1) < h->prof_ev.size();           \
  if (prof_) HIPCHK(hipEventRecord(h->prof_ev[2 * h->prof_used], s))
#define PROF_STOP(KIND)                                                                           \
  if (prof_) { HIPCHK(hipEventRecord(h->prof_ev[2 * h->prof_used + 1], s)); h->prof_kind[h->prof_used++] = (KIND); }

  // ---- basis-point side of every block (side stream) -------------------------------------------------------------
  if (ov) {
    HIPCHK(hipEventRecord(h->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(sb, h->ev_fork, 0));
    HIPCHK(hipStreamWaitEvent(sk, h->ev_fork, 0));
  }
  // block 0 of the head path (template queries, fixed anchors): see poem_handle_s::anchor_tables
  const bool tables = template_queries && h->anchor_tables && h->precision == POEM_PRECISION_FP32;
  auto bps_side = [&](int i) -> int {
    const auto& f = h->fused[i];
    // F1: keys / values of both BERT cross attentions and of the vector cross attention, straight from the basis-point
    // features (embedding and fc1 composed in); the four BERT blocks leave the GEMM as MFMA fragment images
    const size_t seg = (size_t)BS * C;
    float* outs[6] = {p.y1[i], p.y1[i] + seg, p.y1[i] + 2 * seg, p.y1[i] + 3 * seg, p.y1[i] + 4 * seg, p.y1[i] + 5 * seg};
    const int modes[6] = {1, 2, 1, 2, 0, 0};
    h->kv_presplit[i] = poem_gemm_split_applies(f.w[0], BS, C, C) != 0;      // split GEMM -> the K / V images are split too
    const bool anchored = tables && i == 0;
    HIPCHK(poem_launch_gemm_segs(pt_feats, C, f.w[0], f.b[0], BS, C, POEM_ACT_NONE, C, anchored ? 4 : 6, outs, modes, sb));
    if (anchored) {
      // the vector cross attention of block 0 reads only the 32 anchor rows of (kc | vc): project just those (the same
      // fma chain per element as the full GEMM's rows)
      HIPCHK(poem_launch_gather_anchor_rows(pt_feats, C, h->anchor_idx, S, p.anch_x[1], B, C, p.ident, sb));
      HIPCHK(poem_launch_gemm(p.anch_x[1], C, (const float*)f.w[0] + (size_t)4 * C * C, f.b[0] + 4 * C, nullptr, 0,
                              p.anch_kv[1], 2 * C, B * 32, 2 * C, C, POEM_ACT_NONE, sb));
    }
    if (ov) HIPCHK(hipEventRecord(h->ev_bps[i], sb));
    return POEM_OK;
  };
  if (ov) {
    for (int i = 0; i < c.nblocks; ++i) {
      const int rc = bps_side(i);
      if (rc != POEM_OK) return rc;
    }
  }
  const float* feats = feats_in;
  bool knn_issued[9] = {};
  for (int i = 0; i < c.nblocks; ++i) {
    const int bb = h->block_base(i);
    const float* xyz = p.xyz[i];
    // neighbours (block 0: the fixed anchors for both attentions -- Q2)
    const int* idx_s = h->anchor_idx;
    const int* idx_c = h->anchor_idx;
    const float* anchor = h->anchor;
    int shared = 1;
    if (i > 0) {
      if (!knn_issued[i]) {   // (chain mode issues them right behind the coordinate update of block i-1)
      if (ov) {
        HIPCHK(hipEventRecord(h->ev_xyz[i], s));          // xyz_i is final here (written at the end of block i-1)
        HIPCHK(hipStreamWaitEvent(sk, h->ev_xyz[i], 0));
      }
      // the large search first: it gets its CUs before the persistent attention kernel of this block takes them all
      // (+0.8 %; both searches at once on two streams, or both right behind the xyz update of the previous block, lose)
      HIPCHK(poem_launch_knn(xyz, pt_xyz, p.idx_cross[i], B, Q, S, h->knn_fma, sk));
      HIPCHK(poem_launch_knn(xyz, xyz, p.idx_self[i], B, Q, Q, h->knn_fma, sk));
      if (ov) HIPCHK(hipEventRecord(h->ev_knn[i], sk));
      }
      idx_s = p.idx_self[i];
      idx_c = p.idx_cross[i];
      anchor = nullptr;
      shared = 0;
    }
    if (!ov) {
      const int rc = bps_side(i);
      if (rc != POEM_OK) return rc;
    }
    const bool chain = h->chains && h->precision == POEM_PRECISION_FP32 && poem_chain_supported(C) != 0;
    // F2: qe = embedding(feats) | query projection of the first attention (composed with the embedding)
    const float* hidden = p.qeqp;      // residual of the first attention: qe
    int ldh = 2 * C, hidden_mod = 0, q_batch = Q;
    const float* q0 = p.qeqp + C;
    if (tables && i == 0) {
      // every sample's block-0 query features are the learned embedding table: F2 on its Q rows, once
      HIPCHK(poem_launch_gemm_split(h->R(T_QEMB), C, h->fused[i].w[1], h->fused[i].b[1], nullptr, 0, p.qeqp0, 2 * C, Q,
                                    2 * C, C, POEM_ACT_NONE, 2 * C, POEM_ACT_NONE, s));
      if (chain) {       // ... and read by every sample in place: queries with batch stride 0, residual rows modulo Q
        hidden = p.qeqp0; hidden_mod = Q; q0 = p.qeqp0 + C; q_batch = 0;
      } else {
        HIPCHK(poem_launch_broadcast(p.qeqp0, p.qeqp, (long)Q * 2 * C, B, s));
      }
    } else if (!(chain && i > 0)) {    // (chain mode: block i-1's last chain already wrote p.qeqp)
      HIPCHK(poem_launch_gemm_split(feats, C, h->fused[i].w[1], h->fused[i].b[1], nullptr, 0, p.qeqp, 2 * C, BQ, 2 * C, C,
                                    POEM_ACT_NONE, 2 * C, POEM_ACT_NONE, s));
    }
    const int vsb = bb + B_VS;
    if (chain) {
      const int a1 = bb + B_A1, a2 = bb + B_A2;
      if (ov) HIPCHK(hipStreamWaitEvent(s, h->ev_bps[i], 0));
      // the chain combines the attention's split-key partials while it fills its tile (no attn_combine launch, no ctx round trip)
      const void *part_o = nullptr, *part_ml = nullptr;
      int pchunks = 0;
      float pkc2 = 0.f;
      poem_cross_attention_partials(B, Q, S, C, c.heads, p.attn_scratch, &part_o, &part_ml, &pchunks, &pkc2);
      const bool comb = h->chain_combine && poem_chain_combines(C, c.heads, pchunks) != 0;
      auto from_partials = [&](ChainArgs& a) {
        if (!comb) return;
        a.x = nullptr; a.part_o = (const float4*)part_o; a.part_ml = (const float2*)part_ml;
        a.pc_heads = c.heads; a.pc_chunks = pchunks; a.pc_nq = Q; a.pc_kc2 = pkc2;
      };
      HIPCHK(poem_launch_cross_attention_imgq(q0, 2 * C, q_batch, p.y1[i], p.y1[i] + (size_t)BS * C, comb ? nullptr : p.ctx, B, Q, S,
                                              C, c.heads, p.attn_scratch, s));
      ChainArgs ca{};
      ca.tile_p = h->chain_tile;
      ca.kind = 0; ca.M = BQ; ca.x = p.ctx; ca.ldx = C;
      from_partials(ca);
      ca.w1 = (const float4*)h->P(a1 + 6); ca.b1 = h->R(a1 + 7); ca.res = hidden; ca.ldres = ldh; ca.res_mod = hidden_mod;
      ca.ln_g = h->R(a1 + 8); ca.ln_b = h->R(a1 + 9); ca.eps = c.ln_eps; ca.y1 = p.h_attn; ca.ldy1 = C;
      ca.w2 = (const float4*)h->P(a2 + 0); ca.b2 = h->R(a2 + 1); ca.n2 = 1; ca.y2 = p.qp; ca.ldy2 = C;
      HIPCHK(poem_launch_chain(&ca, C, s));
      HIPCHK(poem_launch_cross_attention_imgq(p.qp, C, Q, p.y1[i] + (size_t)2 * BS * C, p.y1[i] + (size_t)3 * BS * C,
                                              comb ? nullptr : p.ctx, B, Q, S, C, c.heads, p.attn_scratch, s));
      ChainArgs cb{};
      cb.tile_p = h->chain_tile;
      cb.kind = 0; cb.M = BQ; cb.x = p.ctx; cb.ldx = C;
      from_partials(cb);
      cb.w1 = (const float4*)h->P(a2 + 6); cb.b1 = h->R(a2 + 7); cb.res = p.h_attn; cb.ldres = C; cb.res_mod = 0;
      cb.ln_g = h->R(a2 + 8); cb.ln_b = h->R(a2 + 9); cb.eps = c.ln_eps; cb.y1 = p.h_cross[i]; cb.ldy1 = C;
      // F3: (w_qs | w_ks | w_vs) o fc1 on h_cross; block 0 on the tables needs qg for every row and (kg | v) for the anchor rows only
      cb.w2 = (const float4*)h->fused[i].w[2]; cb.b2 = h->fused[i].b[2]; cb.n2 = (tables && i == 0) ? 1 : 3; cb.y2 = p.y3; cb.ldy2 = 3 * C;
      HIPCHK(poem_launch_chain(&cb, C, s));
      hidden = p.h_cross[i];
      ldh = C;
      if (tables && i == 0) {
        HIPCHK(poem_launch_gather_anchor_rows(hidden, C, h->anchor_idx, Q, p.anch_x[0], B, C, nullptr, s));
        HIPCHK(poem_launch_gemm(p.anch_x[0], C, (const float*)h->fused[i].w[2] + (size_t)C * C, h->fused[i].b[2] + C,
                                nullptr, 0, p.anch_kv[0], 2 * C, B * 32, 2 * C, C, POEM_ACT_NONE, s));
      }
    } else {
    for (int a = 0; a < 2; ++a) {
      const int ab = bb + (a == 0 ? B_A1 : B_A2);
      float* hout = a == 0 ? p.h_attn : p.h_cross[i];
      const float* qptr = p.qeqp + C;
      int ldq = 2 * C;
      if (a == 1) {
        GEMM(hidden, ldh, ab + 0, ab + 1, nullptr, 0, p.qp, C, BQ, C, C, POEM_ACT_NONE);
        qptr = p.qp;
        ldq = C;
      }
      if (ov && a == 0) HIPCHK(hipStreamWaitEvent(s, h->ev_bps[i], 0));
      if (h->precision == POEM_PRECISION_SPLIT_F16X3_ALL) poem_cross_attention_split(h->kv_presplit[i] ? 2 : 1);
      HIPCHK(poem_launch_cross_attention_img(qptr, ldq, p.y1[i] + (size_t)(2 * a) * BS * C,
                                             p.y1[i] + (size_t)(2 * a + 1) * BS * C, p.ctx, B, Q, S, C, c.heads,
                                             p.attn_scratch, s));
      GEMM(p.ctx, C, ab + 6, ab + 7, hidden, ldh, p.att, C, BQ, C, C, POEM_ACT_NONE);
      HIPCHK(poem_launch_layernorm(p.att, h->R(ab + 8), h->R(ab + 9), hout, BQ, C, c.ln_eps, s));
      hidden = hout;
      ldh = C;
    }
    // vector self-attention over the queries
    // F3: (w_qs | w_ks | w_vs) o fc1 on h_cross
    if (tables && i == 0) {
      // block 0 gathers keys / values from the 32 anchor rows only: qg for every row, (kg | v) for the anchor rows
      HIPCHK(poem_launch_gemm(hidden, C, h->fused[i].w[2], h->fused[i].b[2], nullptr, 0, p.y3, 3 * C, BQ, C, C,
                              POEM_ACT_NONE, s));
      HIPCHK(poem_launch_gather_anchor_rows(hidden, C, h->anchor_idx, Q, p.anch_x[0], B, C, nullptr, s));
      HIPCHK(poem_launch_gemm(p.anch_x[0], C, (const float*)h->fused[i].w[2] + (size_t)C * C, h->fused[i].b[2] + C,
                              nullptr, 0, p.anch_kv[0], 2 * C, B * 32, 2 * C, C, POEM_ACT_NONE, s));
    } else
    HIPCHK(poem_launch_gemm_split(hidden, C, h->fused[i].w[2], h->fused[i].b[2], nullptr, 0, p.y3, 3 * C, BQ, 3 * C, C,
                                  POEM_ACT_NONE, 3 * C, POEM_ACT_NONE, s));
    }   // !chain
    if (ov && i > 0) HIPCHK(hipStreamWaitEvent(s, h->ev_knn[i], 0));
    {
    PROF_START();
    if (h->precision != POEM_PRECISION_FP32) {
      const auto& sw = h->split[2 * i];
      HIPCHK(poem_launch_vector_attention_split(xyz, xyz, anchor, idx_s, shared, p.y3, p.y3 + C, p.y3 + 2 * C, Q,
                                                h->R(vsb + 4), h->R(vsb + 5), sw.w[0], h->R(vsb + 7), sw.w[1], sw.w[2],
                                                sw.scales, p.rs, B, Q, C, 3 * C, 3 * C, 3 * C, s));
    } else if (tables && i == 0) {
      if (ov && h->tables_pending) HIPCHK(hipStreamWaitEvent(s, h->ev_tab, 0));
      HIPCHK(poem_launch_vector_attention_anchored(p.ident, p.y3, p.anch_kv[0], p.anch_kv[0] + C, 32, h->P(vsb + 10),
                                                   p.tab_g[0], p.tab_p[0], p.rs, B, Q, C, 3 * C, 2 * C, 2 * C, s));
    } else
    HIPCHK(poem_launch_vector_attention(xyz, xyz, anchor, idx_s, shared, p.y3, p.y3 + C, p.y3 + 2 * C, Q, h->R(vsb + 4),
                                        h->R(vsb + 5), h->P(vsb + 6), h->R(vsb + 7), h->fused[i].w[5], h->R(vsb + 9),
                                        h->P(vsb + 10), h->R(vsb + 11), p.rs, B, Q, C, 3 * C, 3 * C, 3 * C, 1, s));
    PROF_STOP(tables && i == 0 ? 1 : 0);
    }
    // vector cross-attention over the basis points
    const int vcb = bb + B_VC;
    if (chain) {       // f_self = fc2(r) + h_cross ; qc = (W_g1 w_qs) f_self + (W_g1 b_d2 + b_g1)
      ChainArgs cc{};
      cc.tile_p = h->chain_tile;
      cc.kind = 1; cc.M = BQ; cc.x = p.rs; cc.ldx = C;
      cc.w1 = (const float4*)h->P(vsb + 2); cc.b1 = h->R(vsb + 3); cc.res = hidden; cc.ldres = C; cc.res_mod = 0;
      cc.y1 = p.f_self[i]; cc.ldy1 = C;
      cc.w2 = (const float4*)h->fused[i].w[4]; cc.b2 = h->fused[i].b[4]; cc.n2 = 1; cc.y2 = p.qc; cc.ldy2 = C;
      HIPCHK(poem_launch_chain(&cc, C, s));
    } else {
    GEMM(p.rs, C, vsb + 2, vsb + 3, hidden, C, p.f_self[i], C, BQ, C, C, POEM_ACT_NONE);
    HIPCHK(poem_launch_gemm(p.f_self[i], C, h->fused[i].w[4], h->fused[i].b[4], nullptr, 0, p.qc, C, BQ, C, C, POEM_ACT_NONE, s));
    }
    {
    PROF_START();
    if (h->precision != POEM_PRECISION_FP32) {
      const auto& sw = h->split[2 * i + 1];
      HIPCHK(poem_launch_vector_attention_split(xyz, pt_xyz, anchor, idx_c, shared, p.qc, p.y1[i] + 4 * (size_t)BS * C,
                                                p.y1[i] + 5 * (size_t)BS * C, S, h->R(vcb + 4), h->R(vcb + 5), sw.w[0],
                                                h->R(vcb + 7), sw.w[1], sw.w[2], sw.scales, p.rc, B, Q, C, C, C, C, s));
    } else if (tables && i == 0) {
      HIPCHK(poem_launch_vector_attention_anchored(p.ident, p.qc, p.anch_kv[1], p.anch_kv[1] + C, 32, h->P(vcb + 10),
                                                   p.tab_g[1], p.tab_p[1], p.rc, B, Q, C, C, 2 * C, 2 * C, s));
    } else
    HIPCHK(poem_launch_vector_attention(xyz, pt_xyz, anchor, idx_c, shared, p.qc, p.y1[i] + 4 * (size_t)BS * C,
                                        p.y1[i] + 5 * (size_t)BS * C, S, h->R(vcb + 4),
                                        h->R(vcb + 5), h->P(vcb + 6), h->R(vcb + 7), h->fused[i].w[6], h->R(vcb + 9),
                                        h->P(vcb + 10), h->R(vcb + 11), p.rc, B, Q, C, C, C, C, 1, s));
    PROF_STOP(tables && i == 0 ? 1 : 0);
    }
    // xyz update
    // F4: reg_branch.0 (relu) | intermediate.dense (gelu) share f_cross
    // The last block's feed-forward output (intermediate -> output -> LayerNorm) feeds nothing: PtEmbedTRv4.forward returns
    // the coordinate stack only (ptEmb_transformer.py:115-121,371-376 upstream; the reference evaluates it and drops it).  It
    // is computed only when something reads it: the parametric tail (medium_MANO) or the debug taps.
    const bool feats_dead = i == c.nblocks - 1 && !c.parametric && !h->taps;
    if (chain) {
      // f_cross = fc2(r) + f_self ; reg_branch -> xyz_{i+1} ; feed forward + LayerNorm -> feats ; next block's F2
      ChainArgs cd{};
      cd.tile_p = h->chain_tile;
      cd.kind = 2; cd.M = BQ; cd.x = p.rc; cd.ldx = C;
      cd.w1 = (const float4*)h->P(vcb + 2); cd.b1 = h->R(vcb + 3); cd.res = p.f_self[i]; cd.ldres = C; cd.res_mod = 0;
      cd.y1 = p.f_cross[i]; cd.ldy1 = C; cd.eps = c.ln_eps;
      cd.wf4 = (const float4*)h->fused[i].w[3]; cd.bf4 = h->fused[i].b[3];
      cd.wreg2 = h->R(bb + B_REG2_W); cd.breg2 = h->R(bb + B_REG2_B); cd.xyz_in = xyz; cd.xyz_out = p.xyz[i + 1];
      HIPCHK(poem_launch_chain(&cd, C, s));       // D1: xyz_{i+1} is final behind it (the next block's searches wait for it)
      if (feats_dead) break;
      if (ov && h->knn_early && i + 1 < c.nblocks) {
        HIPCHK(hipEventRecord(h->ev_xyz[i + 1], s));
        HIPCHK(hipStreamWaitEvent(sk, h->ev_xyz[i + 1], 0));
        HIPCHK(poem_launch_knn(p.xyz[i + 1], pt_xyz, p.idx_cross[i + 1], B, Q, S, h->knn_fma, sk));
        HIPCHK(poem_launch_knn(p.xyz[i + 1], p.xyz[i + 1], p.idx_self[i + 1], B, Q, Q, h->knn_fma, sk));
        HIPCHK(hipEventRecord(h->ev_knn[i + 1], sk));
        knn_issued[i + 1] = true;
      }
      ChainArgs ce{};
      ce.tile_p = h->chain_tile;
      ce.kind = 3; ce.M = BQ; ce.x = p.f_cross[i]; ce.ldx = C; ce.eps = c.ln_eps;
      ce.wf4 = (const float4*)h->fused[i].w[3]; ce.bf4 = h->fused[i].b[3];
      ce.wout = (const float4*)h->P(bb + B_OUT_W); ce.bout = h->R(bb + B_OUT_B);
      ce.ln2_g = h->R(bb + B_LN_W); ce.ln2_b = h->R(bb + B_LN_B); ce.y3 = p.feats[i]; ce.ldy3 = C;
      if (i + 1 < c.nblocks) {
        ce.w2 = (const float4*)h->fused[i + 1].w[1]; ce.b2 = h->fused[i + 1].b[1]; ce.n2 = 2; ce.y2 = p.qeqp; ce.ldy2 = 2 * C;
      }
      HIPCHK(poem_launch_chain(&ce, C, s));
      feats = p.feats[i];
      if (c.parametric && i == c.nblocks - 1) {
        HIPCHK(poem_launch_q3_flatten(feats, h->R(bb + B_FLAT_W), h->R(bb + B_FLAT_B), p.q3t, B, Q, C, s));
        HIPCHK(poem_launch_narrow_linear(p.q3t, C, h->R(bb + B_MANO_W), h->R(bb + B_MANO_B), nullptr, p.par, B, C, 106, s));
        HIPCHK(poem_launch_rot6d_to_aa(p.par, pose_aa, betas, B, s));
      }
      continue;
    }
    GEMM(p.rc, C, vcb + 2, vcb + 3, p.f_self[i], C, p.f_cross[i], C, BQ, C, C, POEM_ACT_NONE);
    if (feats_dead) {
      HIPCHK(poem_launch_gemm(p.f_cross[i], C, h->fused[i].w[3], h->fused[i].b[3], nullptr, 0, p.y4, 5 * C, BQ, C, C,
                              POEM_ACT_RELU, s));
      HIPCHK(poem_launch_narrow_linear(p.y4, 5 * C, h->R(bb + B_REG2_W), h->R(bb + B_REG2_B), xyz, p.xyz[i + 1], BQ, C, 3, s));
      break;
    }
    HIPCHK(poem_launch_gemm_split(p.f_cross[i], C, h->fused[i].w[3], h->fused[i].b[3], nullptr, 0, p.y4, 5 * C, BQ, 5 * C, C,
                                  POEM_ACT_RELU, C, POEM_ACT_GELU, s));
    HIPCHK(poem_launch_narrow_linear(p.y4, 5 * C, h->R(bb + B_REG2_W), h->R(bb + B_REG2_B), xyz, p.xyz[i + 1], BQ, C, 3, s));
    // feed forward (second Linear; its input is y4[:, C:5C])
    GEMM(p.y4 + C, 5 * C, bb + B_OUT_W, bb + B_OUT_B, p.f_cross[i], C, p.ffo, C, BQ, C, 4 * C, POEM_ACT_NONE);
    HIPCHK(poem_launch_layernorm(p.ffo, h->R(bb + B_LN_W), h->R(bb + B_LN_B), p.feats[i], BQ, C, c.ln_eps, s));
    feats = p.feats[i];
    if (c.parametric && i == c.nblocks - 1) {
      HIPCHK(poem_launch_q3_flatten(feats, h->R(bb + B_FLAT_W), h->R(bb + B_FLAT_B), p.q3t, B, Q, C, s));
      HIPCHK(poem_launch_narrow_linear(p.q3t, C, h->R(bb + B_MANO_W), h->R(bb + B_MANO_B), nullptr, p.par, B, C, 106, s));
      HIPCHK(poem_launch_rot6d_to_aa(p.par, pose_aa, betas, B, s));
    }
  }
  if (ov) {   // every side-stream product has been consumed behind an event; join so the caller's stream owns the tail
    HIPCHK(hipEventRecord(h->ev_join_bps, sb));
    HIPCHK(hipEventRecord(h->ev_join_knn, sk));
    HIPCHK(hipStreamWaitEvent(s, h->ev_join_bps, 0));
    HIPCHK(hipStreamWaitEvent(s, h->ev_join_knn, 0));
  }
#undef GEMM
#undef GEMM_ON
#undef PROF_START
#undef PROF_STOP
  return POEM_OK;
}

static void register_taps(poem_handle_t h, const Plan& p, int B, int BN, bool sampling) {
  const poem_config_t& c = h->cfg;
  const int64_t C = c.embed, S = c.nsample, Q = c.nquery, HW = c.feat_h * c.feat_w;
  const int64_t BS = B * S, BQ = B * Q;
  h->tapmap.clear();
  if (!h->taps) return;
  auto put = [&](const std::string& k, const void* ptr, int64_t n) { h->tapmap[k] = {ptr, n}; };
  if (sampling) {
    put("x", p.x, (int64_t)BN * C * HW);
    if (!(h->fused_sampling && h->precision == POEM_PRECISION_FP32 && poem_sample_merge_supported((int)C, (int)S, (int)HW)))
      put("g", p.g, (int64_t)BN * C * S);
    put("bps_feat", p.bps_feat, BS * C);
    put("pt_xyz", p.pt_xyz, BS * 3);
  }
  put("query_xyz", p.xyz[0], BQ * 3);
  for (int i = 0; i < c.nblocks; ++i) {
    const std::string pre = "b" + std::to_string(i) + ".";
    put(pre + "h_cross", p.h_cross[i], BQ * C);
    put(pre + "f_self", p.f_self[i], BQ * C);
    put(pre + "f_cross", p.f_cross[i], BQ * C);
    put(pre + "feats", p.feats[i], BQ * C);
    put(pre + "xyz", p.xyz[i + 1], BQ * 3);
    if (i > 0) {
      put(pre + "idx_self", p.idx_self[i], BQ * 32);
      put(pre + "idx_cross", p.idx_cross[i], BQ * 32);
    }
  }
}

extern "C" {

int poem_abi_version(void) { return 1; }
int poem_last_hip_error(void) { return g_last_hip_error; }
const char* poem_error_string(int code) {
  switch (code) {
    case POEM_OK: return "ok";
    case POEM_E_ARG: return "bad argument";
    case POEM_E_WORKSPACE: return "workspace too small";
    case POEM_E_LAUNCH: return "HIP runtime/launch error";
    case POEM_E_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown";
  }
}

int poem_num_weight_tensors(const poem_config_t* cfg) {
  if (check_config(cfg) != POEM_OK) return POEM_E_UNSUPPORTED;
  return (int)tensor_table(*cfg).size();
}

int64_t poem_weight_tensor_numel(const poem_config_t* cfg, int index) {
  if (check_config(cfg) != POEM_OK) return POEM_E_UNSUPPORTED;
  auto t = tensor_table(*cfg);
  if (index < 0 || index >= (int)t.size()) return POEM_E_ARG;
  return (int64_t)t[index].rows * t[index].cols;
}

static size_t pe_views(int max_views) { return (size_t)max_views * (max_views + 1) / 2; }

size_t poem_packed_bytes(const poem_config_t* cfg) {
  if (check_config(cfg) != POEM_OK) return 0;
  size_t total = 0;
  for (auto& s : tensor_table(*cfg))
    if (s.pack) total += align_up(packed_bytes_linear(s.rows, s.cols), 256);
  const size_t hw = (size_t)cfg->feat_h * cfg->feat_w;
  {   // fused images F1..F4 per block + their concatenated biases (F1, F4)
    const size_t C = cfg->embed;
    const size_t per = align_up(packed_bytes_linear(6 * C, C), 256) + align_up(packed_bytes_linear(2 * C, C), 256) +
                       align_up(packed_bytes_linear(3 * C, C), 256) + align_up(packed_bytes_linear(5 * C, C), 256) +
                       align_up(6 * C * 4, 256) + align_up(2 * C * 4, 256) + align_up(3 * C * 4, 256) + align_up(5 * C * 4, 256) +
                       3 * align_up(packed_bytes_linear(C, C), 256) + 2 * align_up(C * 4, 256);
    total += per * cfg->nblocks;
    total += align_up((6 * C * C + 2 * C * C + 8 * C) * 4, 256) + 256;   // raw composites (init-time scratch, at the arena's end)
  }
  total += align_up((size_t)poem_handle_s::IDX_CAP * 4, 256);                              // per-view index arrays
  total += align_up(pe_views(cfg->max_views) * cfg->embed * hw * 4, 256);              // folded positional table
  total += align_up(pe_views(cfg->max_views) * (3 * cfg->embed / 2) * hw * 4, 256);    // sine scratch (init only)
  return total;
}

size_t poem_packed_linear_bytes(int out_features, int in_features) {
  if (in_features % 8) return 0;
  return packed_bytes_linear(out_features, in_features);
}

int poem_pack_linear(const float* w, int out_features, int in_features, void* packed, void* stream) {
  if (!w || !packed || in_features % 8 || out_features <= 0) return POEM_E_ARG;
  HIPCHK(poem_launch_pack_linear(w, out_features, in_features, packed, (hipStream_t)stream));
  return POEM_OK;
}

int poem_create(const poem_config_t* cfg, const void* const* raw_host, int n, const float* bps, const float* anchor,
                const int32_t* anchor_idx, const float* template_xyz, void* packed, size_t packed_bytes, void* stream,
                poem_handle_t* out) {
  int rc = check_config(cfg);
  if (rc != POEM_OK) return rc;
  if (!raw_host || !bps || !anchor || !anchor_idx || !template_xyz || !packed || !out) return POEM_E_ARG;
  auto* h = new poem_handle_s();
  h->cfg = *cfg;
  h->specs = tensor_table(*cfg);
  if (n != (int)h->specs.size() || packed_bytes < poem_packed_bytes(cfg)) { poem_destroy(h); return POEM_E_ARG; }
  hipStream_t s = (hipStream_t)stream;
  char* cur = (char*)packed;
  h->packed_base = (const char*)packed;
  h->packed_size = packed_bytes;
  if (cfg->embed >= 128) {
    if (hipMalloc((void**)&h->gemm_split, packed_bytes) != hipSuccess ||
        hipMalloc((void**)&h->gemm_scales, (packed_bytes / 256 + 1) * sizeof(float)) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError(); poem_destroy(h); return POEM_E_LAUNCH;
    }
  }
  // mirror of a freshly packed fp32 image at `at`: the split image of the same row-major weight
  auto mirror = [&](const float* w_rows, int rows, int cols, const char* at) -> hipError_t {
    if (!h->gemm_split || cols % 16) return hipSuccess;
    const size_t off = (size_t)(at - (const char*)packed);
    return poem_launch_pack_split_tiles(w_rows, rows, cols, h->gemm_split + off, h->gemm_scales + off / 256, cols / 2, s);
  };
  h->raw.resize(n);
  h->packed.assign(n, nullptr);
  for (int i = 0; i < n; ++i) {
    if (!raw_host[i]) { poem_destroy(h); return POEM_E_ARG; }
    h->raw[i] = (const float*)raw_host[i];
    if (h->specs[i].pack) {
      hipError_t e = poem_launch_pack_linear(h->raw[i], h->specs[i].rows, h->specs[i].cols, cur, s);
      if (e == hipSuccess) e = mirror(h->raw[i], h->specs[i].rows, h->specs[i].cols, cur);
      if (e != hipSuccess) { g_last_hip_error = (int)e; poem_destroy(h); return POEM_E_LAUNCH; }
      h->packed[i] = cur;
      cur += align_up(packed_bytes_linear(h->specs[i].rows, h->specs[i].cols), 256);
    }
  }
  h->bps = bps; h->anchor = anchor; h->anchor_idx = anchor_idx; h->tmpl = template_xyz;
  const int C = cfg->embed, hw = cfg->feat_h * cfg->feat_w;
  if (C >= 128) {
    h->split.resize(2 * cfg->nblocks);
    if (hipMalloc(&h->split_mem, (size_t)2 * cfg->nblocks * ((size_t)3 * C * C * 4 + 256)) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError(); poem_destroy(h); return POEM_E_LAUNCH;
    }
  }
  {
    h->fused.resize(cfg->nblocks);
    // init-time scratch for raw composites: rows (<= 6C x C), T (C x C), t2 (C x C), bias vectors
    float* raw_rows = (float*)((char*)packed + ((packed_bytes - align_up((size_t)(6 * C * C + 2 * C * C + 8 * C) * 4, 256)) & ~(size_t)255));
    float* raw_T = raw_rows + (size_t)6 * C * C;
    float* raw_tb = raw_T + (size_t)2 * C * C;       // C floats (+ spare)
    bool ok = true;
    auto LOK = [&](hipError_t e) { ok = ok && e == hipSuccess; };
    // rows [slot*C, (slot+1)*C) of raw_rows = A . Bm ; bias slot likewise = A . b1 + b2
    auto comp = [&](int slot, const float* A, const float* Bm, const float* b1, const float* b2, float* bias_out) {
      LOK(poem_launch_compose_weight(A, Bm, raw_rows + (size_t)slot * C * C, C, C, C, s));
      LOK(poem_launch_compose_bias(A, b1, b2, bias_out + (size_t)slot * C, C, C, s));
    };
    auto pack_rows = [&](int nslots, const void** wout) {
      *wout = cur;
      LOK(poem_launch_pack_linear(raw_rows, nslots * C, C, cur, s));
      LOK(mirror(raw_rows, nslots * C, C, cur));
      cur += align_up(packed_bytes_linear(nslots * C, C), 256);
    };
    for (int b = 0; b < cfg->nblocks && ok; ++b) {
      const int bb = h->block_base(b);
      const int a1 = bb + B_A1, a2 = bb + B_A2, vs = bb + B_VS, vc = bb + B_VC;
      auto& f = h->fused[b];
      const float* We = h->raw[bb + B_EMB_W];
      const float* be = h->raw[bb + B_EMB_B];
      // ---- F1: basis-point side
      float* b0 = (float*)cur; cur += align_up((size_t)6 * C * 4, 256);
      comp(0, h->raw[a1 + 2], We, be, h->raw[a1 + 3], b0);
      comp(1, h->raw[a1 + 4], We, be, h->raw[a1 + 5], b0);
      comp(2, h->raw[a2 + 2], We, be, h->raw[a2 + 3], b0);
      comp(3, h->raw[a2 + 4], We, be, h->raw[a2 + 5], b0);
      // T = fc1 . embedding, tb = fc1 . be + b_fc1 ; then (W_g1 w_ks) . T and w_vs . T (no bias of their own)
      float* raw_T2 = raw_T + (size_t)C * C;
      LOK(poem_launch_compose_weight(h->raw[vc + 0], We, raw_T, C, C, C, s));
      LOK(poem_launch_compose_bias(h->raw[vc + 0], be, h->raw[vc + 1], raw_tb, C, C, s));
      LOK(poem_launch_compose_weight(h->raw[vc + 8], h->raw[vc + 13], raw_T2, C, C, C, s));      // W_g1 w_ks
      comp(4, raw_T2, raw_T, raw_tb, nullptr, b0);
      comp(5, h->raw[vc + 14], raw_T, raw_tb, nullptr, b0);
      f.b[0] = b0;
      pack_rows(6, &f.w[0]);
      // ---- F2: query side, embedding | attn.query o embedding
      float* b1v = (float*)cur; cur += align_up((size_t)2 * C * 4, 256);
      LOK(hipMemcpyAsync(raw_rows, We, (size_t)C * C * 4, hipMemcpyDeviceToDevice, s));
      LOK(hipMemcpyAsync(b1v, be, (size_t)C * 4, hipMemcpyDeviceToDevice, s));
      comp(1, h->raw[a1 + 0], We, be, h->raw[a1 + 1], b1v);
      f.b[1] = b1v;
      pack_rows(2, &f.w[1]);
      // ---- F3: (W_g1 w_qs | W_g1 w_ks | w_vs) o fc1 of the vector self attention; the query part carries
      //      cvec = W_g1 b_d2 + b_g1 (vecattn.hip, composed form)
      float* b2v = (float*)cur; cur += align_up((size_t)3 * C * 4, 256);
      float* cvec = raw_tb + 2 * C;
      LOK(poem_launch_compose_bias(h->raw[vs + 8], h->raw[vs + 7], h->raw[vs + 9], cvec, C, C, s));
      LOK(poem_launch_compose_weight(h->raw[vs + 8], h->raw[vs + 12], raw_T, C, C, C, s));       // W_g1 w_qs
      comp(0, raw_T, h->raw[vs + 0], h->raw[vs + 1], cvec, b2v);
      LOK(poem_launch_compose_weight(h->raw[vs + 8], h->raw[vs + 13], raw_T, C, C, C, s));       // W_g1 w_ks
      comp(1, raw_T, h->raw[vs + 0], h->raw[vs + 1], nullptr, b2v);
      comp(2, h->raw[vs + 14], h->raw[vs + 0], h->raw[vs + 1], nullptr, b2v);
      f.b[2] = b2v;
      pack_rows(3, &f.w[2]);
      // ---- [4]: query of the vector cross attention, (W_g1 w_qs) f_self + (W_g1 b_d2 + b_g1)
      float* b4v = (float*)cur; cur += align_up((size_t)C * 4, 256);
      LOK(poem_launch_compose_bias(h->raw[vc + 8], h->raw[vc + 7], h->raw[vc + 9], b4v, C, C, s));
      LOK(poem_launch_compose_weight(h->raw[vc + 8], h->raw[vc + 12], raw_rows, C, C, C, s));
      f.b[4] = b4v;
      pack_rows(1, &f.w[4]);
      // ---- [5], [6]: W_g1 W_d2 of the two vector attentions
      LOK(poem_launch_compose_weight(h->raw[vs + 8], h->raw[vs + 6], raw_rows, C, C, C, s));
      pack_rows(1, &f.w[5]);
      LOK(poem_launch_compose_weight(h->raw[vc + 8], h->raw[vc + 6], raw_rows, C, C, C, s));
      pack_rows(1, &f.w[6]);
      f.b[5] = f.b[6] = nullptr;
      if (h->split_mem) {                      // split images: W_d2, W_g1 W_d2 (re-composed into raw_rows), W_g2
        const size_t img = (size_t)C * C * 4;
        for (int a = 0; a < 2; ++a) {
          const int vb = a == 0 ? vs : vc;
          char* base = (char*)h->split_mem + ((size_t)(2 * b + a)) * (3 * img + 256);
          float* sc = (float*)(base + 3 * img);
          LOK(poem_launch_pack_split(h->raw[vb + 6], C, base, sc + 0, s));
          LOK(poem_launch_compose_weight(h->raw[vb + 8], h->raw[vb + 6], raw_rows, C, C, C, s));
          LOK(poem_launch_pack_split(raw_rows, C, base + img, sc + 1, s));
          LOK(poem_launch_pack_split(h->raw[vb + 10], C, base + 2 * img, sc + 2, s));
          h->split[2 * b + a] = {{base, base + img, base + 2 * img}, sc};
        }
      }
      // F4: reg_branch.0 | intermediate.dense share f_cross; intermediate.dense is (4C, C): four C-row slabs of the raw tensor
      f.w[3] = cur;
      ok = ok && poem_launch_pack_linear(h->raw[bb + B_REG0_W], C, C, cur, s) == hipSuccess;
      LOK(mirror(h->raw[bb + B_REG0_W], C, C, cur));
      cur += packed_bytes_linear(C, C);
      ok = ok && poem_launch_pack_linear(h->raw[bb + B_INT_W], 4 * C, C, cur, s) == hipSuccess;
      LOK(mirror(h->raw[bb + B_INT_W], 4 * C, C, cur));
      cur += packed_bytes_linear(4 * C, C);
      cur = (char*)packed + align_up((size_t)(cur - (char*)packed), 256);
      f.b[3] = (const float*)cur;
      ok = ok && hipMemcpyAsync(cur, h->raw[bb + B_REG0_B], (size_t)C * 4, hipMemcpyDeviceToDevice, s) == hipSuccess;
      ok = ok && hipMemcpyAsync(cur + (size_t)C * 4, h->raw[bb + B_INT_B], (size_t)4 * C * 4, hipMemcpyDeviceToDevice, s) == hipSuccess;
      cur += align_up((size_t)5 * C * 4, 256);
    }
    if (!ok) { g_last_hip_error = (int)hipGetLastError(); poem_destroy(h); return POEM_E_LAUNCH; }
  }
  h->idx_dev = (int32_t*)cur;
  cur += align_up((size_t)poem_handle_s::IDX_CAP * 4, 256);
  h->pe_table = (float*)cur;
  cur += align_up(pe_views(cfg->max_views) * C * hw * 4, 256);
  float* sine = (float*)cur;
  rc = poem_pe_table(h->P(T_ADAPT_W), h->R(T_ADAPT_B), C, cfg->feat_h, cfg->feat_w, cfg->max_views, sine, h->pe_table,
                     stream);
  if (rc != POEM_OK) { poem_destroy(h); return rc; }
  {
    bool ok = hipStreamCreateWithFlags(&h->bps_stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->knn_stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) == hipSuccess;
    auto mk = [&](hipEvent_t* e) { ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess; };
    mk(&h->ev_fork); mk(&h->ev_join_bps); mk(&h->ev_join_knn); mk(&h->ev_tab); mk(&h->ev_fork0);
    for (int i = 0; i < 8; ++i) { mk(&h->ev_bps[i]); mk(&h->ev_xyz[i]); mk(&h->ev_knn[i]); }
    if (!ok) { poem_destroy(h); return POEM_E_LAUNCH; }
  }
  {   // block-0 anchor tables (see poem_handle_s::tables_cached): handle-owned, built here once
    const size_t tf = poem_vector_attention_table_floats(cfg->nquery, C);
    const size_t cx = align_up((size_t)cfg->nquery * 3, 64);
    if (hipMalloc((void**)&h->tab_mem, (cx + 4 * tf) * sizeof(float)) != hipSuccess) {
      g_last_hip_error = (int)hipGetLastError(); poem_destroy(h); return POEM_E_LAUNCH;
    }
    h->c_canon_xyz = h->tab_mem;
    Plan tp{};
    tp.canon_xyz = h->c_canon_xyz;
    for (int k = 0; k < 2; ++k) {
      tp.tab_g[k] = h->c_tab_g[k] = h->tab_mem + cx + (size_t)(2 * k) * tf;
      tp.tab_p[k] = h->c_tab_p[k] = h->tab_mem + cx + (size_t)(2 * k + 1) * tf;
    }
    rc = build_anchor_tables(h, tp, s, true);
    if (rc != POEM_OK) { poem_destroy(h); return rc; }
  }
  *out = h;
  return POEM_OK;
}

void poem_destroy(poem_handle_t h) {
  if (!h) return;
  for (auto e : h->prof_ev) (void)hipEventDestroy(e);
  auto de = [](hipEvent_t e) { if (e) (void)hipEventDestroy(e); };
  de(h->ev_fork); de(h->ev_join_bps); de(h->ev_join_knn); de(h->ev_tab); de(h->ev_fork0);
  for (int i = 0; i < 8; ++i) { de(h->ev_bps[i]); de(h->ev_xyz[i]); de(h->ev_knn[i]); }
  if (h->tab_mem) (void)hipFree(h->tab_mem);
  if (h->split_mem) (void)hipFree(h->split_mem);
  if (h->gemm_split) (void)hipFree(h->gemm_split);
  if (h->gemm_scales) (void)hipFree(h->gemm_scales);
  for (auto& g : h->graph_cache) (void)hipGraphExecDestroy(g.exec);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  if (h->bps_stream) (void)hipStreamDestroy(h->bps_stream);
  if (h->knn_stream) (void)hipStreamDestroy(h->knn_stream);
  delete h;
}

int poem_set_overlap(poem_handle_t h, int enable) {
  if (!h) return POEM_E_ARG;
  h->overlap = enable != 0;
  return POEM_OK;
}

int poem_set_chains(poem_handle_t h, int enable) {
  if (!h) return POEM_E_ARG;
  h->chains = enable != 0;
  return POEM_OK;
}

int poem_set_option(poem_handle_t h, const char* name, int value) {
  if (!h || !name) return POEM_E_ARG;
  const std::string k(name);
  if (k == "overlap") h->overlap = value != 0;
  else if (k == "anchor_tables") h->anchor_tables = value != 0;
  else if (k == "chains") h->chains = value != 0;
  else if (k == "knn_early") h->knn_early = value != 0;
  else if (k == "fused_sampling") h->fused_sampling = value != 0;
  else if (k == "chain_combine") h->chain_combine = value != 0;
  else if (k == "tables_first") h->tables_first = value != 0;
  else if (k == "tables_cached") h->tables_cached = value != 0;
  else if (k == "knn_fma") h->knn_fma = value != 0;
  else if (k == "graphs") h->graphs = value != 0;
  else if (k == "chain_tile") { if (value < 0 || value > 3) return POEM_E_ARG; h->chain_tile = value; }
  else return POEM_E_ARG;
  return POEM_OK;
}

int poem_set_anchor_tables(poem_handle_t h, int enable) {
  if (!h) return POEM_E_ARG;
  h->anchor_tables = enable != 0;
  return POEM_OK;
}

int poem_set_precision(poem_handle_t h, int mode) {
  if (!h || mode < POEM_PRECISION_FP32 || mode > POEM_PRECISION_SPLIT_F16X3_ALL) return POEM_E_ARG;
  if (mode != POEM_PRECISION_FP32 && (!h->split_mem || !h->gemm_split)) return POEM_E_UNSUPPORTED;      // embed < 128
  h->precision = mode;
  return POEM_OK;
}

int poem_enable_taps(poem_handle_t h, int enable) {
  if (!h) return POEM_E_ARG;
  h->taps = enable != 0;
  return POEM_OK;
}

int64_t poem_tap(poem_handle_t h, const char* name, void* dst, int64_t dst_elems, void* stream) {
  if (!h || !name) return POEM_E_ARG;
  auto it = h->tapmap.find(name);
  if (it == h->tapmap.end()) return POEM_E_ARG;
  if (dst) {
    if (dst_elems < it->second.elems) return POEM_E_ARG;
    HIPCHK(hipMemcpyAsync(dst, it->second.p, (size_t)it->second.elems * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  }
  return it->second.elems;
}

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

int poem_pe_table(const void* adapt_w_packed, const float* adapt_b, int embed, int fh, int fw, int max_views,
                  float* scratch_sine, float* table, void* stream) {
  if (!adapt_w_packed || !adapt_b || !scratch_sine || !table || embed % 2 || (fh * fw) % 32) return POEM_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(poem_launch_sine_pe(scratch_sine, embed / 2, fh, fw, max_views, s));
  HIPCHK(poem_launch_conv1x1(scratch_sine, adapt_w_packed, adapt_b, nullptr, nullptr, table, nullptr, (int)pe_views(max_views),
                             3 * embed / 2, embed, fh * fw, s));
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

int poem_mano_lbs(const float* pose_aa, const float* betas, const float* v_template, const float* shapedirs,
                  const float* posedirs, const float* j_regressor, const float* weights, int batch, int center_idx,
                  float* verts, float* joints, void* stream) {
  if (!pose_aa || !betas || !v_template || !shapedirs || !posedirs || !j_regressor || !weights || !verts || !joints ||
      batch <= 0 || center_idx < -1 || center_idx > 20)
    return POEM_E_ARG;
  HIPCHK(poem_launch_mano_lbs(pose_aa, betas, v_template, shapedirs, posedirs, j_regressor, weights, verts, joints, batch,
                              center_idx, (hipStream_t)stream));
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

// ---- whole path (entry points) ------------------------------------------------------------------------------

size_t poem_workspace_bytes(poem_handle_t h, int batch, int total_views) {
  if (!h || batch <= 0 || total_views < batch || h->cfg.nblocks > 8) return 0;
  return make_plan(h->cfg, batch, total_views, nullptr).bytes;
}

int poem_head_forward(poem_handle_t h, const float* mlvl_feat, const float* cam_intr, const float* cam_extr,
                      const int32_t* view_offsets_host, int batch, const float* reference_joints, int img_w, int img_h,
                      float* out_xyz, float* pose_aa, float* betas, void* workspace, size_t workspace_bytes,
                      void* stream) {
  if (!h || !mlvl_feat || !cam_intr || !cam_extr || !view_offsets_host || batch <= 0 || !reference_joints || !out_xyz ||
      !workspace || img_w <= 0 || img_h <= 0)
    return POEM_E_ARG;
  const poem_config_t& c = h->cfg;
  if (c.nblocks > 8) return POEM_E_UNSUPPORTED;
  if (c.parametric && (!pose_aa || !betas)) return POEM_E_ARG;
  // SPLIT_F16X3_ALL: every panel GEMM enqueued by this call whose weight lies in this handle's packed arena takes the split
  // image at the same offset (gemm.hip); cleared on every way out
  // (the context is thread-local host state and only a call that installed it clears it: an fp32 forward never touches it)
  struct SplitCtx {
    bool set = false;
    explicit SplitCtx(poem_handle_t hh) {
      if (hh->precision == POEM_PRECISION_SPLIT_F16X3_ALL) {
        set = true;
        poem_gemm_split_context(hh->packed_base, hh->packed_size, hh->gemm_split, hh->gemm_scales);
        poem_cross_attention_split(1);
        const int dh = hh->cfg.embed / hh->cfg.heads;
        poem_gemm_split_images(dh == 32 || dh == 64);          // the head dims the split cross attention takes
      }
    }
    ~SplitCtx() {
      if (!set) return;
      poem_gemm_split_context(nullptr, 0, nullptr, nullptr);
      poem_cross_attention_split(0);
      poem_gemm_split_images(0);
    }
  } split_ctx(h);
  const int B = batch, BN = view_offsets_host[B];
  if (view_offsets_host[0] != 0 || BN < B) return POEM_E_ARG;
  std::vector<int32_t> vs(BN), pei(BN);
  for (int b = 0; b < B; ++b) {
    const int n = view_offsets_host[b + 1] - view_offsets_host[b];
    if (n < 1 || n > c.max_views) return POEM_E_ARG;
    for (int k = 0; k < n; ++k) {
      vs[view_offsets_host[b] + k] = b;
      pei[view_offsets_host[b] + k] = n * (n - 1) / 2 + k;
    }
  }
  Plan p = make_plan(c, B, BN, workspace);
  if (workspace_bytes < p.bytes) return POEM_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int C = c.embed, S = c.nsample, Q = c.nquery, HW = c.feat_h * c.feat_w;
  const int BS = B * S;

  {
    const size_t o1 = align_up((size_t)B + 1, 64), o2 = o1 + align_up((size_t)BN, 64), tot = o2 + align_up((size_t)BN, 64);
    if (h->idx_dev && tot <= (size_t)poem_handle_s::IDX_CAP) {
      std::vector<int32_t> cur(tot, 0);
      std::copy(view_offsets_host, view_offsets_host + B + 1, cur.begin());
      std::copy(vs.begin(), vs.end(), cur.begin() + o1);
      std::copy(pei.begin(), pei.end(), cur.begin() + o2);
      if (cur != h->idx_host) {      // (stream-ordered behind the previous forward's kernels, which may still read the old layout)
        HIPCHK(hipMemcpyAsync(h->idx_dev, cur.data(), tot * sizeof(int32_t), hipMemcpyHostToDevice, s));
        h->idx_host.swap(cur);
      }
      p.offs = h->idx_dev;
      p.view_sample = h->idx_dev + o1;
      p.pe_index = h->idx_dev + o2;
    } else {
      HIPCHK(hipMemcpyAsync(p.offs, view_offsets_host, (B + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(p.view_sample, vs.data(), BN * sizeof(int32_t), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(p.pe_index, pei.data(), BN * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    // the host vectors die at return: pageable H2D copies complete (are staged) before hipMemcpyAsync returns.
  }

  h->tables_pending = false;
  if (h->anchor_tables && h->precision == POEM_PRECISION_FP32) {
    if (h->tables_cached && h->tab_mem) {
      p.canon_xyz = h->c_canon_xyz;
      for (int k = 0; k < 2; ++k) { p.tab_g[k] = h->c_tab_g[k]; p.tab_p[k] = h->c_tab_p[k]; }
    } else {
      const int rc = build_anchor_tables(h, p, s);
      if (rc != POEM_OK) return rc;
    }
  }
  // ---- sampling stage ------------------------------------------------------------------------------------------
  const bool prof_fe = h->prof_on && (size_t)(2 * h->prof_used + 1) < h->prof_ev.size();
  const int prof_fe_slot = h->prof_used;
  if (prof_fe) { HIPCHK(hipEventRecord(h->prof_ev[2 * prof_fe_slot], s)); h->prof_kind[h->prof_used++] = POEM_PROF_SAMPLING; }
  const bool fused_fe = h->fused_sampling && h->precision == POEM_PRECISION_FP32 && poem_sample_merge_supported(C, S, HW) != 0;
  HIPCHK(poem_launch_conv1x1(mlvl_feat, h->P(T_INPROJ_W), h->R(T_INPROJ_B), h->pe_table, p.pe_index,
                             (fused_fe && !h->taps) ? nullptr : p.x, fused_fe ? p.xt : nullptr, BN, c.in_channels, C, HW, s));
  HIPCHK(poem_launch_prep_xyz(reference_joints, h->bps, h->tmpl, p.centre, p.pt_xyz, p.xyz[0], B, S, Q, c.radius, s));
  if (fused_fe) {
    float* inv = p.uv + (size_t)BN * S * 2;
    HIPCHK(poem_launch_invert_extr(cam_extr, inv, BN, s));
    HIPCHK(poem_launch_project_table(h->bps, p.centre, p.view_sample, cam_intr, inv, p.ptab, nullptr, BN, C, c.feat_h, c.feat_w,
                                     S, img_w, img_h, s));
  }
  // Everything from here to the de-normalisation reads and writes workspace / handle memory only.
  auto body = [&](hipStream_t st, float* pose_dst, float* betas_dst) -> int {
#define GEMM(X, LDX, WI, BI, RES, LDR, Y, LDY, M, N, K, ACT) \
  HIPCHK(poem_launch_gemm(X, LDX, h->P(WI), (BI) >= 0 ? h->R(BI) : nullptr, RES, LDR, Y, LDY, M, N, K, ACT, st))
    if (fused_fe) {
      SampleMergeArgs sm{};
      sm.xt = p.xt; sm.tab = (const float4*)p.ptab; sm.view_sample = p.view_sample; sm.offs = p.offs;
      sm.w0 = (const float4*)h->P(T_M00_W); sm.b0 = h->R(T_M00_B); sm.w1 = (const float4*)h->P(T_M02_W); sm.b1 = h->R(T_M02_B);
      sm.h2 = p.h2; sm.q1 = p.q1; sm.views = BN; sm.S = S; sm.hw = HW; sm.h2_tiled = 1;
      // (per-forward table build only) The build on the neighbour-search stream holds 68 KB of LDS per block, and next to the
      // MFMA-dense sample_merge waves its blocks linger: a CU that hosts one takes a single sample_merge block (2 x 66.5 KB no
      // longer fit) and the persistent grid runs in two rounds -- 2.55 instead of 1.69 ms in a third of the forwards.  The
      // build overlaps input_proj / the projection; sample_merge waits for it (+0.06 ms on the critical path).
      if (h->tables_first && h->tables_pending) HIPCHK(hipStreamWaitEvent(st, h->ev_tab, 0));
      HIPCHK(poem_launch_sample_merge(&sm, C, st));
      MergeTailArgs mt{};
      mt.h2 = p.h2; mt.q1 = p.q1; mt.offs = p.offs;
      mt.w0 = (const float4*)h->P(T_M10_W); mt.b0 = h->R(T_M10_B); mt.w1 = (const float4*)h->P(T_M12_W); mt.b1 = h->R(T_M12_B);
      mt.out = p.bps_feat; mt.B = B; mt.S = S; mt.h2_tiled = 1;
      HIPCHK(poem_launch_merge_tail(&mt, C, st));
    } else {
      HIPCHK(poem_launch_project_sample(p.x, h->bps, p.centre, p.view_sample, cam_intr, cam_extr, p.uv + (size_t)BN * S * 2,
                                        p.uv, p.g, BN, C, c.feat_h, c.feat_w, S, img_w, img_h, st));
      // merge MLP 0 on the Q1 rows == the (BN*S, C) row-major view of g's memory
      GEMM(p.g, C, T_M00_W, T_M00_B, nullptr, 0, p.h1, C, BN * S, C, C, POEM_ACT_RELU);
      GEMM(p.h1, C, T_M02_W, T_M02_B, nullptr, 0, p.h2, C / 2, BN * S, C / 2, C, POEM_ACT_NONE);
      HIPCHK(poem_launch_merge_reduce(p.h2, p.offs, p.mm, B, S, C / 2, st));
      GEMM(p.mm, C / 2, T_M10_W, T_M10_B, nullptr, 0, p.mh, C / 2, BS, C / 2, C / 2, POEM_ACT_RELU);
      GEMM(p.mh, C / 2, T_M12_W, T_M12_B, nullptr, 0, p.y, C, BS, C, C / 2, POEM_ACT_NONE);
      HIPCHK(poem_launch_merge_finalize(p.g, p.y, p.offs, p.bps_feat, B, S, C, st));
    }
#undef GEMM
    if (prof_fe) HIPCHK(hipEventRecord(h->prof_ev[2 * prof_fe_slot + 1], st));
    // ---- decoder -------------------------------------------------------------------------------------------------
    HIPCHK(poem_launch_broadcast(h->R(T_QEMB), p.feats0, (long)Q * C, B, st));
    return run_decoder(h, p, p.feats0, p.pt_xyz, p.bps_feat, B, pose_dst, betas_dst, st, true);
  };
  // ---- graph replay or plain launches ----------------------------------------------------------------------------------
  bool replayed = false;
  if (h->graphs && !h->graph_broken && fused_fe && !h->prof_on && !h->tables_pending && h->cap_stream) {
    std::vector<int64_t> key = {B, BN, img_w, img_h, (int64_t)(uintptr_t)workspace, h->precision, h->anchor_tables, h->chains,
                                h->fused_sampling, h->tables_first, h->chain_combine, h->knn_early, h->overlap, h->chain_tile,
                                h->tables_cached, h->knn_fma, h->taps, c.parametric};
    key.insert(key.end(), view_offsets_host, view_offsets_host + B + 1);
    poem_handle_s::GraphEntry* hit = nullptr;
    for (auto& g : h->graph_cache)
      if (g.key == key) { hit = &g; break; }
    if (!hit) {
      // capture the body on the internal stream (the caller's may be the legacy default stream, which cannot be captured);
      // the side-stream forks / joins inside run_decoder become edges of the same graph
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      bool ok = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeRelaxed) == hipSuccess;
      if (ok) {
        const int rc = body(h->cap_stream, p.g_pose, p.g_betas);
        const hipError_t e = hipStreamEndCapture(h->cap_stream, &graph);
        ok = rc == POEM_OK && e == hipSuccess && graph != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) (void)hipGraphDestroy(graph);
      if (!ok) {
        (void)hipGetLastError();
        h->graph_broken = true;               // plain launches from now on (results are the same either way)
      } else {
        if (h->graph_cache.size() >= poem_handle_s::GRAPH_CAP) {      // evict the least recently used
          size_t lru = 0;
          for (size_t i = 1; i < h->graph_cache.size(); ++i)
            if (h->graph_cache[i].stamp < h->graph_cache[lru].stamp) lru = i;
          (void)hipGraphExecDestroy(h->graph_cache[lru].exec);
          h->graph_cache.erase(h->graph_cache.begin() + lru);
        }
        h->graph_cache.push_back({key, exec, 0});
        hit = &h->graph_cache.back();
      }
    }
    if (hit) {
      hit->stamp = ++h->graph_clock;
      HIPCHK(hipGraphLaunch(hit->exec, s));
      if (c.parametric) {
        HIPCHK(hipMemcpyAsync(pose_aa, p.g_pose, (size_t)B * 48 * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(betas, p.g_betas, (size_t)B * 10 * sizeof(float), hipMemcpyDeviceToDevice, s));
      }
      replayed = true;
    }
  }
  if (!replayed) {
    const int rc = body(s, pose_aa, betas);
    if (rc != POEM_OK) return rc;
  }
  HIPCHK(poem_launch_finalize(p.xyz[1], p.centre, out_xyz, c.nblocks, B, Q, c.radius, s));

  register_taps(h, p, B, BN, true);
  return POEM_OK;
}

int poem_decoder_forward(poem_handle_t h, const float* query_xyz, const float* query_feat, const float* pt_xyz,
                         const float* pt_feats, int batch, float* out_xyz_norm, float* pose_aa, float* betas,
                         void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !query_xyz || !query_feat || !pt_xyz || !pt_feats || batch <= 0 || !out_xyz_norm || !workspace)
    return POEM_E_ARG;
  const poem_config_t& c = h->cfg;
  if (c.nblocks > 8) return POEM_E_UNSUPPORTED;
  if (c.parametric && (!pose_aa || !betas)) return POEM_E_ARG;
  Plan p = make_plan(c, batch, batch, workspace);
  if (workspace_bytes < p.bytes) return POEM_E_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)batch * c.nquery * 3;
  HIPCHK(hipMemcpyAsync(p.xyz[0], query_xyz, n * 4, hipMemcpyDeviceToDevice, s));
  const int rc = run_decoder(h, p, query_feat, pt_xyz, pt_feats, batch, pose_aa, betas, s);
  if (rc != POEM_OK) return rc;
  HIPCHK(hipMemcpyAsync(out_xyz_norm, p.xyz[1], n * 4 * c.nblocks, hipMemcpyDeviceToDevice, s));
  register_taps(h, p, batch, batch, false);
  return POEM_OK;
}

int poem_profile_enable(poem_handle_t h, int max_launches) {
  if (!h || max_launches < 0) return POEM_E_ARG;
  for (auto e : h->prof_ev) (void)hipEventDestroy(e);
  h->prof_ev.clear();
  h->prof_used = 0;
  h->prof_on = max_launches > 0;
  h->prof_kind.assign((size_t)max_launches, 0);
  for (int i = 0; i < 2 * max_launches; ++i) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    h->prof_ev.push_back(e);
  }
  return POEM_OK;
}

static int profile_sum(poem_handle_t h, int kind, int* launches, float* total_ms) {
  float tot = 0.f;
  int n = 0;
  for (int i = 0; i < h->prof_used; ++i) {
    if (h->prof_kind[i] != kind) continue;
    HIPCHK(hipEventSynchronize(h->prof_ev[2 * i + 1]));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
    tot += ms;
    ++n;
  }
  *launches = n;
  *total_ms = tot;
  return POEM_OK;
}

int poem_profile_read(poem_handle_t h, int* launches, float* total_ms, int reset) {
  if (!h || !launches || !total_ms) return POEM_E_ARG;
  const int rc = profile_sum(h, 0, launches, total_ms);
  if (rc == POEM_OK && reset) h->prof_used = 0;
  return rc;
}

int poem_profile_read_anchored(poem_handle_t h, int* launches, float* total_ms) {
  if (!h || !launches || !total_ms) return POEM_E_ARG;
  return profile_sum(h, 1, launches, total_ms);
}

int poem_profile_read_stage(poem_handle_t h, int stage, int* launches, float* total_ms) {
  if (!h || !launches || !total_ms || stage < 0 || stage > 7) return POEM_E_ARG;
  return profile_sum(h, stage, launches, total_ms);
}

int poem_finalize_parametric(poem_handle_t h, const float* mano_verts, const float* mano_joints,
                             const float* reference_joints, int batch, float* out_xyz, void* stream) {
  if (!h || !mano_verts || !mano_joints || !reference_joints || !out_xyz || batch <= 0) return POEM_E_ARG;
  const poem_config_t& c = h->cfg;
  float* last = out_xyz + (size_t)(c.nblocks - 1) * batch * c.nquery * 3;
  HIPCHK(poem_launch_finalize_param(mano_verts, mano_joints, reference_joints, last, batch, c.nquery, (hipStream_t)stream));
  return POEM_OK;
}

}  // extern "C"

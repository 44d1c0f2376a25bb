// Host-side state of libpoem_hip.so shared by its translation units: the handle (packed weights, composed Linears, constant
// tables, streams / events, option switches, graph cache), the workspace plan, and the stage functions of the launch sequence.
//   handle.cpp   poem_create / poem_destroy, options, taps, profile read-out        plan.cpp     workspace plan, tap registry
//   decoder.cpp  PtEmbedTRv4.forward: the launch sequence of the three blocks       forward.cpp  poem_head_forward / poem_decoder_forward
//   ops.cpp      the operator-level entry points (thin argument checks around the launchers of launchers.h)
#pragma once
#include "../../include/poem_hip.h"
#include "launchers.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

extern thread_local int g_last_hip_error;      // handle.cpp

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

// POEM_TRACE=<file> in the environment: one line per phase of a forward appended to that file (diagnosis of crashes inside
// the HIP runtime on boxes without a debugger; a file because test runners capture stderr); read once.
static inline FILE* poem_trace_file() {
  static FILE* f = getenv("POEM_TRACE") ? fopen(getenv("POEM_TRACE"), "a") : nullptr;
  return f;
}
#define POEM_TRACE(...) do { if (FILE* tf_ = poem_trace_file()) { fprintf(tf_, __VA_ARGS__); fputc('\n', tf_); fflush(tf_); } } while (0)

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

// first of the four PETR tensors (position_encoder.0.weight, .0.bias, .2.weight, .2.bias), behind every block's slots
static inline int petr_slot(const poem_config_t& c) { return T_HEAD_COUNT + c.nblocks * (c.parametric ? B_COUNT_PARAM : B_COUNT); }
static inline size_t pe_views(int max_views) { return (size_t)max_views * (max_views + 1) / 2; }   // view slots of the folded positional table
std::vector<TensorSpec> tensor_table(const poem_config_t& c);      // handle.cpp
int check_config(const poem_config_t* c);

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
  char* native16 = nullptr;          // mirror of the packed arena: native 16x16x4 images of the packed weights (chain16.hip), C = 128 / 256
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
  // the cross attention merges its four split-key partials inside the kernel and writes the context rows itself (attn.hip
  // xattn_kernel MERGE; fp32 mode, head dim 64, 4096 keys): no partials in HBM, chain kind A fills its tile from ctx.
  // Bit-identical; default OFF: measured 0.5 % slower end to end (the attention +27 us per launch -- four K/V chunk
  // streams per CU instead of one overflow the 32 KB L1 -- against -12 us per chain launch; LABNOTES R3.5)
  // (round 4: -1 = for batches of one or two samples, where the attention is one round of items either way and the chain behind
  //  it is a latency chain whose fill shrinks from four partials to one row: B = 1 / 2 -2.1 / -3.0 %, B = 4 +0.5 %)
  int xattn_merge = -1;
  bool knn_early = true;     // chain mode: issue block i+1's neighbour searches right behind block i's coordinate update
  // hipGraph replay of the step's launch list (everything between the four kernels that read the caller's inputs and the
  // one that writes the caller's output touches workspace / handle memory only): captured once per (batch, view layout,
  // workspace, option set) on an internal stream -- side-stream forks and joins become graph edges -- and replayed with one
  // hipGraphLaunch per forward instead of ~45 launches + ~25 event calls (host enqueue 0.35 -> ~0.1 ms per forward, which is
  // what a small batch's latency sees).  Not used while the HIP-event profile or the per-forward table build is on.
  bool graphs = true;
  bool graph_broken = false;         // a capture failed once on this handle: stay on plain launches
  hipStream_t cap_stream = nullptr;
  int stream_device = 0;             // device whose stream pool (handle.cpp) the three streams belong to
  struct GraphEntry { std::vector<int64_t> key; hipGraphExec_t exec; uint64_t stamp; uint64_t shape; hipStream_t last_stream = nullptr; bool launched = false; };
  std::vector<GraphEntry> graph_cache;
  std::vector<std::vector<int64_t>> graph_seen;      // keys met once: a key is captured at its second forward
  bool graph_eager = false;                          // capture at the first forward of a key (tests / benches that want it)
  uint64_t graph_clock = 0;
  static constexpr size_t GRAPH_CAP = 12;
  // counters (poem_graph_stats)
  int64_t graph_captures = 0, graph_instantiations = 0, graph_replays = 0, plain_forwards = 0, layout_uploads = 0;
  // launch-count / dependency shortcuts, one bit each (A/B; all bit-identical): 1 = one input launch (coordinates + inverse extrinsics
  // + projection table) and no embedding broadcast where block 0 runs on the tables; 2 = block 0's anchor keys / values out of the
  // chain's rows (small batches).  (Measured and dropped: block 0's F1 as two launches, +1 %; a small batch's later F1 GEMMs on
  // a quarter of the CUs, +4 % at B = 4.)
  int small_batch = 3;
  int group_xcd = 1;         // sample_group_kernel's XCD-aware unit order (A/B switch)
  int group_min_views = 0;   // sample_group_kernel: 0 = by the chip (units >= 2 per CU), > 0 = this many views, -1 = never (forward.cpp)
  // The MANO layer of the parametric tail inside the forward (poem_attach_mano): the prepared asset table of poem_mano_prepare
  // (caller-owned device memory, must outlive the handle's forwards) and the layer's centre joint.  nullptr: the forward
  // returns (pose, betas) and the caller runs its own layer + poem_finalize_parametric (rounds 1-5).
  const float* mano_table = nullptr;
  int mano_center = 9;
  int knn_query = 0;         // N_NEIGHBOR_QUERY when it differs from cfg.knn (= N_NEIGHBOR); 0: the same.  Both 1..32: counts below 32 mask
                             // the vector attention's last columns (vecattn.hip MODE 3); block 0 takes the 32 anchors either way (Q2)
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
  // block 0's F2 on the anchor-table path acts on the learned query embedding, the same Q rows for every sample: a function of the
  // weights only -- folded at poem_create like the tables (round 4; it was a 45-75 us GEMM in front of the first cross attention)
  float* c_qeqp0 = nullptr;
  hipEvent_t ev_bps[8] = {}, ev_xyz[8] = {}, ev_knn[8] = {}, ev_def[8] = {};
  // When the basis-point side (F1) of block i + 1 is issued: 0 = every block's up front, beside block 0 (large batches: the
  // matrix pipe is the limit either way); 1 / 2 / 3 = behind block i's first / second cross attention / its chain -- a small
  // batch leaves most CUs idle there, and block 0's own cross attention no longer shares the chip with the later blocks' GEMMs
  int bps_defer = 0;
  // F1 of blocks >= 1 as two launches -- the four attention images (4C columns) and the vector cross attention's (k | v) rows
  // (2C columns): 8 and 4 panels divide an XCD's 32 CUs evenly (12 do not), so both take the XCD-aware map
  bool f1_split = true;
  // one-query blocks of the full vector attention (vecattn.hip): -1 = for small batches (B * Q <= 16 x CUs: the busiest CU gets
  // 7 queries instead of 8 at B = 2; measured B = 1 / 2 / 4 -0.7 / -1.7 / -2.0 %), 0 never, 1 / 2 always (3 / 2 waves per SIMD); same bits
  int va_p1 = -1;
  // D2 issued in front of the next block's neighbour searches (decoder.cpp tail(): which successor of D1 the launch graph keeps on
  // D1's hardware queue): 0 = the searches first (rounds 1-4), 1 = D2 first (measured -1.3 / -0.5 / -0.4 % at B = 1 / 2 / 32);
  // scheduling only
  int d2_first = 1;
  // fewer cross-stream waits on the query-side chain (decoder.cpp DecoderRun::wmerge): bit mask 1 = (a), 2 = (b), 4 = (c) there;
  // -1 = (a) + (b); scheduling only
  int wait_merge = -1;
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

// ---- workspace plan (plan.cpp) ---------------------------------------------------------------------------------------------
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
  float *petr_f, *petr_h, *petr_tab;   // PETR_EMBEDDING: frustum features (BN, 3D, HW), hidden (BN, 2C, HW), per-view table (BN, C, HW)
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
  float *mano_verts, *mano_joints;          // the attached MANO layer's output (B,778,3) / (B,21,3)
  float *canon_xyz, *tab_g[2], *tab_p[2];   // block-0 anchor tables (self, cross) of the head path
  float *anch_x[2], *anch_kv[2], *qeqp0;    // block 0: anchor rows of the key/value sources, their (k | v) rows; F2 on Q rows
  int32_t* ident;
  size_t bytes;
};

Plan make_plan(const poem_config_t& c, int B, int BN, void* base);
void register_taps(poem_handle_t h, const Plan& p, int B, int BN, bool sampling);

// handle.cpp: a retired exec is parked, not destroyed (runtime bug, see there), and offered to the next capture of the same shape
int poem_process_switches();      // handle.cpp: the process-wide launcher switches, for the launch-graph key
void poem_park_graph_exec(hipGraphExec_t e, uint64_t shape, int device, hipStream_t last_stream, bool launched);
hipGraphExec_t poem_reuse_graph_exec(hipGraph_t graph, uint64_t shape, int device);      // a parked exec updated to `graph`, or nullptr

// ---- launch sequence (decoder.cpp) -------------------------------------------------------------------------------------------
int build_anchor_tables(poem_handle_t h, Plan& p, hipStream_t s, bool at_create = false);
int run_decoder(poem_handle_t h, Plan& p, const float* feats_in, const float* pt_xyz, const float* pt_feats, int B, float* pose_aa,
                float* betas, hipStream_t s, bool template_queries = false);

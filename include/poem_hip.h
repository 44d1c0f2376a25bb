/* poem_hip.h -- C ABI of the MI355X-native POEM-v2 point-embedded decoder (libpoem_hip.so).
 *
 * Plain pointers and sizes only; every pointer is a DEVICE pointer unless the name ends in _host.  All entry
 * points are stream-ordered, allocate nothing, never throw, and return 0 on success or a negative POEM_E_* code.
 * Tensors are fp32, row-major/contiguous; indices are int32.  hipStream_t is passed as void*.
 *
 * Threading contract.  The operator-level entry points are re-entrant: they keep no host state, and two host threads may
 * call them concurrently on different streams.  A handle (poem_create) is used by ONE host thread at a time (its side
 * streams, events and cached index upload are per handle; different handles on different threads -- one thread per GPU --
 * are independent).  The opt-in split-precision modes carry their per-call context in thread-local host state that only the
 * call which installed it clears, so a forward on one thread never changes the arithmetic of a forward on another.
 * poem_last_hip_error is per thread.
 *
 * What each entry point replaces in the upstream reference (paths relative to the reference root) -- the
 * reference has no FFI of its own (it is pure Python on torch / pytorch3d / transformers), so these are the
 * native calls its hot path makes today:
 *
 *   poem_gemm                 torch.nn.Linear forward: every Linear of the path
 *                             (lib/models/bricks/pt_metro_transformer.py:180-181 embedding, :88-91 FFN, :25,38
 *                             reg_branch; point_transformers.py:49-56,86-87,138-142 fc1/fc2/w_qs/w_ks/w_vs;
 *                             ptEmb_head.py:701-707 merge_net_feature; BertSelfAttention query/key/value,
 *                             BertSelfOutput.dense)
 *   poem_layernorm            BertSelfOutput.LayerNorm / BertOutput.LayerNorm (transformers, call sites
 *                             pt_metro_transformer.py:57-74,88-91)
 *   poem_input_proj           nn.Conv2d 1x1 input_proj + positional table add (ptEmb_head.py:835,853-870)
 *   poem_pe_table             SinePositionalEncoding3D + adapt_pos3d (petr_transformer.py:434-469,
 *                             ptEmb_head.py:857-858), constant-folded per view count
 *   poem_project_sample       generate_grid_sample_proj + F.grid_sample (lib/utils/collation.py:48-65,
 *                             lib/utils/transform.py:898-930, ptEmb_head.py:873-883,900-901)
 *   poem_merge_reduce /       merge_features_mv / merge_features_sv around the two merge MLPs, on the Q1
 *   poem_merge_finalize       re-interpreted rows (ptEmb_head.py:745-771,910-926)
 *   poem_cross_attention      BertSelfAttention scores/softmax/context for encoder_hidden_states
 *                             (pt_metro_transformer.py:57-72; transformers v4 semantics)
 *   poem_knn                  pytorch3d.ops.knn_points(K=32) (point_transformers.py:83,134)
 *   poem_vector_attention     ptTransformerBlock._forward / ptTransformerBlock_CrossAttn._forward from
 *                             fc_delta to the softmax-weighted sum (point_transformers.py:88-95,144-151)
 *   poem_reg_update           reg_branch second Linear + xyz residual (pt_metro_transformer.py:38)
 *   poem_triangulate_dlt      batch_triangulate_dlt_torch + the ragged per-sample loop (lib/utils/triangulation.py:5-45,
 *                             lib/models/POEM.py:284-299) -- the stage that produces reference_joints
 *   poem_conv3x3 /            ConvBlock (Conv2d 3x3 + BatchNorm(eval) + ReLU, lib/models/bricks/conv.py:4-45) and the glue
 *   poem_upsample2_concat_pad around it in PtEmbedMultiviewStereoV2.feat_decode / uv_decode (lib/models/POEM.py:167-211:
 *   poem_pool_conv1x1_sigmoid F.interpolate x2 + torch.cat, max_pool2d + uv_out + sigmoid); poem_heatmap_uv is the read-out of
 *   poem_heatmap_uv           heatmap_stage (POEM.py:213-222, integral_heatmap2d) -- the stage that produces the head's
 *                             mlvl_feat and the per-view 2-D joints
 *   poem_pa_epe /             PAEval.feed + align_w_scale (lib/metrics/pa_eval.py:45-83,104-124) and _PCKMetric.feed
 *   poem_pck_accumulate       (lib/metrics/pck.py:36-96) -- device-side evaluation metrics
 *   poem_mano_to_openpose     mano_to_openpose (lib/utils/transform.py:836-872) -- joints from mesh for the metrics (N3)
 *   poem_warp_affine          cv2.warpAffine + colour jitter + to_tensor / normalize of SimpleTransform2D.__call__
 *                             (lib/utils/transform.py:153-170) and the mirror warp of process_data_item
 *                             (lib/data_wds/multiview_wds.py:112-118) -- the image side of the input pipeline
 *   poem_rot6d_to_axis_angle  rot6d_to_aa of get_parametric_output (pt_metro_transformer.py:144-146, lib/utils/transform.py:448-466)
 *   poem_mano_lbs             manotorch ManoLayer.forward (pt_metro_transformer.py:120-124,147-148; ptEmb_head.py:732-736,886-892)
 *   poem_head_forward         POEM_Generalized_Head.forward + PtEmbedTRv4.forward (ptEmb_head.py:825-964,
 *                             lib/models/layers/ptEmb_transformer.py:371-376)
 */
#ifndef POEM_HIP_H
#define POEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POEM_OK 0
#define POEM_E_ARG (-1)        /* bad argument (null pointer, unsupported size, misaligned) */
#define POEM_E_WORKSPACE (-2)  /* workspace too small */
#define POEM_E_LAUNCH (-3)     /* HIP launch/runtime error (see poem_last_hip_error) */
#define POEM_E_UNSUPPORTED (-4)

#define POEM_ACT_NONE 0
#define POEM_ACT_RELU 1
#define POEM_ACT_GELU 2 /* erf form */

typedef struct poem_config {
  int32_t embed;       /* C: EMBED_DIMS = POINTS_FEAT_DIM = INPUT_FEAT_DIM (32..1024, multiple of 32) */
  int32_t in_channels; /* IN_CHANNELS (160), multiple of 8 */
  int32_t nsample;     /* S: N_SAMPLE (4096); S % C == 0 (Q1 re-interpretation) */
  int32_t nquery;      /* Q: 799 */
  int32_t heads;       /* NUM_ATTENTION_HEADS (4) */
  int32_t nblocks;     /* N_BLOCKS (3) */
  int32_t knn;         /* N_NEIGHBOR (ptEmb_transformer.py:30), 1..32; N_NEIGHBOR_QUERY (:31) is the same unless
                          poem_set_option(h, "knn_query", k) says otherwise.  The release configs set 32 for both; counts below 32
                          run the masked vector attention (the search's first k of its 32 nearest).  Block 0 takes the 32 fixed
                          anchors whatever the keys say (assets/anchor.npy, point_transformers.py:10-32). */
  int32_t parametric;  /* TRANSFORMER.PARAMETRIC_OUTPUT */
  int32_t feat_h, feat_w; /* backbone feature map (16x16) */
  int32_t max_views;   /* largest views-per-sample the positional table is folded for */
  float radius;        /* RADIUS_SAMPLE (0.1) */
  float ln_eps;        /* BertConfig.layer_norm_eps (1e-12) */
  /* ABI 2: the positional-encoding switches of the reference's constructor.  No release config changes them. */
  int32_t pe_normalize;   /* POSITIONAL_ENCODING.NORMALIZE (1; lib/models/layers/petr_transformer.py:451-457) */
  int32_t petr_embedding; /* PETR_EMBEDDING (0; lib/models/heads/ptEmb_head.py:692,865-867): position_encoder of the cameras'
                           * frustum points is added to the positional embedding; four more weight tensors, listed last */
  int32_t depth_num;      /* DEPTH_NUM (32); with petr_embedding: 3 * depth_num must be a multiple of 8 */
  int32_t lid;            /* LID (0): linear-increasing depth bins (ptEmb_head.py:122-126) */
  int32_t reserved0;
  double depth_start, depth_end; /* DEPTH_START, DEPTH_END (0.0, 1.2) -- doubles, as the reference's Python scalars are:      */
  double position_range[6];      /* POSITION_RANGE (xmin ymin zmin xmax ymax zmax); each is rounded to fp32 where the reference's
                                  * tensor-scalar arithmetic rounds it */
} poem_config_t;

typedef struct poem_handle_s* poem_handle_t;

/* ---- library ------------------------------------------------------------------------------------------ */
int poem_abi_version(void);
int poem_last_hip_error(void);            /* hipError_t of the last failing runtime call on this thread */
const char* poem_error_string(int code);

/* ---- weights ------------------------------------------------------------------------------------------- */
/* Number of raw tensors / their element counts in canonical order (the order of
 * poem_v2_amd.weights.live_key_shapes == the reference's state_dict order of the live tensors). */
int poem_num_weight_tensors(const poem_config_t* cfg);
int64_t poem_weight_tensor_numel(const poem_config_t* cfg, int index);
/* Bytes of the packed device image (MFMA-fragment order weights + folded tables) the handle keeps. */
size_t poem_packed_bytes(const poem_config_t* cfg);
/* raw: host array of `n` device pointers to the raw fp32 tensors in canonical order.
 * bps (S,3), anchor (32,3) fp32, anchor_idx (32) int32, template_xyz (Q,3) fp32 metres: device pointers.
 * packed: caller-owned device buffer of poem_packed_bytes(); must outlive the handle. */
int poem_create(const poem_config_t* cfg, const void* const* raw_host, int n, const float* bps, const float* anchor,
                const int32_t* anchor_idx, const float* template_xyz, void* packed, size_t packed_bytes,
                void* stream, poem_handle_t* out);
void poem_destroy(poem_handle_t h);

/* ---- whole path ---------------------------------------------------------------------------------------- */
/* Bytes of the caller-owned workspace of a forward of `batch` samples with `total_views` views in all.  Passing
 * total_views = batch * cfg.max_views (the worst case of the batch size) makes poem_head_forward lay the workspace out
 * for that capacity whatever the batch's own view counts are: every pointer of the forward is then a function of the batch
 * size alone and ONE captured launch graph per batch size serves every view layout (a stream of ragged batches, the
 * reference's collation_random_n_views, lib/utils/collation.py:7-25).  A workspace sized for the batch's own total works
 * too; its graph is then shared by the layouts with that total only. */
size_t poem_workspace_bytes(poem_handle_t h, int batch, int total_views);
/* Diagnostics of the launch-graph cache, n >= 9 slots: [0] graph execs cached by this handle, [1] stream captures,
 * [2] hipGraphInstantiate calls, [3] forwards replayed from a graph, [4] forwards issued as plain launches,
 * [5] view-layout uploads (a forward whose layout equals the previous one uploads nothing), [6] retired execs parked in the
 * process, [7] parked execs taken over by a later capture (hipGraphExecUpdate), [8] updates the runtime refused, [9] (n >= 10)
 * parked execs passed over because their last launch was still running.  Retired execs are never destroyed (a runtime
 * workaround, csrc/handle.cpp): they are parked per device and re-used, so [6] is bounded by the largest number of execs ever
 * cached at once per (device, shape) -- 12 per handle -- not by the handles created or layouts met; a server that cycles many
 * model shapes should watch [6].  A parked exec is offered for update only on its own device and only after the launch it
 * was retired behind has completed.
 * A forward never blocks the host: the layout travels in a kernel's argument segment. */
int poem_graph_stats(poem_handle_t h, int64_t* out, int n);
/* view_offsets_host: B+1 prefix sums of views per sample (host memory, the reference's cam_view_num).
 * cam_extr is camera->master (as in the reference's img_metas["cam_extr"]).
 * out_xyz: (nblocks, B, Q, 3) metres in the master frame (all_coords_preds).
 * Parametric configs: out_xyz holds the pre-MANO stack; pose_aa (B,48) and betas (B,10) are written and the caller
 * finishes with its MANO layer + poem_finalize_parametric. */
int poem_head_forward(poem_handle_t h, const float* mlvl_feat, const float* cam_intr, const float* cam_extr,
                      const int32_t* view_offsets_host, int batch, const float* reference_joints, int img_w,
                      int img_h, float* out_xyz, float* pose_aa, float* betas, void* workspace,
                      size_t workspace_bytes, void* stream);
/* Decoder only (PtEmbedTRv4.forward, lib/models/layers/ptEmb_transformer.py:371-376): normalised inputs
 * query_xyz (B,Q,3), query_feat (B,Q,C), pt_xyz (B,S,3), pt_feats (B,S,C) -> out_xyz_norm (nblocks,B,Q,3). */
int poem_decoder_forward(poem_handle_t h, const float* query_xyz, const float* query_feat, const float* pt_xyz,
                         const float* pt_feats, int batch, float* out_xyz_norm, float* pose_aa, float* betas,
                         void* workspace, size_t workspace_bytes, void* stream);
/* Stream overlap (default on): the basis-point-side projections of all blocks and the neighbour searches run on two
 * internal side streams ordered against `stream` by events; everything is joined back before the call returns its
 * last launch, so the caller only ever synchronises its own stream.  0 = issue everything on `stream` in order. */
int poem_set_overlap(poem_handle_t h, int enable);
/* Query-side row-tile chains (csrc/chain.hip), default ON for embed in {128, 256, 512} in fp32 mode: the Linears, residual
 * adds and LayerNorms between the attention kernels of a decoder block (pt_metro_transformer.py:56-91,34-40) run as four chain
 * launches per block with the activations resident in LDS; 0 = one launch per operator (the round-1 sequence; A/B and tests). */
int poem_set_chains(poem_handle_t h, int enable);
/* Scheduling switches by name, for A/B measurements (results are unaffected): "overlap", "anchor_tables", "chains" as the
 * setters above; "knn_early" (default 1): in chain mode the neighbour searches of block i+1 are issued right behind block i's
 * coordinate update instead of at the top of block i+1; "fused_sampling" (default 1, fp32 mode, embed in {128, 256, 512}):
 * F.grid_sample + the Q1 view + merge_features_mv / _sv (lib/models/heads/ptEmb_head.py:900-926,745-771) as the two kernels of
 * csrc/merge.hip -- the sampled tensor (sum N, C, S) and merge_net[0]'s hidden layer never reach HBM, the debug tap "g"
 * does not exist; 0 = the operator sequence (poem_project_sample, poem_gemm x4, poem_merge_reduce / _finalize), same
 * results to fp32 round-off (the cross-view dot products reduce in another order); "chain_combine" (default 1, chain mode,
 * 4 heads): the chain kernel behind a cross attention merges the attention's split-key partials while it fills its tile
 * instead of a separate combine launch writing the context rows (bit-identical); "xattn_merge" (default -1 = for batches of one or two samples; fp32 mode, head
 * dim 64, 4096 keys): the cross attention kernel merges its four split-key partials through LDS and writes the context rows
 * itself -- no partials in HBM, "chain_combine" then has nothing to do (bit-identical; at batch 32 0.5 % slower end to end,
 * at batch 1 / 2 2-3 % faster); "tables_first" (default 1): the fused
 * sampling kernel is ordered behind the block-0 anchor-table build of the neighbour-search stream (a CU that hosts a table
 * block takes one sampling block instead of two; results unaffected); "graphs" (default 1): replay the launch list of a
 * forward as a hipGraph, keyed by (batch size, workspace, options) -- NOT by the view layout; "graph_eager" (default 0): capture
 * at the first forward of a key instead of the second.
 * Round 3: "tables_cached" (default 1): read the block-0 anchor tables folded at poem_create (0 = rebuild them on every
 * forward: same kernel, same inputs, bit-identical); "chain_tile" (default 0 = per launch): row-tile height of the chain
 * kernels, 1 = 32 rows, 2 = 64 rows (csrc/chain.hip), 3 = 16-row units on v_mfma_f32_16x16x4_f32 (csrc/chain16.hip) -- any
 * choice gives the same bits; "knn_fma" (default 0): neighbour distances rounded as pytorch3d's CUDA kernel (poem_knn_ex below)
 * -- the one switch that is NOT round-off neutral by design.
 * "knn_query" (round 6; default 0 = poem_config_t.knn): N_NEIGHBOR_QUERY, 1..32 -- part of the model's configuration, not an A/B switch.
 * Round 4, all bit-identical: "gemm_xcd_map" (default 1; process-wide): panel GEMM blocks of one XCD own a row range and all its
 * column panels (csrc/gemm.hip); "f1_split" (default 1): the basis-point GEMM of blocks >= 1 as a 4C- and a 2C-column launch;
 * "gemm_kslab" (default 1; process-wide): K >= 512 Linears on the K-slab kernel; "bps_defer" (default 0): 1 / 2 / 3 = the
 * basis-point GEMM of block i+1 behind block i's first / second cross attention / its chain instead of up front; "va_p1"
 * (default -1 = small batches): one-query blocks of the full vector attention (0 never, 1 / 2 always with 3 / 2 waves per SIMD);
 * "xattn_merge" -1 / 0 / 1 as above; "xattn_tail" (round 6, default 1; process-wide; bit-identical): the remainder items of a
 * cross-attention launch -- 4 of a CU pair's 100 at the headline batch, a 13th item on half of the SIMDs -- run as channel-tile
 * halves on every SIMD (csrc/attn.hip xattn_half_item); "xattn_half" (default 1; process-wide): a single sample's merged cross attention on
 * 32-channel-tile items (twice the blocks, 48 instead of 64 MFMAs per key tile each); "small_batch" (default 3), a bit mask of launch-count / dependency shortcuts: 1 = one input
 * launch (coordinates + inverse extrinsics + projection table) and no query-embedding broadcast where block 0 runs on the anchor
 * tables, 2 = block 0's anchor keys / values read out of the rows its chain projects (batches of <= 5 samples).
 * Round 5, bit-identical: "group_min_views" (default 0): the samples whose view count divides 8 run the whole sampling stage in
 * ONE kernel (csrc/merge.hip sample_group_kernel: bilinear sampling, merge_net[0], the cross-view dot / weighted sum and
 * merge_net[1]; the hidden rows of merge_features_mv, ptEmb_head.py:745-762, never reach HBM) when the batch has at least this
 * many views -- 0 = enough to give every CU two of the kernel's 8-tile units, -1 = never (two-kernel form for every sample);
 * which samples go where is decided on the device from the view layout; "group_xcd" (default 1): that kernel's units in
 * XCD-aware order (a view's feature planes and projection table are fetched by one L2); "d2_first" (default 1): the
 * feed-forward chain of a block is issued in front of the next block's neighbour searches, so the launch graph keeps it on the
 * hardware queue of the chain before it (a launch that waits for another queue costs 5-13 us in a replayed graph) and the
 * searches take the side queue; "wait_merge" (default -1 = 3; bit mask): 1 = the first cross attention is issued in front of
 * the side stream's remaining launches (same reason), 2 = the main stream waits for the later blocks' basis-point GEMMs once,
 * where block 0's vector cross attention waits for its anchor rows anyway, 4 (lab) = it waits for a block's neighbour searches
 * at the block's first cross attention.  Scheduling only.
 * The switches marked process-wide are launcher statics: setting one on any handle sets it for the process (the launch-graph
 * key carries the process's values).  Unknown names and out-of-range values return POEM_E_ARG. */
int poem_set_option(poem_handle_t h, const char* name, int value);
/* Block-0 anchor tables of poem_head_forward (default on, fp32 mode).  In the first decoder block every query's
 * neighbours are the 32 fixed anchors (anchor_points, lib/models/bricks/point_transformers.py:10-32) and every sample's
 * query coordinates are the hand template (lib/models/heads/ptEmb_head.py:886-894,935: ((c + t) - c) / r, i.e. t / r up
 * to the rounding of c + t), so fc_delta's output and fc_gamma.0's positional term of both vector attentions
 * (point_transformers.py:88-91,144-147) are computed once per forward from t / r for the Q x 32 (query, anchor) pairs
 * and shared by all samples; the per-sample kernel runs one C x C GEMM per neighbour column instead of three.  Results
 * differ from the per-sample form by fp32 round-off of the inputs only and do not depend on the batch.  0 = per-sample
 * form everywhere (the exact arithmetic of the reference).  poem_decoder_forward never uses the tables. */
int poem_set_anchor_tables(poem_handle_t h, int enable);
/* Arithmetic of the three C x C per-neighbour GEMMs inside the fused vector attention (everything else is fp32 always):
 *   POEM_PRECISION_FP32 (default): v_mfma_f32_32x32x2_f32, exact fp32 products, k-ordered fma chains;
 *   POEM_PRECISION_SPLIT_F16X3 (opt-in, embed >= 128): hi/lo f16 splits of both operands on the f16 matrix cores
 *     (w_hi x_hi + w_hi x_lo + w_lo x_hi, fp32 accumulation; csrc/vecattn_split.hip) -- products carry ~22 significant
 *     bits instead of 24; measured MPVPE against the reference fixtures is unchanged at the 1e-5 mm level.  The
 *     ReLU outputs inside the attention MLPs saturate at 937.5 (f16 range after the x64 pre-scale).
 * Returns POEM_E_UNSUPPORTED when the handle has no split images (embed < 128). */
#define POEM_PRECISION_FP32 0
#define POEM_PRECISION_SPLIT_F16X3 1
/*   POEM_PRECISION_SPLIT_F16X3_ALL (opt-in): additionally every Linear that runs on the panel GEMM kernel (all of them
 *     except the K = 4C feed-forward output Linear and the narrow ones) uses the same hi/lo scheme -- weights split per 32-row
 *     tile at handle creation, activations scaled by 16, split in registers (saturating at |x| = 3750); and the two
 *     contractions of the cross attention for head dims 32 / 64 (from the same fp32 K/V fragment images, split in
 *     registers).  Softmaxes, LayerNorms, sampling, neighbour searches, the K = 4C Linear: exact fp32. */
#define POEM_PRECISION_SPLIT_F16X3_ALL 2
int poem_set_precision(poem_handle_t h, int mode);
/* Operator level of the split panel GEMM: image (ceil(n/32)*32 * k * 4 bytes) and scales (ceil(n/32) floats, device) from
 * poem_pack_split_gemm; y = act(x w^T + bias) + residual as poem_gemm.  POEM_E_UNSUPPORTED for shapes the panel kernel
 * does not take (k % 16, n % 32, a 32-column panel beyond 128 KiB of LDS). */
/* poem_cross_attention with the four split-key partials of a query tile merged inside the kernel (four waves of one block,
 * through LDS, chunk order: the bits of poem_cross_attention) -- what poem_head_forward runs under "xattn_merge" = 1.
 * POEM_E_UNSUPPORTED unless embed / heads == 64 and nk / 32 splits into four chunks of >= 8 key tiles (nk = 4096). */
int poem_cross_attention_merged(const float* q, const float* k, const float* v, float* ctx, int batch, int nq, int nk, int embed,
                                int heads, void* scratch, size_t scratch_bytes, void* stream);
/* poem_cross_attention with both contractions as hi/lo f16 splits (head dims 32 and 64; POEM_E_UNSUPPORTED otherwise);
 * same arguments, scratch and partial/combine structure as poem_cross_attention. */
int poem_cross_attention_split_f16x3(const float* q, const float* k, const float* v, float* ctx, int batch, int nq, int nk,
                                     int embed, int heads, void* scratch, size_t scratch_bytes, void* stream);
int poem_pack_split_gemm(const float* w, int out_features, int in_features, void* image, float* scales, void* stream);
int poem_gemm_split(const float* x, int ldx, const void* image, const float* scales, const float* bias, const float* residual,
                    int ldr, float* y, int ldy, int m, int n, int k, int act, void* stream);
int poem_finalize_parametric(poem_handle_t h, const float* mano_verts, const float* mano_joints,
                             const float* reference_joints, int batch, float* out_xyz, void* stream);
/* Timing of the dominant kernel (the fused vector attention) with HIP events recorded on the launch stream around
 * each of its launches inside poem_head_forward / poem_decoder_forward.  enable(h, n) allocates n event pairs
 * (0 disables); read() synchronises the recorded pairs and returns their count and summed duration. */
int poem_profile_enable(poem_handle_t h, int max_launches);
int poem_profile_read(poem_handle_t h, int* launches, float* total_ms, int reset);
/* The same for the anchored (table) launches of block 0 (poem_set_anchor_tables), which poem_profile_read leaves out:
 * call it before a resetting poem_profile_read. */
int poem_profile_read_anchored(poem_handle_t h, int* launches, float* total_ms);
/* Generic form: the spans poem_head_forward times between HIP events on the caller's stream, by kind.  (Read before a
 * resetting poem_profile_read.) */
#define POEM_PROF_VECATTN 0        /* the full fused vector attention (what poem_profile_read sums) */
#define POEM_PROF_VECATTN_ANCHORED 1
#define POEM_PROF_SAMPLING 2       /* input_proj .. merge finalize: the sampling front end of one forward */
int poem_profile_read_stage(poem_handle_t h, int stage, int* launches, float* total_ms);
/* Debug taps: copies of intermediate tensors of the LAST poem_head_forward on this handle (device->device).
 * name: "x","g","bps_feat","pt_xyz","query_xyz","b<i>.h_cross","b<i>.f_self","b<i>.f_cross","b<i>.xyz",
 * "b<i>.feats","b<i>.idx_self","b<i>.idx_cross".  Returns number of elements or <0. */
int64_t poem_tap(poem_handle_t h, const char* name, void* dst, int64_t dst_elems, void* stream);
int poem_enable_taps(poem_handle_t h, int enable);

/* ---- individual operators (same kernels the whole path launches) ----------------------------------------- */
size_t poem_packed_linear_bytes(int out_features, int in_features);
int poem_pack_linear(const float* w, int out_features, int in_features, void* packed, void* stream);
/* Y[M,N] = act(X[M,K] . W^T + bias) + residual ; ldx/ldr/ldy in floats; bias/residual may be NULL. */
int poem_gemm(const float* x, int ldx, const void* w_packed, const float* bias, const float* residual, int ldr,
              float* y, int ldy, int M, int N, int K, int act, void* stream);
/* Layout-aware GEMM.  Layouts: POEM_LAYOUT_RM (row-major, ld* in floats) or POEM_LAYOUT_PA ("packed activation":
 * the fragment order of poem_pack_linear applied to the activation rows; buffers hold ceil(M/32)*32 rows; the
 * residual shares the output layout).  PA->PA chains are how the whole path runs its Linears (coalesced 1 KiB
 * operand loads and result stores). */
#define POEM_LAYOUT_RM 0
#define POEM_LAYOUT_PA 1
int poem_gemm_ex(const float* x, int ldx, const void* w_packed, const float* bias, const float* residual, int ldr,
                 float* y, int ldy, int M, int N, int K, int act, int in_layout, int out_layout, void* stream);
/* row-major (rows, cols) <-> PA; poem_packed_linear_bytes(rows, cols) gives the PA size. */
int poem_pack_rows(const float* x, int rows, int cols, void* packed, void* stream);
int poem_unpack_rows(const void* packed, int rows, int cols, float* x, void* stream);
int poem_layernorm(const float* x, const float* gamma, const float* beta, float* y, int rows, int cols, float eps,
                   void* stream);
/* table: (sum_{N=1..max_views} N, C, H*W) */
int poem_pe_table(const void* adapt_w_packed, const float* adapt_b, int embed, int h, int w, int max_views,
                  float* scratch_sine, float* table, void* stream);
/* the same with SinePositionalEncoding3D's `normalize` argument (petr_transformer.py:451-457): 0 = plain cumulative counts */
int poem_pe_table_ex(const void* adapt_w_packed, const float* adapt_b, int embed, int h, int w, int max_views, int normalize,
                     float* scratch_sine, float* table, void* stream);
/* The input of position_encoder (BasePointEmbedHead.position_embeding, lib/models/heads/ptEmb_head.py:113-181;
 * inverse_sigmoid lib/utils/transform.py:1145-1161): out (views, 3 * depth_num, feat_h, feat_w), channel 3 d + axis = the
 * inverse sigmoid of the position_range-normalised master-frame coordinate of pixel (x, y)'s frustum point at depth d.
 * img0, img1 = img_metas["inp_img_shape"][0], [1].  Reads cfg's feat_h, feat_w, depth_num, lid, depth_*, position_range. */
int poem_frustum_features(const poem_config_t* cfg, const float* cam_intr, const float* cam_extr, int views, int img0, int img1,
                          float* out, void* stream);
/* x[v,c,p] = W[c,:] . feat[v,:,p] + b[c] + table[pe_index[v], c, p] */
int poem_input_proj(const float* feat, const void* w_packed, const float* bias, const float* table,
                    const int32_t* pe_index, float* x, int views, int in_channels, int embed, int hw, void* stream);
/* g[v,c,s] = bilinear sample of x[v,c] at the projection of (bps[s] + centre[view_sample[v]]) into view v.
 * uv_scratch: (views, S, 2) floats (pixel-space sample coordinates ix, iy). */
int poem_project_sample(const float* x, const float* bps, const float* centre, const int32_t* view_sample,
                        const float* cam_intr, const float* cam_extr, float* uv_scratch, float* g, int views,
                        int embed, int fh, int fw, int nsample, int img_w, int img_h, void* stream);
int poem_merge_reduce(const float* h2, const int32_t* view_offsets, float* m, int batch, int nsample, int half,
                      void* stream);
int poem_merge_finalize(const float* g, const float* y, const int32_t* view_offsets, float* out, int batch,
                        int nsample, int embed, void* stream);
/* q (B,Q,C) k,v (B,S,C) -> ctx (B,Q,C); softmax(q k^T / sqrt(C/heads)) v per head  (S % 32 == 0, C % 32 == 0,
 * C/heads in {8,16,32,64,128,256}).  k and v are first re-laid into MFMA fragment images, then the key axis is processed
 * in fixed chunks whose partial (O, m, l) triples are merged in fixed order (a sample's result does not depend on the
 * batch it travels in); images and partials live in `scratch` (poem_cross_attention_scratch_bytes() bytes, 16-byte
 * aligned).  Inside the decoder the projection GEMM writes the images itself. */
size_t poem_cross_attention_scratch_bytes(int batch, int nq, int nk, int embed, int heads);
int poem_cross_attention(const float* q, const float* k, const float* v, float* ctx, int batch, int nq, int nk,
                         int embed, int heads, void* scratch, size_t scratch_bytes, void* stream);
/* Ragged batched DLT triangulation -- replaces lib/utils/triangulation.py:5-45 (batch_triangulate_dlt_torch) and the
 * per-sample loop of lib/models/POEM.py:284-299: uv (BN,J,2) pixels, cam_intr (BN,3,3), cam_mat (BN,4,4),
 * view_offsets (B+1) DEVICE int32 prefix sums of views per sample -> out_xyz (B,J,3) in the master frame.
 * invert=1: cam_mat is the batch's cam_extr (camera->master) and is inverted here (POEM.py:286);
 * invert=0: cam_mat is already master->camera (the `Extrs` argument of the reference function). */
int poem_triangulate_dlt(const float* uv, const float* cam_intr, const float* cam_mat, const int32_t* view_offsets,
                         int batch, int njoints, int invert, float* out_xyz, void* stream);
/* Heat-map read-out in front of the triangulation (tail of heatmap_stage, lib/models/POEM.py:213-222, with
 * integral_heatmap2d, lib/models/integal_pose.py:194-218): heatmaps (BN,J,Hh,Wh) non-negative (sigmoid outputs) ->
 * uv (BN,J,2) in image pixels: normalise by (sum + 1e-6), expectation of (x/Wh, y/Hh), scale by (img_w, img_h). */
int poem_heatmap_uv(const float* heatmaps, float* uv, int views, int njoints, int hm_h, int hm_w, float img_w, float img_h,
                    void* stream);
/* Convolutional glue between the backbone's multi-level features and the head -- replaces
 * PtEmbedMultiviewStereoV2.feat_decode / uv_decode (lib/models/POEM.py:167-211, HRNet branch) built from
 * lib/models/bricks/conv.py ConvBlocks.  The Python mirror (poem_v2_amd/decode.py) chains them exactly as the
 * reference methods do.
 * poem_pack_conv3x3: OIHW (cout,cin,3,3) weights -> MFMA fragment image (poem_conv3x3_packed_bytes bytes); cin % 8 == 0.
 * poem_conv3x3: in (views,cin,h+2,w+2) with a zero border; y = conv(in)*scale[c] + shift[c] (conv bias and eval-mode
 *   BatchNorm folded by the caller; arrays padded to a multiple of 32 channels), optional ReLU, optional lateral add
 *   residual (views,cout,h/stride,w/stride) AFTER the activation (POEM.py:185-186); element (n,c,y,x) is written to
 *   out[n*out_view_stride + c*out_ch_stride + y*out_row_stride + x + out_offset] so that the result can land inside
 *   the next conv's zero-bordered input.  stride 1|2 (padding 1), (h/stride)*(w/stride) % 32 == 0.
 * poem_upsample2_concat_pad: out (views, ca+cb, h+2*pad, w+2*pad) = [bilinear x2 (align_corners=False) of
 *   a (views,ca,h/2,w/2) | b (views,cb,h,w)] with a zero border of pad (0|1) pixels (F.interpolate + torch.cat,
 *   POEM.py:189,203-204); ca or cb may be 0.
 * poem_pool_conv1x1_sigmoid: x (views,c,h,w) -> max_pool2d(2,2) -> 1x1 conv (j x c, bias) -> sigmoid ->
 *   heatmaps (views,j,h/2,w/2)  (POEM.py:206-207); c <= 64, j <= 32. */
size_t poem_conv3x3_packed_bytes(int cout, int cin);
int poem_pack_conv3x3(const float* w_oihw, int cout, int cin, void* packed, void* stream);
int poem_conv3x3(const float* in_padded, const void* w_packed, const float* scale, const float* shift,
                 const float* residual, float* out, int views, int cin, int cout, int h, int w, int stride, int relu,
                 int64_t out_view_stride, int out_ch_stride, int out_row_stride, int out_offset, void* stream);
int poem_upsample2_concat_pad(const float* a, int ca, const float* b, int cb, float* out, int views, int h, int w, int pad,
                              void* stream);
/* A stride-2 ConvBlock of feat_decode (POEM.py:183-189: padding 1, stride 2) straight from the UNBORDERED input
 * (views,cin,h,w): LDS-staged, 16-channel matrix tiles; out / residual addressing as poem_conv3x3.  POEM_E_UNSUPPORTED for
 * shapes other than HRNet-W40's three (cout, h = w) = (80, 64), (160, 32), (320, 16): use poem_conv3x3(stride 2) on a
 * zero-bordered copy then. */
int poem_conv3x3_down2(const float* in, const void* w_packed, const float* scale, const float* shift, const float* residual,
                       float* out, int views, int cin, int cout, int h, int w, int relu, int64_t out_view_stride,
                       int out_ch_stride, int out_row_stride, int out_offset, void* stream);
/* Process-wide A/B switches of the decode operators (not per handle: these operators take no handle).  Results do not depend on
 * them.  "s2_staging_wave" (default 1): poem_conv3x3_down2 as persistent blocks of four MFMA waves plus a fifth wave that does
 * all the LDS-DMA staging; 0 = the round-3 kernel in which every wave stages and multiplies; 2..4 = the default kernel with
 * that many blocks per CU instead of the rule in decode.hip (measurement only; 0 / 1 restore the rule).
 * "row_stager" (default 3; bit 0: at w = 64, bit 1: at w = 32): poem_upcat_conv3x3 stages its fused input by rows (wave =
 * channel, lane = (row group, column), the source rows of a chunk loaded once) and runs two blocks per CU; 0 = the per-float
 * stager of the other widths.
 * "pin32" (default 1): the fused-input convolution with five 32-channel tiles (uv_decode's first, 480 -> 160) runs its taps as a
 * pinned two-stage pipeline (next tap's weights and operands requested before this tap's MFMAs); 0 = compiler-scheduled taps.
 * "pool_fused" (default 1): poem_upcat_conv3x3_pool_head takes the shapes it supports; 0 = it returns POEM_E_UNSUPPORTED (A/B of
 * the two-launch form).
 * POEM_E_ARG for an unknown name or an out-of-range value. */
int poem_set_decode_option(const char* name, int value);
/* feat_decode's tail in one launch (POEM.py:190-193: F.interpolate(x, scale_factor=2, mode="bilinear") followed by feat_in, a
 * 1x1 convolution with bias): in (views,cin,h,w) -> out (views,cout,2h,2w).  The convolution is applied BEFORE the
 * interpolation (they commute: both linear, the interpolation weights sum to one) on the matrix cores with the input and the
 * low-resolution result in LDS.  w_packed: poem_pack_linear image of the (cout, cin) weight.  POEM_E_UNSUPPORTED unless
 * h = w = 8, cin % 8 == 0, cout % 32 == 0, (cin + cout) * 256 B <= 160 KB: use poem_input_proj + poem_upsample2_concat_pad then. */
int poem_conv1x1_upsample2(const float* in, const void* w_packed, const float* bias, float* out, int views, int cin, int cout,
                           int h, int w, void* stream);
/* One uv_decode stage in one launch (POEM.py:203-205: F.interpolate x2, torch.cat, ConvBlock): stride-1 conv3x3 of
 * [bilinear x2 of a_half (views,ca,h/2,w/2) | b_full (views,cb,h,w)] with the concatenation, the zero border and the
 * upsampling applied while the input halo is staged in LDS -- the concatenated tensor never exists.  Same epilogue and output
 * addressing as poem_conv3x3.  POEM_E_UNSUPPORTED for shapes the LDS-staged kernel does not take (cout > 160, w > 64,
 * 256 % w != 0, ...): call poem_upsample2_concat_pad + poem_conv3x3 then. */
int poem_upcat_conv3x3(const float* a_half, int ca, const float* b_full, int cb, const void* w_packed, const float* scale,
                       const float* shift, float* out, int views, int cout, int h, int w, int relu, int64_t out_view_stride,
                       int out_ch_stride, int out_row_stride, int out_offset, void* stream);
int poem_pool_conv1x1_sigmoid(const float* x, const float* w, const float* bias, float* heatmaps, int views, int c, int j,
                              int h, int w_, void* stream);
/* uv_decode's last stage AND the read-out head in one launch (POEM.py:203-207 upstream: F.interpolate x2, torch.cat, ConvBlock,
 * then max_pool2d(2, 2) + uv_out + sigmoid): poem_upcat_conv3x3 whose epilogue pools its own rows, contracts them with
 * head_w (j x cout, row-major) + head_b and writes sigmoid(.) to heatmaps (views, j, h/2, w/2) -- the convolution's
 * (views, cout, h, w) output, which nothing else reads, never reaches HBM.  Same bits as poem_upcat_conv3x3 +
 * poem_pool_conv1x1_sigmoid.  POEM_E_UNSUPPORTED unless w == 64, 33 <= cout <= 48, j <= 32 and ca, cb are multiples of 8:
 * call the two operators then. */
int poem_upcat_conv3x3_pool_head(const float* a_half, int ca, const float* b_full, int cb, const void* w_packed, const float* scale,
                                 const float* shift, const float* head_w, const float* head_b, float* heatmaps, int views, int cout,
                                 int j, int h, int w, int relu, void* stream);
/* Device-side evaluation metrics (replace the host loops of lib/metrics/pa_eval.py:45-83,104-124 and
 * lib/metrics/pck.py:36-96).
 * poem_pa_epe: pred, gt (B,P,3) -> out (B,2) = per-sample (Procrustes-aligned mean distance, plain mean distance);
 *   alignment = PAEval.align_w_scale (scipy orthogonal_procrustes with scale, no reflection handling).
 * poem_pck_accumulate: adds this batch to counts (P,steps) [#dist <= linspace(val_min,val_max,steps)[t]],
 *   dist_sum (P) fp64 and n (P); the caller zeroes them at reset and derives pck / auc / epe from them;
 *   dist_out (B,P) optionally receives this batch's distances (NULL to skip). */
int poem_pa_epe(const float* pred, const float* gt, float* out, int batch, int npoints, void* stream);
int poem_pck_accumulate(const float* pred, const float* gt, int batch, int npoints, double val_min, double val_max,
                        int steps, uint32_t* counts, double* dist_sum, uint32_t* n, float* dist_out, void* stream);
/* mano_to_openpose (lib/utils/transform.py:836-872; called on predicted and ground-truth vertices by testing_step,
 * lib/models/POEM.py:602-603): j_regressor (16,nverts) MANO's th_J_regressor, verts (B,nverts,3) -> joints (B,21,3) in
 * OpenPose order (16 regressed joints + the 5 finger-tip vertices, re-ordered).  nverts must be 778. */
int poem_mano_to_openpose(const float* j_regressor, const float* verts, float* joints, int batch, int nverts, void* stream);
/* MANO linear blend skinning: the `ManoLayer(pose_aa, betas)` call of the medium_MANO tail
 * (lib/models/bricks/pt_metro_transformer.py:120-124,147-148: manotorch ManoLayer(joint_rot_mode="axisang", use_pca=False,
 * flat_hand_mean=True, center_idx=9)) and the head's zero-pose template (lib/models/heads/ptEmb_head.py:732-736,886-892).
 * pose_aa (B,48) axis-angle of the 16 joints, betas (B,10); assets as device buffers: v_template (778,3),
 * shapedirs (778,3,10), posedirs (778,3,135), j_regressor (16,778), weights (778,16)  [MANO_RIGHT.pkl fields; licence-gated,
 * never read from disk here].  -> verts (B,778,3), joints (B,21,3) in the 21-joint hand order (16 skeleton joints + the
 * finger-tip vertices 745,317,444,556,673), both minus joint center_idx (-1: not centred).  manotorch is absent from the
 * reference tree: restated from the published model in manopth / manotorch's evaluation order -- parity unpinned. */
/* The rotation half of get_parametric_output (pt_metro_transformer.py:144-146 -> rot6d_to_aa, lib/utils/transform.py:448-466:
 * pytorch3d rotation_6d_to_matrix -> matrix_to_quaternion -> quaternion_to_axis_angle): params (B,106) = 16 six-dimensional
 * rotations then 10 betas -> pose_aa (B,48), betas (B,10). */
int poem_rot6d_to_axis_angle(const float* params, float* pose_aa, float* betas, int batch, void* stream);
/* Round 6: the five asset arrays are re-laid once into a TABLE (poem_mano_table_bytes() bytes of device memory, 16-byte
 * aligned; poem_mano_prepare at layer creation -- coefficient-major blend shapes, joint regression composed with template and
 * shape basis in fp64, joint-major skinning weights: csrc/mano.hip) and every call takes that table: one launch over
 * (13 vertex tiles x batch) blocks instead of one block per sample. */
size_t poem_mano_table_bytes(void);
int poem_mano_prepare(const float* v_template, const float* shapedirs, const float* posedirs, const float* j_regressor,
                      const float* weights, void* table, void* stream);
int poem_mano_lbs(const float* pose_aa, const float* betas, const void* table, int batch, int center_idx, float* verts,
                  float* joints, void* stream);
/* The MANO layer INSIDE the forward of a parametric handle (get_parametric_output, pt_metro_transformer.py:139-151, and the last
 * layer of ptEmb_head.py:953-958): with a table attached, poem_head_forward runs Q3 -> Linears -> rot6d -> poem_mano_lbs in its
 * captured launch graph and writes the last layer of out_xyz as nan_to_num(joints | verts) + centre itself;
 * poem_decoder_forward replaces the last layer's rows by (joints | verts).  pose_aa / betas are returned as before.  The table is
 * caller-owned and must stay valid while attached; table = NULL detaches (the caller then runs its own layer and
 * poem_finalize_parametric).  POEM_E_UNSUPPORTED on a handle without PARAMETRIC_OUTPUT. */
int poem_attach_mano(poem_handle_t h, const void* table, int center_idx);
/* Crop / warp / normalise every view of a batch in one launch (replaces the per-view host chain of
 * lib/utils/transform.py:153-170: cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 0) -> colour jitter -> to_tensor ->
 * normalize(0.5, 1)).  src: the raw uint8 HxWx3 images back to back in one device blob; src_offsets (views) byte offset
 * of each image; src_hw (views,2) = (height, width); m_inv (views,6) fp64 = the INVERSE (destination -> source) 2x3 map,
 * i.e. what cv::warpAffine derives from the matrix it is given; gain (views,3) fp64 per-channel colour gains or NULL.
 * Outputs (either may be NULL, not both): out_f32 (views,3,out_h,out_w) = p / 255 - 0.5; out_u8 (views,out_h,out_w,3).
 * Arithmetic is OpenCV's 8-bit fixed-point bilinear path (1/32-pixel coordinates, 15-bit weights): integer, bit-exact. */
int poem_warp_affine(const uint8_t* src, const int64_t* src_offsets, const int32_t* src_hw, const double* m_inv,
                     const double* gain, float* out_f32, uint8_t* out_u8, int views, int out_h, int out_w, void* stream);
/* Operator-level entry points of the opt-in split-precision vector attention (see poem_set_precision).
 * poem_pack_split_linear: w (embed,embed) row-major fp32 -> image (embed*embed*4 bytes: hi | lo f16 fragments of w * scale)
 *   and *scale (device float, the power of two the image carries).
 * poem_vector_attention_split: the COMPOSED form of poem_vector_attention -- qg = W_g1 q + (W_g1 b_d2 + b_g1),
 *   kg = W_g1 k (row stride embed), images of W_d2, W_g1 W_d2 and W_g2, scales = their three scales (device). */
int poem_pack_split_linear(const float* w, int embed, void* image, float* scale, void* stream);
int poem_vector_attention_split(const float* query_xyz, const float* src_xyz, const float* anchor_xyz, const int32_t* idx,
                                int shared_idx, const float* qg, const float* kg, const float* v, int nsrc, const float* wd1,
                                const float* bd1, const void* wd2_image, const float* bd2, const void* wg1d2_image,
                                const void* wg2_image, const float* scales, float* out, int batch, int nq, int embed,
                                void* stream);
/* idx (B,Q,32) int32: 32 nearest src points per query, ascending squared L2, ties -> lower index. */
int poem_knn(const float* query_xyz, const float* src_xyz, int32_t* idx, int batch, int nq, int nsrc, void* stream);
/* The same with a choice of the distance's rounding.  fma_contract = 0: ((dx*dx + dy*dy) + dz*dz), every operation rounded --
 * pytorch3d's CPU kernel (knn_cpu.cpp, built without FMA), the path BASELINE's parity bar is stated against, and what
 * poem_knn / poem_head_forward use.  fma_contract = 1: fma(dz, dz, fma(dy, dy, dx*dx)) -- pytorch3d's CUDA kernel
 * (knn.cu `dist += diff * diff` under nvcc's default -fmad=true), for comparing with results produced on a CUDA box;
 * whole path: poem_set_option(h, "knn_fma", 1).  The two differ by <= 1 ulp per distance and only reorder near-ties. */
int poem_knn_ex(const float* query_xyz, const float* src_xyz, int32_t* idx, int batch, int nq, int nsrc, int fma_contract,
                void* stream);
/* Vector attention core.  q (B,Q,C); k,v (B,NS,C) gathered by idx; idx (B,Q,32) or (32) when shared_idx!=0;
 * neighbour coordinates: anchor_xyz (32,3) when non-NULL, else src_xyz[b, idx];
 * wd1 (C,3)+bd1 raw; wd2, wg1, wg2 packed (C,C) + biases.  out r (B,Q,C) = sum_j softmax_j(a/sqrt(C)) * (v_j + pos_j). */
int poem_vector_attention(const float* query_xyz, const float* src_xyz, const float* anchor_xyz, const int32_t* idx,
                          int shared_idx, const float* q, const float* k, const float* v, int nsrc,
                          const float* wd1, const float* bd1, const void* wd2_packed, const float* bd2,
                          const void* wg1_packed, const float* bg1, const void* wg2_packed, const float* bg2,
                          float* out, int batch, int nq, int embed, void* stream);
/* xyz_out = xyz_in + r . W^T + b   (W (3,C) raw) */
int poem_reg_update(const float* r, const float* w, const float* b, const float* xyz_in, float* xyz_out, int rows,
                    int embed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POEM_HIP_H */

// poem_head_forward (POEM_Generalized_Head.forward, lib/models/heads/ptEmb_head.py:825-964 upstream) and poem_decoder_forward
// (PtEmbedTRv4.forward, lib/models/layers/ptEmb_transformer.py:371-376).  One forward =
//
//   view_layout()   per-view index arrays (CSR of the ragged batch) into handle-owned device memory, when they changed: built
//                   by a kernel from offsets carried in its arguments -- the host never waits for the previous forward
//   inputs()        the four kernels that read the CALLER's tensors: input_proj (+ folded positional table), normalised
//                   coordinates, inverted extrinsics, projection table                                   [:835-883 upstream]
//   body()          sampling + Q1 + merge MLPs -> bps_feat ; query features ; the three decoder blocks (decoder.cpp)
//                   -- workspace / handle memory only, so it replays as a hipGraph                       [:900-942]
//   finalize        nan_to_num, x radius + centre into the caller's output                                [:944-958]
#include "engine.h"

namespace {

// SPLIT_F16X3_ALL: every panel GEMM enqueued by the call whose weight lies in the handle's packed arena takes the split image at
// the same offset (gemm.hip).  Thread-local host state; only a call that installed it clears it (an fp32 forward never touches it).
struct SplitContext {
  bool set = false;
  explicit SplitContext(poem_handle_t h) {
    if (h->precision != POEM_PRECISION_SPLIT_F16X3_ALL) return;
    set = true;
    poem_gemm_split_context(h->packed_base, h->packed_size, h->gemm_split, h->gemm_scales);
    poem_cross_attention_split(1);
    const int dh = h->cfg.embed / h->cfg.heads;
    poem_gemm_split_images(dh == 32 || dh == 64);          // the head dims the split cross attention takes
  }
  ~SplitContext() {
    if (!set) return;
    poem_gemm_split_context(nullptr, 0, nullptr, nullptr);
    poem_cross_attention_split(0);
    poem_gemm_split_images(0);
  }
};

struct HeadRun {
  poem_handle_t h;
  const poem_config_t& c;
  Plan p;
  const int plan_views;
  const float *mlvl_feat, *cam_intr, *cam_extr, *reference_joints;
  const int32_t* offs_host;
  const int B, BN, C, S, Q, HW, BS, img_w, img_h;
  hipStream_t s;
  bool fused_fe = false;     // merge.hip front end (else the operator sequence of sample.hip + gemm.hip)
  bool prof_fe = false;      // HIP-event pair around the sampling stage
  int prof_slot = 0;

  // plan_views: the view capacity the workspace plan (and every captured launch) is laid out for: batch * max_views when the
  // caller's workspace holds that plan -- one graph per batch size, whatever the view layout -- else this batch's own total.
  HeadRun(poem_handle_t h_, const float* feat, const float* intr, const float* extr, const int32_t* offs, int batch, int plan_views_,
          const float* ref_joints, int w, int hgt, void* workspace, hipStream_t s_)
      : h(h_), c(h_->cfg), p(make_plan(h_->cfg, batch, plan_views_, workspace)), plan_views(plan_views_), mlvl_feat(feat), cam_intr(intr), cam_extr(extr),
        reference_joints(ref_joints), offs_host(offs), B(batch), BN(offs[batch]), C(c.embed), S(c.nsample), Q(c.nquery),
        HW(c.feat_h * c.feat_w), BS(batch * c.nsample), img_w(w), img_h(hgt), s(s_) {}

  // ---- view_offsets | view_sample[v] = sample of view v | pe_index[v] = its slot in the folded positional table.  Kept in
  // handle-owned device memory and rebuilt only when the layout changes, by view_layout_kernel from offsets that ride in the
  // kernel-argument segment: stream-ordered behind the previous forward's readers, and the host returns at once.  (A pageable
  // H2D copy -- rounds 1-3 -- blocks the host until the stream reaches it, i.e. until the PREVIOUS forward has finished: with a
  // fresh layout per batch, which is what the reference's collation produces (lib/utils/collation.py:7-25,
  // lib/data_wds/multiview_wds.py:86-95 upstream), the host could never run ahead of the GPU.)
  int view_layout() {
    for (int b = 0; b < B; ++b) {
      const int n = offs_host[b + 1] - offs_host[b];
      if (n < 1 || n > c.max_views) return POEM_E_ARG;
    }
    // (array positions are functions of the batch size and the plan's capacity only: a captured launch holds these pointers)
    const size_t cap = (size_t)std::max(plan_views, BN);
    const size_t o1 = align_up((size_t)B + 1, 64), o2 = o1 + align_up(cap, 64), tot = o2 + align_up(cap, 64);
    const bool owned = h->idx_dev && tot <= (size_t)poem_handle_s::IDX_CAP;
    if (owned) {
      p.offs = h->idx_dev;
      p.view_sample = h->idx_dev + o1;
      p.pe_index = h->idx_dev + o2;
      if (h->idx_host.size() == (size_t)B + 2 && h->idx_host[B + 1] == (int32_t)cap && std::equal(offs_host, offs_host + B + 1, h->idx_host.begin()))
        return POEM_OK;
    }
    h->idx_host.clear();       // (a failed upload leaves no signature behind)
    if (B <= POEM_LAYOUT_MAX_BATCH && BN <= 65535) {
      ViewLayoutArgs a;
      a.offs = p.offs; a.view_sample = p.view_sample; a.pe_index = p.pe_index; a.B = B;
      for (int b = 0; b <= B; ++b) a.off16[b] = (unsigned short)offs_host[b];
      HIPCHK(poem_launch_view_layout(&a, s));
      ++h->layout_uploads;
    } else {     // (the host vectors die at return: pageable H2D copies are staged before hipMemcpyAsync returns)
      std::vector<int32_t> vs(BN), pei(BN);
      for (int b = 0; b < B; ++b)
        for (int k = 0, n = offs_host[b + 1] - offs_host[b]; k < n; ++k) {
          vs[offs_host[b] + k] = b;
          pei[offs_host[b] + k] = n * (n - 1) / 2 + k;
        }
      HIPCHK(hipMemcpyAsync(p.offs, offs_host, (B + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(p.view_sample, vs.data(), BN * sizeof(int32_t), hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(p.pe_index, pei.data(), BN * sizeof(int32_t), hipMemcpyHostToDevice, s));
      ++h->layout_uploads;
    }
    if (owned) {      // (workspace arrays are the caller's scratch: never trusted to persist)
      h->idx_host.assign(offs_host, offs_host + B + 1);
      h->idx_host.push_back((int32_t)cap);
    }
    return POEM_OK;
  }

  // ---- block-0 anchor tables: the handle's (folded at poem_create) or, tables_cached = 0, rebuilt on the side stream
  int anchor_tables() {
    h->tables_pending = false;
    if (!h->anchor_tables || h->precision != POEM_PRECISION_FP32) return POEM_OK;
    if (h->tables_cached && h->tab_mem) {
      p.canon_xyz = h->c_canon_xyz;
      for (int k = 0; k < 2; ++k) { p.tab_g[k] = h->c_tab_g[k]; p.tab_p[k] = h->c_tab_p[k]; }
      return POEM_OK;
    }
    return build_anchor_tables(h, p, s);
  }

  // ---- the kernels that read the caller's tensors
  int inputs() {
    prof_fe = h->prof_on && (size_t)(2 * h->prof_used + 1) < h->prof_ev.size();
    prof_slot = h->prof_used;
    if (prof_fe) { HIPCHK(hipEventRecord(h->prof_ev[2 * prof_slot], s)); h->prof_kind[h->prof_used++] = POEM_PROF_SAMPLING; }
    fused_fe = h->fused_sampling && h->precision == POEM_PRECISION_FP32 && poem_sample_merge_supported(C, S, HW) != 0;
    const float* table = h->pe_table;
    const int32_t* table_index = p.pe_index;
    if (c.petr_embedding) {
      // PETR_EMBEDDING (ptEmb_head.py:865-867): posi_embed = adapt_pos3d(sine) + position_encoder(frustum features of the
      // batch's cameras) -- a per-VIEW table built here from the caller's cameras; input_proj then adds it as it adds the folded one
      const int ps = petr_slot(c), K0 = 3 * c.depth_num;
      HIPCHK(poem_launch_frustum_features(cam_intr, cam_extr, p.petr_f, BN, c.feat_h, c.feat_w, c.depth_num, c.lid != 0, c.depth_start,
                                          c.depth_end, c.position_range, img_w, img_h, s));      // (img_w, img_h) = inp_img_shape[0], [1]
      HIPCHK(poem_launch_conv1x1_ex(p.petr_f, h->P(ps), h->R(ps + 1), nullptr, nullptr, p.petr_h, nullptr, BN, K0, 2 * C, HW, 1, s));
      HIPCHK(poem_launch_conv1x1_ex(p.petr_h, h->P(ps + 2), h->R(ps + 3), h->pe_table, p.pe_index, p.petr_tab, nullptr, BN, 2 * C, C, HW, 0, s));
      table = p.petr_tab;
      table_index = nullptr;      // row v of the table is view v's
    }
    HIPCHK(poem_launch_conv1x1(mlvl_feat, h->P(T_INPROJ_W), h->R(T_INPROJ_B), table, table_index,
                               (fused_fe && !h->taps) ? nullptr : p.x, fused_fe ? p.xt : nullptr, BN, c.in_channels, C, HW, s));
    if (fused_fe && (h->small_batch & 1)) {      // coordinates, inverted extrinsics and the projection table in one launch (merge.hip)
      HIPCHK(poem_launch_input_tables(h->bps, reference_joints, h->tmpl, p.view_sample, cam_intr, cam_extr, p.ptab, p.ptab + (size_t)plan_views * S * 4, p.centre, p.pt_xyz,
                                      p.xyz[0], BN, B, S, Q, c.feat_h, c.feat_w, img_w, img_h, c.radius, s));
      return POEM_OK;
    }
    HIPCHK(poem_launch_prep_xyz(reference_joints, h->bps, h->tmpl, p.centre, p.pt_xyz, p.xyz[0], B, S, Q, c.radius, s));
    if (fused_fe) {
      float* inv = p.uv + (size_t)BN * S * 2;
      HIPCHK(poem_launch_invert_extr(cam_extr, inv, BN, s));
      HIPCHK(poem_launch_project_table(h->bps, p.centre, p.view_sample, cam_intr, inv, p.ptab, p.ptab + (size_t)plan_views * S * 4, nullptr, BN, C, c.feat_h, c.feat_w, S,
                                       img_w, img_h, s));
    } else {      // operator front end: the projection (the caller's cameras) here, the sampling inside the body
      HIPCHK(poem_launch_project_uv(h->bps, p.centre, p.view_sample, cam_intr, cam_extr, p.uv + (size_t)BN * S * 2, p.uv, BN, c.feat_h,
                                    c.feat_w, S, img_w, img_h, s));
    }
    return POEM_OK;
  }

  // ---- F.grid_sample + the Q1 view + merge_features_mv / _sv -> bps_feat
  int sampling(hipStream_t st) {
    if (fused_fe) {
      SampleMergeArgs sm{};
      sm.xt = p.xt; sm.tab = (const float4*)p.ptab; sm.tabo = (const uint2*)(p.ptab + (size_t)plan_views * S * 4);
      sm.view_sample = p.view_sample; sm.offs = p.offs;
      sm.w0 = (const float4*)h->P(T_M00_W); sm.b0 = h->R(T_M00_B); sm.w1 = (const float4*)h->P(T_M02_W); sm.b1 = h->R(T_M02_B);
      sm.h2 = p.h2; sm.q1 = p.q1; sm.S = S; sm.hw = HW; sm.h2_tiled = 1;
      sm.views = plan_views; sm.B = B; sm.views_dev = p.offs + B;      // the grid is sized for the plan, the kernel reads the batch's own count
      // The samples whose view count divides 8 go through sample_group_kernel (the whole stage in one kernel: merge_net[0]'s
      // hidden rows never reach HBM) when the batch has enough views to fill the chip with its 8-tile units; the two-kernel
      // form takes the rest.  Which samples go where is decided on the device from the layout (the launch graph is keyed by the
      // batch size only); the two forms give the same bits.
      const bool group_ok = h->group_min_views >= 0 && (S / C) % 8 == 0;
      sm.group_min_views = group_ok ? (h->group_min_views > 0 ? h->group_min_views : 2 * poem_device_cu_count() / ((C / (C == 512 ? 32 : 64)) * (S / C / 8)))
                                    : 0x7fffffff;
      // (per-forward table build only) The build on the neighbour-search stream holds 68 KB of LDS per block, and next to the
      // MFMA-dense sample_merge waves its blocks linger: a CU that hosts one takes a single sample_merge block (2 x 66.5 KB no
      // longer fit) and the persistent grid runs in two rounds.  sample_merge waits for the build (+0.06 ms on the critical path).
      if (h->tables_first && h->tables_pending) HIPCHK(hipStreamWaitEvent(st, h->ev_tab, 0));
      MergeTailArgs mt{};
      mt.h2 = p.h2; mt.q1 = p.q1; mt.offs = p.offs;
      mt.w0 = (const float4*)h->P(T_M10_W); mt.b0 = h->R(T_M10_B); mt.w1 = (const float4*)h->P(T_M12_W); mt.b1 = h->R(T_M12_B);
      mt.out = p.bps_feat; mt.B = B; mt.S = S; mt.h2_tiled = 1;
      mt.group_min_views = sm.group_min_views; mt.views_dev = sm.views_dev; mt.views = sm.views;
      sm.xcd_order = h->group_xcd;
      // (a batch size whose view capacity stays below the threshold can never take the grouped kernel: no launch at all --
      //  its early exit still costs ~6 us in front of a small batch's forward; the graph is keyed by the batch size, so is this)
      if (group_ok && plan_views >= sm.group_min_views) {
        SampleGroupArgs sg{};
        sg.sm = sm; sg.w2 = mt.w0; sg.b2 = mt.b0; sg.w3 = mt.w1; sg.b3 = mt.b1; sg.out = p.bps_feat;
        HIPCHK(poem_launch_sample_group(&sg, C, st));
      }
      HIPCHK(poem_launch_sample_merge(&sm, C, st));
      HIPCHK(poem_launch_merge_tail(&mt, C, st));
      return POEM_OK;
    }
    auto gemm = [&](const float* X, int ldx, int wi, int bi, float* Y, int ldy, int M, int N, int K, int act) -> int {
      HIPCHK(poem_launch_gemm(X, ldx, h->P(wi), h->R(bi), nullptr, 0, Y, ldy, M, N, K, act, st));
      return POEM_OK;
    };
    HIPCHK(poem_launch_grid_sample(p.x, p.uv, p.g, BN, C, c.feat_h, c.feat_w, S, st));
    // merge MLP 0 on the Q1 rows == the (BN*S, C) row-major view of g's memory
    int rc = gemm(p.g, C, T_M00_W, T_M00_B, p.h1, C, BN * S, C, C, POEM_ACT_RELU);
    if (rc == POEM_OK) rc = gemm(p.h1, C, T_M02_W, T_M02_B, p.h2, C / 2, BN * S, C / 2, C, POEM_ACT_NONE);
    if (rc != POEM_OK) return rc;
    HIPCHK(poem_launch_merge_reduce(p.h2, p.offs, p.mm, B, S, C / 2, st));
    rc = gemm(p.mm, C / 2, T_M10_W, T_M10_B, p.mh, C / 2, BS, C / 2, C / 2, POEM_ACT_RELU);
    if (rc == POEM_OK) rc = gemm(p.mh, C / 2, T_M12_W, T_M12_B, p.y, C, BS, C, C / 2, POEM_ACT_NONE);
    if (rc != POEM_OK) return rc;
    HIPCHK(poem_launch_merge_finalize(p.g, p.y, p.offs, p.bps_feat, B, S, C, st));
    return POEM_OK;
  }

  // ---- everything between the inputs and the de-normalisation: workspace / handle memory only
  int body(hipStream_t st, float* pose_dst, float* betas_dst) {
    if (const int rc = sampling(st); rc != POEM_OK) return rc;
    if (prof_fe) HIPCHK(hipEventRecord(h->prof_ev[2 * prof_slot + 1], st));
    // query_feat_embedding for every sample -- read only where block 0 does not run on the anchor tables (there F2 is evaluated
    // once on the embedding table's own Q rows, decoder.cpp query_projection)
    if (!(h->anchor_tables && h->precision == POEM_PRECISION_FP32 && (h->small_batch & 1)))
      HIPCHK(poem_launch_broadcast(h->R(T_QEMB), p.feats0, (long)Q * C, B, st));
    return run_decoder(h, p, p.feats0, p.pt_xyz, p.bps_feat, B, pose_dst, betas_dst, st, true);
  }

  // ---- hipGraph replay of body(): captured once per (batch size, plan capacity, workspace, option set) on the handle's capture
  // stream (the caller's may be the legacy default stream, which cannot be captured); the side-stream forks / joins of
  // decoder.cpp become edges of the graph.  The VIEW LAYOUT is not part of the key: every launch of the body takes its per-view
  // arrays, its view count included, from device memory (view_layout()), and its pointers from a plan laid out for plan_views
  // -- so a stream of ragged batches whose layout changes every batch, which is what the reference's collation produces
  // (lib/utils/collation.py:7-25 upstream), replays ONE graph per batch size.  A key is captured the second time it is seen (a
  // one-off shape -- the short last batch of an epoch -- never pays capture + instantiate), and an exec retired by eviction or
  // by a destroyed handle is re-used through hipGraphExecUpdate before a new one is instantiated (handle.cpp: execs are never
  // destroyed).  -> 1 replayed, 0 not eligible / not yet captured / capture failed (plain launches), < 0 error.
  int replay(void* workspace, float* pose_aa, float* betas) {
    if (!h->graphs || h->graph_broken || h->prof_on || h->tables_pending || !h->cap_stream) return 0;
    // (the operator front end -- embed widths the fused sampling kernels do not take, POEM-huge -- sizes its launches by the
    //  batch's own view total: that total joins the key there)
    const std::vector<int64_t> key = {B, plan_views, fused_fe ? -1 : BN, (int64_t)(uintptr_t)workspace, h->precision, h->anchor_tables, h->chains,
                                      h->fused_sampling, h->tables_first, h->chain_combine, h->knn_early, h->overlap, h->chain_tile,
                                      h->tables_cached, h->knn_fma, h->knn_query, h->taps, c.parametric, h->xattn_merge, h->small_batch, h->bps_defer, poem_process_switches(), h->f1_split, h->va_p1, h->group_min_views, h->group_xcd, h->d2_first, h->wait_merge,
                                      (int64_t)(uintptr_t)h->mano_table, h->mano_center};
    poem_handle_s::GraphEntry* hit = nullptr;
    for (auto& g : h->graph_cache)
      if (g.key == key) { hit = &g; break; }
    if (!hit) {
      if (!h->graph_eager) {
        bool seen = false;
        for (auto& k : h->graph_seen) seen = seen || k == key;
        if (!seen) {
          if (h->graph_seen.size() >= 64) h->graph_seen.erase(h->graph_seen.begin());
          h->graph_seen.push_back(key);
          return 0;
        }
      }
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      POEM_TRACE("capture begin h=%p B=%d cached=%zu", (void*)h, B, h->graph_cache.size());
      bool ok = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeRelaxed) == hipSuccess;
      if (ok) {
        const int rc = body(h->cap_stream, p.g_pose, p.g_betas);
        POEM_TRACE("capture body rc=%d", rc);
        const hipError_t e = hipStreamEndCapture(h->cap_stream, &graph);
        POEM_TRACE("capture end e=%d graph=%p", (int)e, (void*)graph);
        ok = rc == POEM_OK && e == hipSuccess && graph != nullptr;
      }
      ++h->graph_captures;
      if (ok) {
        exec = poem_reuse_graph_exec(graph, graph_shape_of(c, key), h->stream_device);
        if (!exec) {
          ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
          ++h->graph_instantiations;
        }
      }
      POEM_TRACE("instantiate ok=%d exec=%p", (int)ok, (void*)exec);
      if (graph) (void)hipGraphDestroy(graph);
      POEM_TRACE("graph destroyed");
      if (!ok) {
        (void)hipGetLastError();
        h->graph_broken = true;               // plain launches from now on (results are the same either way)
        return 0;
      }
      if (h->graph_cache.size() >= poem_handle_s::GRAPH_CAP) {      // evict the least recently used
        size_t lru = 0;
        for (size_t i = 1; i < h->graph_cache.size(); ++i)
          if (h->graph_cache[i].stamp < h->graph_cache[lru].stamp) lru = i;
        poem_park_graph_exec(h->graph_cache[lru].exec, h->graph_cache[lru].shape, h->stream_device, h->graph_cache[lru].last_stream,
                             h->graph_cache[lru].launched);      // (not destroyed: handle.cpp)
        h->graph_cache.erase(h->graph_cache.begin() + lru);
      }
      h->graph_cache.push_back({key, exec, 0, graph_shape_of(c, key), nullptr, false});
      hit = &h->graph_cache.back();
    }
    hit->stamp = ++h->graph_clock;
    POEM_TRACE("graph launch exec=%p", (void*)hit->exec);
    HIPCHK(hipGraphLaunch(hit->exec, s));
    hit->last_stream = s; hit->launched = true;
    ++h->graph_replays;
    POEM_TRACE("graph launched");
    if (c.parametric) {      // the captured tail wrote pose / shape into the workspace
      HIPCHK(hipMemcpyAsync(pose_aa, p.g_pose, (size_t)B * 48 * sizeof(float), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(betas, p.g_betas, (size_t)B * 10 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return 1;
  }

  // What a retired exec must share with a fresh capture to be offered for update: the model shape and the WHOLE key except the
  // workspace pointer -- batch size and switches pick kernel instantiations and optional launches (the debug taps keep the
  // last block's feed-forward alive), and the runtime refuses an update whose nodes changed kind or function.
  static uint64_t graph_shape_of(const poem_config_t& c, const std::vector<int64_t>& key) {
    uint64_t v = 1469598103934665603ull;
    auto mix = [&](uint64_t x) { v = (v ^ x) * 1099511628211ull; };
    mix((uint64_t)c.embed); mix((uint64_t)c.nblocks); mix((uint64_t)c.heads); mix((uint64_t)c.nsample); mix((uint64_t)c.nquery);
    mix((uint64_t)c.in_channels); mix((uint64_t)c.feat_h * 65536u + (uint64_t)c.feat_w); mix((uint64_t)c.max_views);
    for (size_t i = 0; i < key.size(); ++i)
      if (i != 3) mix((uint64_t)key[i]);          // [3] = the workspace pointer
    return v ? v : 1;
  }
};

}  // namespace

extern "C" {

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
  if (view_offsets_host[0] != 0 || view_offsets_host[batch] < batch) return POEM_E_ARG;
  SplitContext split_ctx(h);
  // A workspace that holds the plan of batch * max_views views (poem_workspace_bytes(h, batch, batch * max_views)) makes every
  // pointer of the forward a function of the batch size alone; a smaller one is laid out for this batch's own view total.
  const int total_views = view_offsets_host[batch];
  const long cap_views = (long)batch * c.max_views;
  int plan_views = total_views;
  if (cap_views > total_views && cap_views < (1l << 30) && workspace_bytes >= make_plan(c, batch, (int)cap_views, nullptr).bytes)
    plan_views = (int)cap_views;
  HeadRun run(h, mlvl_feat, cam_intr, cam_extr, view_offsets_host, batch, plan_views, reference_joints, img_w, img_h, workspace,
              (hipStream_t)stream);
  if (workspace_bytes < run.p.bytes) return POEM_E_WORKSPACE;
  POEM_TRACE("head_forward h=%p B=%d C=%d", (void*)h, batch, c.embed);
  int rc = run.view_layout();
  if (rc == POEM_OK) rc = run.anchor_tables();
  if (rc == POEM_OK) rc = run.inputs();
  POEM_TRACE("inputs rc=%d", rc);
  if (rc != POEM_OK) return rc;
  rc = run.replay(workspace, pose_aa, betas);
  POEM_TRACE("replay rc=%d", rc);
  if (rc < 0) return rc;
  if (rc == 0) ++h->plain_forwards;
  if (rc == 0 && (rc = run.body(run.s, pose_aa, betas)) != POEM_OK) return rc;
  const bool mano = c.parametric && h->mano_table;      // the last layer = the attached MANO layer's output + centre
  HIPCHK(poem_launch_finalize(run.p.xyz[1], run.p.centre, out_xyz, c.nblocks, batch, c.nquery, c.radius, mano ? run.p.mano_verts : nullptr,
                              mano ? run.p.mano_joints : nullptr, run.s));
  register_taps(h, run.p, batch, run.BN, true);
  POEM_TRACE("head_forward done");
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
  if (c.parametric && h->mano_table)      // get_parametric_output: the last layer's rows are the MANO layer's (pt_metro_transformer.py:149-150)
    HIPCHK(poem_launch_param_rows(p.mano_verts, p.mano_joints, out_xyz_norm + (size_t)(c.nblocks - 1) * n, batch, c.nquery, s));
  register_taps(h, p, batch, batch, false);
  return POEM_OK;
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

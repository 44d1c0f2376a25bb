// The launch sequence of the decoder (PtEmbedTRv4.forward -> three point_METRO_blocks, lib/models/layers/ptEmb_transformer.py:
// 115-121,371-376 and lib/models/bricks/pt_metro_transformer.py:153-200 upstream), one method per stage of a block:
//
//   basis_point_side(i)   F1: keys / values of both BERT cross attentions + of the vector cross attention   (side stream)
//   neighbour_searches(i) knn_points x2 on xyz_i                                                             (side stream)
//   query_projection(i)   F2: embedding | first attention's query                                            (block 0 / operator mode)
//   cross_attentions(i)   BertAttention x2 -> h_cross, then F3 (q | k | v of the vector self attention)
//   vector_self(i)        ptTransformerBlock -> r_s ; fc2 + residual -> f_self ; composed cross query
//   vector_cross(i)       ptTransformerBlock_CrossAttn -> r_c
//   tail(i)               fc2 + residual -> f_cross ; reg_branch -> xyz_{i+1} ; feed forward + LayerNorm -> feats ; next F2
//
// Three HIP streams.  The caller's stream carries the query side (the critical path).  Everything that depends only on the
// basis-point features is issued up front on `bps_stream` for EVERY block; the neighbour searches of block i (they need only
// xyz_i) go to `knn_stream`.  Events fork / join inside one call, so the caller only ever synchronises its own stream -- and a
// stream capture of the call (forward.cpp) turns the forks and joins into graph edges.
#include "engine.h"

// Block-0 anchor tables (poem_handle_s::anchor_tables): at poem_create into handle-owned memory (default), or -- tables_cached
// = 0 -- once per forward on the neighbour-search stream, forked at the very top of poem_head_forward.
int build_anchor_tables(poem_handle_t h, Plan& p, hipStream_t s, bool at_create) {
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

namespace {

struct DecoderRun {
  poem_handle_t h;
  Plan& p;
  const poem_config_t& c;
  const float *pt_xyz, *pt_feats;
  float *pose_aa, *betas;
  const int B, C, S, Q, BS, BQ;
  const bool ov;            // side streams in use
  hipStream_t s, sb, sk;    // query side (caller's), basis-point side, neighbour searches
  const bool tables;        // block 0 on the anchor tables (head path, fp32)
  const bool chain;         // query-side row-tile chains (chain.hip / chain16.hip) instead of one launch per operator
  // state handed from stage to stage of a block
  const float* feats;       // the block's input query features
  const float* hidden;      // residual of the attention in flight
  int ldh = 0, hidden_mod = 0, q_batch = 0;
  const float* q0 = nullptr;             // first attention's query projection
  const int *idx_s = nullptr, *idx_c = nullptr;
  const float* anchor = nullptr;
  int shared = 0;
  bool knn_issued[9] = {};
  // wait_merge (chain mode with side streams; scheduling only).  A launch that waits for another stream costs 5-13 us in a
  // replayed launch graph even when the other side finished long ago (LABNOTES R5.3), and a small batch is a chain of ~40
  // dependent launches: (a) the first cross attention is issued in FRONT of the side stream's remaining launches (block 0's
  // anchor rows, the later blocks' F1), so the runtime keeps F1(0) -> attention on one hardware queue; (b) the main stream waits
  // for the later blocks' F1 once, where block 0's vector cross attention waits for its anchor rows anyway; (c, lab bit, off) it waits for
  // block i's neighbour searches where block i's first cross attention starts instead of in front of the vector self attention
  // (nothing at B <= 3, +0.5-1 % from four samples on: the searches no longer overlap the first cross attention).
  // Measured (a) + (b): B = 1 / 2 / 4 -2.0 / -1.4 / -1.2 %, B >= 8 +-0.
  bool wmerge = false;
  int wmask = 0;              // 1 = (a), 2 = (b), 4 = (c)
  bool side_rest_pending = false;
  bool bps_waited[9] = {}, knn_waited[9] = {};
  bool anchor_from_y3 = false;      // block 0 on the tables: (kg | v) of the anchor rows are rows of p.y3 (small batches)

  DecoderRun(poem_handle_t h_, Plan& p_, const float* feats_in, const float* pt_xyz_, const float* pt_feats_, int B_, float* pose,
             float* bet, hipStream_t s_, bool template_queries)
      : h(h_), p(p_), c(h_->cfg), pt_xyz(pt_xyz_), pt_feats(pt_feats_), pose_aa(pose), betas(bet), B(B_), C(c.embed), S(c.nsample),
        Q(c.nquery), BS(B_ * c.nsample), BQ(B_ * c.nquery), ov(h_->overlap && h_->bps_stream && h_->knn_stream), s(s_),
        sb(ov ? h_->bps_stream : s_), sk(ov ? h_->knn_stream : s_),
        tables(template_queries && h_->anchor_tables && h_->precision == POEM_PRECISION_FP32),
        chain(h_->chains && h_->precision == POEM_PRECISION_FP32 && poem_chain_supported(c.embed) != 0), feats(feats_in),
        hidden(nullptr) {}

  int gemm(const float* X, int ldx, int wi, int bi, const float* res, int ldr, float* Y, int ldy, int M, int N, int K, int act) {
    HIPCHK(poem_launch_gemm(X, ldx, h->P(wi), bi >= 0 ? h->R(bi) : nullptr, res, ldr, Y, ldy, M, N, K, act, s));
    return POEM_OK;
  }
  bool prof_begin() {
    const bool on = h->prof_on && (size_t)(2 * h->prof_used + 1) < h->prof_ev.size();
    if (on && hipEventRecord(h->prof_ev[2 * h->prof_used], s) != hipSuccess) return false;
    return on;
  }
  int prof_end(bool on, int kind) {
    if (on) { HIPCHK(hipEventRecord(h->prof_ev[2 * h->prof_used + 1], s)); h->prof_kind[h->prof_used++] = (char)kind; }
    return POEM_OK;
  }
  ChainArgs chain_args(int kind) const {
    ChainArgs a{};
    a.kind = kind; a.M = BQ; a.tile_p = h->chain_tile; a.eps = c.ln_eps;
    a.native_delta = h->native16 ? (long long)(h->native16 - h->packed_base) : 0;
    return a;
  }

  // ---- fork / join of the side streams ---------------------------------------------------------------------------------
  int fork() {
    if (!ov) return POEM_OK;
    HIPCHK(hipEventRecord(h->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(sb, h->ev_fork, 0));
    HIPCHK(hipStreamWaitEvent(sk, h->ev_fork, 0));
    return POEM_OK;
  }
  int join() {     // every side-stream product has been consumed behind an event; join so the caller's stream owns the tail
    if (!ov) return POEM_OK;
    HIPCHK(hipEventRecord(h->ev_join_bps, sb));
    HIPCHK(hipEventRecord(h->ev_join_knn, sk));
    HIPCHK(hipStreamWaitEvent(s, h->ev_join_bps, 0));
    HIPCHK(hipStreamWaitEvent(s, h->ev_join_knn, 0));
    return POEM_OK;
  }

  // ---- F1: keys / values of both BERT cross attentions and of the vector cross attention, straight from the basis-point
  // features (embedding and fc1 composed in); the four BERT blocks leave the GEMM as MFMA fragment images
  int basis_point_side(int i) {
    const auto& f = h->fused[i];
    const size_t seg = (size_t)BS * C;
    float* outs[6] = {p.y1[i], p.y1[i] + seg, p.y1[i] + 2 * seg, p.y1[i] + 3 * seg, p.y1[i] + 4 * seg, p.y1[i] + 5 * seg};
    const int modes[6] = {1, 2, 1, 2, 0, 0};
    h->kv_presplit[i] = poem_gemm_split_applies(f.w[0], BS, C, C) != 0;      // split GEMM -> the K / V images are split too
    const bool anchored = tables && i == 0;
    if (!anchored && h->f1_split && C % 128 == 0) {
      HIPCHK(poem_launch_gemm_segs(pt_feats, C, f.w[0], f.b[0], BS, C, POEM_ACT_NONE, C, 4, outs, modes, sb));
      HIPCHK(poem_launch_gemm_segs(pt_feats, C, (const float*)f.w[0] + (size_t)4 * C * C, f.b[0] + 4 * C, BS, C, POEM_ACT_NONE, C, 2, outs + 4,
                                   modes + 4, sb));
    } else {
      HIPCHK(poem_launch_gemm_segs(pt_feats, C, f.w[0], f.b[0], BS, C, POEM_ACT_NONE, C, anchored ? 4 : 6, outs, modes, sb));
    }
    if (ov) HIPCHK(hipEventRecord(h->ev_bps[i], sb));      // the cross attentions wait for the K / V images only
    if (anchored && !side_rest_pending) return basis_point_anchor_rows(f);
    return POEM_OK;
  }
  int basis_point_anchor_rows(const poem_handle_s::Fused& f) {
    {
      // the vector cross attention of block 0 reads only the 32 anchor rows of (kc | vc): project just those (the same
      // fma chain per element as the full GEMM's rows); its own event: the first cross attention does not wait for them
      HIPCHK(poem_launch_gather_anchor_rows(pt_feats, C, h->anchor_idx, S, p.anch_x[1], B, C, p.ident, sb));
      HIPCHK(poem_launch_gemm(p.anch_x[1], C, (const float*)f.w[0] + (size_t)4 * C * C, f.b[0] + 4 * C, nullptr, 0,
                              p.anch_kv[1], 2 * C, B * 32, 2 * C, C, POEM_ACT_NONE, sb));
      if (ov) HIPCHK(hipEventRecord(h->ev_xyz[0], sb));    // (ev_xyz[0]: block 0 has no neighbour search to use it for)
    }
    return POEM_OK;
  }

  bool merge_in_attention() const { return h->xattn_merge >= 0 ? h->xattn_merge != 0 : B <= 2; }

  // F1 of block i + 1 behind stage `at` of block i's cross attentions (poem_handle_s::bps_defer)
  int defer_at() const {
    return (ov && chain) ? h->bps_defer : 0;
  }
  int deferred_basis_point_side(int i, int at) {
    if (defer_at() != at || i + 1 >= c.nblocks) return POEM_OK;
    HIPCHK(hipEventRecord(h->ev_def[i], s));
    HIPCHK(hipStreamWaitEvent(sb, h->ev_def[i], 0));
    return basis_point_side(i + 1);
  }

  // ---- neighbours of block i >= 1 from xyz_i (block 0: the fixed anchors for both attentions -- Q2).  The large search
  // first: it gets its CUs before the persistent attention kernel of the block takes them all.
  // `recorded`: ev_xyz[i] has already been recorded on s where xyz_i became final (tail(): the searches are then ISSUED behind
  // the chain launch that follows -- see there)
  int neighbour_searches(int i, bool recorded = false) {
    if (knn_issued[i]) return POEM_OK;
    if (ov) {
      if (!recorded) HIPCHK(hipEventRecord(h->ev_xyz[i], s));            // xyz_i is final here
      HIPCHK(hipStreamWaitEvent(sk, h->ev_xyz[i], 0));
    }
    HIPCHK(poem_launch_knn(p.xyz[i], pt_xyz, p.idx_cross[i], B, Q, S, h->knn_fma, sk));
    HIPCHK(poem_launch_knn(p.xyz[i], p.xyz[i], p.idx_self[i], B, Q, Q, h->knn_fma, sk));
    if (ov) HIPCHK(hipEventRecord(h->ev_knn[i], sk));
    knn_issued[i] = true;
    return POEM_OK;
  }

  // ---- F2: qe = embedding(feats) | query projection of the first attention (composed with the embedding)
  int query_projection(int i) {
    hidden = p.qeqp;                   // residual of the first attention: qe
    ldh = 2 * C; hidden_mod = 0; q_batch = Q;
    q0 = p.qeqp + C;
    if (tables && i == 0) {
      // every sample's block-0 query features are the learned embedding table: F2 on its Q rows, once -- at poem_create (the
      // handle's copy; same launch, same inputs: bit-identical), or here when the per-forward tables are asked for
      const float* qe0 = p.qeqp0;
      if (h->tables_cached && h->c_qeqp0 && h->precision == POEM_PRECISION_FP32)
        qe0 = h->c_qeqp0;
      else
        HIPCHK(poem_launch_gemm_split(h->R(T_QEMB), C, h->fused[i].w[1], h->fused[i].b[1], nullptr, 0, p.qeqp0, 2 * C, Q, 2 * C, C,
                                      POEM_ACT_NONE, 2 * C, POEM_ACT_NONE, s));
      if (chain) {       // ... and read by every sample in place: queries with batch stride 0, residual rows modulo Q
        hidden = qe0; hidden_mod = Q; q0 = qe0 + C; q_batch = 0;
      } else {
        HIPCHK(poem_launch_broadcast(qe0, p.qeqp, (long)Q * 2 * C, B, s));
      }
    } else if (!(chain && i > 0)) {    // (chain mode: block i-1's last chain already wrote p.qeqp)
      HIPCHK(poem_launch_gemm_split(feats, C, h->fused[i].w[1], h->fused[i].b[1], nullptr, 0, p.qeqp, 2 * C, BQ, 2 * C, C,
                                    POEM_ACT_NONE, 2 * C, POEM_ACT_NONE, s));
    }
    return POEM_OK;
  }

  // F3 of block 0 on the tables: (kg | v) for the 32 anchor rows only (qg for every row comes from the caller)
  int anchor_rows_f3(int i) {
    HIPCHK(poem_launch_gather_anchor_rows(hidden, C, h->anchor_idx, Q, p.anch_x[0], B, C, nullptr, s));
    HIPCHK(poem_launch_gemm(p.anch_x[0], C, (const float*)h->fused[i].w[2] + (size_t)C * C, h->fused[i].b[2] + C, nullptr, 0,
                            p.anch_kv[0], 2 * C, B * 32, 2 * C, C, POEM_ACT_NONE, s));
    return POEM_OK;
  }

  // ---- the two BERT cross attentions -> h_cross, then F3 = (w_qs | w_ks | w_vs) o fc1 on it
  int cross_attentions(int i) {
    const int bb = h->block_base(i);
    if (chain) {
      const int a1 = bb + B_A1, a2 = bb + B_A2;
      if (ov && !bps_waited[i]) HIPCHK(hipStreamWaitEvent(s, h->ev_bps[i], 0));
      if (ov && (wmask & 4) && h->knn_early && i > 0 && knn_issued[i] && !knn_waited[i]) {      // (c)
        HIPCHK(hipStreamWaitEvent(s, h->ev_knn[i], 0));
        knn_waited[i] = true;
      }
      // the chain combines the attention's split-key partials while it fills its tile (no attn_combine launch, no ctx round trip)
      const void *part_o = nullptr, *part_ml = nullptr;
      int pchunks = 0;
      float pkc2 = 0.f;
      poem_cross_attention_partials(B, Q, S, C, c.heads, p.attn_scratch, &part_o, &part_ml, &pchunks, &pkc2);
      // the kernel merges the partials itself where it can (attn.hip MERGE); else the chain does while it fills its tile
      const bool merged = merge_in_attention() && poem_cross_attention_merges(S, C, c.heads) != 0;
      const bool comb = !merged && h->chain_combine && poem_chain_combines(C, c.heads, pchunks) != 0;
      auto attention = [&](const float* q, int ldq, int qb, const float* kimg, const float* vimg) -> hipError_t {
        if (merged) return poem_launch_cross_attention_merged(q, ldq, qb, kimg, vimg, p.ctx, B, Q, S, C, c.heads, s);
        return poem_launch_cross_attention_imgq(q, ldq, qb, kimg, vimg, comb ? nullptr : p.ctx, B, Q, S, C, c.heads, p.attn_scratch, s);
      };
      auto from_partials = [&](ChainArgs& a) {
        if (!comb) return;
        a.x = nullptr; a.part_o = (const float4*)part_o; a.part_ml = (const float2*)part_ml;
        a.pc_heads = c.heads; a.pc_chunks = pchunks; a.pc_nq = Q; a.pc_kc2 = pkc2;
      };
      HIPCHK(attention(q0, 2 * C, q_batch, p.y1[i], p.y1[i] + (size_t)BS * C));
      if (side_rest_pending) {                                               // (a)
        side_rest_pending = false;
        if (tables && i == 0)
          if (const int rc = basis_point_anchor_rows(h->fused[0]); rc != POEM_OK) return rc;
        for (int k = 1; k < c.nblocks; ++k)
          if (const int rc = basis_point_side(k); rc != POEM_OK) return rc;
      }
      if (const int rc = deferred_basis_point_side(i, 1); rc != POEM_OK) return rc;
      ChainArgs ca = chain_args(0);
      ca.x = p.ctx; ca.ldx = C;
      from_partials(ca);
      ca.w1 = (const float4*)h->P(a1 + 6); ca.b1 = h->R(a1 + 7); ca.res = hidden; ca.ldres = ldh; ca.res_mod = hidden_mod;
      ca.ln_g = h->R(a1 + 8); ca.ln_b = h->R(a1 + 9); ca.y1 = p.h_attn; ca.ldy1 = C;
      ca.w2 = (const float4*)h->P(a2 + 0); ca.b2 = h->R(a2 + 1); ca.n2 = 1; ca.y2 = p.qp; ca.ldy2 = C;
      HIPCHK(poem_launch_chain(&ca, C, s));
      HIPCHK(attention(p.qp, C, Q, p.y1[i] + (size_t)2 * BS * C, p.y1[i] + (size_t)3 * BS * C));
      if (const int rc = deferred_basis_point_side(i, 2); rc != POEM_OK) return rc;
      ChainArgs cb = chain_args(0);
      cb.x = p.ctx; cb.ldx = C;
      from_partials(cb);
      cb.w1 = (const float4*)h->P(a2 + 6); cb.b1 = h->R(a2 + 7); cb.res = p.h_attn; cb.ldres = C; cb.res_mod = 0;
      cb.ln_g = h->R(a2 + 8); cb.ln_b = h->R(a2 + 9); cb.y1 = p.h_cross[i]; cb.ldy1 = C;
      // block 0 on the tables needs qg for every row and (kg | v) for the anchor rows only
      // (a small batch lets the chain project all three for every row -- two more GEMM phases on tiles that are latency chains
      //  anyway -- and block 0's vector self attention picks its 32 anchor rows out of them: no gather + 32-row GEMM launches)
      anchor_from_y3 = tables && i == 0 && (h->small_batch & 2) && (long)BQ <= 16L * poem_device_cu_count();
      cb.w2 = (const float4*)h->fused[i].w[2]; cb.b2 = h->fused[i].b[2]; cb.n2 = (tables && i == 0 && !anchor_from_y3) ? 1 : 3; cb.y2 = p.y3; cb.ldy2 = 3 * C;
      HIPCHK(poem_launch_chain(&cb, C, s));
      if (const int rc = deferred_basis_point_side(i, 3); rc != POEM_OK) return rc;
      hidden = p.h_cross[i];
      ldh = C;
      if (tables && i == 0 && !anchor_from_y3) return anchor_rows_f3(i);
      return POEM_OK;
    }
    // operator sequence: query projection, attention, out-proj + residual, LayerNorm -- twice; then F3
    for (int a = 0; a < 2; ++a) {
      const int ab = bb + (a == 0 ? B_A1 : B_A2);
      float* hout = a == 0 ? p.h_attn : p.h_cross[i];
      const float* qptr = p.qeqp + C;
      int ldq = 2 * C;
      if (a == 1) {
        const int rc = gemm(hidden, ldh, ab + 0, ab + 1, nullptr, 0, p.qp, C, BQ, C, C, POEM_ACT_NONE);
        if (rc != POEM_OK) return rc;
        qptr = p.qp;
        ldq = C;
      }
      if (ov && a == 0) HIPCHK(hipStreamWaitEvent(s, h->ev_bps[i], 0));
      if (h->precision == POEM_PRECISION_SPLIT_F16X3_ALL) poem_cross_attention_split(h->kv_presplit[i] ? 2 : 1);
      if (h->precision == POEM_PRECISION_FP32 && merge_in_attention() && poem_cross_attention_merges(S, C, c.heads))
        HIPCHK(poem_launch_cross_attention_merged(qptr, ldq, Q, p.y1[i] + (size_t)(2 * a) * BS * C, p.y1[i] + (size_t)(2 * a + 1) * BS * C,
                                                  p.ctx, B, Q, S, C, c.heads, s));
      else
        HIPCHK(poem_launch_cross_attention_img(qptr, ldq, p.y1[i] + (size_t)(2 * a) * BS * C, p.y1[i] + (size_t)(2 * a + 1) * BS * C,
                                               p.ctx, B, Q, S, C, c.heads, p.attn_scratch, s));
      const int rc = gemm(p.ctx, C, ab + 6, ab + 7, hidden, ldh, p.att, C, BQ, C, C, POEM_ACT_NONE);
      if (rc != POEM_OK) return rc;
      HIPCHK(poem_launch_layernorm(p.att, h->R(ab + 8), h->R(ab + 9), hout, BQ, C, c.ln_eps, s));
      hidden = hout;
      ldh = C;
    }
    if (tables && i == 0) {
      HIPCHK(poem_launch_gemm(hidden, C, h->fused[i].w[2], h->fused[i].b[2], nullptr, 0, p.y3, 3 * C, BQ, C, C, POEM_ACT_NONE, s));
      return anchor_rows_f3(i);
    }
    HIPCHK(poem_launch_gemm_split(hidden, C, h->fused[i].w[2], h->fused[i].b[2], nullptr, 0, p.y3, 3 * C, BQ, 3 * C, C, POEM_ACT_NONE,
                                  3 * C, POEM_ACT_NONE, s));
    return POEM_OK;
  }

  // ---- vector self attention over the queries -> r_s ; f_self = fc2(r_s) + h_cross ; qc = composed cross query on f_self
  int vector_self(int i) {
    const int vsb = h->block_base(i) + B_VS;
    const float* xyz = p.xyz[i];
    if (ov && i > 0 && !knn_waited[i]) HIPCHK(hipStreamWaitEvent(s, h->ev_knn[i], 0));
    const bool prof = prof_begin();
    const int kq = shared ? 32 : (h->knn_query ? h->knn_query : c.knn);      // N_NEIGHBOR_QUERY (block 0: the 32 anchors)
    if (h->precision != POEM_PRECISION_FP32) {
      if (kq != 32) return POEM_E_UNSUPPORTED;      // the split-precision kernel has no masked form
      const auto& sw = h->split[2 * i];
      HIPCHK(poem_launch_vector_attention_split(xyz, xyz, anchor, idx_s, shared, p.y3, p.y3 + C, p.y3 + 2 * C, Q, h->R(vsb + 4),
                                                h->R(vsb + 5), sw.w[0], h->R(vsb + 7), sw.w[1], sw.w[2], sw.scales, p.rs, B, Q, C,
                                                3 * C, 3 * C, 3 * C, s));
    } else if (tables && i == 0) {
      if (ov && h->tables_pending) HIPCHK(hipStreamWaitEvent(s, h->ev_tab, 0));
      if (anchor_from_y3)      // keys / values straight from the rows the chain projected: neighbour j = query row anchor_idx[j] (Q2)
        HIPCHK(poem_launch_vector_attention_anchored(h->anchor_idx, p.y3, p.y3 + C, p.y3 + 2 * C, Q, h->P(vsb + 10), p.tab_g[0], p.tab_p[0],
                                                     p.rs, B, Q, C, 3 * C, 3 * C, 3 * C, s));
      else {
      if (ov) HIPCHK(hipStreamWaitEvent(s, h->ev_xyz[0], 0));      // p.ident is written on the basis-point stream (basis_point_side(0))
      HIPCHK(poem_launch_vector_attention_anchored(p.ident, p.y3, p.anch_kv[0], p.anch_kv[0] + C, 32, h->P(vsb + 10), p.tab_g[0],
                                                   p.tab_p[0], p.rs, B, Q, C, 3 * C, 2 * C, 2 * C, s));
      }
    } else {
      poem_vecattn_one_query_blocks(h->va_p1);           // (thread-local launcher switches: this handle's values for this launch only)
      poem_vecattn_valid_neighbours(kq);
      const hipError_t ve = poem_launch_vector_attention(xyz, xyz, anchor, idx_s, shared, p.y3, p.y3 + C, p.y3 + 2 * C, Q, h->R(vsb + 4),
                                                         h->R(vsb + 5), h->P(vsb + 6), h->R(vsb + 7), h->fused[i].w[5], h->R(vsb + 9),
                                                         h->P(vsb + 10), h->R(vsb + 11), p.rs, B, Q, C, 3 * C, 3 * C, 3 * C, 1, s);
      poem_vecattn_valid_neighbours(32);
      poem_vecattn_one_query_blocks(0);
      HIPCHK(ve);
    }
    if (const int rc = prof_end(prof, tables && i == 0 ? 1 : 0); rc != POEM_OK) return rc;
    if (chain) {
      ChainArgs cc = chain_args(1);
      cc.x = p.rs; cc.ldx = C;
      cc.w1 = (const float4*)h->P(vsb + 2); cc.b1 = h->R(vsb + 3); cc.res = hidden; cc.ldres = C; cc.res_mod = 0;
      cc.y1 = p.f_self[i]; cc.ldy1 = C;
      cc.w2 = (const float4*)h->fused[i].w[4]; cc.b2 = h->fused[i].b[4]; cc.n2 = 1; cc.y2 = p.qc; cc.ldy2 = C;
      HIPCHK(poem_launch_chain(&cc, C, s));
      return POEM_OK;
    }
    if (const int rc = gemm(p.rs, C, vsb + 2, vsb + 3, hidden, C, p.f_self[i], C, BQ, C, C, POEM_ACT_NONE); rc != POEM_OK) return rc;
    HIPCHK(poem_launch_gemm(p.f_self[i], C, h->fused[i].w[4], h->fused[i].b[4], nullptr, 0, p.qc, C, BQ, C, C, POEM_ACT_NONE, s));
    return POEM_OK;
  }

  // ---- vector cross attention over the basis points -> r_c
  int vector_cross(int i) {
    const int vcb = h->block_base(i) + B_VC;
    const float* xyz = p.xyz[i];
    if (ov && (wmask & 2) && i == 0 && defer_at() == 0)                        // (b)
      for (int k = 1; k < c.nblocks; ++k) {
        HIPCHK(hipStreamWaitEvent(s, h->ev_bps[k], 0));
        bps_waited[k] = true;
      }
    const bool prof = prof_begin();
    const int kc = shared ? 32 : c.knn;                                       // N_NEIGHBOR
    if (h->precision != POEM_PRECISION_FP32) {
      if (kc != 32) return POEM_E_UNSUPPORTED;
      const auto& sw = h->split[2 * i + 1];
      HIPCHK(poem_launch_vector_attention_split(xyz, pt_xyz, anchor, idx_c, shared, p.qc, p.y1[i] + 4 * (size_t)BS * C,
                                                p.y1[i] + 5 * (size_t)BS * C, S, h->R(vcb + 4), h->R(vcb + 5), sw.w[0],
                                                h->R(vcb + 7), sw.w[1], sw.w[2], sw.scales, p.rc, B, Q, C, C, C, C, s));
    } else if (tables && i == 0) {
      if (ov) HIPCHK(hipStreamWaitEvent(s, h->ev_xyz[0], 0));      // the anchor rows of (kc | vc), basis_point_side(0)
      HIPCHK(poem_launch_vector_attention_anchored(p.ident, p.qc, p.anch_kv[1], p.anch_kv[1] + C, 32, h->P(vcb + 10), p.tab_g[1],
                                                   p.tab_p[1], p.rc, B, Q, C, C, 2 * C, 2 * C, s));
    } else {
      poem_vecattn_one_query_blocks(h->va_p1);
      poem_vecattn_valid_neighbours(kc);
      const hipError_t ve = poem_launch_vector_attention(xyz, pt_xyz, anchor, idx_c, shared, p.qc, p.y1[i] + 4 * (size_t)BS * C,
                                                         p.y1[i] + 5 * (size_t)BS * C, S, h->R(vcb + 4), h->R(vcb + 5), h->P(vcb + 6),
                                                         h->R(vcb + 7), h->fused[i].w[6], h->R(vcb + 9), h->P(vcb + 10), h->R(vcb + 11), p.rc, B,
                                                         Q, C, C, C, C, 1, s);
      poem_vecattn_valid_neighbours(32);
      poem_vecattn_one_query_blocks(0);
      HIPCHK(ve);
    }
    return prof_end(prof, tables && i == 0 ? 1 : 0);
  }

  // the parametric tail of the last block (medium_MANO: Q3 flatten, Linears, rot6d -> axis-angle)
  int parametric_tail(int i) {
    const int bb = h->block_base(i);
    HIPCHK(poem_launch_q3_flatten(feats, h->R(bb + B_FLAT_W), h->R(bb + B_FLAT_B), p.q3t, B, Q, C, s));
    HIPCHK(poem_launch_narrow_linear(p.q3t, C, h->R(bb + B_MANO_W), h->R(bb + B_MANO_B), nullptr, p.par, B, C, 106, s));
    HIPCHK(poem_launch_rot6d_to_aa(p.par, pose_aa, betas, B, s));
    // the attached MANO layer (poem_attach_mano): mano_layer(pose_aa, betas) of get_parametric_output (:147-148 upstream) right
    // here -- inside the captured body, no Python callable between the decoder and the de-normalisation
    if (h->mano_table) HIPCHK(poem_launch_mano_lbs(pose_aa, betas, h->mano_table, p.mano_verts, p.mano_joints, B, h->mano_center, s));
    return POEM_OK;
  }

  // ---- f_cross = fc2(r_c) + f_self ; reg_branch -> xyz_{i+1} ; feed forward + LayerNorm -> feats ; (chain mode) next F2.
  // The last block's feed-forward output feeds nothing -- PtEmbedTRv4.forward returns the coordinate stack only
  // (ptEmb_transformer.py:115-121,371-376 upstream; the reference evaluates it and drops it) -- so it is computed only when
  // something reads it: the parametric tail (medium_MANO) or the debug taps.  Returns `done` = the decoder ends here.
  int tail(int i, bool* done) {
    const int bb = h->block_base(i), vcb = bb + B_VC;
    const float* xyz = p.xyz[i];
    const bool last = i == c.nblocks - 1;
    const bool feats_dead = last && !c.parametric && !h->taps;
    *done = feats_dead;
    if (chain) {
      ChainArgs cd = chain_args(2);
      cd.x = p.rc; cd.ldx = C;
      cd.w1 = (const float4*)h->P(vcb + 2); cd.b1 = h->R(vcb + 3); cd.res = p.f_self[i]; cd.ldres = C; cd.res_mod = 0;
      cd.y1 = p.f_cross[i]; cd.ldy1 = C;
      cd.wf4 = (const float4*)h->fused[i].w[3]; cd.bf4 = h->fused[i].b[3];
      cd.wreg2 = h->R(bb + B_REG2_W); cd.breg2 = h->R(bb + B_REG2_B); cd.xyz_in = xyz; cd.xyz_out = p.xyz[i + 1];
      HIPCHK(poem_launch_chain(&cd, C, s));       // D1: xyz_{i+1} is final behind it (the next block's searches wait for it)
      if (feats_dead) return POEM_OK;
      // The next block's searches overlap D2 and the next cross attention.  Which of the two is ISSUED first decides which one
      // the launch graph keeps on D1's hardware queue (the runtime continues a node's queue with its first-created successor and
      // moves the others to another queue: a cross-queue edge costs ~10 us, LABNOTES R5.3): D2 first keeps D1 -> D2 -> next cross
      // attention -- the critical path -- on one queue and sends the searches, which have ~80 us of slack, to the side queue.
      const bool early = ov && h->knn_early && !last;
      const bool d2_first = early && h->d2_first != 0;
      if (early && d2_first) HIPCHK(hipEventRecord(h->ev_xyz[i + 1], s));
      if (early && !d2_first)
        if (const int rc = neighbour_searches(i + 1); rc != POEM_OK) return rc;
      ChainArgs ce = chain_args(3);
      ce.x = p.f_cross[i]; ce.ldx = C;
      ce.wf4 = (const float4*)h->fused[i].w[3]; ce.bf4 = h->fused[i].b[3];
      ce.wout = (const float4*)h->P(bb + B_OUT_W); ce.bout = h->R(bb + B_OUT_B);
      ce.ln2_g = h->R(bb + B_LN_W); ce.ln2_b = h->R(bb + B_LN_B); ce.y3 = p.feats[i]; ce.ldy3 = C;
      if (!last) {
        ce.w2 = (const float4*)h->fused[i + 1].w[1]; ce.b2 = h->fused[i + 1].b[1]; ce.n2 = 2; ce.y2 = p.qeqp; ce.ldy2 = 2 * C;
      }
      HIPCHK(poem_launch_chain(&ce, C, s));
      if (early && d2_first)
        if (const int rc = neighbour_searches(i + 1, true); rc != POEM_OK) return rc;
      feats = p.feats[i];
      return c.parametric && last ? parametric_tail(i) : POEM_OK;
    }
    if (const int rc = gemm(p.rc, C, vcb + 2, vcb + 3, p.f_self[i], C, p.f_cross[i], C, BQ, C, C, POEM_ACT_NONE); rc != POEM_OK) return rc;
    if (feats_dead) {
      HIPCHK(poem_launch_gemm(p.f_cross[i], C, h->fused[i].w[3], h->fused[i].b[3], nullptr, 0, p.y4, 5 * C, BQ, C, C, POEM_ACT_RELU, s));
      HIPCHK(poem_launch_narrow_linear(p.y4, 5 * C, h->R(bb + B_REG2_W), h->R(bb + B_REG2_B), xyz, p.xyz[i + 1], BQ, C, 3, s));
      return POEM_OK;
    }
    // F4: reg_branch.0 (relu) | intermediate.dense (gelu) share f_cross
    HIPCHK(poem_launch_gemm_split(p.f_cross[i], C, h->fused[i].w[3], h->fused[i].b[3], nullptr, 0, p.y4, 5 * C, BQ, 5 * C, C,
                                  POEM_ACT_RELU, C, POEM_ACT_GELU, s));
    HIPCHK(poem_launch_narrow_linear(p.y4, 5 * C, h->R(bb + B_REG2_W), h->R(bb + B_REG2_B), xyz, p.xyz[i + 1], BQ, C, 3, s));
    // feed forward (second Linear; its input is y4[:, C:5C])
    if (const int rc = gemm(p.y4 + C, 5 * C, bb + B_OUT_W, bb + B_OUT_B, p.f_cross[i], C, p.ffo, C, BQ, C, 4 * C, POEM_ACT_NONE); rc != POEM_OK)
      return rc;
    HIPCHK(poem_launch_layernorm(p.ffo, h->R(bb + B_LN_W), h->R(bb + B_LN_B), p.feats[i], BQ, C, c.ln_eps, s));
    feats = p.feats[i];
    return c.parametric && last ? parametric_tail(i) : POEM_OK;
  }

  int run() {
    int rc = fork();
    if (rc != POEM_OK) return rc;
    wmask = (ov && chain) ? (h->wait_merge >= 0 ? h->wait_merge : 3) : 0;
    wmerge = wmask != 0;
    side_rest_pending = (wmask & 1) && defer_at() == 0;
    if (ov)          // the basis-point side of every block up front on its stream (or block 0's only: defer_at(), wait_merge (a))
      for (int i = 0; i < ((defer_at() || side_rest_pending) ? 1 : c.nblocks); ++i)
        if ((rc = basis_point_side(i)) != POEM_OK) return rc;
    for (int i = 0; i < c.nblocks; ++i) {
      idx_s = idx_c = h->anchor_idx;       // block 0: the fixed anchors for both attentions (Q2)
      anchor = h->anchor;
      shared = 1;
      if (i > 0) {
        if ((rc = neighbour_searches(i)) != POEM_OK) return rc;     // (chain mode issued them behind block i-1's D1)
        idx_s = p.idx_self[i]; idx_c = p.idx_cross[i]; anchor = nullptr; shared = 0;
      }
      if (!ov && (rc = basis_point_side(i)) != POEM_OK) return rc;
      bool done = false;
      if ((rc = query_projection(i)) != POEM_OK || (rc = cross_attentions(i)) != POEM_OK || (rc = vector_self(i)) != POEM_OK ||
          (rc = vector_cross(i)) != POEM_OK || (rc = tail(i, &done)) != POEM_OK)
        return rc;
      if (done) break;
    }
    return join();
  }
};

}  // namespace

// Decoder: p.xyz[0] holds the initial normalised query coordinates; writes p.xyz[1..nblocks].
int run_decoder(poem_handle_t h, Plan& p, const float* feats_in, const float* pt_xyz, const float* pt_feats, int B, float* pose_aa,
                float* betas, hipStream_t s, bool template_queries) {
  DecoderRun run(h, p, feats_in, pt_xyz, pt_feats, B, pose_aa, betas, s, template_queries);
  return run.run();
}

// The workspace plan of a forward (caller-owned memory, bump-allocated: poem_workspace_bytes / make_plan) and the registry of
// debug taps over it.
#include "engine.h"

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
  p.h2 = a.take<float>(VS * C / 2);
  p.bps_feat = a.take<float>(BS * C);
  // A forward runs ONE sampling front end: the operator sequence (g, h1, mm, mh, y: 2.2 GB at 32 x 8 views, C = 256) or the
  // fused kernels of merge.hip (xt, ptab, q1: 0.2 GB) -- the two sets share one region (the switch can flip between forwards
  // on the same workspace, so the plan holds the larger)
  const size_t front = a.off;
  p.g = a.take<float>(VS * C);
  p.h1 = a.take<float>(VS * C);
  p.mm = a.take<float>(BS * C / 2);
  p.mh = a.take<float>(BS * C / 2);
  p.y = a.take<float>(BS * C);
  const size_t end_ops = a.off;
  a.off = front;
  p.xt = a.take<float>((size_t)BN * C * HW);
  p.ptab = a.take<float>(VS * 8);
  p.q1 = a.take<float>(BS * C);
  a.off = std::max(a.off, end_ops);
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
  p.mano_verts = a.take<float>((size_t)B * 778 * 3);
  p.mano_joints = a.take<float>((size_t)B * 21 * 3);
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
  if (c.petr_embedding) {
    p.petr_f = a.take<float>((size_t)BN * 3 * c.depth_num * HW);
    p.petr_h = a.take<float>((size_t)BN * 2 * C * HW);
    p.petr_tab = a.take<float>((size_t)BN * C * HW);
  }
  p.bytes = align_up(a.off, 256);
  return p;
}
void register_taps(poem_handle_t h, const Plan& p, int B, int BN, bool sampling) {
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

// ---- whole path: workspace -------------------------------------------------------------------------------------------
size_t poem_workspace_bytes(poem_handle_t h, int batch, int total_views) {
  if (!h || batch <= 0 || total_views < batch || h->cfg.nblocks > 8) return 0;
  return make_plan(h->cfg, batch, total_views, nullptr).bytes;
}
}  // extern "C"

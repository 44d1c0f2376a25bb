"""Lab: how far do the block-0 vector attentions move when their positional terms are computed once from the exact
normalised template t/r instead of each sample's ((c + t) - c)/r (which differs from t/r by the rounding of c + t)?
CPU, inside the oracle; compares against the reference fixture of a release shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import poem_oracle as O
from util import case_setup, load_golden

name = sys.argv[1] if len(sys.argv) > 1 else "medium"
z, meta = load_golden(name)
cfg, w, consts, batch = case_setup(meta["spec"])
feat, metas, rj = batch["mlvl_feat"], batch["img_metas"], batch["reference_joints"]
def run():
    with torch.no_grad():
        return O.head_forward(w, cfg, consts, feat, metas["cam_intr"], metas["cam_extr"], metas["cam_view_num"], rj)["all_coords_preds"]
base = run()
canon = (consts["template"] / cfg.radius)
orig_self, orig_cross = O.vec_attn_self, O.vec_attn_cross
state = {"n": 0}
def vs(w_, pre, xyz, feats, idx, nxyz):
    if ".0.encoder" in pre: xyz = canon[None].expand_as(xyz)
    return orig_self(w_, pre, xyz, feats, idx, nxyz)
def vc(w_, pre, xyz, qf, pf, idx, nxyz, hoist=False):
    if ".0.encoder" in pre: xyz = canon[None].expand_as(xyz)
    return orig_cross(w_, pre, xyz, qf, pf, idx, nxyz, hoist=hoist)
O.vec_attn_self, O.vec_attn_cross = vs, vc
hoisted = run()
ref = torch.from_numpy(z["all_coords_preds"])
def mp(a, b): return float(torch.norm(a[-1, :, 21:] - b[-1, :, 21:], dim=-1).mean() * 1e3)
print(name, "MPVPE mm: oracle vs ref", mp(base, ref), " hoisted vs ref", mp(hoisted, ref), " hoisted vs oracle", mp(hoisted, base))

import sys, os
ROOT='/root/repo'
for p in (ROOT, ROOT+'/oracle', ROOT+'/tests'): sys.path.insert(0,p)
import torch
from util import batch_to, build_hip_head, case_setup, load_golden, run_oracle, stage_report
z, meta = load_golden("medium_g6"); spec = meta["spec"]
cfg, w, consts, batch = case_setup(spec)
head = build_hip_head(spec, "cuda:0")
feat, metas, rj = batch_to(batch, "cuda:0")
eng = head._engine_for(torch.device("cuda:0")); eng.enable_taps(True)
B,C,Q=2,256,799
def run(tag, **opts):
    for k,v in opts.items(): eng.set_option(k,v)
    with torch.no_grad(): out = head(feat, metas, rj)["all_coords_preds"].cpu()
    taps = {k: eng.tap(k, (B,Q,C if 'xyz' not in k else 3)).cpu() for k in ("b1.feats","b2.h_cross","b2.f_self","b2.f_cross","b1.xyz")}
    idx = {k: eng.tap(k, (B,Q,32), torch.int32).cpu() for k in ("b2.idx_self","b2.idx_cross")}
    return out, taps, idx
base = run("base")
ref_fs = torch.from_numpy(z["tap.b2.f_self"])
step = (Q + ref_fs.shape[1] - 1)//ref_fs.shape[1]
print("default: f_self err", float((base[1]["b2.f_self"][:, ::step]-ref_fs).abs().max()))
for name, opts in (("again", {}), ("graphs0", dict(graphs=0)), ("tile2", dict(graphs=0, chain_tile=2)), ("tile1", dict(chain_tile=1)), ("tile3", dict(chain_tile=3)), ("tile0 graphs1", dict(chain_tile=0, graphs=1)), ("overlap0", dict(overlap=0)), ("chains0", dict(overlap=1, chains=0))):
    o = run(name, **opts)
    print(name, "out equal", torch.equal(o[0], base[0]), {k: (torch.equal(o[1][k], base[1][k]), float((o[1][k]-base[1][k]).abs().max())) for k in o[1]}, {k: torch.equal(o[2][k], base[2][k]) for k in o[2]},
          "f_self err vs ref", float((o[1]["b2.f_self"][:, ::step]-ref_fs).abs().max()))

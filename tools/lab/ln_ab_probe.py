"""A/B of two builds of libpoem_hip.so on one fixture: per stage tap, max |A - B| (run as: POEM_HIP_LIB=<so> python ... dump <file>; then diff)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + '/oracle', ROOT + '/tests'): sys.path.insert(0, p)
import torch
if sys.argv[1] == "dump":
    from util import batch_to, build_hip_head, case_setup, load_golden
    z, meta = load_golden(sys.argv[2]); spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, "cuda:0")
    feat, metas, rj = batch_to(batch, "cuda:0")
    eng = head._engine_for(torch.device("cuda:0")); eng.enable_taps(True)
    with torch.no_grad(): out = head(feat, metas, rj)["all_coords_preds"].cpu()
    B, C, Q = len(spec["views"]), spec["embed"], 799
    taps = {f"b{i}.{k}": eng.tap(f"b{i}.{k}", (B, Q, 3 if k == "xyz" else C)).cpu() for i in range(3) for k in ("h_cross", "f_self", "f_cross", "feats", "xyz")}
    taps["out"] = out
    torch.save(taps, sys.argv[3])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k] - b[k]).abs()
        print(f"{k:12s} scale {float(a[k].abs().max()):10.3e}  max|A-B| {float(d.max()):10.3e}  rel {float(d.max() / a[k].abs().max()):9.2e}")

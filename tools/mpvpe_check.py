"""Development tool: MPVPE (mm) of the HIP head vs the reference fixtures, per release shape (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from util import batch_to, build_hip_head, case_setup, load_golden

for name in sys.argv[1:] or ["small", "medium", "large", "huge", "ragged"]:
    z, meta = load_golden(name)
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, "cuda:0")
    feat, metas, rj = batch_to(batch, "cuda:0")
    with torch.no_grad():
        got = head(feat, metas, rj)["all_coords_preds"].cpu()
    ref = torch.from_numpy(z["all_coords_preds"])
    mp = torch.norm(got[-1, :, 21:] - ref[-1, :, 21:], dim=-1).mean(dim=1) * 1e3
    print(f"{name:10s} MPVPE vs reference fixture: {[round(float(v), 6) for v in mp]} mm   max |diff| {float((got - ref).abs().max()) * 1e3:.6f} mm", flush=True)

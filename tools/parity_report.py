"""Stage-level parity report of the default HIP path against the reference fixtures (run on the GPU box; the committed
output is profiles/<tag>_parity.txt).  Per fixture: MPVPE of every decoder layer, HIP vs reference next to oracle vs
reference; per block and stage the path's distance from the reference's own tensor on the rows whose neighbour sets are the
reference's, next to the oracle's; the neighbour-set agreement of blocks 1, 2 and the relative 32nd / 33rd distance gap of
every query that picked another set.  Same helpers as tests/test_hip_parity.py::test_release_shape_stage_taps_vs_reference."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from util import batch_to, build_hip_head, case_setup, load_golden, run_oracle, stage_report  # noqa: E402

names = sys.argv[1:] or ["small", "medium", "large", "huge", "ragged", "mediummano", "small_hot", "medium_hot", "large_hot", "medium_g1", "medium_g4", "medium_g6"]
curve = []
for name in names:
    z, meta = load_golden(name)
    spec = meta["spec"]
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, "cuda:0")
    feat, metas, rj = batch_to(batch, "cuda:0")
    eng = head._engine_for(torch.device("cuda:0"))
    eng.enable_taps(True)
    with torch.no_grad():
        got = head(feat, metas, rj)["all_coords_preds"].cpu()
    taps = {}
    orc = run_oracle(cfg, w, consts, batch, taps=taps)["all_coords_preds"]
    ref = torch.from_numpy(z["all_coords_preds"])
    mp = lambda a, b, l: float(torch.norm(a[l, :, 21:] - b[l, :, 21:], dim=-1).mean()) * 1e3   # noqa: E731
    print(f"== {name}: C = {spec['embed']}, views {spec['views']}, gain {spec.get('gain', 1.0)}"
          + (", reference neighbour distances rounded as pytorch3d's CUDA kernel (fma)" if spec.get("knn_fma") else ""))
    if spec["embed"] == 256 and spec["views"] == [8, 4]:
        curve.append((spec.get("gain", 1.0), bool(spec.get("knn_fma")), float(ref[-1].abs().max()), mp(got, ref, 2), mp(orc, ref, 2)))
    for layer in range(3):
        print(f"   layer {layer}: MPVPE HIP vs reference {mp(got, ref, layer):.3e} mm | oracle vs reference {mp(orc, ref, layer):.3e} mm"
              f" | HIP vs oracle {mp(got, orc, layer):.3e} mm")
    if spec["parametric"] or "tap.b0.h_cross" not in z.files:
        continue
    rep = stage_report(z, spec, lambda n, shp, dt=torch.float32: eng.tap(n, shp, dt).cpu(), taps)
    for k, v in rep["neighbours"].items():
        gaps = ", ".join(f"{g:.1e}" for _, _, g in v["flips"][:6])
        print(f"   {k:10s} neighbour sets equal to the reference's: {100 * v['set_equal']:.3f} %  ({len(v['flips'])} other sets; rel. gap of their 32nd / 33rd distance: {gaps})")
    for k, v in rep["stages"].items():
        print(f"   {k:12s} scale {v['scale']:9.3e}  clean rows {100 * v['clean_rows']:6.2f} %  path {v['path_clean']:.2e}  oracle {v['oracle_clean']:.2e}"
              f"   (all rows: path {v['path_all']:.2e}  oracle {v['oracle_all']:.2e})")
if curve:
    print("== conditioning curve (POEM-medium, views [8, 4], seed 22; last decoder layer): gain of the block Linears | reference "
          "rounding | max |xyz| of the reference (m) | MPVPE HIP vs reference (mm) | CPU restatement vs reference (mm)")
    for gain, fma, mag, a, b in sorted(curve):
        print(f"   gain {gain:3.1f} | {'cuda (fma)' if fma else 'cpu       '} | {mag:9.3f} | {a:.3e} | {b:.3e}"
              + ("   <- above the 1e-3 mm bar for BOTH: two correct fp32 evaluations no longer agree to it" if min(a, b) > 1e-3 else ""))

#!/usr/bin/env python
"""Handle life cycles in one process (lab): build a head, run forwards (graph capture + replay), drop it, repeat -- the
pattern of the GPU test suite.  Run under LD_PRELOAD=tools/lab/segv_bt.so for a native backtrace if the runtime crashes."""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

from util import build_hip_head, batch_to, case_setup  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
opts = [kv.split("=") for kv in sys.argv[2:]]
shapes = [(128, [2, 3]), (256, [8, 2]), (512, [10]), (256, [3, 10, 1, 6]), (128, [1])]
for i in range(N):
    C, views = shapes[i % len(shapes)]
    spec = dict(embed=C, nsample=4096, views=views, seed=100 + i, parametric=False)
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, "cuda:0")
    feat, metas, rj = batch_to(batch, "cuda:0")
    with torch.no_grad():
        head(feat, metas, rj)
        for k, v in opts:
            head._engine.set_option(k, int(v))
        for mode in ("split_f16x3", "fp32", "fp32"):
            head.set_precision(mode)
            out = head(feat, metas, rj)["all_coords_preds"]
    torch.cuda.synchronize()
    print(i, C, views, float(out.abs().max()), flush=True)
    del head, out
    gc.collect()
print("done")

#!/usr/bin/env python
"""CPU probe: what would split-precision matrix-core arithmetic cost in accuracy?  Emulates the three CxC per-neighbour
GEMMs of the vector attention (fc_delta.2, fc_gamma.0, fc_gamma.2 -- 51 % of a step) as hi/lo splits on the f16 / bf16
MFMA with fp32 accumulation, inside the oracle, and reports MPVPE against the plain-fp32 oracle.  Test infrastructure /
lab only -- nothing in the product path uses this."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import poem_oracle as po  # noqa: E402
from util import case_setup, run_oracle  # noqa: E402


def split(x, dt, parts):
    out, r = [], x
    for _ in range(parts):
        h = r.to(dt).float()
        out.append(h)
        r = r - h
    return out


def make_linear(dt, parts, terms, every=False):
    def lin(x, w, b=None):
        if not every and (w.shape[0] != w.shape[1] or x.dim() != 4):   # only the per-(query, neighbour) CxC GEMMs
            return F.linear(x, w, b)
        if w.shape[1] < 16:                                        # 3 -> C first layers stay fp32 in any design
            return F.linear(x, w, b)
        xs, ws = split(x, dt, parts), split(w, dt, parts)
        y = None
        for i in range(parts):
            for j in range(parts):
                if i + j < terms:
                    t = F.linear(xs[i], ws[j])
                    y = t if y is None else y + t
        return y if b is None else y + b
    return lin


def main():
    torch.set_num_threads(8)
    spec = dict(embed=256, nsample=4096, views=[8, 8], seed=0, parametric=False)
    cfg, w, consts, batch = case_setup(spec)
    ref = run_oracle(cfg, w, consts, batch)["all_coords_preds"]
    plain = po.linear
    every = "--all-linears" in sys.argv                            # every nn.Linear of the path, not only the vector attention
    print("scope:", "every Linear with K >= 16" if every else "vector-attention C x C GEMMs", flush=True)
    for name, dt, parts, terms in (("f16 x3 (hi*hi + hi*lo + lo*hi)", torch.float16, 2, 2), ("f16 x4", torch.float16, 2, 3),
                                   ("bf16 x3", torch.bfloat16, 2, 2), ("bf16 x6 (3-way split)", torch.bfloat16, 3, 3),
                                   ("bf16 x1", torch.bfloat16, 1, 1), ("f16 x1", torch.float16, 1, 1)):
        po.linear = make_linear(dt, parts, terms, every)
        try:
            out = run_oracle(cfg, w, consts, batch)["all_coords_preds"]
        finally:
            po.linear = plain
        d = (out[-1, :, 21:] - ref[-1, :, 21:]).norm(dim=-1).mean().item() * 1e3
        print(f"{name:34s} MPVPE vs fp32 oracle {d:.3e} mm", flush=True)


if __name__ == "__main__":
    main()

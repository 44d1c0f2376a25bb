#!/usr/bin/env python
"""Lab: stand-alone timing of the split-precision vector attention (poem_vector_attention_split) at the bench shape
(B=32, Q=799, C=256; self: NS=799, cross: NS=4096) next to the exact kernel.  POEM_HIP_LIB selects the build (A/B runs of
kept .so files on one box).  Not part of the product path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import poem_v2_amd as pk  # noqa: E402
from poem_v2_amd import hip  # noqa: E402


def main():
    C = int(os.environ.get("C", 256))
    B, Q = int(os.environ.get("B", 32)), 799
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)   # noqa: E731
    for NS in (799, 4096):
        qxyz, sxyz = r(B, Q, 3), r(B, NS, 3)
        idx = torch.randint(0, NS, (B, Q, 32), generator=g, dtype=torch.int32).to(dev)
        q, k, v = r(B, Q, C), r(B, NS, C), r(B, NS, C)
        wd1, bd1 = r(C, 3), r(C) * 0.1
        W = [r(C, C) / C ** 0.5 for _ in range(3)]
        b = [r(C) * 0.1 for _ in range(3)]
        imgs = [hip.pack_split_linear(w) for w in W]
        scales = torch.cat([s for _, s in imgs])
        packed = [hip.pack_linear(w) for w in W]

        def split():
            return hip.vector_attention_split(qxyz, sxyz, None, idx, q, k, v, wd1, bd1, imgs[0][0], b[0], imgs[1][0], imgs[2][0], scales)

        def exact():
            return hip.vector_attention(qxyz, sxyz, None, idx, q, k, v, wd1, bd1, packed[0], b[0], packed[1], b[1], packed[2], b[2])

        for name, fn in (("split", split), ("exact", exact)):
            if name == "exact" and os.environ.get("SKIP_EXACT"):
                continue
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"{os.path.basename(hip.LIB_PATH)} C={C} NS={NS} {name}: {e0.elapsed_time(e1) / n:.4f} ms", flush=True)


if __name__ == "__main__":
    main()

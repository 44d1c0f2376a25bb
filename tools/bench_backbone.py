#!/usr/bin/env python
"""A/B of the PyTorch-ROCm HRNet (E2E scope plumbing): memory format and MIOpen find mode.  Not part of the hot path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import poem_v2_amd as pk  # noqa: E402,F401
from poem_v2_amd.backbone import HRNet, seeded_hrnet_state_dict  # noqa: E402

dev = torch.device("cuda:0")
views = int(sys.argv[1]) if len(sys.argv) > 1 else 256
img = pk.inputs.synthetic_images(views, seed=1).to(dev)
for fmt in ("contiguous", "channels_last"):
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        net = HRNet(state_dict=seeded_hrnet_state_dict(0), device=dev)
        x = img
        if fmt == "channels_last":
            x = img.contiguous(memory_format=torch.channels_last)
            for c in net._convs.values():
                c.weight = c.weight.contiguous(memory_format=torch.channels_last)
        t0 = time.perf_counter()
        net(x)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ys = net(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{fmt:14s} miopen-benchmark={bench!s:5s} first call {first:6.1f} s, steady {dt * 1e3:7.1f} ms / {views} images "
              f"({views / dt:7.0f} img/s)  out0 {tuple(ys[0].shape)} {ys[0].is_contiguous()}", flush=True)

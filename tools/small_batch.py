#!/usr/bin/env python
"""Small-per-GPU-batch regime of the head path (the reference's evaluation default is --val_batch_size 2,
lib/opt.py:27-30 upstream; BASELINE configs[4] read as a global batch is 8 samples per GPU): samples/s and ms per forward
of POEM-<model>, N views, for each batch size, on resident inputs.

  python tools/small_batch.py                          # B = 1 2 4 8 16 32, medium, 8 views
  python tools/small_batch.py --batches 2 --steps 30   # one size (what tools/run/gpu_small.sh traces)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import poem_v2_amd as pk  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4, 8, 16, 32])
    ap.add_argument("--model", default="medium")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--ragged", action="store_true", help="views per sample ~ U{2..10} (seed 5)")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE")
    ap.add_argument("--streams", type=int, default=1, help="issue consecutive forwards round-robin on this many torch streams")
    ap.add_argument("--sync-each", action="store_true", help="also time with a device sync after every forward (latency)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    C = pk.weights.MODEL_EMBED[args.model]
    out = {}
    for B in args.batches:
        views = [args.views] * B
        if args.ragged:
            views = np.random.RandomState(5).randint(2, 11, size=B).tolist()
        head, batches, _ = bench.make_leg(C, views, False, dev, 0, rotate=4, seed0=3000)
        with torch.no_grad():
            head(*batches[0][:3])
        for kv in args.option:
            k_, v_ = kv.split("=")
            head.set_option(k_, int(v_))
        if args.streams > 1:
            sec = bench.time_leg_streams(head, batches, steps=args.steps, warmup=args.warmup, nstreams=args.streams)
        else:
            sec = bench.time_leg(head, batches, steps=args.steps, warmup=args.warmup)
        rec = {"ms_per_forward": sec * 1e3, "samples_per_s": B / sec, "views_total": int(sum(views))}
        if args.sync_each:
            with torch.no_grad():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    head(*batches[i % len(batches)][:3])
                    torch.cuda.synchronize()
                rec["ms_per_forward_synced"] = (time.perf_counter() - t0) / args.steps * 1e3
            # host time to enqueue one forward (the stream is kept busy, so this is pure host cost)
            with torch.no_grad():
                t0 = time.perf_counter()
                for i in range(args.steps):
                    head(*batches[i % len(batches)][:3])
                rec["host_enqueue_ms"] = (time.perf_counter() - t0) / args.steps * 1e3
                torch.cuda.synchronize()
        out[f"B{B}"] = rec
        print(f"B={B:3d}  {rec['ms_per_forward']:8.3f} ms/forward  {rec['samples_per_s']:8.1f} samples/s"
              + (f"  synced {rec['ms_per_forward_synced']:.3f} ms  host enqueue {rec['host_enqueue_ms']:.3f} ms" if args.sync_each else ""),
              flush=True)
        del head, batches
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

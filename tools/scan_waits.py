#!/usr/bin/env python
"""Scan the gfx950 ISA of csrc/*.hip for vector-memory waits that can serialise on STORES: on gfx9 `s_waitcnt vmcnt(N)` counts
loads and stores alike, in order, so a load issued behind a store cannot be waited for without waiting for the store's
acknowledgement (round 6: the F1 GEMM's K-image epilogue was 32 such round trips in a row).  Per kernel: every basic block in which
a store is followed by a load and a vmcnt wait smaller than the number of VMEM operations issued since that store."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "poem-v2_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
for f in files:
    out = f"/tmp/scan_{f}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out,
                    os.path.join(CSRC, f)], check=True, capture_output=True)
    text = open(out).read()
    for m in re.finditer(r"\n(_Z[^\n:]+):[^\n]*\n(.*?)s_endpgm", text, re.S):
        name, body = m.group(1), m.group(2)
        hits, since_store, pending = 0, None, 0
        nmfma = body.count("v_mfma")
        for line in body.split("\n"):
            l = line.strip()
            if re.match(r"^\.LBB", l):
                since_store = None
            if re.match(r"^(global|buffer|flat|scratch)_store", l):
                since_store = 0
            elif re.match(r"^(global|buffer|flat)_load", l):
                if since_store is not None:
                    since_store += 1
            elif l.startswith("s_waitcnt") and "vmcnt" in l and since_store:
                n = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
                if n < since_store:      # waits for a load issued behind the store -> waits for the store
                    hits += 1
                    since_store = None
        if hits:
            print(f"{f:18s} {hits:4d} store-then-load waits   mfma {nmfma:5d}   {name[:90]}")

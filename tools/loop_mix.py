#!/usr/bin/env python
"""Instruction mix of the MFMA-carrying basic blocks of one kernel (ISA from hipcc -S): per block the counts of MFMA, other VALU,
LDS, vector-memory, scalar, s_waitcnt (and which counters / values) and s_nop -- where the non-MFMA issue slots of a loop go."""
import re
import subprocess
import sys
import os
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], sys.argv[2]
out = "/tmp/loop_mix.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", out,
                os.path.join(ROOT, "poem-v2_amd", "csrc", src)], check=True, capture_output=True)
text = open(out).read()
for m in re.finditer(r"\n(_Z[^\n:]+):[^\n]*\n(.*?)s_endpgm", text, re.S):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    print(name[:110])
    blocks, cur, label = [], [], "entry"
    for line in body.split("\n"):
        l = line.strip()
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((label, cur)); label, cur = l.split(":")[0], []
        elif l and not l.startswith(";") and not l.startswith("."):
            cur.append(l)
    blocks.append((label, cur))
    for label, ins in blocks:
        n = sum(i.startswith("v_mfma") for i in ins)
        if n < 8:
            continue
        c = Counter()
        waits = []
        for i in ins:
            op = i.split()[0]
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith("v_"): c["valu"] += 1
            elif op.startswith("ds_"): c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
            elif op == "s_waitcnt": c["wait"] += 1; waits.append(i.replace("s_waitcnt ", ""))
            elif op == "s_nop": c["nop"] += 1
            elif op.startswith("s_"): c["salu"] += 1
        top = Counter(i.split()[0] for i in ins if i.startswith("v_") and not i.startswith("v_mfma")).most_common(6)
        print(f"  {label:12s} {dict(c)}  valu top: {top}\n      waits: {waits[:14]}")

"""Turn gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the committed profiles/<tag>_* files:
   <tag>_bench.json               the bench line of the profiled run
   <tag>_bench_kernel_stats.csv   rocprofv3 --stats per-kernel summary
   <tag>_pmc.json                 per-kernel HBM bytes per launch (FETCH_SIZE x2 correction, WRITE_SIZE) and MFMA busy"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return {k: {"launches": n, "avg": v / n} for k, (n, v) in acc.items()}


out = {"round_tag": tag,
       "commands": ["rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --steps 1 --warmup 1 --headline-only",
                    "rocprofv3 --kernel-trace --pmc WRITE_SIZE -- (same)",
                    "rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- (same)"],
       "correction": "gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM): read bytes = 2 x "
                     "FETCH_SIZE[KiB] x 1024; WRITE_SIZE[KiB] x 1024 taken as is",
       "kernels": {}}
f = glob.glob(os.path.join(src, "pmc_fetch", "**", "*counter_collection.csv"), recursive=True)
w = glob.glob(os.path.join(src, "pmc_write", "**", "*counter_collection.csv"), recursive=True)
m = glob.glob(os.path.join(src, "pmc_mfma", "**", "*counter_collection.csv"), recursive=True)
fetch = per_kernel(f[0], "FETCH_SIZE") if f else {}
write = per_kernel(w[0], "WRITE_SIZE") if w else {}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    e = {"launches": fetch.get(k, write.get(k))["launches"]}
    if k in fetch:
        e["read_bytes_per_launch"] = 2.0 * fetch[k]["avg"] * 1024
    if k in write:
        e["write_bytes_per_launch"] = write[k]["avg"] * 1024
    if k in fetch and k in write:
        e["hbm_bytes_per_launch"] = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
    out["kernels"][k] = e
if m:
    busy, mf, gui = per_kernel(m[0], "SQ_BUSY_CYCLES"), per_kernel(m[0], "SQ_VALU_MFMA_BUSY_CYCLES"), per_kernel(m[0], "GRBM_GUI_ACTIVE")
    for k, e in out["kernels"].items():
        if k in mf and k in gui and gui[k]["avg"] > 0:
            e["SQ_VALU_MFMA_BUSY_CYCLES"] = mf[k]["avg"]
            e["GRBM_GUI_ACTIVE"] = gui[k]["avg"]
            if k in busy:
                e["SQ_BUSY_CYCLES"] = busy[k]["avg"]
# round 6: issue / wait accounting and the L2 -> memory write requests (tools/collect_pmc.sh passes pmc_issue, pmc_wr), and the
# derived figures the reviews quote: MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 GRBM_GUI_ACTIVE)
for sub in ("pmc_issue", "pmc_wr"):
    g = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    if not g:
        continue
    names = sorted({r["Counter_Name"] for r in csv.DictReader(open(g[0]))})
    for cn in names:
        pk = per_kernel(g[0], cn)
        for k, e in out["kernels"].items():
            if k in pk:
                e[cn] = pk[k]["avg"]
for k, e in out["kernels"].items():
    if e.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * e["GRBM_GUI_ACTIVE"]), 4)
    if e.get("SQ_INSTS_MFMA"):
        e["valu_per_mfma"] = round((e.get("SQ_INSTS_VALU", 0.0) - e["SQ_INSTS_MFMA"]) / e["SQ_INSTS_MFMA"], 3)      # SQ_INSTS_VALU counts the MFMAs too
ar = os.path.join(src, "args.txt")
if os.path.exists(ar):
    out["bench_args"] = open(ar).read().strip() or "(headline)"
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
st = glob.glob(os.path.join(src, "stats_full", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(dst, f"{tag}_bench_full_kernel_stats.csv"))
bf = os.path.join(src, "bench_full.json")
if os.path.exists(bf):
    lines = [l for l in open(bf).read().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(dst, f"{tag}_bench_full.json"), "w").write(lines[-1] + "\n")
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    lines = [l for l in open(bj).read().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(dst, f"{tag}_bench.json"), "w").write(lines[-1] + "\n")
va = [k for k in out["kernels"] if k.startswith("vecattn_kernel")]
print(json.dumps({k: out["kernels"][k] for k in va}, indent=1))

"""Timeline of ONE default-path forward out of a rocprofv3 --kernel-trace CSV of `python bench.py` (the n-th forward,
default 8th = inside the timed region of the fp32 leg): per-kernel totals + (start us, duration us, queue, kernel)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a forward starts with input_proj (conv1x1_lds_kernel; conv1x1_kernel for shapes it does not take) -- or, when the block-0
# anchor tables are rebuilt per forward (poem_set_option tables_cached=0), with the table fork (canon_xyz_kernel)
names = [r["Kernel_Name"].replace("void ", "") for r in rows]
n_canon = sum(n.startswith("canon_xyz_kernel") for n in names)
first = "canon_xyz_kernel" if n_canon > 4 else ("conv1x1_lds_kernel" if any(n.startswith("conv1x1_lds_kernel") for n in names) else "conv1x1_kernel")
starts = [i for i, n in enumerate(names) if n.startswith(first)]
step = rows[starts[nth]:starts[nth + 1]]
last = max(j for j, r in enumerate(step) if r["Kernel_Name"].startswith("finalize_kernel"))
step = step[:last + 1]
t0 = int(step[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in step)
qs = {}
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:50]
print(f"# kernel timeline of one 32-sample step of the default path (rocprofv3 --kernel-trace of `python bench.py`, forward {nth + 1};")
print("# us from step start, duration us, HIP queue, kernel)")
print(f"# wall {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels")
tot = defaultdict(lambda: [0, 0.0])
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[name(r)][0] += 1
    tot[name(r)][1] += d
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"#   {n:50s} x{c:3d}  {d / 1e3:8.3f} ms")
for r in step:
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs) + 1)
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} q{q}   {name(r)}")

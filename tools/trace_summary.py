"""Summarise a rocprofv3 --kernel-trace CSV: per-step timeline (last step), per-kernel totals, stream overlap."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last step: starts at the last conv1x1_kernel<4> launch that is followed by project/prep
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void conv1x1_kernel") ]
i0 = starts[-1]
step = rows[i0:]
# cut at finalize_kernel
for j, r in enumerate(step):
    if r["Kernel_Name"].startswith("finalize_kernel"):
        step = step[:j + 1]
        break
t0 = int(step[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in step)
print(f"last step: {len(step)} kernels, wall {(t1 - t0) / 1e6:.3f} ms")
tot = defaultdict(lambda: [0, 0.0])
busy = 0.0
for r in step:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[n][0] += 1
    tot[n][1] += d
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:62s} x{c:3d}  {d / 1e3:8.3f} ms")
print(f"  sum of kernel durations {sum(v[1] for v in tot.values()) / 1e3:.3f} ms")
if len(sys.argv) > 2:
    for r in step:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:50]
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} q{r.get('Queue_Id', '?'):>3} {n}")

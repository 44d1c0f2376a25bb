"""Per-kernel averages of arbitrary PMC counters out of a rocprofv3 --pmc counter_collection CSV (development tool).
   python tools/pmc_kernels.py <counter_collection.csv> [kernel-name substring ...]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:56]
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    e = acc[k][r["Counter_Name"]]
    e[0] += 1
    e[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k.ljust(58), "  ".join(f"{c}={v / n:.4g}" for c, (n, v) in sorted(cs.items())), f"(launches {max(n for n, _ in cs.values())})")

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/decode_prof; mkdir -p $O
python tools/bench_ops.py decode 2>&1 | grep -v amdgpu.ids | tee $O/bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o s -- python tools/bench_ops.py decode > /dev/null 2> $O/err.txt
F=$(find $O/p -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY' | tee $O/stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
rm -rf $O/p

#!/bin/bash
# N1 operators alone: op bench, rocprofv3 kernel stats, and PMC passes (MFMA busy; HBM bytes) of the stride-2 ConvBlocks.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/decode_prof; mkdir -p $O
python tools/bench_ops.py down2 decode 2>&1 | grep -v amdgpu.ids | tee $O/bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o s -- python tools/bench_ops.py decode > /dev/null 2> $O/err.txt
F=$(find $O/p -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY' | tee $O/stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
rm -rf $O/p
for pass in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc -o pmc -- python tools/bench_ops.py down2 > /dev/null 2>> $O/err.txt
  C=$(find $O/pmc -name "*counter_collection.csv" | head -1)
  python tools/pmc_kernels.py "$C" conv3x3_s2 | tee -a $O/pmc.txt
  rm -rf $O/pmc
done

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "cross_attention or batch_properties or c5_ragged" 2>&1 | tail -4
for i in 1 2 3; do
for O in 0 1; do
  timeout 300 python bench.py --steps 20 --warmup 5 --headline-only --cpu-samples 0 --option xattn_tail=$O 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('xattn_tail=$O', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms')"
done; done
O=gpurun_out/r6_tail; mkdir -p $O
for P in 0 1; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$P -o b -- python bench.py --steps 10 --warmup 3 --headline-only --cpu-samples 0 --option xattn_tail=$P > /dev/null 2> $O/err$P.txt
python - $P <<'PY'
import csv,glob,sys
f=glob.glob(f'gpurun_out/r6_tail/s{sys.argv[1]}/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'xattn' in r['Name']: print(sys.argv[1], r['Name'][:50], r['Calls'], r['AverageNs'])
PY
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_mano; mkdir -p $O
timeout 900 python -m pytest tests/test_mano.py tests/test_hip_parity.py -x -q -m gpu -k "mano or parametric or c3" 2>&1 | tail -15
for i in 1 2; do
for M in medium medium_MANO; do
  timeout 300 python bench.py --steps 20 --warmup 5 --headline-only --model $M --cpu-samples 0 2> $O/err_$M.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$M', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms')"
done; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python bench.py --steps 10 --warmup 3 --headline-only --model medium_MANO --cpu-samples 0 > $O/bench_mano.json 2> $O/err_prof.txt
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r6_mano/stats/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if any(k in n for k in ('mano','q3_','rot6d','narrow','finalize')): print(n[:60], r['Calls'], r['AverageNs'])
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete

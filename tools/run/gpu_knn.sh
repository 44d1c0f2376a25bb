#!/bin/bash
# knn.hip: parity tests + stand-alone timing + headline
set -u
OUT=gpurun_out/knn_${1:-a}
mkdir -p $OUT
python -m pytest tests/test_hip_parity.py -q -m gpu -k "knn" -x > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python tools/bench_ops.py knn > $OUT/knn.txt 2>&1; cat $OUT/knn.txt
python - > $OUT/knn_small.txt 2>&1 <<'P'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, tools.bench_ops as bo
for B in (1, 2, 8):
    bo.knn_case(B=B)
P
cat $OUT/knn_small.txt
python bench.py --headline-only > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['mpvpe_vs_oracle_mm'] if 'mpvpe_vs_oracle_mm' in d else '')"

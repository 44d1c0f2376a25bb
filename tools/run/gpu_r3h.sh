#!/bin/bash
# round 3, step h: decode tests + decode timing + small-batch table + headline
set -u
OUT=gpurun_out/r3h
mkdir -p $OUT
python -m pytest tests/test_decode.py -q -m gpu -x > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
python tools/bench_ops.py decode 2>&1 | grep -v amdgpu.ids | head -3 | tee $OUT/decode.txt
python tools/small_batch.py --batches 1 2 8 --steps 30 2>/dev/null | head -4 | tee $OUT/small.txt
python bench.py --headline-only > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"

#!/bin/bash
# round 3, step i: merged cross attention -- tests + A/B + timeline
set -u
OUT=gpurun_out/r3i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_decode.py -q -m gpu -x -k "cross_attention or full_size_batch or conv1x1_upsample2 or chain" > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt
for m in 1 0 1 0; do
python bench.py --headline-only --option xattn_merge=$m > $OUT/bench_m$m.json 2> $OUT/bench.err; python -c "
import json;d=json.loads(open('$OUT/bench_m$m.json').read().strip().splitlines()[-1]);print('xattn_merge=$m',d['value'],d['ms_per_step'],d.get('mpvpe_vs_oracle_mm'))"
done
python tools/small_batch.py --batches 1 2 8 --steps 30 2>/dev/null | head -3 | tee $OUT/small.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --headline-only > /dev/null 2> $OUT/trace.err
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T 9 > $OUT/timeline_B32.txt 2>> $OUT/trace.err
rm -rf $OUT/trace
head -24 $OUT/timeline_B32.txt

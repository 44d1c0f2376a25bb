#!/bin/bash
set -u
OUT=gpurun_out/r3x; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for late in 0 1 0 1; do
  echo "bps_late=$late"; python tools/small_batch.py --batches 1 2 4 8 --steps 40 --option bps_late=$late 2>/dev/null | head -4
done | tee $OUT/ab.txt
python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "full_size_batch or head_release or decoder_entry" 2>&1 | tail -2
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/small_batch.py --batches 2 --steps 12 --warmup 3 > /dev/null 2> $OUT/trace.err
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T 10 > $OUT/timeline_B2.txt 2>> $OUT/trace.err; rm -rf $OUT/trace
sed -n 24,50p $OUT/timeline_B2.txt

#!/bin/bash
# Small-batch regime: table over B + kernel-trace timelines at B = 2 and B = 8 (gpurun_out/small_$1/)
set -u
TAG=${1:-r03a}
OUT=gpurun_out/small_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
python tools/small_batch.py --sync-each > $OUT/table.txt 2> $OUT/table.err
python tools/small_batch.py --ragged --batches 8 --sync-each >> $OUT/table.txt 2>> $OUT/table.err
for B in 1 2 8; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_B$B -o t -- python tools/small_batch.py --batches $B --steps 12 --warmup 3 > $OUT/trace_B$B.txt 2> $OUT/trace_B$B.err
  T=$(find $OUT/trace_B$B -name "*kernel_trace.csv" | head -1)
  python tools/step_timeline.py $T 10 > $OUT/timeline_B$B.txt 2>> $OUT/trace_B$B.err
  rm -rf $OUT/trace_B$B
done
cat $OUT/table.txt
head -30 $OUT/timeline_B2.txt

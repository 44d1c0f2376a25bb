#!/bin/bash
# Kernel-trace timelines of one forward at the given batch sizes: tools/run/gpu_trace_small.sh TAG "1 2" [--option k=v ...]
set -u
TAG=${1:-t}; BS=${2:-"1 2"}; shift 2
OUT=gpurun_out/small_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
for B in $BS; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_B$B -o t -- python tools/small_batch.py --batches $B --steps 12 --warmup 3 "$@" > $OUT/trace_B$B.txt 2> $OUT/trace_B$B.err
  T=$(find $OUT/trace_B$B -name "*kernel_trace.csv" | head -1)
  python tools/step_timeline.py $T 10 > $OUT/timeline_B$B.txt 2>> $OUT/trace_B$B.err
  rm -rf $OUT/trace_B$B
done

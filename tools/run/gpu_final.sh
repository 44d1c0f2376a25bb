#!/bin/bash
# final numbers of a round: the driver's bench invocation unprofiled (wall time recorded) + a kernel timeline of one step
set -u
TAG=${1:-r03}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T0=$(date +%s)
python bench.py > $OUT/bench_unprofiled.json 2> $OUT/bench.err
echo "bench wall $(( $(date +%s) - T0 )) s" | tee $OUT/wall.txt
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python bench.py --headline-only > /dev/null 2> $OUT/trace.err
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T 9 > $OUT/timeline_B32.txt 2>> $OUT/trace.err
rm -rf $OUT/trace
head -24 $OUT/timeline_B32.txt
tail -c 600 $OUT/bench_unprofiled.json

#!/bin/bash
# round 6, final numbers: the driver's invocation unprofiled, kernel stats, PMC (headline, c4, c5), step timeline, small batches
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
F=gpurun_out/final_r06; mkdir -p $F
T0=$(date +%s)
python bench.py > $F/bench_unprofiled.json 2> $F/bench.err
echo "bench wall $(( $(date +%s) - T0 )) s" | tee $F/wall.txt
bash tools/collect_profiles.sh r06 > $F/collect.log 2>&1
bash tools/collect_pmc.sh r06_c4 --model large --views 10 --batch 16 > /dev/null 2>&1
bash tools/collect_pmc.sh r06_c5 --views-range 2 10 --batch 64 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $F/trace -o t -- python bench.py --headline-only --cpu-samples 0 > /dev/null 2> $F/trace.err
T=$(find $F/trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T 9 > $F/timeline_B32.txt 2>> $F/trace.err
rm -rf $F/trace
python tools/small_batch.py --batches 1 2 4 8 16 32 --steps 30 --warmup 5 > $F/small_batch_table.txt 2> $F/small.err
bash tools/run/gpu_trace_small.sh r06 "1 2 8" > /dev/null 2>&1
tail -c 1500 $F/bench_unprofiled.json; cat $F/small_batch_table.txt | head -8

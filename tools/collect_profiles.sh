#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of the bench (headline leg alone, and the default command
# with every leg) + the PMC passes of tools/collect_pmc.sh on the headline.  Everything lands in gpurun_out/prof_$1/ ;
# tools/pmc_summary.py turns it into the profiles/ files that are committed.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
# (i) the headline leg alone: the per-kernel averages of --stats are then the headline's (the default command also runs the
#     opt-in / A-B / extra-config / E2E legs, whose launches of the same kernels have other shapes and modes)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --headline-only > $OUT/bench.json 2> $OUT/bench.err
# (ii) the default command, every leg (the driver's invocation)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_full -o bench -- python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
rm -f $OUT/stats_full/*kernel_trace.csv $OUT/stats/*kernel_trace.csv
# (iii) counters, separate passes
bash tools/collect_pmc.sh $TAG > /dev/null 2>&1
find $OUT -name "*.db" -delete
ls $OUT | head -30
tail -2 $OUT/bench.json | cut -c1-400

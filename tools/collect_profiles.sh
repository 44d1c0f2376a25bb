#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of the default bench + two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Everything lands in
# gpurun_out/prof_$1/ ; tools/pmc_summary.py turns it into the profiles/ files that are committed.
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
rm -f $OUT/stats_full/*kernel_trace.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python bench.py --steps 1 --warmup 1 --headline-only --option graphs=0 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python bench.py --steps 1 --warmup 1 --headline-only --option graphs=0 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- python bench.py --steps 1 --warmup 1 --headline-only --option graphs=0 > /dev/null 2> $OUT/pmc_mfma.err
find $OUT -name "*.db" -delete
ls -la $OUT $OUT/*/ | head -40
tail -2 $OUT/bench.json

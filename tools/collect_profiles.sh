#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of the default bench + two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Everything lands in
# gpurun_out/prof_$1/ ; tools/pmc_summary.py turns it into the profiles/ files that are committed.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python bench.py --steps 1 --warmup 1 --cpu-samples 0 --no-e2e > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python bench.py --steps 1 --warmup 1 --cpu-samples 0 --no-e2e > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- python bench.py --steps 1 --warmup 1 --cpu-samples 0 --no-e2e > /dev/null 2> $OUT/pmc_mfma.err
find $OUT -name "*.db" -delete
ls -la $OUT $OUT/*/ | head -40
tail -2 $OUT/bench.json

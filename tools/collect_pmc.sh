#!/bin/bash
# The three PMC passes of tools/collect_profiles.sh alone (HBM bytes and MFMA-busy per kernel of the headline leg): gpurun_out/prof_$1/
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python bench.py --steps 1 --warmup 1 --headline-only --option graphs=0 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python bench.py --steps 1 --warmup 1 --headline-only --option graphs=0 > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pmc -- python bench.py --steps 1 --warmup 1 --headline-only --option graphs=0 > /dev/null 2> $OUT/pmc_mfma.err
find $OUT -name "*.db" -delete
ls $OUT/*/ | head

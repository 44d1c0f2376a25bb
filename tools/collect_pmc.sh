#!/bin/bash
# PMC passes of one bench configuration (HBM bytes, MFMA-busy, issue / wait accounting per kernel): gpurun_out/prof_$TAG/
#   tools/collect_pmc.sh TAG [bench.py arguments of the configuration, default: the headline]
# Separate passes (FETCH_SIZE and WRITE_SIZE cannot share one: MI355X_MICROARCH.md "rocprofv3 PMC slots"); plain launches
# (graphs=0) of ONE step so that every kernel launch is its own dispatch record.  tools/pmc_summary.py TAG -> profiles/TAG_pmc.json
set -u
TAG=${1:-r01}; shift || true
ARGS="$*"
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
echo "$ARGS" > $OUT/args.txt
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -o pmc -- python bench.py --steps 1 --warmup 1 --headline-only --cpu-samples 0 --option graphs=0 $ARGS > /dev/null 2> $OUT/$1.err; }
run pmc_fetch FETCH_SIZE
run pmc_write WRITE_SIZE
run pmc_mfma "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
run pmc_issue "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM"
run pmc_wr "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
find $OUT -name "*.db" -delete
ls $OUT/*/ | head -30; tail -3 $OUT/pmc_wr.err

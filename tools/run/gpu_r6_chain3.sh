#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
REPS=1 bash tools/run/gpu_r6_ab.sh d2def tools/lab/libpoem_new.so tools/lab/libpoem_nat.so | sed 's/^/default /'
BENCH_ARGS="--option chain_tile=3" REPS=1 bash tools/run/gpu_r6_ab.sh d2c16 tools/lab/libpoem_new.so tools/lab/libpoem_nat.so | sed 's/^/chain_tile=3 /'
done

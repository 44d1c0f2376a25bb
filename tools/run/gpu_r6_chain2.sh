#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_chain2; mkdir -p $O
for v in base nat prio new; do
  B=tools/lab/c16_lab_$v; [ $v == new ] && B=tools/lab/c16_lab
  for i in 1 2; do timeout 120 $B 32 | grep "per launch" | cut -c1-60 | sed "s/^/$v /"; done
done > $O/lab.txt 2>&1
cat $O/lab.txt
timeout 120 tools/lab/c16_lab 32 trace > $O/c16_trace_new_B32.txt 2>&1
REPS=3 bash tools/run/gpu_r6_ab.sh chain tools/lab/libpoem_base.so tools/lab/libpoem_nat.so tools/lab/libpoem_prio.so tools/lab/libpoem_new.so

#!/bin/bash
# round 6: chain16 at B = 32 with more, smaller tiles per CU (lab override POEM_C16_LAYERS): does a third / fourth tile fill the older tile's tail?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_layers; mkdir -p $O; : > $O/layers.txt
for B in 32 24; do
for L in 2 3 4 7; do
  echo "=== B=$B layers >= $L" >> $O/layers.txt
  POEM_C16_LAYERS=$L timeout 120 tools/lab/c16_lab $B 2>&1 | grep -E "us per launch|physical CUs|blocks: first" >> $O/layers.txt
done
done
cat $O/layers.txt

#!/bin/bash
# round 6: chain16 at C = 512 (POEM-large, c4: B = 16): two co-resident tiles of 2 units (shipped) vs one tile of up to 4 units per CU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_c512; mkdir -p $O; : > $O/c512.txt
for B in 16 32 8; do
  echo "=== C=512 B=$B shipped (MAXRU 2, two tiles per CU)" >> $O/c512.txt
  C16_LAB_C=512 timeout 120 tools/lab/c16_lab $B 2>&1 | grep -E "us per launch|physical CUs" >> $O/c512.txt
  echo "=== C=512 B=$B one tile of up to 4 units per CU" >> $O/c512.txt
  C16_LAB_C=512 POEM_C16_RU512=4 timeout 120 tools/lab/c16_lab $B 2>&1 | grep -E "us per launch|physical CUs" >> $O/c512.txt
done
cat $O/c512.txt

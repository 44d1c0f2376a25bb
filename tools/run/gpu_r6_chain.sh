#!/bin/bash
# round 6: chain16 at B = 32 -- issue-cost lab, per-wave phase trace of the two co-resident tiles of CU 0, PMC accounting
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_chain; mkdir -p $O
timeout 300 tools/lab/c16_issue_lab > $O/issue_lab.txt 2>&1
timeout 300 tools/lab/c16_lab 32 trace > $O/c16_trace_B32.txt 2>&1
timeout 300 tools/lab/c16_lab 24 trace > $O/c16_trace_B24.txt 2>&1
timeout 600 bash tools/pmc_lab.sh "tools/lab/c16_lab 32" > $O/c16_pmc_B32.txt 2>&1
cat $O/issue_lab.txt
head -60 $O/c16_trace_B32.txt

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -q -m gpu --durations=25 2>&1 | tail -40 > gpurun_out/r6_fullsuite2.txt; tail -3 gpurun_out/r6_fullsuite2.txt
bash tools/collect_pmc.sh r06_c4 --model large --views 10 --batch 16 > /dev/null 2>&1
bash tools/collect_pmc.sh r06_c5 --views-range 2 10 --batch 64 > /dev/null 2>&1
bash tools/collect_pmc.sh r06h > /dev/null 2>&1
ls gpurun_out/prof_r06_c4/ gpurun_out/prof_r06_c5 gpurun_out/prof_r06h | head -40; tail -2 gpurun_out/prof_r06h/pmc_wr.err gpurun_out/prof_r06h/pmc_issue.err

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3c
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "full_size or errors_are_loud or mediummano or ragged" > gpurun_out/r3c/pytest.txt 2>&1
tail -5 gpurun_out/r3c/pytest.txt
python tools/small_batch.py --sync-each > gpurun_out/r3c/table_graph.txt 2> gpurun_out/r3c/table.err
python tools/small_batch.py --sync-each --option graphs=0 > gpurun_out/r3c/table_nograph.txt 2>> gpurun_out/r3c/table.err
head -7 gpurun_out/r3c/table_graph.txt; head -7 gpurun_out/r3c/table_nograph.txt; tail -3 gpurun_out/r3c/table.err

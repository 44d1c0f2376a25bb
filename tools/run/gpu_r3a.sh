#!/bin/bash
# quick check of a change: the batch-property tests + the small-batch table
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "full_size or errors_are_loud or anchor_tables or row_tile" > gpurun_out/r3a/pytest.txt 2>&1
tail -5 gpurun_out/r3a/pytest.txt
python tools/small_batch.py --sync-each > gpurun_out/r3a/table.txt 2> gpurun_out/r3a/table.err
cat gpurun_out/r3a/table.txt | head -8

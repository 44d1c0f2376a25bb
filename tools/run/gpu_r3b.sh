#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3b
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "knn or conditioning" > gpurun_out/r3b/pytest.txt 2>&1
tail -15 gpurun_out/r3b/pytest.txt
timeout 1200 python tools/parity_report.py > gpurun_out/r3b/parity.txt 2> gpurun_out/r3b/parity.err
tail -12 gpurun_out/r3b/parity.txt

#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3g; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "conditioning or both_matrix or row_tile" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 1200 python tools/parity_report.py > $O/parity.txt 2> $O/parity.err
tail -10 $O/parity.txt

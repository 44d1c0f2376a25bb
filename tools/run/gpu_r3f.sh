#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_decode.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
bash tools/run/gpu_decode_prof.sh 2>&1 | grep -v "at::native\|miopen\|MIOpen\|igemm"

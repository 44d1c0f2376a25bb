#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "ragged or graph or grouped or fused_sampling or nan_features or two_streams or retired or parked" 2>&1 | tail -4
for i in 1 2 3; do
for O in "" "--option xattn_merge=1"; do
  timeout 300 python bench.py --steps 20 --warmup 5 --headline-only --cpu-samples 0 $O 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$O]', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms')"
done; done

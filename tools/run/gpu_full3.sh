#!/bin/bash
# full -m gpu suite + the default bench + a B=32 kernel trace / timeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r3full}
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/pytest.txt 2>&1
tail -5 gpurun_out/$TAG/pytest.txt
timeout 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -c 1500 gpurun_out/$TAG/bench.json
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/trace -o t -- python bench.py --headline-only --steps 10 --warmup 3 > gpurun_out/$TAG/trace_bench.json 2> gpurun_out/$TAG/trace.err
T=$(find gpurun_out/$TAG/trace -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $T 8 > gpurun_out/$TAG/timeline_B32.txt 2>> gpurun_out/$TAG/trace.err
rm -rf gpurun_out/$TAG/trace
head -30 gpurun_out/$TAG/timeline_B32.txt

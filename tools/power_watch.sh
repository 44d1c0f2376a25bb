#!/bin/bash
# Sample power / clocks with rocm-smi while the bench runs (is the step power-limited?)
cd "$GRAFT_REPO_ROOT"
rocm-smi --showmaxpower 2>&1 | grep -i "max" | head -3
(python bench.py --steps ${STEPS:-600} --warmup 3 --cpu-samples 0 > /tmp/bench_pw.json 2>/dev/null) &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "socket\|sclk\|junction" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';' >> /tmp/pw.log
  echo >> /tmp/pw.log
  sleep 0.25
done
grep -v "Power (W): 2[0-9][0-9]\.\|Power (W): 1[0-9][0-9]\." /tmp/pw.log | tail -25
tail -1 /tmp/bench_pw.json | cut -c1-200

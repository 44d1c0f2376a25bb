#!/bin/bash
# A/B the default bench between kept builds of libpoem_hip.so on ONE box (box-to-box spread is ~2 %): 
#   tools/ab.sh libA.so libB.so ...   (each may be prefixed with VAR=value, settings; separated by ':')
REPS=${REPS:-3}
for i in $(seq $REPS); do
  for SPEC in "$@"; do
    L=${SPEC##*:}; ENVS=${SPEC%:*}; [ "$ENVS" == "$SPEC" ] && ENVS=""
    env $(echo $ENVS | tr ',' ' ') POEM_HIP_LIB=$PWD/$L python bench.py --steps 10 --warmup 3 --cpu-samples 0 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$SPEC', round(d['value'],1), 'samples/s  vecattn', round(d['roofline']['avg_launch_ms'],4), 'ms')"
  done
done

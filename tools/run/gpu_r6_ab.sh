#!/bin/bash
# round 6: A/B of kept library builds on ONE box: tools/run/gpu_r6_ab.sh <tag> lib1.so lib2.so ...   (headline leg only, interleaved)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
O=gpurun_out/r6_ab_$TAG; mkdir -p $O; : > $O/ab.txt
for i in $(seq ${REPS:-3}); do
  for L in "$@"; do
    POEM_HIP_LIB=$PWD/$L timeout 300 python bench.py --steps 20 --warmup 5 --headline-only ${BENCH_ARGS:-} 2> $O/err.txt | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms  vecattn', round(d['roofline']['avg_launch_ms'],4), 'ms')" >> $O/ab.txt
  done
done
cat $O/ab.txt

#!/bin/bash
# A/B of poem_set_option switches on ONE box, interleaved: tools/ab_opt.sh "name=value[,name=value]" ...  ("-" = defaults)
REPS=${REPS:-3}
EXTRA=${EXTRA:---headline-only --steps 20 --warmup 5}
for i in $(seq $REPS); do
  for SPEC in "$@"; do
    OPTS=""
    if [ "$SPEC" != "-" ]; then for kv in $(echo $SPEC | tr ',' ' '); do OPTS="$OPTS --option $kv"; done; fi
    python bench.py $EXTRA $OPTS 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$SPEC', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms  sampling', round(d.get('sampling_stage',{}).get('avg_ms',0),4), ' vecattn', round(d['roofline']['avg_launch_ms'],4))"
  done
done

#!/bin/bash
# A/B of one poem_set_option switch over the batch sizes, interleaved on one box: tools/lab/ab_opt_small.sh NAME "V0 V1" ["1 2 4 8 16 32"] [reps]
NAME=$1; VALS=${2:-"0 1"}; BS=${3:-"1 2 4 8 16 32"}; REPS=${4:-2}
for rep in $(seq $REPS); do
for v in $VALS; do
  echo "$NAME=$v"; python tools/small_batch.py --batches $BS --steps 60 --option $NAME=$v 2>&1 | grep "^B="
done
done

#!/bin/bash
# A/B of kept builds of libpoem_hip.so over the batch sizes on one box: tools/lab/ab_lib_small.sh "1 2 4" libA.so libB.so ...
BS=$1; shift
for rep in 1 2; do
for L in "$@"; do
  echo "$L"; POEM_HIP_LIB=$PWD/$L python tools/small_batch.py --batches $BS --steps 60 2>&1 | grep "^B="
done
done

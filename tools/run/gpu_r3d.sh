#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_transform.py -x -q -m gpu -k "both_matrix_shapes or full_size_batch or pipeline_matches" > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
for T in 0 3; do
  python tools/small_batch.py --option chain_tile=$T > $O/table_tile$T.txt 2>> $O/table.err
  head -6 $O/table_tile$T.txt
done
for T in 0 3; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace$T -o t -- python tools/small_batch.py --batches 32 --steps 12 --warmup 3 --option chain_tile=$T > $O/trace$T.txt 2> $O/trace$T.err
  F=$(find $O/trace$T -name "*kernel_trace.csv" | head -1)
  python tools/step_timeline.py $F 10 > $O/timeline_B32_tile$T.txt 2>> $O/trace$T.err
  rm -rf $O/trace$T
  head -24 $O/timeline_B32_tile$T.txt
done

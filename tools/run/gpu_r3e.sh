#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3e; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "both_matrix_shapes or full_size or row_tile or decoder_entry or small or ragged_views" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
python tools/small_batch.py > $O/table.txt 2>> $O/table.err; head -6 $O/table.txt

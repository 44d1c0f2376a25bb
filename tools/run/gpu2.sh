cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "last_block or anchor_tables or release_shapes or tiny_stage or full_size or ragged_views or stage_taps" 2>&1 | tail -5
REPS=3 bash tools/ab.sh tools/lab/so/tabtop.so tools/lab/so/deadffn.so

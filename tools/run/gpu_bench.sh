# quick A/B on one box: un-profiled default bench lines (no CPU leg)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2 3; do timeout 300 python bench.py --cpu-samples 0 --steps 20 --warmup 3 --no-e2e 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['anchored_block0']['avg_launch_ms'],4))"; done

# final un-profiled numbers of the round: the full default bench line + five short runs on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python bench.py 2> gpurun_out/bench_unprofiled.err | tail -1 > gpurun_out/bench_unprofiled.json
for i in 1 2 3 4 5; do timeout 300 python bench.py --cpu-samples 0 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],3), round(r['avg_launch_ms'],4), round(r['frac'],4), round(r['anchored_block0']['avg_launch_ms'],4), round(d['pyramid_scope']['value'],1))"; done | tee gpurun_out/bench_runs.txt
cut -c1-300 gpurun_out/bench_unprofiled.json

# the other BASELINE.json configurations at their per-GPU load (un-profiled, head scope only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() { timeout 400 python bench.py --cpu-samples 0 --steps 10 --warmup 3 --no-e2e "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value'],1), 'samples/s', round(d['ms_per_step'],2), 'ms')"; }
run --model large --views 10 --batch 16
run --model medium_MANO --views 8 --batch 32
run --views-range 2 10 --batch 64
run --model small --views 2 --batch 32
run --model huge --views 8 --batch 8
run --model large --views 10 --batch 16 --anchor-tables 0

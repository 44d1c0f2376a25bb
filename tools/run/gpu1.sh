cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "anchor_tables or release_shapes or tiny_stage or full_size or ragged_views" 2>&1 | tail -15 > gpurun_out/t1.log
timeout 300 python bench.py --cpu-samples 0 --steps 20 --warmup 3 2> gpurun_out/b1.err | tail -1 > gpurun_out/b1.json
timeout 300 python bench.py --cpu-samples 0 --steps 20 --warmup 3 --anchor-tables 0 2> gpurun_out/b0.err | tail -1 > gpurun_out/b0.json
cat gpurun_out/t1.log; python - <<'PY'
import json
for f in ("b1","b0"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read()); print(f, d["value"], d["ms_per_step"], json.dumps(d.get("roofline"))[:900])
    except Exception as e: print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-1500:])
PY

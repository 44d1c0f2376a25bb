# kernel trace of the default bench (no CPU / e2e legs) -> gpurun_out/trace/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --steps 10 --warmup 3 --cpu-samples 0 --no-e2e > gpurun_out/trace/bench.json 2> gpurun_out/trace/err.log
find gpurun_out/trace -name "*.db" -delete; ls -la gpurun_out/trace

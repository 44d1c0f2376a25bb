# Round-end rehearsal on the GPU box: smoke(), the whole -m gpu suite, the bench under torch.distributed.run (1 rank, RCCL).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3 ) > gpurun_out/smoke.log
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/gpu_tests.log
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 3 --cpu-samples 0 2>&1 | tail -2 ) > gpurun_out/bench_dist1.log
cat gpurun_out/smoke.log gpurun_out/gpu_tests.log; cut -c1-400 gpurun_out/bench_dist1.log

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/e2e
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/e2e/gputests.log
timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 5 --warmup 1 > gpurun_out/e2e/bench_w20.json 2> gpurun_out/e2e/bench_w20.err
timeout 1200 python tests/manual/gpu_chunk_sweep.py 6000 > gpurun_out/e2e/sweep3.log 2>&1
cat gpurun_out/e2e/gputests.log; grep "^N " gpurun_out/e2e/sweep3.log

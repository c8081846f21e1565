cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/e2e
timeout 1200 python tests/manual/gpu_chunk_sweep.py 1000 6000 > gpurun_out/e2e/sweep8.log 2>&1
LTR_DEBUG=1 timeout 600 python tests/manual/gpu_calc_hap_aln_probs_rate.py 6000 > gpurun_out/e2e/rate6000.log 2> gpurun_out/e2e/rate6000.err
grep -v "^\[ltr  " gpurun_out/e2e/rate6000.err | tail -14 > gpurun_out/e2e/rate6000.timeline; grep "^\[ltr  " gpurun_out/e2e/rate6000.err | grep -v "launched" | tail -40 > gpurun_out/e2e/rate6000.plan; rm gpurun_out/e2e/rate6000.err
grep "^N " gpurun_out/e2e/sweep8.log; cat gpurun_out/e2e/rate6000.log gpurun_out/e2e/rate6000.timeline gpurun_out/e2e/rate6000.plan

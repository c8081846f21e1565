# round-5 profile set: rocprofv3 kernel stats + PMC passes of the default bench (config 3) and the catalogue, per-dispatch counters of a
# 1250-locus shard and of the exact-only pass, wave clocks of the plan kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r05_prof}; mkdir -p $O
bash profiles/collect.sh r05 > $O/collect.log 2>&1
bash profiles/collect.sh r05_catalogue catalogue > $O/collect_catalogue.log 2>&1
bash profiles/pmc_dispatch.sh shard1250 'plan_kernel|wg_kernel|ltr_dp_kernel' --loci 1250 > $O/pmc_shard1250.log 2>&1
bash profiles/pmc_dispatch.sh exact 'ltr_dp_kernel<|wgx' --pair-packing 4 > $O/pmc_exact.log 2>&1
bash profiles/pmc_dispatch_prog.sh neighbours 'nw_|short' tests/manual/gpu_neighbours.py > $O/pmc_neighbours.log 2>&1
timeout 300 python tests/manual/gpu_wave_clock.py config3 8 > $O/wave_clock_1250.log 2>&1
timeout 300 python tests/manual/gpu_wave_clock.py config3 1 > $O/wave_clock_10000.log 2>&1
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$O/trace_neighbours -o run --output-format csv -- python3 $ROOT/tests/manual/gpu_neighbours.py > $ROOT/$O/trace_neighbours.log 2>&1
cd $ROOT; find $O gpurun_out/prof_* gpurun_out/pmc_* -name "*kernel_trace.csv" -size +2M -delete
tail -c 1500 gpurun_out/prof_r05/summary.json; cat gpurun_out/pmc_shard1250/dispatches.txt | head -12

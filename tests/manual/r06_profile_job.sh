# round-6 profile set: rocprofv3 kernel stats + PMC passes of the default bench (config 3), per-dispatch counters of a 1250-locus shard,
# of BASELINE config 5 under both first passes, of config5hifi through the exact lists, and of the neighbours; wave clocks
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06_prof}; mkdir -p $O
bash profiles/collect.sh r06 > $O/collect.log 2>&1
bash profiles/pmc_dispatch.sh shard1250 'plan_kernel|wg_kernel|ltr_dp_kernel' --loci 1250 > $O/pmc_shard1250.log 2>&1
bash profiles/pmc_dispatch.sh config5_thresholds 'wg_kernel|wgx|ltr_dp_kernel' --workload config5 --debug wg_first_pass=2 > $O/pmc_config5_thr.log 2>&1
bash profiles/pmc_dispatch.sh config5_certificates 'wg_kernel|wgx|ltr_dp_kernel' --workload config5 --debug wg_first_pass=1 > $O/pmc_config5_cert.log 2>&1
bash profiles/pmc_dispatch.sh config5hifi_exact 'wgx|ltr_dp_kernel<' --workload config5hifi --pair-packing 4 > $O/pmc_config5hifi_exact.log 2>&1
bash profiles/pmc_dispatch.sh config5hifi_thresholds 'wg_kernel|wgx' --workload config5hifi --debug wg_first_pass=2 > $O/pmc_config5hifi_thr.log 2>&1
bash profiles/pmc_dispatch_prog.sh neighbours 'nw_|short' tests/manual/gpu_neighbours.py > $O/pmc_neighbours.log 2>&1
timeout 300 python tests/manual/gpu_wave_clock.py config3 8 > $O/wave_clock_1250.log 2>&1
timeout 300 python tests/manual/gpu_wave_clock.py config3 1 > $O/wave_clock_10000.log 2>&1
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$O/trace_config5 -o run --output-format csv -- python3 $ROOT/bench.py --workload config5 --no-cpu-baseline --no-end-to-end --no-neighbours --no-verify --steps 3 --warmup 2 > $ROOT/$O/trace_config5.log 2>&1
cd $ROOT; find $O gpurun_out/prof_* gpurun_out/pmc_* -name "*kernel_trace.csv" -size +2M -delete
tail -c 1500 gpurun_out/prof_r06/summary.json; for t in shard1250 config5_thresholds config5_certificates config5hifi_exact config5hifi_thresholds; do echo == $t; head -14 gpurun_out/pmc_$t/dispatches.txt; done

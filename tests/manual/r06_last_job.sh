# round 6, last GPU run: the default bench line with the committed counter summary attached (profiles/r06/pmc_traffic.json is of this
# build), and the ring protocol of the workgroup kernels -- certificate, threshold first pass, exact lists -- under stress builds
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06_last}; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_config3.json 2> $O/bench_config3.err; cp bench_detail.json $O/bench_detail_config3.json
bash tests/manual/gpu_wg_ring_stress.sh > $O/wg_ring_stress.log 2>&1
tail -1 $O/bench_config3.json | cut -c1-1500; cat $O/wg_ring_stress.log

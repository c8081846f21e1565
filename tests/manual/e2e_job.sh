cd $GRAFT_REPO_ROOT; O=gpurun_out/nw; mkdir -p $O
timeout 600 python -m pytest tests/test_nw_hap_to_ref.py -m gpu -x -q 2>&1 | tail -5 > $O/nwtest.log
timeout 600 python tests/manual/gpu_nw_rate.py 1000 > $O/nw_rate.log 2>&1
timeout 600 python tests/manual/gpu_nw_rate.py 4000 >> $O/nw_rate.log 2>&1
cat $O/nwtest.log; tail -4 $O/nw_rate.log

cd $GRAFT_REPO_ROOT; O=gpurun_out/real; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_real_reads.py -m gpu -x -q 2>&1 | tail -30 > $O/test.log
timeout 600 python examples/real_reads_trio.py $O/trio.vcf.gz > $O/trio.log 2>&1
cat $O/test.log; cat $O/trio.log

# long differential fuzz of the final build (no source change: same source_id as profiles/r05)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r05_long_fuzz}; mkdir -p $O
python -c "from longtr_amd import _lib; print('source_id', _lib.source_id())" > $O/long_fuzz.log 2>&1
for s in 101 102 103; do timeout 400 python tests/manual/gpu_fuzz.py 300 $s 2>&1 | tail -1; done >> $O/long_fuzz.log 2>&1
for s in 41 42; do timeout 300 python tests/manual/gpu_nw_fuzz.py 200 $s 2>&1 | tail -1; done >> $O/long_fuzz.log 2>&1
for s in 15 16; do timeout 300 python tests/manual/gpu_short_fuzz.py 150 $s 2>&1 | tail -1; done >> $O/long_fuzz.log 2>&1
cat $O/long_fuzz.log

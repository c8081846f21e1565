# long differential fuzz of the final build (no library source change: same source_id as profiles/r06), + the default bench line with the final bench.py
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06_long_fuzz}; mkdir -p $O
python -c "from longtr_amd import _lib; print('source_id', _lib.source_id())" > $O/long_fuzz.log 2>&1
for s in 201 202 203 204; do timeout 500 python tests/manual/gpu_fuzz.py 400 $s 2>&1 | tail -1; done >> $O/long_fuzz.log 2>&1
for s in 41 42; do timeout 300 python tests/manual/gpu_nw_fuzz.py 200 $s 2>&1 | tail -1; done >> $O/long_fuzz.log 2>&1
for s in 15 16; do timeout 300 python tests/manual/gpu_short_fuzz.py 150 $s 2>&1 | tail -1; done >> $O/long_fuzz.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_config3.json 2> $O/bench_config3.err; cp bench_detail.json $O/bench_detail_config3.json
cat $O/long_fuzz.log; tail -1 $O/bench_config3.json | cut -c1-900

# round 6, closing run on the final tree: the -m gpu suite and smoke as the driver runs them, then a long differential fuzz of the final build
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06_end}; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 > $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
python -c "from longtr_amd import _lib; print('source_id', _lib.source_id())" > $O/long_fuzz.log 2>&1
for s in 301 302 303; do timeout 500 python tests/manual/gpu_fuzz.py 400 $s 2>&1 | tail -1; done >> $O/long_fuzz.log 2>&1
for s in 43; do timeout 300 python tests/manual/gpu_nw_fuzz.py 150 $s 2>&1 | tail -1; done >> $O/long_fuzz.log 2>&1
for s in 17; do timeout 300 python tests/manual/gpu_short_fuzz.py 120 $s 2>&1 | tail -1; done >> $O/long_fuzz.log 2>&1
cat $O/gputests.log; tail -1 $O/smoke.log; cat $O/long_fuzz.log

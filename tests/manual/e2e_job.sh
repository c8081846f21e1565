cd $GRAFT_REPO_ROOT; O=gpurun_out/final7; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 900 python bench.py > $O/bench_config3.json 2> $O/bench_config3.err
tail -3 $O/gputests.log; tail -1 $O/smoke.log; python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/final7/bench_config3.json') if l.startswith('{')][-1])
print('%.4e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'mism', d['oracle_check']['mismatches'], 'e2e', d['loci_per_s_end_to_end'], d['end_to_end']['frac_of_resident_rate'], 'x1', d['speedup_vs_cpu_1thread'], 'xN', d['speedup_vs_cpu_ncores'])
P

cd $GRAFT_REPO_ROOT; O=gpurun_out/c5hifi; mkdir -p $O; ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --workload config5hifi --no-cpu-baseline --no-end-to-end --no-neighbours --no-verify --steps 2 --warmup 1"
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  NAME=$(echo "$SET" | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $SET -d "$ROOT/$O/pmc_$NAME" -o run --output-format csv -- $BENCH > "$ROOT/$O/pmc_$NAME.log" 2>&1
done
cd $ROOT; python - <<'P'
import csv,glob,collections
tot=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/c5hifi/pmc_*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'ltr_dp_wg_kernel<10' in k:
            tot['wg10'][r['Counter_Name']]+=float(r['Counter_Value']); n['wg10'][r['Counter_Name']]+=1
for c,v in tot['wg10'].items(): print(c, v/n['wg10'][c])
P

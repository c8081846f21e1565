# PMC + kernel-trace of the packed kernels on the catalogue workload, per DISPATCH (gpurun; program directly after --)
cd $GRAFT_REPO_ROOT; O=gpurun_out/pmc_pack; rm -rf $O; mkdir -p $O; ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --workload catalogue --loci ${LOCI:-100000} --no-cpu-baseline --no-end-to-end --no-neighbours --no-verify --steps 1 --warmup 0 --debug fan_lanes=1"
timeout 600 rocprofv3 --kernel-trace -d "$ROOT/$O/trace" -o run --output-format csv -- $BENCH > "$ROOT/$O/trace.log" 2>&1
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"; do
  NAME=$(echo "$SET" | tr ' ' '+')
  timeout 600 rocprofv3 --pmc $SET -d "$ROOT/$O/pmc_$NAME" -o run --output-format csv -- $BENCH > "$ROOT/$O/pmc_$NAME.log" 2>&1
done
cd $ROOT
python3 - <<'P'
import csv, glob, re, collections
def short(name):
    m = re.search(r"(ltr_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][:40]
# per kernel: list of dispatches in order -> dict of values
seq = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in glob.glob('gpurun_out/pmc_pack/trace/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    cnt = collections.Counter()
    for r in rows:
        k = short(r['Kernel_Name']); i = cnt[k]; cnt[k] += 1
        seq[k][i]['us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        seq[k][i]['grid'] = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)
for d in glob.glob('gpurun_out/pmc_pack/pmc_*'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        rows = list(csv.DictReader(open(f)))
        ids = collections.defaultdict(list)
        for r in rows:
            k = short(r['Kernel_Name'])
            did = int(r['Dispatch_Id'])
            if did not in ids[k]: ids[k].append(did)
        for r in rows:
            k = short(r['Kernel_Name']); i = sorted(ids[k]).index(int(r['Dispatch_Id']))
            seq[k][i][r['Counter_Name']] = seq[k][i].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
print("kernel dispatch# grid us VALU SALU LDS waves valu_issue(4cyc) wait_frac")
for k in sorted(seq):
    if 'pack' not in k: continue
    for i in sorted(seq[k]):
        v = seq[k][i]
        gui = v.get('GRBM_GUI_ACTIVE', 0) / 8.0
        print(k, i, v.get('grid'), '%.1f' % v.get('us', 0), '%.3g' % v.get('SQ_INSTS_VALU', 0), '%.3g' % v.get('SQ_INSTS_SALU', 0), '%.3g' % v.get('SQ_INSTS_LDS', 0),
              '%.0f' % v.get('SQ_WAVES', 0), '%.2f' % (v.get('SQ_INSTS_VALU', 0) * 4 / (1024 * gui) if gui else 0), '%.2f' % (v.get('SQ_WAIT_INST_ANY', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1)))
P

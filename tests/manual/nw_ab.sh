cd $GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for so in "" abtest/nw3.so abtest/nw4.so; do
  if [ -n "$so" ]; then export LTR_GPU_LIB=$GRAFT_REPO_ROOT/$so; else unset LTR_GPU_LIB; fi
  rm -rf /tmp/nwtrace; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/nwtrace -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/manual/gpu_nw_rate.py 4000 > /tmp/nw.log 2>&1
  echo "lib [$so]: $(tail -1 /tmp/nw.log | grep -o 'haplotypes.*' | cut -c1-150)"
  python3 - <<'P'
import csv,glob
f=glob.glob('/tmp/nwtrace/**/*kernel_stats.csv', recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    if 'nw_wave' in r['Name']: tot+=float(r['AverageNs'])/1e6
print('   sum of the wave kernels per call: %.2f ms'%tot)
P
done

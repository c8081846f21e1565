# NW kernel variants (abtest/nw_*.so from tests/manual/ab_build.sh), interleaved on one box: call and kernel time from ltr_timers
cd $GRAFT_REPO_ROOT
cat > /tmp/nwv.py <<'P'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from longtr_amd import _lib, synth
loci, desc = synth.config_loci("config3", n_loci=3000)
ctx = _lib.Context(0)
packed = ctx.pack_haplotypes([L.blocks() for L in loci])
cells = float(sum(len(L.haplotypes[0]) * sum(len(h) for h in L.haplotypes) for L in loci))
ctx.haplotype_align_to_ref_packed(packed, decode=False)
ctx.timers(reset=True)
t0 = time.perf_counter()
for _ in range(5): ctx.haplotype_align_to_ref_packed(packed, decode=False)
dt = (time.perf_counter() - t0) / 5
kms = ctx.timers(reset=True)["nw_kernel_ms"] / 5
print(f"call {dt*1e3:.2f} ms, kernels {kms:.2f} ms, {cells/(kms*1e-3):.3e} cells/s in the kernels")
P
for rep in 1 2; do
for so in ""; do   # (add abtest/<name>.so built by tests/manual/ab_build.sh to compare variants)
  if [ -n "$so" ]; then export LTR_GPU_LIB=$GRAFT_REPO_ROOT/$so; else unset LTR_GPU_LIB; fi
  echo "lib [${so:-in-tree}]: $(timeout 300 python3 /tmp/nwv.py 2>&1 | tail -1)"
done; done

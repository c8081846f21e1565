"""Manual GPU check: when the wavefronts of the plan kernel leave (ltr_ctx_set_debug "wave_clock"): the tail of the one launch.
    python tests/manual/gpu_wave_clock.py [workload] [shards] [a,b,c,d,e,f,g]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from longtr_amd import _abi, _lib, shard, synth
WL = sys.argv[1] if len(sys.argv) > 1 else "config3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
NL = synth._DEFAULT_N[WL]
parts = shard.shard_by_cost(shard.header_time_costs(synth.config_headers(WL, n_loci=NL)), N)
loci, _ = synth.config_loci(WL, n_loci=NL, ids=parts[0])
batch, _ = synth.pack_loci(loci)
ctx = _lib.Context(0)
if len(sys.argv) > 3:
    ctx.set_params(_abi.make_params(tuple(float(x) for x in sys.argv[3].split(","))))
ctx.set_debug("wave_clock", 1)
plan = ctx.plan(batch)
for rep in range(3):
    t0 = time.perf_counter(); plan.execute(); plan.wait(); dt = time.perf_counter() - t0
    wc = plan.wave_clocks().astype(np.float64) / 100.0      # microseconds
    first, last = wc[:, 0], wc[:, 1]
    z = first.min()
    span = last.max() - z
    q = np.percentile(last - z, [0, 1, 10, 50, 90, 99, 100])
    busy = (last - first).sum() / (len(wc) * span)
    rp, rt = wc[:, 2] * 100.0, wc[:, 3]     # (undo the division by 100 for the pair count)
    late = np.argsort(-last)[:8]
    print(f"pass {rep}: host {dt*1e3:.2f} ms; kernel span {span/1e3:.3f} ms over {len(wc)} wavefronts; start spread {np.ptp(first)/1e3:.3f} ms; "
          f"wavefronts leave at (ms) min {q[0]/1e3:.2f} p1 {q[1]/1e3:.2f} p10 {q[2]/1e3:.2f} p50 {q[3]/1e3:.2f} p90 {q[4]/1e3:.2f} p99 {q[5]/1e3:.2f} max {q[6]/1e3:.2f}; "
          f"wave-time / (waves x span) = {busy:.4f}; exact body: {int(rp.sum())} pairs, {rt.sum()/1e3:.2f} wave-ms, longest {rt.max()/1e3:.2f} ms; "
          f"the 8 last wavefronts: left at {[round(float(x - z)/1e3, 2) for x in last[late]]}, exact pairs {[int(x) for x in rp[late]]}, exact ms {[round(float(x)/1e3, 2) for x in rt[late]]}", flush=True)
# per entry of the table: the wave-time of all wavefronts together, and the cells it bought.  NOT a per-class efficiency: the three
# wavefronts of a SIMD sit in different entries (the start shares spread them over the table) and the issue arbiter favours the oldest
# wavefront -- the one with the lowest id, i.e. the one that started highest in the table -- so the wide strips at the top run at
# more than a third of their SIMD and the entries further down at less; the sum is what the pass costs.
ents = plan.plan_entries()
tot_ms = sum(float(plan.entry_ticks[i]) for i in range(len(ents))) / 1e5
print(f"entries of the plan kernel's table (walk order), last pass: wave-time of all wavefronts {tot_ms:.0f} ms = {tot_ms / len(wc):.2f} ms x {len(wc)}")
print("entry  kind                 W    pairs      cells   wave-ms   share of wave-time   share of cells   cells per wave-us")
tot_cells = sum(e["cells"] for e in ents) or 1.0
for i, e in enumerate(ents):
    wave_ms = float(plan.entry_ticks[i]) / 1e5
    print(f"{i:5d}  {e['kind']:18s} {e['strip_width']:3d} {e['pairs']:8d} {e['cells']:10.3e} {wave_ms:9.1f} {wave_ms / tot_ms:14.3f} {e['cells'] / tot_cells:18.3f} "
          f"{(e['cells'] / (wave_ms * 1e3) if wave_ms > 0 else 0.0):15.0f}")
lg = sorted(plan.redo_log)
print(f"pairs that took the exact body in the last pass (n, m, n - m): {[(n, m, n - m) for n, m in lg]}")
plan.close()

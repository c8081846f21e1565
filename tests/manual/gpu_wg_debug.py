"""Manual GPU probe: one tiny batch per scheduling mode, each in its own bounded subprocess."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from longtr_amd import _abi, _lib, synth
import oracle_lib as ol
mode, m = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
core = synth._rand_seq(rng, m).tobytes()
read = bytearray(core); read[m // 2] = ord("A") if read[m // 2] != ord("A") else ord("C")
hap = synth._rand_seq(rng, 30).tobytes() + core + synth._rand_seq(rng, 30).tobytes()
b = _abi.PackedBatch([([bytes(read)], [hap])] * int(sys.argv[3]))
ctx = _lib.Context(0)
ctx.set_pair_packing(mode)
ctx.set_debug("trace", 1)
print("mode", mode, "m", m, "pairs", b.ll_size, flush=True)
ll, _ = ctx.align_batch(b)
want = ol.oracle_align_long(hap, bytes(read), ctx.params, rolling=True)
print("  gpu", ll[:3], "oracle", want, "OK" if (ll == want).all() else "MISMATCH", flush=True)
''' % (ROOT, ROOT)
for mode, m, k in [(3, 100, 1), (2, 100, 1), (2, 100, 300), (2, 700, 3), (0, 1500, 1), (0, 1500, 50), (0, 4000, 2), (0, 9000, 2)]:
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, str(mode), str(m), str(k)], capture_output=True, text=True, timeout=40)
        print(r.stdout, r.stderr[-1500:] if r.returncode else "", "rc", r.returncode, flush=True)
    except subprocess.TimeoutExpired as e:
        print("TIMEOUT mode", mode, "m", m, "k", k, "\n", (e.stdout or b"")[-800:], "\n", (e.stderr or b"")[-2500:], flush=True)

"""Manual GPU check: differential fuzz of ltr_haplotype_align_to_ref (NeedlemanWunsch::Align + adjust_indels + the M / I / D string)
against the C restatement: random loci -- repeat lengths 1 .. 1300 (every strip-width class and the workgroup kernel), 1 - 9 alleles
with substitutions, indels, N and lower-case bases sprinkled in -- in batches (several classes side by side, longest pairs first).
    python tests/manual/gpu_nw_fuzz.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from longtr_amd import _lib, synth

T = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(SEED)
ctx = _lib.Context(0)


def mutate(a):
    a = bytearray(a)
    for _ in range(int(rng.integers(0, 4))):
        if not a:
            break
        k = int(rng.integers(0, len(a)))
        op = int(rng.integers(0, 5))
        if op == 0: a[k] = b"ACGT"[int(rng.integers(0, 4))]
        elif op == 1: del a[k:k + int(rng.integers(1, 8))]
        elif op == 2: a[k:k] = bytes(b"ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 8))))
        elif op == 3: a[k] = ord("N")
        else: a[k] = ord(chr(a[k]).lower())
    return bytes(a) if a else b"A"


t0 = time.time()
batches = pairs = 0
while time.time() - t0 < T:
    loci = []
    for k in range(int(rng.integers(20, 120))):
        r = rng.random()
        tr = int(rng.integers(1, 60)) if r < 0.3 else (int(rng.integers(60, 1300)) if r < 0.97 else int(rng.integers(1300, 2600)))
        L = synth.synth_locus(rng, tr, int(rng.integers(1, 40)), int(rng.integers(1, 9)), 1)
        for q in range(1, len(L.alleles)):
            if rng.random() < 0.5:
                L.alleles[q] = mutate(L.alleles[q])
        if rng.random() < 0.1:
            L.alleles[0] = mutate(L.alleles[0])
        loci.append(L)
    got = ctx.haplotype_align_to_ref([L.blocks() for L in loci])
    for L, infos in zip(loci, got):
        haps = L.haplotypes
        assert len(infos) == len(haps)
        for h, info in zip(haps, infos):
            want = ol.oracle_nw_aln_info(haps[0], h, L.start, L.start + 35)
            assert info == want, (SEED, batches, len(haps[0]), len(h))
            pairs += 1
    batches += 1
print(f"NW fuzz ok: {batches} batches, {pairs} (reference, haplotype) pairs identical to the restatement, {T:.0f} s, seed {SEED}")

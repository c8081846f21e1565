"""Manual GPU check: per-locus latency of the compiled adapter (integration/GpuHapAligner.h: construction + process_reads +
destruction per locus, the literal call at seq_stutter_genotyper.cpp:517-523) on the golden loci.
    python tests/manual/gpu_adapter_latency.py [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import adapter_util as au, golden_util as gu
d = gu.load("process_locus")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for _ in range(3):
    r = au.run("latency", d["params"], list(d["loci"]), reps)
    print("adapter latency:", r.stdout.strip(), r.stderr.strip()[-300:], flush=True)

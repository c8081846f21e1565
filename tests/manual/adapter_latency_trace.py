import sys, os, subprocess
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import adapter_util as au, golden_util as gu
d=gu.load("process_locus")
loci=list(d["loci"])[:2]*3
r=subprocess.run([au.BIN,"latency","3","trace"], input="".join(au.locus_text(d["params"],x) for x in loci), capture_output=True, text=True)
print(r.stdout[-300:]); print("\n".join(r.stderr.splitlines()[-45:]))

"""What hipcc made of the DP kernels, read off the gfx950 assembly (no GPU needed: hipcc cross-compiles here).

`analyse(tu)` compiles one kernel translation unit of longtr_amd/csrc with the library's own flags to assembly
(`--cuda-device-only -S`) and returns, per function (kernels and the real calls behind them alike):

  vgprs, sgprs, scratch     the ".. Function info" block hipcc prints behind every function (NumVgprs, NumSgprs, ScratchSize)
  sgpr_spills, vgpr_spills  kernels only (the .amdhsa / remark figures are per kernel)
  loops                     every natural loop (a backward branch to a label): first / last line, FP64 add/max count, scratch_ /
                            buffer_ accesses, v_readlane / v_writelane (SGPR spill traffic), global atomics
  atomics                   global_atomic_* instructions of the whole function
  step_loops                the loops that hold >= 20 v_add_f64 + v_max_f64: the wavefront steps of the DP

Used by tests/test_isa_budget.py (register budgets, "no scratch access inside a step loop", "one atomic per queue pop") and by
hand (`python tests/isa_util.py ltr_k_one.hip`)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "longtr_amd", "csrc")


def compile_flags():
    sys.path.insert(0, ROOT)
    from longtr_amd import _lib
    return [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-pthread", "-Wall")]


def assembly(tu, extra=(), cache_dir=None):
    """The device assembly of csrc/<tu> (cached by source mtime under cache_dir)."""
    src = os.path.join(CSRC, tu)
    out = None
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
        out = os.path.join(cache_dir, tu + ("." + "_".join(extra).replace("/", "_").replace("=", "_") if extra else "") + ".s")
        newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".hip")))
        if os.path.exists(out) and os.path.getmtime(out) >= newest:
            return open(out).read()
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + compile_flags() + list(extra) + ["--cuda-device-only", "-S", src, "-o", out or "/dev/stdout"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    return open(out).read() if out else r.stdout


_FP64 = re.compile(r"^\s*v_(add|max|min)_f64\b")
_SCRATCH = re.compile(r"^\s*scratch_(load|store)")            # (spill code is flat-scratch on gfx950; buffer_ loads are the kernels' own: the haplotype rows)
_LANE = re.compile(r"^\s*v_(readlane|writelane)_b32")
_ATOMIC = re.compile(r"^\s*(global|flat)_atomic_")
_BRANCH = re.compile(r"^\s*s_c?branch\w*\s+(\.LBB\d+_\d+)")
_LABEL = re.compile(r"^(\.LBB\d+_\d+):")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(anonymous namespace\)::", "", n) for n in out]


def parse(asm):
    lines = asm.splitlines()
    funcs = {}
    i = 0
    starts = []
    for i, l in enumerate(lines):
        m = re.match(r"^\s*\.type\s+(\S+),@function", l)
        if m:
            starts.append((i, m.group(1)))
    for (s, name), nxt in zip(starts, starts[1:] + [(len(lines), None)]):
        body = lines[s:nxt[0]]
        info = {"mangled": name}
        for l in body:
            m = re.match(r"^; (NumVgprs|NumSgprs|ScratchSize|NumAgprs): (\d+)", l)
            if m:
                info[{"NumVgprs": "vgprs", "NumSgprs": "sgprs", "ScratchSize": "scratch", "NumAgprs": "agprs"}[m.group(1)]] = int(m.group(2))
            m = re.match(r"^; (sgpr_spill_count|vgpr_spill_count)\s*[:=]?\s*(\d+)", l.replace("\t", " "))
            if m:
                info[{"sgpr_spill_count": "sgpr_spills", "vgpr_spill_count": "vgpr_spills"}[m.group(1)]] = int(m.group(2))
            m = re.search(r"\.(sgpr|vgpr)_spill_count:\s*(\d+)", l)
            if m:
                info[m.group(1) + "_spills"] = int(m.group(2))
        label_at = {}
        for k, l in enumerate(body):
            m = _LABEL.match(l)
            if m:
                label_at[m.group(1)] = k
        loops = []
        for k, l in enumerate(body):
            m = _BRANCH.match(l)
            if m and m.group(1) in label_at and label_at[m.group(1)] <= k:
                a = label_at[m.group(1)]
                seg = body[a:k + 1]
                loops.append({"first": a, "last": k, "fp64": sum(1 for x in seg if _FP64.match(x)),
                              "scratch": sum(1 for x in seg if _SCRATCH.match(x)), "lane_moves": sum(1 for x in seg if _LANE.match(x)),
                              "atomics": sum(1 for x in seg if _ATOMIC.match(x)), "instructions": sum(1 for x in seg if re.match(r"^\s+[a-z]\w+", x) and not x.strip().startswith("."))})
        info["loops"] = loops
        info["atomics"] = sum(1 for x in body if _ATOMIC.match(x))
        info["scratch_accesses"] = sum(1 for x in body if _SCRATCH.match(x))
        # innermost step loops: >= 20 FP64 add/max and no other such loop nested inside
        big = [L for L in loops if L["fp64"] >= 20]
        info["step_loops"] = [L for L in big if not any(M is not L and M["first"] >= L["first"] and M["last"] <= L["last"] for M in big)]
        funcs[name] = info
    names = demangle(list(funcs))
    return {re.sub(r"\((?:KernelArgs|long|int|unsigned).*$", "", n): v for n, v in zip(names, funcs.values())}


def kernel_spills(asm):
    """{demangled kernel name: (sgpr spills, vgpr spills)} from the .amdhsa metadata at the end of the assembly."""
    out, cur = {}, None
    for l in asm.splitlines():
        m = re.match(r"^\s*- \.agpr_count:|^\s*\.name:\s+(\S+)", l)
        m2 = re.match(r"^\s*\.name:\s+(\S+)", l)
        if m2:
            cur = m2.group(1)
            out.setdefault(cur, {})
        for key in ("sgpr_spill_count", "vgpr_spill_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            m3 = re.match(r"^\s*\.%s:\s+(\d+)" % key, l)
            if m3:
                out.setdefault("__pending__", {})[key] = int(m3.group(1))
        if re.match(r"^\s*\.symbol:\s+(\S+)\.kd", l):
            sym = re.match(r"^\s*\.symbol:\s+(\S+)\.kd", l).group(1)
            out[sym] = dict(out.get("__pending__", {}))
            out["__pending__"] = {}
    out.pop("__pending__", None)
    names = [k for k in out if out[k]]
    dem = demangle(names)
    return {re.sub(r"\((?:KernelArgs).*$", "", d): out[k] for d, k in zip(dem, names)}


def pop_sites(asm, mangled):
    """Every global atomic of function `mangled`: {returns (sc0: its value is used), guarded (issued under an exec mask that an
    s_and_saveexec / s_mov exec within the 10 instructions before it cut down -- one lane's add --, exec restored behind it), in_short_loop
    (inside a loop of fewer than 40 instructions that is closed by a branch on EXEC: a waterfall loop over lanes)}."""
    lines = asm.splitlines()
    start = None
    for i, l in enumerate(lines):
        if l.startswith(mangled + ":"):
            start = i
            break
    if start is None:
        return []
    end = start
    while end < len(lines) and not lines[end].startswith(".Lfunc_end"):
        end += 1
    body = lines[start:end]
    label_at = {}
    for k, l in enumerate(body):
        m = _LABEL.match(l)
        if m:
            label_at[m.group(1)] = k
    loops = []
    for k, l in enumerate(body):
        m = _BRANCH.match(l)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= k:
            loops.append((label_at[m.group(1)], k))
    is_instr = lambda x: bool(re.match(r"^\s+[a-z]\w+", x)) and not x.strip().startswith(".")
    out = []
    for k, l in enumerate(body):
        if not _ATOMIC.match(l):
            continue
        before = [x for x in body[max(0, k - 40):k] if is_instr(x)][-10:]
        after = [x for x in body[k + 1:k + 40] if is_instr(x)][:6]
        # the exec mask was cut down right before it (s_and_saveexec, or s_and + s_mov exec) and is restored right behind it
        guarded = any(("s_and_saveexec_b64" in x) or re.search(r"s_mov_b64 exec, s\[", x) for x in before) and any(re.search(r"s_or_b64 exec, exec", x) for x in after)
        # a waterfall loop over lanes: a short loop closed by a branch on EXEC (s_cbranch_execnz) around the atomic
        short = any(a <= k <= b and "execnz" in body[b] and sum(1 for x in body[a:b + 1] if is_instr(x)) < 40 for a, b in loops)
        out.append({"line": k, "returns": " sc0" in l, "guarded": guarded, "in_short_loop": short, "text": l.strip()})
    return out


def analyse(tu, extra=(), cache_dir=None):
    asm = assembly(tu, extra, cache_dir)
    f = parse(asm)
    for name, meta in kernel_spills(asm).items():
        if name in f:
            f[name].update({k: v for k, v in meta.items()})
    return f


if __name__ == "__main__":
    res = analyse(sys.argv[1], tuple(sys.argv[2:]), cache_dir=os.path.join("/tmp", "ltr_isa_cache"))
    for n, v in res.items():
        sl = v["step_loops"]
        print(f"{n[:74]:74s} V={v.get('vgprs')} S={v.get('sgprs')} scratch={v.get('scratch')} sS={v.get('sgpr_spill_count')} vS={v.get('vgpr_spill_count')} "
              f"atomics={v['atomics']} scratchAcc={v['scratch_accesses']} stepLoops={[(L['instructions'], L['fp64'], L['scratch'], L['lane_moves']) for L in sl]}")

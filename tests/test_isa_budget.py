"""What hipcc made of the DP kernels, checked in the build container (no GPU: hipcc cross-compiles gfx950 here).

The kernels of the hot path (reference: HapAligner::align_seq_to_hap, src/SeqAlignment/HapAligner.cpp:236-343, its inner loop
:282-307) are written against a register budget and a handful of code-generation facts that nothing in the language guarantees:
the strip width W is chosen so that a wavefront step keeps its 4W carried doubles in registers at the occupancy its
__launch_bounds__ asks for; the step loops touch no scratch; a work-queue pop is ONE lane's atomic whose value is broadcast --
which the source gets from hipcc in two ways (ltr_dp_kernel.hpp, pop_one: the wave-wide `atomicAdd(queue, lane == 0 ? 1 : 0)`
that the atomic optimizer turns into one lane's add while the queue is a kernel argument, the explicit lane-0 form with a
laundered address for loop-carried queues); rounds 3 - 4 met forms of the pop that never left their loop on MI355X, and round
5's attempt at ONE explicit form with the optimizer off hung the exact and NW kernels.  A toolchain bump can silently turn any
of these into spills or a hang; this test reads the assembly (tests/isa_util.py) and goes red instead.

Ceilings are the figures of the committed build plus a little slack; tighten them when a kernel improves."""
import os
import re
from concurrent.futures import ThreadPoolExecutor

import pytest

import isa_util

CACHE = os.path.join("/tmp", "ltr_isa_cache")
TUS = ["ltr_k_one.hip", "ltr_k_pack.hip", "ltr_k_plan.hip", "ltr_k_wg.hip", "ltr_k_wgt.hip", "ltr_k_exact.hip"]


@pytest.fixture(scope="module")
def isa():
    with ThreadPoolExecutor(max_workers=min(len(TUS), os.cpu_count() or 1)) as ex:
        texts = list(ex.map(lambda tu: isa_util.assembly(tu, cache_dir=CACHE), TUS))
    out = {}
    for tu, asm in zip(TUS, texts):
        f = isa_util.parse(asm)
        for name, meta in isa_util.kernel_spills(asm).items():
            if name in f:
                f[name].update(meta)
        for name, info in f.items():
            info["pops"] = isa_util.pop_sites(asm, info["mangled"])
        out[tu] = f
    return out


def _targs(name):
    m = re.search(r"<(.*)>", name)
    return [a.strip() for a in m.group(1).split(",")] if m else []


def _select(funcs, base):
    return {n: v for n, v in funcs.items() if re.search(r"\b%s<" % base, n)}


def vgpr_budget(waves_per_simd):
    """Registers a wavefront may use for `waves_per_simd` to be resident: 512 per SIMD lane, allocated in blocks of 8."""
    return (512 // waves_per_simd) // 8 * 8


def lb_onewave(w):          # LTR_LB, ltr_dp_types.h
    return 5 if w <= 6 else (4 if w <= 10 else 3)


def lb_pack(w):             # LTR_PACK_LB, ltr_dp_pack.hpp
    return 5 if w <= 6 else (4 if w <= 12 else (3 if w <= 20 else 2))


def budget_violations(isa_one):
    """Certificate kernels of the one-wave family that use more registers than their occupancy leaves."""
    bad = []
    for n, v in _select(isa_one, "ltr_dp_kernel").items():
        a = _targs(n)
        if a[1] != "false":
            continue
        w = int(a[0])
        if v["vgprs"] > vgpr_budget(lb_onewave(w)):
            bad.append((n, v["vgprs"], vgpr_budget(lb_onewave(w))))
    return bad


def test_register_budgets(isa):
    assert budget_violations(isa["ltr_k_one.hip"]) == []
    seen = 0
    for n, v in _select(isa["ltr_k_one.hip"], "ltr_dp_kernel").items():
        a = _targs(n)
        if a[1] == "false":
            seen += 1
    assert seen == 40                                              # W = 1 .. 20, symmetric and general model
    for n, v in _select(isa["ltr_k_pack.hip"], "ltr_dp_pack_kernel").items():
        w = int(_targs(n)[0])
        assert v["vgprs"] <= vgpr_budget(lb_pack(w)), (n, v["vgprs"])
    # everything a three-waves-per-SIMD launch calls: 168 registers, the callee's included
    for tu, bases in (("ltr_k_one.hip", ["class_walk_call", "ltr_dp_multi_kernel"]), ("ltr_k_pack.hip", ["pack_walk_call", "ltr_dp_pack_multi_kernel"]),
                      ("ltr_k_plan.hip", ["plan_class_call", "plan_pack_call", "plan_chain_call", "plan_plain_pair_call", "redo_thr_call", "redo_generic_call", "ltr_dp_plan_kernel"])):
        for base in bases:
            sel = _select(isa[tu], base)
            assert sel, base
            for n, v in sel.items():
                assert v["vgprs"] <= 168, (n, v["vgprs"])
    plan = _select(isa["ltr_k_plan.hip"], "ltr_dp_plan_kernel")
    assert sorted(_targs(n)[0] for n in plan) == ["false", "true"]     # the symmetric-model instance and the general one (round 6)
    for pk in plan.values():
        assert pk["group_segment_fixed_size"] <= 160 * 1024 // 3      # three workgroups per CU: emission table + thresholds + notes
    assert len(_select(isa["ltr_k_plan.hip"], "plan_class_call")) == 40 and len(_select(isa["ltr_k_plan.hip"], "plan_pack_call")) == 40
    assert len(_select(isa["ltr_k_plan.hip"], "redo_thr_call")) == 10                                  # W = 4, 8, 12, 16, 20 for either model
    # workgroup kernels: four waves at three per SIMD (168), eight waves up to W = 18 at four per SIMD (128)
    for n, v in _select(isa["ltr_k_wg.hip"], "ltr_dp_wg_kernel").items():
        w, nw = int(_targs(n)[0]), int(_targs(n)[1])
        if nw == 4:
            assert v["vgprs"] <= 168, (n, v["vgprs"])
        elif nw == 8:
            assert v["vgprs"] <= (128 if w <= 18 else 168), (n, v["vgprs"])
    # threshold kernels as a first pass (ltr_dp_wg_kernel<W, NW, true, true>, round 6), even strip widths: four waves -- their LDS
    # (emission table + thresholds + rings) admits two workgroups a CU, so up to 256 registers; eight waves -- 128 up to W = 10
    # (two workgroups a CU = four waves per SIMD), 168 beyond
    wgt = _select(isa["ltr_k_wgt.hip"], "ltr_dp_wg_kernel")
    assert sorted((int(_targs(n)[1]), int(_targs(n)[0])) for n in wgt) == [(4, w) for w in range(6, 21, 2)] + [(8, w) for w in range(8, 21, 2)]
    for n, v in wgt.items():
        w, nw = int(_targs(n)[0]), int(_targs(n)[1])
        assert _targs(n)[3] == "true"
        assert v["vgprs"] <= (256 if nw == 4 else (128 if w <= 10 else 168)), (n, v["vgprs"])
        assert v["group_segment_fixed_size"] <= 160 * 1024 // 2, (n, v["group_segment_fixed_size"])
    # the long pairs' exact lists (ltr_dp_wgx_kernel: the threshold bodies; rounds 2-5: running maxima, 127 / 173 spilled SGPRs at
    # two waves per SIMD).  The eight-wave list's reads of up to 5121 bases go through the first-pass kernel of 10 columns (above:
    # 128 registers, four waves per SIMD); no step loop of these kernels touches scratch
    wgx = _select(isa["ltr_k_exact.hip"], "ltr_dp_wgx_kernel")
    assert sorted(tuple(int(a) for a in _targs(n)) for n in wgx) == [(4, 6, 10, 14), (8, 12, 16, 20)]
    for n, v in wgx.items():
        a = tuple(int(x) for x in _targs(n))
        assert v["vgprs"] <= (256 if a[0] == 4 else 168), (n, v["vgprs"])
        for L in v["step_loops"]:
            assert L["scratch"] == 0, (n, L)
    # exact kernels (ltr_dp_kernel<W, true, ..>): W = 4 four waves per SIMD asked / three got, 10 and 16 three, 20 two
    for n, v in _select(isa["ltr_k_exact.hip"], "ltr_dp_kernel").items():
        w = int(_targs(n)[0])
        assert v["vgprs"] <= {4: 168, 8: 168, 10: 168, 16: 168, 20: 256}[w], (n, v["vgprs"])


def test_nw_kernel_budgets():
    """ltr_nw_wave_kernel<W> (haplotype -> reference-haplotype NW, ltr_nw.hip): strips of up to 16 columns are built for three
    waves per SIMD (LTR_NW_LB3_MAXW), 20 columns for two; only the 16-column body may touch scratch (a few words, measured faster
    than two waves per SIMD), and every pop of the task queue is one atomic."""
    f = isa_util.analyse("ltr_nw.hip", cache_dir=CACHE)
    sel = _select(f, "ltr_nw_wave_kernel")
    assert sorted(int(_targs(n)[0]) for n in sel) == [4, 8, 12, 16, 20]
    for n, v in sel.items():
        w = int(_targs(n)[0])
        assert v["vgprs"] <= (168 if w <= 16 else 256), (n, v["vgprs"])
        assert v["scratch"] <= (0 if w != 16 else 64), (n, v["scratch"])
        assert v["atomics"] == 1, (n, v["atomics"])


def test_no_scratch_access_inside_a_wavefront_step(isa):
    """The step loop (4W carried doubles, one row of the DP per trip) must not spill: a scratch access there is on the critical
    path of every cell.  Known exceptions, as built: the eight-wave workgroup kernels of W = 17 / 18 at four waves per SIMD."""
    known = {("ltr_dp_wg_kernel", 17, 8): 2, ("ltr_dp_wg_kernel", 18, 8): 5}
    checked = 0
    for tu, bases in (("ltr_k_one.hip", ["ltr_dp_kernel", "class_walk_call"]), ("ltr_k_pack.hip", ["ltr_dp_pack_kernel", "pack_walk_call"]),
                      ("ltr_k_plan.hip", ["plan_class_call", "plan_pack_call"]), ("ltr_k_wg.hip", ["ltr_dp_wg_kernel"]), ("ltr_k_wgt.hip", ["ltr_dp_wg_kernel"])):
        for base in bases:
            for n, v in _select(isa[tu], base).items():
                a = _targs(n)
                if base == "ltr_dp_kernel" and a[1] != "false":
                    continue
                w = int(a[0])
                if w == 1:
                    continue                                   # (one column per lane: 11 FP64 operations a step -- the detector's "step loop" is the pair loop)
                allowed = known.get((base, w, int(a[1])), 0) if (base == "ltr_dp_wg_kernel" and tu == "ltr_k_wg.hip") else 0
                if tu == "ltr_k_wgt.hip" and (w, int(a[1])) == (10, 8):
                    allowed = 3      # the threshold body of 10 columns at 128 registers with its band-skipping copies: the read's three row offsets are reloaded every step (measured faster than without the copies: profiles/r06/band_ab.log)
                assert v["step_loops"], n
                # (a wavefront step is >= 11 FP64 operations per column of the strip: a loop with fewer is set-up code the
                # layout put between a spin loop's label and its backward branch)
                # (the threshold kernels: the detector also takes stretches of the pair loop -- the peeled last step with the two statistics
                # counters behind it -- for loops of their own: a wavefront step never holds an atomic)
                steps = [L for L in v["step_loops"] if L["fp64"] >= 10 * w and (tu != "ltr_k_wgt.hip" or L["atomics"] == 0)]
                assert steps, n
                for L in steps:
                    assert L["scratch"] <= allowed, (n, L)
                    assert L["atomics"] == 0, (n, L)
                    checked += 1
    assert checked > 250
    # the chained walk (off by default): its steady loops spill nothing up to 13 columns a strip, two to five table offsets beyond
    for n, v in _select(isa["ltr_k_plan.hip"], "plan_chain_call").items():
        w = int(_targs(n)[0])
        assert len(v["step_loops"]) >= 2, n                     # the steady copy (twice: with and without a set-up ahead) and the rotation copy
        for L in v["step_loops"]:
            assert L["scratch"] <= (0 if w <= 13 else 6), (n, L)
            assert L["atomics"] == 0, (n, L)
    # the exact bodies of the plan kernel: none up to W = 16, the W = 20 body (168 registers for 20 strips + thresholds) one or two
    # (the general model's 15-operation cell: up to four)
    for n, v in _select(isa["ltr_k_plan.hip"], "redo_thr_call").items():
        w = int(_targs(n)[0])
        for L in v["step_loops"]:
            assert L["scratch"] <= (0 if w <= 16 else (2 if _targs(n)[1] == "true" else 4)), (n, L)


def test_spill_ceilings(isa):
    """Scratch bytes per lane and spilled SGPRs of the certificate kernels: the committed build's figures + slack."""
    for n, v in _select(isa["ltr_k_one.hip"], "ltr_dp_kernel").items():
        a = _targs(n)
        if a[1] != "false" or a[2] != "true":
            continue
        w = int(a[0])
        assert v["scratch"] <= (48 if w <= 15 else 128), (n, v["scratch"])
        assert v["sgpr_spill_count"] <= 18 + 3 * w, (n, v["sgpr_spill_count"])       # (as built: 16 at W = 1 .. 52 at W = 20; the two buffer descriptors of the step loads are eight scalar registers)
        assert v["vgpr_spill_count"] <= (48 if w == 1 else 12), (n, v["vgpr_spill_count"])
    for n, v in _select(isa["ltr_k_pack.hip"], "ltr_dp_pack_kernel").items():
        if _targs(n)[1] != "true":
            continue
        assert v["scratch"] <= 16 and v["sgpr_spill_count"] <= 24 and v["vgpr_spill_count"] <= 4, (n, v["scratch"], v["sgpr_spill_count"])
    # what a call saves on entry (callee-saved registers) + its own spills: per lane, as built 204 .. 368 bytes
    for n, v in _select(isa["ltr_k_plan.hip"], "plan_class_call").items():
        assert v["scratch"] <= 400, (n, v["scratch"])
    for n, v in _select(isa["ltr_k_plan.hip"], "plan_pack_call").items():
        assert v["scratch"] <= 300, (n, v["scratch"])
    for pk in _select(isa["ltr_k_plan.hip"], "ltr_dp_plan_kernel").values():
        assert pk["sgpr_spill_count"] <= 115                        # (as built: 109 / 101 -- the walk over the entry table is cold code)


def test_a_queue_pop_is_one_lanes_atomic(isa):
    """Every returning global atomic of the kernels that pop: issued under an exec mask saved by s_and_saveexec (one lane), its
    value broadcast by v_readfirstlane (or written by that lane alone), not inside a wavefront step and not inside a short loop
    (a waterfall loop over lanes would be the 64-lane same-address atomic again)."""
    expect = {  # function base -> returning atomics in its body (pop; + list append of a failed certificate; the packed walk pops ahead)
        ("ltr_k_one.hip", "ltr_dp_kernel"): 2, ("ltr_k_one.hip", "class_walk_call"): 2,
        ("ltr_k_pack.hip", "ltr_dp_pack_kernel"): 4, ("ltr_k_pack.hip", "pack_walk_call"): 4,
        ("ltr_k_plan.hip", "plan_class_call"): 1, ("ltr_k_plan.hip", "plan_pack_call"): 2,
        ("ltr_k_wg.hip", "ltr_dp_wg_kernel"): 2,
    }
    total = 0
    for (tu, base), n_expected in expect.items():
        sel = _select(isa[tu], base)
        assert sel
        for n, v in sel.items():
            a = _targs(n)
            if base == "ltr_dp_kernel" and a[1] != "false":
                continue
            if base == "ltr_dp_wg_kernel" and a[1] == "1":
                continue                                       # (the one-wave latency variant, pair_packing = 2 only: never chosen by rule)
            pops = v["pops"]
            if base == "ltr_dp_wg_kernel":
                # (+ the two counters the context learns its first pass from: no return value, one lane, once per pair)
                assert all(p["guarded"] or not p["returns"] for p in pops), (n, pops)
                pops = [p for p in pops if p["returns"]]
            assert len(pops) == n_expected, (n, len(pops))
            # (the packed walk's two list appends sit in a region several last lanes of a group may share: not one lane's by construction)
            n_guarded = sum(1 for p in pops if p["guarded"])
            assert n_guarded >= (2 if base in ("ltr_dp_pack_kernel", "pack_walk_call") else n_expected), (n, pops)
            for p in pops:
                assert not p["in_short_loop"], (n, p)
                total += 1
    assert total > 300
    # the plan kernel itself: the pop of its entries of kind 2, the statistics counters; guarded as well
    for pk in _select(isa["ltr_k_plan.hip"], "ltr_dp_plan_kernel").values():
        assert 1 <= len([p for p in pk["pops"] if p["returns"]]) <= 3 and any(p["guarded"] for p in pk["pops"]) and not any(p["in_short_loop"] for p in pk["pops"])
    # the exact kernels pop too
    for n, v in _select(isa["ltr_k_exact.hip"], "ltr_dp_kernel").items():
        assert v["pops"] and all(p["guarded"] and not p["in_short_loop"] for p in v["pops"]), n


def test_the_budget_check_goes_red_when_the_launch_bounds_are_changed_by_hand():
    """-DLTR_LB=2 (two waves per SIMD asked for): the register allocator takes up to 256 registers and the wide strips use them."""
    asm = isa_util.assembly("ltr_k_one.hip", extra=("-DLTR_LB=2",), cache_dir=CACHE)
    f = isa_util.parse(asm)
    bad = budget_violations(f)
    assert bad and all(v > b for _, v, b in bad), bad[:3]

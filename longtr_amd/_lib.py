"""Loader + thin Python binding of libltr_gpu.so (the C-ABI of include/ltr_gpu.h).

The product path is the HIP library.  There is no Python/numpy/torch compute
fallback here: a missing library raises at import-of-use time, a missing GPU
makes every compute call raise LtrError(LTR_ERR_NO_DEVICE).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("LTR_GPU_LIB") or os.path.join(CSRC, "libltr_gpu.so")   # override: A/B builds only
SOURCES = ["ltr_gpu.hip", "ltr_short.hip", "ltr_nw.hip", "ltr_host.cpp", "ltr_genotype.cpp", "ltr_vcf.cpp", "ltr_prep.cpp", "ltr_io.cpp", "ltr_bam.cpp"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-honor-nans", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall"]
LINK_LIBS = ["-lz"]                                         # BGZF blocks of the VCF writer (ltr_io.cpp)

# every symbol include/ltr_gpu.h declares
EXPORTS = [
    "ltr_default_params", "ltr_default_stutter_params", "ltr_ctx_set_stutter_params", "ltr_ctx_set_pair_packing", "ltr_ctx_set_debug", "ltr_ctx_wg_first_pass", "ltr_ctx_create", "ltr_ctx_destroy", "ltr_ctx_set_params", "ltr_last_error",
    "ltr_ctx_device_info", "ltr_align_batch", "ltr_plan_create", "ltr_plan_destroy", "ltr_plan_num_pairs",
    "ltr_plan_ll_size", "ltr_plan_cells", "ltr_plan_input_bytes", "ltr_plan_execute", "ltr_plan_fetch",
    "ltr_plan_last_kernel_ms", "ltr_num_kernels", "ltr_kernel_lanes_per_pair", "ltr_kernel_family", "ltr_plan_set_timing", "ltr_plan_kernel_stats", "ltr_plan_kernel_ranges", "ltr_plan_debug_wave_clocks", "ltr_plan_debug_entries", "ltr_plan_kernel_class", "ltr_process_reads", "ltr_calc_hap_aln_probs", "ltr_haplotype_num_combs", "ltr_haplotype_seq",
    "ltr_trim_alignment", "ltr_pool_reads", "ltr_scatter_pool_probs", "ltr_posteriors", "ltr_plan_posteriors", "ltr_extract_genotypes", "ltr_ctx_timers", "ltr_haps_to_alleles", "ltr_unused_alleles", "ltr_remap_haplotypes",
    "ltr_remap_aln_probs", "ltr_default_vcf_options", "ltr_get_alleles", "ltr_vcf_record", "ltr_vcf_header", "ltr_haplotype_aln_info_capacity",
    "ltr_haplotype_align_to_ref", "ltr_left_align_reads", "ltr_phasing_priors", "ltr_read_set_size", "ltr_read_set_alignments",
    "ltr_read_set_alignment_strings", "ltr_read_set_deleted", "ltr_read_set_source", "ltr_read_set_sample", "ltr_read_set_n_p1s",
    "ltr_read_set_n_p2s", "ltr_read_set_fail_count", "ltr_read_set_free", "ltr_extract_sequence", "ltr_build_haplotype",
    "ltr_hap_result_blocks", "ltr_hap_result_failure", "ltr_hap_result_unplaced_reads", "ltr_hap_result_samples_needing_clustering",
    "ltr_hap_result_free", "ltr_version", "ltr_abi_version", "ltr_ctx_timers_n", "ltr_ctx_short_kernel_split", "ltr_ctx_set_host_threads", "ltr_ctx_host_threads", "ltr_host_threads_rule", "ltr_debug_parallel_threads", "ltr_debug_prep_ahead_rule", "ltr_debug_num_classes", "ltr_debug_class_info", "ltr_debug_classify", "ltr_debug_sort_by_class", "ltr_debug_pair_costs", "ltr_debug_threshold_table", "ltr_debug_calc_seed_base",
    "ltr_read_regions", "ltr_region_set_size", "ltr_region_set_lines_read", "ltr_region_set_order", "ltr_region_set_free", "ltr_region_chrom",
    "ltr_region_name", "ltr_region_motif", "ltr_region_period_str", "ltr_region_start", "ltr_region_stop", "ltr_region_period",
    "ltr_fasta_open", "ltr_fasta_close", "ltr_fasta_num_seqs", "ltr_fasta_seq_name", "ltr_fasta_seq_len", "ltr_fasta_fetch", "ltr_fasta_contig_lines",
    "ltr_vcf_writer_open", "ltr_vcf_writer_header", "ltr_vcf_writer_add_record", "ltr_vcf_writer_close",
    "ltr_bam_open", "ltr_bam_close", "ltr_bam_num_refs", "ltr_bam_ref_name", "ltr_bam_ref_len", "ltr_bam_num_read_groups", "ltr_bam_read_group_id",
    "ltr_bam_read_group_sample", "ltr_bam_read_group_library", "ltr_bam_read_group_file", "ltr_bam_set_region", "ltr_bam_next",
    "ltr_bam_aux_int", "ltr_bam_aux_float", "ltr_bam_aux_char", "ltr_bam_aux_string",
]


FAMILIES = {0: "one-wave", 1: "packed", 2: "workgroup", 3: "exact"}      # ltr_kernel_family()


class LtrError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"ltr error {code}: {msg}")
        self.code = code


KERNEL_TUS = ["ltr_k_one.hip", "ltr_k_pack.hip", "ltr_k_plan.hip", "ltr_k_wg.hip", "ltr_k_wgt.hip", "ltr_k_exact.hip", "ltr_plan.cpp"]      # one family of DP kernels each
SOURCES = ["ltr_gpu.hip"] + KERNEL_TUS + SOURCES[1:]
OBJ_DIR = os.path.join(CSRC, "build")


def build(force=False, extra_flags=()):
    """hipcc cross-compiles for gfx950 without a GPU; the .so stays in-tree.  Every source is its own
    translation unit (objects under csrc/build/, rebuilt when the source or any header is newer), compiled
    side by side on the host cores, then linked."""
    import glob
    from concurrent.futures import ThreadPoolExecutor
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "ltr_gpu.h")]
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags)
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ_DIR, s + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])
    objs = [os.path.join(OBJ_DIR, s + ".o") for s in SOURCES]
    if not jobs and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        for r in ex.map(lambda cmd: subprocess.run(cmd), jobs):
            if r.returncode != 0:
                raise subprocess.CalledProcessError(r.returncode, r.args)
    open(stamp, "w").write(" ".join(flags))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + LINK_LIBS + ["-o", LIB_PATH], check=True)
    return LIB_PATH


def source_id():
    """Build-path-independent id of the library: SHA-256 (16 hex digits) over the compile flags and the bytes of every source and
    header the .so is built from, in name order.  The .so's own hash changes with the directory it was built in (paths in its
    debug / assert strings); committed counter summaries (profiles/) are matched to a run by THIS id."""
    import glob
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    files = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.hip"))
                   + glob.glob(os.path.join(CSRC, "*.cpp")) + [os.path.join(HERE, "..", "include", "ltr_gpu.h")], key=os.path.basename)
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the alignment path)")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.ltr_version.restype = C.c_char_p
    L.ltr_default_params.argtypes = [C.POINTER(_abi.AlignParams)]
    L.ltr_default_params.restype = None
    L.ltr_default_stutter_params.argtypes = [C.POINTER(_abi.StutterParams)]
    L.ltr_default_stutter_params.restype = None
    L.ltr_ctx_set_stutter_params.argtypes = [vp, C.POINTER(_abi.StutterParams)]
    L.ltr_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ltr_ctx_destroy.argtypes = [vp]
    L.ltr_ctx_destroy.restype = None
    L.ltr_ctx_set_params.argtypes = [vp, C.POINTER(_abi.AlignParams)]
    L.ltr_last_error.argtypes = [vp]
    L.ltr_last_error.restype = C.c_char_p
    L.ltr_ctx_device_info.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ltr_align_batch.argtypes = [vp, C.POINTER(_abi.LocusBatch), vp, vp]
    L.ltr_plan_create.argtypes = [vp, C.POINTER(_abi.LocusBatch), C.POINTER(vp)]
    L.ltr_plan_destroy.argtypes = [vp]
    L.ltr_plan_destroy.restype = None
    L.ltr_plan_num_pairs.argtypes = [vp]
    L.ltr_plan_num_pairs.restype = i64
    L.ltr_plan_ll_size.argtypes = [vp]
    L.ltr_plan_ll_size.restype = i64
    L.ltr_plan_cells.argtypes = [vp]
    L.ltr_plan_cells.restype = dbl
    L.ltr_plan_input_bytes.argtypes = [vp]
    L.ltr_plan_input_bytes.restype = dbl
    L.ltr_plan_execute.argtypes = [vp, vp, vp]
    L.ltr_plan_fetch.argtypes = [vp, vp, vp]
    L.ltr_plan_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.ltr_plan_set_timing.argtypes = [vp, C.c_int]
    L.ltr_plan_kernel_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(i64), C.POINTER(dbl),
                                        C.POINTER(C.c_float)]
    L.ltr_plan_kernel_ranges.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(i64)]
    L.ltr_process_reads.argtypes = [vp, C.POINTER(_abi.HaplotypeBlocks), vp, C.POINTER(_abi.Alignment), i32, i32,
                                    vp, vp, vp]
    L.ltr_calc_hap_aln_probs.argtypes = [vp, C.POINTER(_abi.Locus), i64, C.POINTER(vp), C.POINTER(vp)]
    L.ltr_haplotype_num_combs.argtypes = [C.POINTER(_abi.HaplotypeBlocks)]
    L.ltr_haplotype_num_combs.restype = i64
    L.ltr_haplotype_seq.argtypes = [C.POINTER(_abi.HaplotypeBlocks), i64, vp, i64]
    L.ltr_haplotype_seq.restype = i64
    L.ltr_trim_alignment.argtypes = [C.POINTER(_abi.Alignment), i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.ltr_pool_reads.argtypes = [C.POINTER(vp), vp, i32, vp]
    L.ltr_pool_reads.restype = i32
    L.ltr_scatter_pool_probs.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.ltr_plan_posteriors.argtypes = [vp, C.POINTER(_abi.PosteriorBatch), vp, vp, vp]
    L.ltr_posteriors.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, C.POINTER(dbl)]
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """ltr_ctx: one per GPU (pass LOCAL_RANK)."""

    def __init__(self, device=0, params=None):
        self._h = C.c_void_p()
        rc = lib().ltr_ctx_create(int(device), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise LtrError(rc, "ltr_ctx_create failed (no HIP device? the alignment path has no CPU fallback)")
        self.params = _abi.default_params()
        if params is not None:
            self.set_params(params)

    def _check(self, rc):
        if rc != 0:
            raise LtrError(rc, lib().ltr_last_error(self._h).decode())

    def set_params(self, params):
        self._check(lib().ltr_ctx_set_params(self._h, C.byref(params)))
        self.params = params

    def set_pair_packing(self, mode):
        """-1: two pairs per wavefront when the batch is large (default); 0 never; 1 whenever the read fits."""
        self._check(lib().ltr_ctx_set_pair_packing(self._h, int(mode)))

    def set_debug(self, key, value):
        """ltr_ctx_set_debug: measurement switches (fan_lanes, fan_pairs, chunks, chunk_streams, chunk_growth, trace)."""
        lib().ltr_ctx_set_debug.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        self._check(lib().ltr_ctx_set_debug(self._h, key.encode(), float(value)))

    def set_host_threads(self, n):
        """ltr_ctx_set_host_threads: the process's host-thread budget (0 = the rule: affinity mask, cgroup quota, LOCAL_WORLD_SIZE)."""
        self._check(lib().ltr_ctx_set_host_threads(self._h, int(n)))

    def host_threads(self):
        lib().ltr_ctx_host_threads.argtypes = [C.c_void_p]
        return int(lib().ltr_ctx_host_threads(self._h))

    def short_kernel_split(self, reset=True):
        """ltr_ctx_short_kernel_split: device ms of the seeded path's launches (needs set_debug("short_split", 1))."""
        out = (C.c_double * 4)()
        lib().ltr_ctx_short_kernel_split.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        self._check(lib().ltr_ctx_short_kernel_split(self._h, out, int(reset)))
        return list(out)

    def wg_first_pass(self):
        """ltr_ctx_wg_first_pass: (mode, pairs the last read execute's first pass could not finish, pairs it scored)."""
        lib().ltr_ctx_wg_first_pass.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        a, b = C.c_int64(0), C.c_int64(0)
        rc = lib().ltr_ctx_wg_first_pass(self._h, C.byref(a), C.byref(b))
        if rc < 0:
            self._check(rc)
        return rc, a.value, b.value

    def set_stutter_params(self, sp):
        self._check(lib().ltr_ctx_set_stutter_params(self._h, C.byref(sp)))

    def device_info(self):
        buf = C.create_string_buffer(64)
        ncu, mhz = C.c_int(0), C.c_int(0)
        self._check(lib().ltr_ctx_device_info(self._h, buf, 64, C.byref(ncu), C.byref(mhz)))
        return dict(arch=buf.value.decode(), n_cu=ncu.value, clock_mhz=mhz.value)

    def align_batch(self, batch, out_ll=None):
        """ltr_align_batch: host in, host out.  Returns (ll, seed)."""
        ll = np.full(max(batch.ll_size, 1), np.nan, dtype=np.float64) if out_ll is None else out_ll
        seed = np.full(max(batch.n_reads, 1), -1, dtype=np.int32)
        self._check(lib().ltr_align_batch(self._h, C.byref(batch.struct), _p(ll), _p(seed)))
        return ll[:batch.ll_size], seed[:batch.n_reads]

    def plan(self, batch):
        return Plan(self, batch)

    def process_reads(self, blocks, alns, realign_hap=None, realign_read=None, init_read_index=0):
        """HapAligner::process_reads for one locus (raw alignments + haplotype blocks)."""
        ph = _abi.PackedHaplotype(blocks)
        pa = _abi.PackedAlignments(alns)
        H = ph.num_combs
        probs = np.full((init_read_index + len(alns)) * H, np.nan, dtype=np.float64)
        seeds = np.full(init_read_index + len(alns), -12345, dtype=np.int32)
        rh = None if realign_hap is None else np.ascontiguousarray(realign_hap, dtype=np.uint8)
        rr = None if realign_read is None else np.ascontiguousarray(realign_read, dtype=np.uint8)
        self._check(lib().ltr_process_reads(self._h, C.byref(ph.struct), _p(rh), pa.array, len(alns),
                                            init_read_index, _p(rr), _p(probs), _p(seeds)))
        return probs.reshape(-1, H), seeds

    def pack_haplotypes(self, loci_blocks):
        """ctypes image of the haplotypes of many loci for haplotype_align_to_ref_packed (+ the output buffers)."""
        phs = [_abi.PackedHaplotype(b) for b in loci_blocks]
        arr = (C.POINTER(_abi.HaplotypeBlocks) * max(len(phs), 1))(*[C.pointer(p.struct) for p in phs])
        L = lib()
        L.ltr_haplotype_aln_info_capacity.restype = C.c_int64
        L.ltr_haplotype_aln_info_capacity.argtypes = [C.c_void_p, C.c_int64]
        cap = L.ltr_haplotype_aln_info_capacity(arr, len(phs))
        if cap < 0:
            raise LtrError(int(cap), "ltr_haplotype_aln_info_capacity")
        nh = sum(p.num_combs for p in phs)
        return dict(phs=phs, arr=arr, n=len(phs), cap=int(cap), buf=C.create_string_buffer(max(int(cap), 1)),
                    off=np.zeros(nh + 1, dtype=np.int64))

    def haplotype_align_to_ref_packed(self, packed, decode=True):
        L = lib()
        L.ltr_haplotype_align_to_ref.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p]
        self._check(L.ltr_haplotype_align_to_ref(self._h, packed["arr"], packed["n"], packed["buf"], packed["cap"], _p(packed["off"])))
        if not decode:
            return None
        raw, off, out, h = packed["buf"].raw, packed["off"], [], 0
        for p in packed["phs"]:
            out.append([raw[off[h + k]:off[h + k + 1]].decode() for k in range(p.num_combs)])
            h += p.num_combs
        return out

    def haplotype_align_to_ref(self, loci_blocks):
        """ltr_haplotype_align_to_ref: per locus the list of hap_aln_info_ strings (Haplotype::next() order)."""
        return self.haplotype_align_to_ref_packed(self.pack_haplotypes(loci_blocks))

    def timers(self, reset=False):
        """ltr_ctx_timers: the reference's hap-build / hap-align / posterior clocks (+ DP kernel device time)."""
        t = _abi.Timers()
        lib().ltr_ctx_timers.argtypes = [C.c_void_p, C.POINTER(_abi.Timers), C.c_int]
        self._check(lib().ltr_ctx_timers(self._h, C.byref(t), int(bool(reset))))
        return {f: getattr(t, f) for f, _ in _abi.Timers._fields_}

    @staticmethod
    def pack_loci(loci, out_init=None, contiguous=False):
        """ctypes image of a list of (blocks, alns[, second_mate[, masks]]) for ltr_calc_hap_aln_probs;
        masks = dict(realign_to_hap=, realign_pool=, copy_read=) (each optional).  out_init: optional list of
        [R x H] arrays the output matrices start from (cells a mask leaves untouched keep these values).
        contiguous: the per-locus matrices are slices of ONE array ("flat", offsets "flat_off"): what a rank hands to the gather."""
        keep, arr = [], (_abi.Locus * max(len(loci), 1))()
        outs = []
        flat = flat_off = None
        if contiguous:
            sizes = [len(item[1]) * _abi.PackedHaplotype(item[0]).num_combs for item in loci]
            flat_off = np.zeros(len(loci) + 1, dtype=np.int64)
            flat_off[1:] = np.cumsum(sizes)
            flat = np.full(max(int(flat_off[-1]), 1), np.nan, dtype=np.float64)
        pp = (C.c_void_p * max(len(loci), 1))()
        sp = (C.c_void_p * max(len(loci), 1))()
        for i, item in enumerate(loci):
            blocks, alns = item[0], item[1]
            sm = item[2] if len(item) > 2 else None
            masks = item[3] if len(item) > 3 and item[3] else {}
            ph, pa = _abi.PackedHaplotype(blocks), _abi.PackedAlignments(alns)
            u8 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.uint8)
            u8p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None
            smv = u8(sm)
            mk = [u8(masks.get(k)) for k in ("realign_to_hap", "realign_pool", "copy_read")]
            probs = np.full(len(alns) * ph.num_combs, np.nan, dtype=np.float64) if flat is None else flat[flat_off[i]:flat_off[i + 1]]
            if out_init is not None:
                probs[:] = np.asarray(out_init[i], dtype=np.float64).ravel()
            seeds = np.full(max(len(alns), 1), -12345, dtype=np.int32)
            keep.append((ph, pa, smv, mk))
            arr[i].hap = C.pointer(ph.struct)
            arr[i].alns = pa.array
            arr[i].n_alns = len(alns)
            arr[i].second_mate = u8p(smv)
            arr[i].realign_to_hap, arr[i].realign_pool, arr[i].copy_read = u8p(mk[0]), u8p(mk[1]), u8p(mk[2])
            pp[i] = probs.ctypes.data
            sp[i] = seeds.ctypes.data
            outs.append((probs.reshape(len(alns), ph.num_combs), seeds[:len(alns)]))
        return dict(n=len(loci), arr=arr, pp=pp, sp=sp, outs=outs, keep=keep, flat=flat, flat_off=flat_off)

    def calc_hap_aln_probs_packed(self, packed):
        self._check(lib().ltr_calc_hap_aln_probs(self._h, packed["arr"], packed["n"], packed["pp"], packed["sp"]))
        return packed["outs"]

    def calc_hap_aln_probs(self, loci):
        """ltr_calc_hap_aln_probs.  loci: list of (blocks, alns[, second_mate]).  Returns per locus
        (log_aln_probs [R x H], seed_positions [R])."""
        return self.calc_hap_aln_probs_packed(self.pack_loci(loci))

    def posteriors(self, ll, log_p1, log_p2, sample_label, n_samples, haploid=False):
        ll = np.array(ll, dtype=np.float64, copy=True)
        R, H = ll.shape
        p1 = np.ascontiguousarray(log_p1, dtype=np.float64)
        p2 = np.ascontiguousarray(log_p2, dtype=np.float64)
        sl = np.ascontiguousarray(sample_label, dtype=np.int32)
        post = np.zeros(n_samples * H * H, dtype=np.float64)
        stl = np.zeros(n_samples, dtype=np.float64)
        gts = np.zeros(2 * n_samples, dtype=np.int32)
        tot = C.c_double(0.0)
        self._check(lib().ltr_posteriors(self._h, n_samples, R, H, _p(ll), _p(p1), _p(p2), _p(sl), int(haploid),
                                         _p(post), _p(stl), _p(gts), C.byref(tot)))
        return dict(post=post.reshape(n_samples, H, H), sample_total_ll=stl, gts=gts.reshape(n_samples, 2),
                    total_ll=tot.value, clamped_ll=ll)

    def close(self):
        if self._h:
            lib().ltr_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- on-disk formats (ltr_io.cpp; host only, no GPU needed) -----------------------------------------
def read_regions(path, max_regions=0xffffffff, chrom_limit=None, order=False):
    """ltr_read_regions (+ ltr_region_set_order): list of dicts chrom / start / stop / motif / name / period / period_str;
    the reference's "Region file contains N regions" count comes back as the second value."""
    L = lib()
    L.ltr_read_regions.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
    for f in ("chrom", "name", "motif", "period_str"):
        getattr(L, "ltr_region_" + f).restype = C.c_char_p
        getattr(L, "ltr_region_" + f).argtypes = [C.c_void_p, C.c_int64]
    for f in ("start", "stop", "period"):
        getattr(L, "ltr_region_" + f).restype = C.c_int32
        getattr(L, "ltr_region_" + f).argtypes = [C.c_void_p, C.c_int64]
    L.ltr_region_set_size.restype = C.c_int64
    L.ltr_region_set_size.argtypes = [C.c_void_p]
    L.ltr_region_set_lines_read.argtypes = [C.c_void_p]
    L.ltr_region_set_order.argtypes = [C.c_void_p]
    L.ltr_region_set_free.argtypes = [C.c_void_p]
    h = C.c_void_p()
    err = C.create_string_buffer(4096)
    rc = L.ltr_read_regions(os.fsencode(path), int(max_regions), None if not chrom_limit else chrom_limit.encode(), C.byref(h), err, len(err))
    if rc != 0:
        raise LtrError(rc, err.value.decode(errors="replace"))
    try:
        if order:
            L.ltr_region_set_order(h)
        out = []
        for i in range(L.ltr_region_set_size(h)):
            out.append(dict(chrom=L.ltr_region_chrom(h, i).decode(), start=L.ltr_region_start(h, i), stop=L.ltr_region_stop(h, i),
                            motif=L.ltr_region_motif(h, i).decode(), name=L.ltr_region_name(h, i).decode(),
                            period=L.ltr_region_period(h, i), period_str=L.ltr_region_period_str(h, i).decode()))
        return out, int(L.ltr_region_set_lines_read(h))
    finally:
        L.ltr_region_set_free(h)


class Fasta:
    """ltr_fasta: FastaReader (one indexed FASTA file or a directory of *.fa)."""

    def __init__(self, path):
        L = lib()
        L.ltr_fasta_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        L.ltr_fasta_close.argtypes = [C.c_void_p]
        L.ltr_fasta_num_seqs.restype = C.c_int64; L.ltr_fasta_num_seqs.argtypes = [C.c_void_p]
        L.ltr_fasta_seq_name.restype = C.c_char_p; L.ltr_fasta_seq_name.argtypes = [C.c_void_p, C.c_int64]
        L.ltr_fasta_seq_len.restype = C.c_int64; L.ltr_fasta_seq_len.argtypes = [C.c_void_p, C.c_char_p]
        L.ltr_fasta_fetch.restype = C.c_int64
        L.ltr_fasta_fetch.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_char_p, C.c_int64, C.c_char_p, C.c_int]
        L.ltr_fasta_contig_lines.restype = C.c_int64; L.ltr_fasta_contig_lines.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        self._h = C.c_void_p()
        err = C.create_string_buffer(4096)
        rc = L.ltr_fasta_open(os.fsencode(path), C.byref(self._h), err, len(err))
        if rc != 0:
            self._h = None
            raise LtrError(rc, err.value.decode(errors="replace"))

    def names(self):
        return [lib().ltr_fasta_seq_name(self._h, i).decode() for i in range(lib().ltr_fasta_num_seqs(self._h))]

    def seq_len(self, chrom):
        return int(lib().ltr_fasta_seq_len(self._h, chrom.encode()))

    def fetch(self, chrom, start, end):
        """0-based, end inclusive (FastaReader::get_sequence)."""
        cap = max(int(end) - max(int(start), 0) + 2, 1)
        buf = C.create_string_buffer(cap)
        err = C.create_string_buffer(1024)
        n = lib().ltr_fasta_fetch(self._h, chrom.encode(), int(start), int(end), buf, cap, err, len(err))
        if n < 0:
            raise LtrError(int(n), err.value.decode(errors="replace"))
        return buf.raw[:n].decode()

    def contig_lines(self):
        buf = C.create_string_buffer(1 << 20)
        n = lib().ltr_fasta_contig_lines(self._h, buf, len(buf))
        if n < 0:
            raise LtrError(int(n), "ltr_fasta_contig_lines")
        return buf.raw[:n].decode()

    def close(self):
        if self._h:
            lib().ltr_fasta_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def vcf_header(fasta_path, full_command, contig_lines, sample_names, options=None):
    """ltr_vcf_header = Genotyper::get_vcf_header (genotyper.cpp:258-336)."""
    L = lib()
    L.ltr_vcf_header.restype = C.c_int64
    L.ltr_vcf_header.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_char_p), C.c_int32, C.c_char_p, C.c_int64]
    names = (C.c_char_p * max(len(sample_names), 1))(*[n.encode() for n in sample_names])
    cap = 1 << 16
    cl = None if contig_lines is None else contig_lines.encode()
    cap += len(cl or b"") + len(fasta_path) + len(full_command) + sum(len(n) + 1 for n in sample_names)
    buf = C.create_string_buffer(cap)
    n = L.ltr_vcf_header(fasta_path.encode(), full_command.encode(), cl, C.byref(options) if options is not None else None, names,
                         len(sample_names), buf, cap)
    if n < 0:
        raise LtrError(int(n), "ltr_vcf_header")
    return buf.raw[:n].decode()


class VcfWriter:
    """ltr_vcf_writer: the reference's position-ordered VCFWriter (BGZF for *.gz / *.bgz paths)."""

    def __init__(self, path):
        L = lib()
        L.ltr_vcf_writer_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.ltr_vcf_writer_header.argtypes = [C.c_void_p, C.c_char_p]
        L.ltr_vcf_writer_add_record.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_char_p]
        L.ltr_vcf_writer_close.argtypes = [C.c_void_p]
        self._h = C.c_void_p()
        rc = L.ltr_vcf_writer_open(os.fsencode(path), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise LtrError(rc, "ltr_vcf_writer_open")

    def header(self, text):
        rc = lib().ltr_vcf_writer_header(self._h, text.encode())
        if rc != 0:
            raise LtrError(rc, "ltr_vcf_writer_header")

    def add_record(self, chrom, pos, text):
        rc = lib().ltr_vcf_writer_add_record(self._h, chrom.encode(), int(pos), text.encode())
        if rc != 0:
            raise LtrError(rc, "ltr_vcf_writer_add_record")

    def close(self):
        if self._h:
            h, self._h = self._h, None
            rc = lib().ltr_vcf_writer_close(h)
            if rc != 0:
                raise LtrError(rc, "ltr_vcf_writer_close")


class BamRecord(C.Structure):
    _fields_ = [("name", C.c_char_p), ("file_index", C.c_int32), ("ref_id", C.c_int32), ("pos", C.c_int32), ("end_pos", C.c_int32),
                ("mapq", C.c_int32), ("flag", C.c_int32), ("mate_ref_id", C.c_int32), ("mate_pos", C.c_int32), ("tlen", C.c_int32),
                ("length", C.c_int32), ("bases", C.c_char_p), ("quals", C.c_char_p), ("n_cigar", C.c_int32), ("cigar_type", C.c_char_p),
                ("cigar_num", C.POINTER(C.c_int32)), ("aux", C.POINTER(C.c_uint8)), ("aux_len", C.c_int32)]


class Bam:
    """ltr_bam: indexed BAM files read as one stream (BamCramMultiReader)."""

    def __init__(self, paths, merge_by_position=True):
        L = lib()
        L.ltr_bam_open.argtypes = [C.POINTER(C.c_char_p), C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
        L.ltr_bam_close.argtypes = [C.c_void_p]
        L.ltr_bam_num_refs.argtypes = [C.c_void_p]
        L.ltr_bam_ref_name.restype = C.c_char_p; L.ltr_bam_ref_name.argtypes = [C.c_void_p, C.c_int32]
        L.ltr_bam_ref_len.restype = C.c_int64; L.ltr_bam_ref_len.argtypes = [C.c_void_p, C.c_int32]
        L.ltr_bam_num_read_groups.argtypes = [C.c_void_p]
        for f in ("id", "sample", "library"):
            getattr(L, "ltr_bam_read_group_" + f).restype = C.c_char_p
            getattr(L, "ltr_bam_read_group_" + f).argtypes = [C.c_void_p, C.c_int32]
        L.ltr_bam_read_group_file.argtypes = [C.c_void_p, C.c_int32]
        L.ltr_bam_set_region.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32]
        L.ltr_bam_next.argtypes = [C.c_void_p, C.POINTER(BamRecord)]
        L.ltr_bam_aux_int.argtypes = [C.POINTER(BamRecord), C.c_char_p, C.POINTER(C.c_int64)]
        L.ltr_bam_aux_string.restype = C.c_char_p; L.ltr_bam_aux_string.argtypes = [C.POINTER(BamRecord), C.c_char_p]
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        self._h = C.c_void_p()
        err = C.create_string_buffer(2048)
        rc = L.ltr_bam_open(arr, len(paths), int(bool(merge_by_position)), C.byref(self._h), err, len(err))
        if rc != 0:
            self._h = None
            raise LtrError(rc, err.value.decode(errors="replace"))

    def refs(self):
        L = lib()
        return [(L.ltr_bam_ref_name(self._h, i).decode(), int(L.ltr_bam_ref_len(self._h, i))) for i in range(L.ltr_bam_num_refs(self._h))]

    def read_groups(self):
        L = lib()
        return [dict(id=L.ltr_bam_read_group_id(self._h, i).decode(), sample=L.ltr_bam_read_group_sample(self._h, i).decode(),
                     library=L.ltr_bam_read_group_library(self._h, i).decode(), file=L.ltr_bam_read_group_file(self._h, i))
                for i in range(L.ltr_bam_num_read_groups(self._h))]

    def fetch(self, chrom, start, end, tags=()):
        """SetRegion + GetNextAlignment until the stream ends: list of dicts."""
        L = lib()
        rc = L.ltr_bam_set_region(self._h, chrom.encode(), int(start), int(end))
        if rc != 0:
            raise LtrError(rc, "ltr_bam_set_region")
        out, rec = [], BamRecord()
        while True:
            rc = L.ltr_bam_next(self._h, C.byref(rec))
            if rc == 0:
                return out
            if rc < 0:
                raise LtrError(rc, "ltr_bam_next")
            d = dict(name=rec.name.decode(), file=rec.file_index, ref_id=rec.ref_id, pos=rec.pos, end_pos=rec.end_pos, mapq=rec.mapq, flag=rec.flag,
                     mate_ref_id=rec.mate_ref_id, mate_pos=rec.mate_pos, tlen=rec.tlen, seq=rec.bases[:rec.length].decode() if rec.length else "",
                     qual=rec.quals[:rec.length].decode("latin-1") if rec.length else "",
                     cigar=[(rec.cigar_type[k:k + 1].decode(), rec.cigar_num[k]) for k in range(rec.n_cigar)])
            for t in tags:
                v = C.c_int64()
                if L.ltr_bam_aux_int(C.byref(rec), t.encode(), C.byref(v)):
                    d[t] = int(v.value)
                else:
                    sv = L.ltr_bam_aux_string(C.byref(rec), t.encode())
                    if sv is not None:
                        d[t] = sv.decode()
            out.append(d)

    def close(self):
        if self._h:
            lib().ltr_bam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """ltr_plan: one packed batch resident in HBM."""

    def __init__(self, ctx, batch):
        self.ctx = ctx
        self.batch = batch
        self._h = C.c_void_p()
        ctx._check(lib().ltr_plan_create(ctx._h, C.byref(batch.struct), C.byref(self._h)))
        self.num_pairs = lib().ltr_plan_num_pairs(self._h)
        self.ll_size = lib().ltr_plan_ll_size(self._h)
        self.cells = lib().ltr_plan_cells(self._h)
        self.input_bytes = lib().ltr_plan_input_bytes(self._h)

    def execute(self, d_out_ptr=None, stream=None):
        self.ctx._check(lib().ltr_plan_execute(self._h, d_out_ptr, stream))

    def fetch(self):
        ll = np.full(max(self.ll_size, 1), np.nan, dtype=np.float64)
        seed = np.full(max(self.batch.n_reads, 1), -1, dtype=np.int32)
        self.ctx._check(lib().ltr_plan_fetch(self._h, _p(ll), _p(seed)))
        return ll[:self.ll_size], seed[:self.batch.n_reads]

    def wait(self):
        self.ctx._check(lib().ltr_plan_fetch(self._h, None, None))

    def last_kernel_ms(self):
        ms, n = C.c_float(0), C.c_int(0)
        self.ctx._check(lib().ltr_plan_last_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def posteriors(self, locus_read_off, pool_index, log_p1, log_p2, sample_label, n_samples, haploid=False):
        """ltr_plan_posteriors: all loci, from the device-resident LL of the last execute.
        Returns (post_flat, post_off[units+1], sample_total_ll, gts[units,2]); units = (locus, sample) in order."""
        lro = np.ascontiguousarray(locus_read_off, dtype=np.int64)
        pi = np.ascontiguousarray(pool_index, dtype=np.int32)
        p1 = np.ascontiguousarray(log_p1, dtype=np.float64)
        p2 = np.ascontiguousarray(log_p2, dtype=np.float64)
        sl = np.ascontiguousarray(sample_label, dtype=np.int32)
        ns = np.ascontiguousarray(n_samples, dtype=np.int32)
        H = np.diff(self.batch.locus_hap_off)
        sizes = np.repeat(H * H, ns)
        off = np.zeros(len(sizes) + 1, dtype=np.int64)
        off[1:] = np.cumsum(sizes)
        post = np.zeros(max(int(off[-1]), 1), dtype=np.float64)
        stl = np.zeros(max(len(sizes), 1), dtype=np.float64)
        gts = np.zeros(2 * max(len(sizes), 1), dtype=np.int32)
        pb = _abi.PosteriorBatch()
        pb.n_loci = len(ns)
        pb.locus_read_off = lro.ctypes.data_as(C.POINTER(C.c_int64))
        pb.n_reads = len(pi)
        pb.pool_index = pi.ctypes.data_as(C.POINTER(C.c_int32))
        pb.log_p1 = p1.ctypes.data_as(C.POINTER(C.c_double))
        pb.log_p2 = p2.ctypes.data_as(C.POINTER(C.c_double))
        pb.sample_label = sl.ctypes.data_as(C.POINTER(C.c_int32))
        pb.n_samples = ns.ctypes.data_as(C.POINTER(C.c_int32))
        pb.haploid = int(haploid)
        self.ctx._check(lib().ltr_plan_posteriors(self._h, C.byref(pb), _p(post), _p(stl), _p(gts)))
        return post[:off[-1]], off, stl[:len(sizes)], gts[:2 * len(sizes)].reshape(-1, 2)

    def set_timing(self, on=True):
        """on: False / True (every launch as it is launched) / 2 (the multi-width one-wave launch class by class)."""
        self.ctx._check(lib().ltr_plan_set_timing(self._h, int(on)))

    def wave_clocks(self):
        """(n_waves, 4) uint64: first / last wall clock (100 MHz), pairs scored with the exact body, ticks spent there -- per
        wavefront of the plan kernel (debug knob wave_clock)."""
        buf = np.zeros(4 * 4 * 4096 + 4096 + 256, dtype=np.uint64)
        lib().ltr_plan_debug_wave_clocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        n = lib().ltr_plan_debug_wave_clocks(self._h, _p(buf), buf.size)
        if n < 0:
            raise LtrError(n, "ltr_plan_debug_wave_clocks")
        k = int(min(buf[4 * n], 4095))
        self.redo_log = [(int(v >> np.uint64(32)), int(v & np.uint64(0xffffffff))) for v in buf[4 * n + 1:4 * n + 1 + k]]   # (n, m) of the pairs that took the exact body
        self.entry_ticks = buf[4 * n + 4096:4 * n + 4096 + 256].copy()       # per entry of the plan kernel's table: ticks of all wavefronts together
        return buf[:4 * n].reshape(n, 4)

    def plan_entries(self):
        """The plan kernel's table in walk order: dicts (kind, strip_width, pairs, cells)."""
        L = lib()
        L.ltr_plan_debug_entries.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        kind, w, npairs, cells = np.zeros(256, np.int32), np.zeros(256, np.int32), np.zeros(256, np.int64), np.zeros(256, np.float64)
        n = L.ltr_plan_debug_entries(self._h, _p(kind), _p(w), _p(npairs), _p(cells), 256)
        if n < 0:
            raise LtrError(n, "ltr_plan_debug_entries")
        names = {0: "one-wave", 1: "packed", 2: "exact body", 3: "one-wave (chained)"}
        return [dict(kind=names.get(int(kind[i]), "?"), strip_width=int(w[i]), pairs=int(npairs[i]), cells=float(cells[i])) for i in range(min(n, 256))]

    def kernel_stats(self):
        """Per strip-width class: dict(strip_width, pairs, cells, ms) of the last execute."""
        out = []
        lib().ltr_plan_kernel_class.argtypes = [C.c_void_p]
        plan_cls = lib().ltr_plan_kernel_class(self._h)
        for k in range(lib().ltr_num_kernels()):
            w, n, c, ms = C.c_int(0), C.c_int64(0), C.c_double(0), C.c_float(0)
            self.ctx._check(lib().ltr_plan_kernel_stats(self._h, k, C.byref(w), C.byref(n), C.byref(c), C.byref(ms)))
            d = dict(strip_width=w.value, pairs=n.value, cells=c.value, ms=ms.value,
                     lanes_per_pair=lib().ltr_kernel_lanes_per_pair(k),
                     family=FAMILIES.get(lib().ltr_kernel_family(k), "?"))
            lanes, widths, npairs = (C.c_int32 * 256)(), (C.c_int32 * 256)(), (C.c_int64 * 256)()
            nr = lib().ltr_plan_kernel_ranges(self._h, k, lanes, widths, npairs)
            if nr > 0:                                  # a launch over several classes: (lanes per pair, strip width, pairs) in launch order
                d["ranges"] = [(int(lanes[i]), int(widths[i]), int(npairs[i])) for i in range(nr)]
            if k == plan_cls:
                d["plan_kernel"] = True                 # the one launch of every one-wave class and packed width (ltr_dp_plan.hpp)
            out.append(d)
        return out

    @staticmethod
    def exact_pairs(stats):
        """Pairs the exact (redo) kernels scored in the execute the stats belong to."""
        return sum(k["pairs"] for k in stats if k["family"] == "exact")

    def close(self):
        if self._h:
            lib().ltr_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- host-only helpers of the C-ABI (no GPU needed) ------------------------------------------
def trim_alignment(aln_dict, repeat_start, repeat_end, indel_flank_len):
    pa = _abi.PackedAlignments([aln_dict])
    lt, rt = C.c_int32(0), C.c_int32(0)
    rc = lib().ltr_trim_alignment(pa.array, repeat_start, repeat_end, indel_flank_len, C.byref(lt), C.byref(rt))
    return rc, lt.value, rt.value


def haplotype_seqs(blocks):
    ph = _abi.PackedHaplotype(blocks)
    n = lib().ltr_haplotype_num_combs(C.byref(ph.struct))
    cap = int(sum(max(len(a) for a in b["alleles"]) for b in blocks)) + 1
    out = []
    buf = np.zeros(cap, dtype=np.uint8)
    for k in range(n):
        ln = lib().ltr_haplotype_seq(C.byref(ph.struct), k, _p(buf), cap)
        if ln < 0:
            raise LtrError(ln, "ltr_haplotype_seq")
        out.append(buf[:ln].tobytes())
    return out


def pool_reads(reads):
    keep = [np.frombuffer(r, dtype=np.uint8).copy() if len(r) else np.zeros(1, dtype=np.uint8) for r in reads]
    ptrs = (C.c_void_p * max(len(reads), 1))(*[k.ctypes.data for k in keep])
    lens = np.asarray([len(r) for r in reads], dtype=np.int32)
    idx = np.zeros(max(len(reads), 1), dtype=np.int32)
    n = lib().ltr_pool_reads(ptrs, _p(lens), len(reads), _p(idx))
    return n, idx[:len(reads)]


def scatter_pool_probs(pool_probs, pool_seeds, pool_index, n_alleles, realign_to_hap=None, copy_read=None,
                       second_mate=None, log_aln_probs=None, seed_positions=None):
    pool_probs = np.ascontiguousarray(pool_probs, dtype=np.float64)
    pool_seeds = np.ascontiguousarray(pool_seeds, dtype=np.int32)
    pool_index = np.ascontiguousarray(pool_index, dtype=np.int32)
    R = len(pool_index)
    out = np.full(R * n_alleles, np.nan, dtype=np.float64) if log_aln_probs is None else log_aln_probs
    seeds = np.full(R, -1, dtype=np.int32) if seed_positions is None else seed_positions
    u8 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.uint8)
    rh, cr, sm = u8(realign_to_hap), u8(copy_read), u8(second_mate)
    rc = lib().ltr_scatter_pool_probs(_p(pool_probs), _p(pool_seeds), _p(pool_index), R, n_alleles, _p(rh), _p(cr),
                                      _p(sm), _p(out), _p(seeds))
    if rc != 0:
        raise LtrError(rc, "ltr_scatter_pool_probs")
    return out.reshape(R, n_alleles), seeds


def extract_genotypes(log_sample_posteriors, sample_total_ll, best_haplotypes, hap_to_allele, n_variants, haploid=False,
                      want=("gls", "gl_diffs", "pls", "phased_gls")):
    """Genotyper::extract_genotypes_and_likelihoods (genotyper.cpp:132-256) through ltr_extract_genotypes.

    log_sample_posteriors [S, H, H], sample_total_ll [S], best_haplotypes [S, 2] (ltr_posteriors' gts),
    hap_to_allele [H].  Returns a dict of numpy arrays (best_gts, log_*_posteriors, gls, gl_diffs, pls, phased_gls)."""
    post = np.ascontiguousarray(log_sample_posteriors, dtype=np.float64)
    S, H = post.shape[0], post.shape[1]
    stl = np.ascontiguousarray(sample_total_ll, dtype=np.float64)
    bh = np.ascontiguousarray(best_haplotypes, dtype=np.int32)
    h2a = np.ascontiguousarray(hap_to_allele, dtype=np.int32)
    if post.shape != (S, H, H) or stl.shape != (S,) or bh.shape != (S, 2) or h2a.shape != (H,):
        raise LtrError(-1, "extract_genotypes: shape mismatch")
    f, arrs = _abi.genotype_field_buffers(S, n_variants, haploid, want)
    L = lib()
    L.ltr_extract_genotypes.restype = C.c_int
    L.ltr_extract_genotypes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.POINTER(_abi.GenotypeFields)]
    rc = L.ltr_extract_genotypes(S, H, n_variants, _p(h2a), 1 if haploid else 0, _p(post), _p(stl), _p(bh), C.byref(f))
    if rc != 0:
        raise LtrError(rc, "ltr_extract_genotypes")
    return arrs


# ---- the genotyper's last steps (host): allele pruning and the VCF record -----------------------
def haps_to_alleles(blocks, block):
    ph = _abi.PackedHaplotype(blocks)
    out = np.zeros(ph.num_combs, dtype=np.int32)
    L = lib()
    L.ltr_haps_to_alleles.argtypes = [C.POINTER(_abi.HaplotypeBlocks), C.c_int32, C.c_void_p]
    rc = L.ltr_haps_to_alleles(C.byref(ph.struct), block, _p(out))
    if rc != 0:
        raise LtrError(rc, "ltr_haps_to_alleles")
    return out


def unused_alleles(best_haplotypes, hap_to_allele, n_block_alleles, sample_has_aligned_read=None, sample_filtered=None):
    bh = np.ascontiguousarray(best_haplotypes, dtype=np.int32)
    h2a = np.ascontiguousarray(hap_to_allele, dtype=np.int32)
    u8 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.uint8)
    ar, fl = u8(sample_has_aligned_read), u8(sample_filtered)
    out = np.zeros(max(n_block_alleles, 1), dtype=np.int32)
    L = lib()
    L.ltr_unused_alleles.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    n = L.ltr_unused_alleles(len(bh), _p(bh), _p(ar), _p(fl), len(h2a), _p(h2a), n_block_alleles, _p(out))
    if n < 0:
        raise LtrError(n, "ltr_unused_alleles")
    return out[:n].tolist()


def remap_haplotypes(old_blocks, new_blocks):
    po, pn = _abi.PackedHaplotype(old_blocks), _abi.PackedHaplotype(new_blocks)
    mapping = np.zeros(po.num_combs, dtype=np.int32)
    realign = np.zeros(pn.num_combs, dtype=np.uint8)
    L = lib()
    L.ltr_remap_haplotypes.argtypes = [C.POINTER(_abi.HaplotypeBlocks), C.POINTER(_abi.HaplotypeBlocks), C.c_void_p, C.c_void_p]
    rc = L.ltr_remap_haplotypes(C.byref(po.struct), C.byref(pn.struct), _p(mapping), _p(realign))
    if rc != 0:
        raise LtrError(rc, "ltr_remap_haplotypes")
    return mapping, realign


def remap_aln_probs(old_ll, mapping, h_new):
    old = np.ascontiguousarray(old_ll, dtype=np.float64)
    R, Ho = old.shape
    m = np.ascontiguousarray(mapping, dtype=np.int32)
    new = np.zeros((R, h_new), dtype=np.float64)
    L = lib()
    L.ltr_remap_aln_probs.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    rc = L.ltr_remap_aln_probs(_p(old), R, Ho, _p(m), h_new, _p(new))
    if rc != 0:
        raise LtrError(rc, "ltr_remap_aln_probs")
    return new


def get_alleles(packed_vcf_locus):
    L = lib()
    L.ltr_get_alleles.argtypes = [C.POINTER(_abi.VcfLocus), C.POINTER(C.c_int32), C.c_char_p, C.c_int64, C.c_void_p]
    buf = C.create_string_buffer(1 << 20)
    off = np.zeros(1024, dtype=np.int64)
    pos = C.c_int32(0)
    n = L.ltr_get_alleles(C.byref(packed_vcf_locus.struct), C.byref(pos), buf, len(buf), _p(off))
    if n < 0:
        raise LtrError(n, "ltr_get_alleles")
    raw = buf.raw
    return pos.value, [raw[off[i]:off[i + 1]].decode() for i in range(n)]


def vcf_record(packed_vcf_locus, options=None):
    """ltr_vcf_record: (VCF line, 1-based position)."""
    L = lib()
    L.ltr_vcf_record.restype = C.c_int64
    L.ltr_vcf_record.argtypes = [C.POINTER(_abi.VcfLocus), C.POINTER(_abi.VcfOptions), C.c_char_p, C.c_int64, C.POINTER(C.c_int32)]
    buf = C.create_string_buffer(1 << 22)
    pos = C.c_int32(0)
    n = L.ltr_vcf_record(C.byref(packed_vcf_locus.struct), None if options is None else C.byref(options), buf, len(buf), C.byref(pos))
    if n < 0:
        raise LtrError(int(n), "ltr_vcf_record")
    return buf.raw[:n].decode(), pos.value


# ---- raw reads -> prepared reads -> candidate haplotypes (host) -----------------------------------
class ReadSet:
    """ltr_read_set: GenotyperBamProcessor::left_align_reads for one locus.  raw: list of dict(pos, end_pos, bases,
    cigar=[(type, num)...], sample=0, hp=0, quals=None, use_for_hap_generation=True)."""

    def __init__(self, raw, n_samples, region_start, region_stop, chrom_seq, chrom_seq_start=0):
        L = lib()
        n = len(raw)
        arr = (_abi.RawAlignment * max(n, 1))()
        self._keep = []
        for i, r in enumerate(raw):
            b = np.frombuffer(r["bases"], dtype=np.uint8).copy() if len(r["bases"]) else np.zeros(1, dtype=np.uint8)
            q = None if r.get("quals") is None else np.frombuffer(r["quals"], dtype=np.uint8).copy()
            ct = bytes(ord(t) for t, _ in r["cigar"])
            cn = np.asarray([k for _, k in r["cigar"]], dtype=np.int32) if r["cigar"] else np.zeros(1, dtype=np.int32)
            self._keep.append((b, q, ct, cn))
            arr[i].pos, arr[i].end_pos = r["pos"], r["end_pos"]
            arr[i].bases = b.ctypes.data_as(C.POINTER(C.c_uint8))
            arr[i].quals = q.ctypes.data_as(C.POINTER(C.c_uint8)) if q is not None else None
            arr[i].length, arr[i].n_cigar = len(r["bases"]), len(r["cigar"])
            arr[i].cigar_type, arr[i].cigar_num = ct, cn.ctypes.data_as(C.POINTER(C.c_int32))
            arr[i].sample, arr[i].haplotype_tag = r.get("sample", 0), r.get("hp", 0)
            arr[i].reverse, arr[i].use_for_hap_generation = int(r.get("reverse", 0)), int(r.get("use_for_hap_generation", 1))
        self.chrom = np.frombuffer(chrom_seq, dtype=np.uint8).copy()
        self._h = C.c_void_p()
        L.ltr_left_align_reads.argtypes = [C.POINTER(_abi.RawAlignment), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                           C.c_int64, C.POINTER(C.c_void_p)]
        rc = L.ltr_left_align_reads(arr, n, n_samples, region_start, region_stop, _p(self.chrom), chrom_seq_start, len(self.chrom), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise LtrError(rc, "ltr_left_align_reads")
        self.n_samples = n_samples
        L.ltr_read_set_size.argtypes = [C.c_void_p]
        self.size = L.ltr_read_set_size(self._h)
        for f, rt in (("alignments", C.POINTER(_abi.Alignment)), ("alignment_strings", C.POINTER(C.c_char_p)), ("deleted", C.POINTER(C.c_uint8)),
                      ("source", C.POINTER(C.c_int32)), ("sample", C.POINTER(C.c_int32)), ("n_p1s", C.POINTER(C.c_int32)), ("n_p2s", C.POINTER(C.c_int32))):
            fn = getattr(L, "ltr_read_set_" + f)
            fn.argtypes, fn.restype = [C.c_void_p], rt
        L.ltr_read_set_fail_count.argtypes = [C.c_void_p]
        self.fail_count = L.ltr_read_set_fail_count(self._h)
        self.alignments_ptr = L.ltr_read_set_alignments(self._h)
        av, st, de, so, sa = (L.ltr_read_set_alignments(self._h), L.ltr_read_set_alignment_strings(self._h), L.ltr_read_set_deleted(self._h),
                              L.ltr_read_set_source(self._h), L.ltr_read_set_sample(self._h))
        self.reads = []
        for i in range(self.size):
            a = av[i]
            self.reads.append(dict(start=a.start, stop=a.stop, seq=bytes(a.seq[:a.seq_len]),
                                   cigar=[(chr(a.cigar_type[k]), a.cigar_num[k]) for k in range(a.n_cigar)],
                                   aln=st[i], deleted=bool(de[i]), source=so[i], sample=sa[i],
                                   qual=None if not a.qual else bytes(a.qual[:a.seq_len])))
        p1, p2 = L.ltr_read_set_n_p1s(self._h), L.ltr_read_set_n_p2s(self._h)
        self.n_p1s, self.n_p2s = [p1[s] for s in range(n_samples)], [p2[s] for s in range(n_samples)]

    def extract_sequence(self, i, region_start, region_end):
        L = lib()
        L.ltr_extract_sequence.restype = C.c_int64
        L.ltr_extract_sequence.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]
        buf = np.zeros(1 << 16, dtype=np.uint8)
        n = L.ltr_extract_sequence(self._h, i, region_start, region_end, _p(buf), len(buf))
        if n < -1:
            raise LtrError(int(n), "ltr_extract_sequence")
        return None if n < 0 else buf[:n].tobytes()

    def build_haplotype(self, region_start, region_stop, period, chrom_seq_start, chrom_len, indel_flank_len=5, ctx=None):
        """ltr_build_haplotype -> dict(blocks=[...] or None, failure, unplaced_reads, samples_needing_clustering)."""
        L = lib()
        L.ltr_build_haplotype.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64,
                                          C.c_int64, C.c_int32, C.POINTER(C.c_void_p)]
        h = C.c_void_p()
        rc = L.ltr_build_haplotype(None if ctx is None else ctx._h, self._h, self.n_samples, region_start, region_stop, period, _p(self.chrom),
                                   chrom_seq_start, len(self.chrom), chrom_len, indel_flank_len, C.byref(h))
        if rc != 0:
            raise LtrError(rc, "ltr_build_haplotype")
        L.ltr_hap_result_blocks.argtypes, L.ltr_hap_result_blocks.restype = [C.c_void_p], C.POINTER(_abi.HaplotypeBlocks)
        L.ltr_hap_result_failure.argtypes, L.ltr_hap_result_failure.restype = [C.c_void_p], C.c_char_p
        L.ltr_hap_result_unplaced_reads.argtypes = [C.c_void_p]
        L.ltr_hap_result_samples_needing_clustering.argtypes = [C.c_void_p]
        L.ltr_hap_result_free.argtypes, L.ltr_hap_result_free.restype = [C.c_void_p], None
        out = dict(failure=L.ltr_hap_result_failure(h).decode(), unplaced_reads=L.ltr_hap_result_unplaced_reads(h),
                   samples_needing_clustering=L.ltr_hap_result_samples_needing_clustering(h), blocks=None)
        bp = L.ltr_hap_result_blocks(h)
        if bp:
            b = bp.contents
            blocks, k = [], 0
            for i in range(b.n_blocks):
                al = []
                for _ in range(b.n_alleles[i]):
                    al.append(bytes(b.allele_bytes[b.allele_off[k]:b.allele_off[k + 1]]))
                    k += 1
                blocks.append(dict(start=b.block_start[i], end=b.block_end[i], is_repeat=bool(b.is_repeat[i]), period=b.period[i], alleles=al))
            out["blocks"] = blocks
        L.ltr_hap_result_free(h)
        return out

    def close(self):
        if self._h:
            lib().ltr_read_set_free.argtypes = [C.c_void_p]
            lib().ltr_read_set_free.restype = None
            lib().ltr_read_set_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def phasing_priors(sample_of_read, haplotype, n_samples):
    """ltr_phasing_priors (SNPBamProcessor::process_phased_reads for unpaired reads): (log_p1, log_p2, phased reads)."""
    so = np.ascontiguousarray(sample_of_read, dtype=np.int32)
    hp = np.ascontiguousarray(haplotype, dtype=np.int32)
    p1, p2 = np.zeros(len(so)), np.zeros(len(so))
    n = C.c_int32(0)
    L = lib()
    L.ltr_phasing_priors.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    rc = L.ltr_phasing_priors(len(so), _p(so), _p(hp), int(n_samples), _p(p1), _p(p2), C.byref(n))
    if rc != 0:
        raise LtrError(rc, "ltr_phasing_priors")
    return p1, p2, n.value

// ltr_dp_plan.hpp -- the PLAN KERNEL: every one-wave class and every packed strip width of a plan in ONE persistent launch
// (included after ltr_dp_kernel.hpp and ltr_dp_pack.hpp by ltr_k_plan.hip).
//
// Replaces HapAligner::align_seq_to_hap (reference src/SeqAlignment/HapAligner.cpp:236-343) for whole small and mid-size plans:
// a GPU's share of BASELINE config 4 at N = 8 (1250 loci, ~30 ms of DP), a chunk of ltr_calc_hap_aln_probs.  Same bodies, same
// bits as the single-class kernels (ltr_dp_kernel<W, false, SYM, true>, ltr_dp_pack_kernel<W, SYM>).
//
// Why.  A launch of the persistent kernels ends in a tail as long as its last pairs (0.5 - 1 ms for reads of 700 - 1300 bases)
// during which its wave slots empty one by one, and a 4-wave workgroup gives its slots back only when all four waves are
// through: measured on MI355X, pass time of a single class = 0.30 ms + its rounds of resident wavefronts, and a 1250-locus plan
// of ~8 launches ran at 0.92 - 0.94 of the 10 000-locus rate -- the multi-width launches (one-wave widths 11 .. 20, packed
// widths 13 .. 20) 24.3 + 6.4 ms for 22.5 + 5.6 ms of vector instructions.  Here the plan is ONE launch: the entries (a
// one-wave class, or one strip width of the packed family with all its lanes-per-pair ranges) sit in a device table, longest
// pairs first; a wavefront starts at the entry its number falls into (PlanEntry::first_wave: the entries share the launch's
// wavefronts in proportion to their modelled work), scores pairs of it until its work counter is drained, then walks the table
// from the top -- so the launch's one tail is made of the SHORTEST pairs of the plan (packed groups of a few dozen microseconds).  A drained entry
// costs one load of its counter, not a call.  A pair whose certificate fails is scored at once by the wavefront that found out
// (ltr_dp_redo.hpp): no exact launches behind the plan for these classes.
//
// A real call per entry (class_walk_call's reasons: every strip width keeps its own register allocation; the kernel arguments
// are read from the kernel's own argument segment; the LDS tables travel as LDS addresses).

// Pairs whose certificate failed are NOTED in a short per-wave list (LDS) and the walk returns to the kernel, which calls the
// exact body for each (redo_dispatch) and comes back if the entry is not drained yet.  The walks themselves stay leaf functions:
// with the calls inside them -- even behind the pair loop -- every wavefront step of the one-wave bodies carried five more
// instructions and three more v_readlane than class_walk_call's (ISA, W = 15: a non-leaf function gives up the SGPRs of its
// return address and frame, and the step loop is 106 SGPRs wide).

template <int W, bool SYM>
__device__ __attribute__((noinline)) int plan_class_call(int64_t kernarg_v, int first_pair_v, int n_pairs_v, int cls_v, unsigned emit_lds_v, unsigned note_lds_v) {
  const KernelArgs& A = *(const KernelArgs*)(KernArgPtr)(uintptr_t)uni64(kernarg_v);
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const double* emit_tab = (const double*)(LdsDoubles)(uintptr_t)(unsigned)uni((int)emit_lds_v);
  int* note = (int*)(LdsInts)(uintptr_t)(unsigned)uni((int)note_lds_v);   // this wave's list
  double* scr = A.scratch + ((size_t)blockIdx.x * kBlockWaves + wave) * 6 * A.scratch_stride;
  const int first_pair = uni(first_pair_v), n_pairs = uni(n_pairs_v);
  uint32_t* queue = A.queue_base + uni(cls_v);
  const double IMP = kImp;
  for (;;) {
    const int q = pop_one(queue, lane);
    if (q >= n_pairs) return kWalkDrained;
    const int pi = first_pair + q;
    const PairDesc* pp = A.pairs + pi;
    const int n = uni(pp->n), m = uni(pp->m), hfl = uni(pp->hap_full_len);
    const int64_t out_idx = uni64(pp->out_idx);
    double r;
    int status = kStatusOk;
    if (hfl <= 60) r = IMP;                                    // HapAligner.cpp:241-244
    else if (abs(n - m) > 600) r = -700.0;                     // :249-252
    else {
      PairCtx P;
      P.hap = A.hap_bytes + uni64(pp->hap_off);
      P.hapc = A.hap_codes + uni64(pp->hap_off);
      P.read = A.read_bytes + uni64(pp->read_off);
      P.n = n; P.m = m; P.dd = n - m;
      const int h0 = uni((int)P.hap[0]), r0 = uni((int)P.read[0]);
      P.emit00 = (h0 == r0) ? (double)A.mc.match : (double)A.mc.mismatch;   // match_matrix[0], :265
      if (m == 1) {
        r = (n == 1) ? dmax(IMP, dmax(IMP, P.emit00)) : -700.0;              // no interior column (see ltr_dp_kernel)
      } else {
        P.e01 = (h0 == uni((int)P.read[1])) ? 1 : 0;           // emission of the whole first column, :276
        r = align_pair<W, kModeCert, SYM, true>(A, P, scr, lane, &status, emit_tab, nullptr);
      }
    }
    if (status == kStatusUncertain) {                          // could not prove "no row aborts": the exact body, from the kernel
      // (at once, not when the entry is drained: measured on MI355X with the wave clocks of a 1250-locus plan -- 99 % of the
      // wavefronts had left after 31.1 ms, the last one after 33.8: exact bodies of 1 - 2 ms each started behind the last pairs)
      if (lane == 0) note[0] = pi;
      return 1;
    } else if (lane == 0) A.out_ll[out_idx] = r;
  }
}

template <int W, bool SYM>
__device__ __attribute__((noinline)) int plan_pack_call(int64_t kernarg_v, int tab_v, unsigned emit_lds_v, unsigned note_lds_v) {
  const KernelArgs& A = *(const KernelArgs*)(KernArgPtr)(uintptr_t)uni64(kernarg_v);
  const int lane = threadIdx.x & 63;
  const double* emit_tab = (const double*)(LdsDoubles)(uintptr_t)(unsigned)uni((int)emit_lds_v);
  int* note = (int*)(LdsInts)(uintptr_t)(unsigned)uni((int)note_lds_v);
  const PackTable* __restrict__ T = A.pk_tabs + uni(tab_v);
  PackRanges R;
#pragma unroll
  for (int r = 0; r < 5; ++r) { R.shift[r] = uni(T->shift[r]); R.first[r] = uni(T->first[r]); R.end[r] = uni(T->end[r]); R.grp_end[r] = uni(T->grp_end[r]); }
  uint32_t* queue = A.queue_base + uni(T->queue_class);
  return pack_walk<W, SYM, true>(A, R, queue, emit_tab, lane, note);
}

template <bool SYM, int W = kWMax>
struct PlanCalls {
  static __device__ __forceinline__ int one(int w, int64_t ka, int first, int np, int cls, unsigned el, unsigned nl) {
    if (w == W) return plan_class_call<W, SYM>(ka, first, np, cls, el, nl);
    return PlanCalls<SYM, W - 1>::one(w, ka, first, np, cls, el, nl);
  }
  static __device__ __forceinline__ int pack(int w, int64_t ka, int tab, unsigned el, unsigned nl) {
    if (w == W) return plan_pack_call<W, SYM>(ka, tab, el, nl);
    return PlanCalls<SYM, W - 1>::pack(w, ka, tab, el, nl);
  }
  // (chained walks and their plain pairs: the strip widths kMultiMinW .. kWMax, whose reads fill a strip of the wave's scratch)
  static __device__ __forceinline__ int chain(int w, int64_t ka, int first, int np, int cls, unsigned el, unsigned nl, unsigned rl) {
    if (W < kMultiMinW || !SYM) return kWalkDrained;             // (the chained walk is an experiment of the symmetric instance: the host never lists kind 3 otherwise)
    if (w == W) return plan_chain_call<(W < kMultiMinW ? kMultiMinW : W), true>(ka, first, np, cls, el, nl, rl);
    return PlanCalls<SYM, W - 1>::chain(w, ka, first, np, cls, el, nl, rl);
  }
  static __device__ __forceinline__ int plain(int w, int64_t ka, int pi, unsigned el) {
    if (W < kMultiMinW || !SYM) return 0;
    if (w == W) return plan_plain_pair_call<(W < kMultiMinW ? kMultiMinW : W), true>(ka, pi, el);
    return PlanCalls<SYM, W - 1>::plain(w, ka, pi, el);
  }
};
template <bool SYM>
struct PlanCalls<SYM, 0> {
  static __device__ __forceinline__ int one(int, int64_t, int, int, int, unsigned, unsigned) { return kWalkDrained; }
  static __device__ __forceinline__ int pack(int, int64_t, int, unsigned, unsigned) { return kWalkDrained; }
  static __device__ __forceinline__ int chain(int, int64_t, int, int, int, unsigned, unsigned, unsigned) { return kWalkDrained; }
  static __device__ __forceinline__ int plain(int, int64_t, int, unsigned) { return 0; }
};
static_assert(kWMax == kPackWMax, "PlanCalls walks both families' strip widths with one recursion");

template <bool SYM>
__global__ __launch_bounds__(64 * kBlockWaves, 3) void ltr_dp_plan_kernel(KernelArgs A) {
  __shared__ __attribute__((aligned(16))) double s_emit[kEmitTabDoubles];
  __shared__ double s_pen[kPenTabDoubles];                     // the exact thresholds of the in-line redo (kModeThr), entry k + kPenHalf
  __shared__ __attribute__((aligned(16))) double s_ring[kBlockWaves][kChainRingDoubles];   // the chained walk's one-row ring (ltr_dp_chain.hpp)
  __shared__ int s_note[kBlockWaves][kRedoNote + 1];           // per wave: pairs waiting for the exact body; [kRedoNote]: a packed group popped but not scored yet
  for (int idx = threadIdx.x; idx < kEmitTabDoubles; idx += 64 * kBlockWaves) {
    const int half = idx >> 11, hcode = (idx >> 9) & 3, quad = (idx >> 1) & 255, k = 2 * half + (idx & 1);
    s_emit[idx] = (hcode == ((quad >> (2 * k)) & 3)) ? (double)A.mc.match : (double)A.mc.mismatch;
  }
  if (A.thr_ok) for (int idx = threadIdx.x; idx < kPenTabDoubles; idx += 64 * kBlockWaves) s_pen[idx] = A.thr_tab[idx];
  if ((threadIdx.x & 63) == 0) s_note[threadIdx.x >> 6][kRedoNote] = -1;
  __syncthreads();
  const unsigned emit_lds = (unsigned)(uintptr_t)(LdsDoubles)s_emit;
  const unsigned pen_lds = (unsigned)(uintptr_t)(LdsDoubles)s_pen;
  const unsigned note_lds = (unsigned)(uintptr_t)(LdsInts)s_note[uni((int)(threadIdx.x >> 6))];
  const unsigned ring_lds = (unsigned)(uintptr_t)(LdsDoubles)s_ring[uni((int)(threadIdx.x >> 6))];
  const int64_t kargs = (int64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int n_e = A.pl_n;
  const unsigned long long t_first = A.wave_clock ? wall_clock64() : 0ull;
  unsigned long long redo_ticks = 0, redo_pairs = 0;
  const int my_wave = (int)blockIdx.x * kBlockWaves + uni((int)(threadIdx.x >> 6));
  // the entry this wavefront starts at: the last one whose first_wave is <= its number
  int start = 0;
  for (int i = 1; i < n_e; ++i) if (uni(A.pl_entries[i].first_wave) <= my_wave) start = i;
  // its own entry first, then the table from the top (longest pairs first)
  for (int j = -1; j < n_e; ++j) {
    const int i = j < 0 ? start : j;
    if (j == start) continue;
    const PlanEntry* __restrict__ E = A.pl_entries + i;
    const int cls = uni(E->queue_class), limit = uni(E->limit);
    // (a drained entry: one load of its counter instead of a call that saves and restores the callee's registers)
    const uint32_t seen = (uint32_t)uni((int)__hip_atomic_load(A.queue_base + cls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (seen >= (uint32_t)limit) continue;
    const int w = uni(E->W), kind = uni(E->kind);
    // (measurement aid: the wall-clock ticks this wavefront spends in entry i, summed per entry behind the per-wave words and the log)
    const unsigned long long t_entry = A.wave_clock ? wall_clock64() : 0ull;
    auto leave_entry = [&]() __attribute__((always_inline)) {
      if (A.wave_clock && (threadIdx.x & 63) == 0)
        atomicAdd(A.wave_clock + 4ull * gridDim.x * kBlockWaves + 4096 + (unsigned)i, wall_clock64() - t_entry);
    };
    if (kind == 2) {
      // pairs that start out with the exact body: bytes outside ACGT (w == 0) or a length difference no certificate can hold
      const int first = uni(E->first), np = uni(E->n_pairs);
      for (;;) {
        const int q = pop_one(A.queue_base + cls, (int)(threadIdx.x & 63));
        if (q >= np) break;
        const int pi = first + q;
        if (w == 0) {
          if ((threadIdx.x & 63) == 0) atomicAdd(A.xcount + kInlineCountOff + kXGeneric, 1u);
          redo_generic_call<SYM>(kargs, pi);
        } else redo_dispatch<SYM>(A, kargs, pi, uni(A.pairs[pi].m) - 1, emit_lds, pen_lds);
      }
      leave_entry();
      continue;
    }
    for (;;) {
      int ret;
      if (kind == 0) ret = PlanCalls<SYM>::one(w, kargs, uni(E->first), uni(E->n_pairs), cls, emit_lds, note_lds);
      else if (kind == 3) ret = PlanCalls<SYM>::chain(w, kargs, uni(E->first), uni(E->n_pairs), cls, emit_lds, note_lds, ring_lds);
      else ret = PlanCalls<SYM>::pack(w, kargs, uni(E->tab), emit_lds, note_lds);
      ret = uni(ret);
      const int noted = ret & (kWalkDrained - 1);
      const int* note = s_note[uni((int)(threadIdx.x >> 6))];
      const unsigned long long t_r = (A.wave_clock && noted) ? wall_clock64() : 0ull;
      for (int k = 0; k < noted; ++k) {
        const int v = uni(note[k]);
        if (A.wave_clock && (threadIdx.x & 63) == 0) {           // (measurement aid: which pairs these are -- (n << 32) | m behind the per-wave words)
          unsigned long long* log = A.wave_clock + 4ull * gridDim.x * kBlockWaves;
          const unsigned long long at = atomicAdd(log, 1ull);
          if (at < 4095ull) log[1 + at] = ((unsigned long long)(unsigned)A.pairs[v & (kNotePlain - 1)].n << 32) | (unsigned)A.pairs[v & (kNotePlain - 1)].m;
        }                              // pair index; bit 31: straight to the generic body (a pair the packed geometry cannot take)
        if (v < 0) redo_generic_call<SYM>(kargs, v & 0x7fffffff);
        else if (v & kNotePlain) {                               // a pair the chained walk could not take: the plain body of this strip width
          const int pi = v & (kNotePlain - 1);
          if (uni(PlanCalls<SYM>::plain(w, kargs, pi, emit_lds))) redo_dispatch<SYM>(A, kargs, pi, uni(A.pairs[pi].m) - 1, emit_lds, pen_lds);
        } else redo_dispatch<SYM>(A, kargs, v, uni(A.pairs[v].m) - 1, emit_lds, pen_lds);
      }
      if (A.wave_clock && noted) { redo_ticks += wall_clock64() - t_r; redo_pairs += (unsigned long long)noted; }
      if (ret & kWalkDrained) break;
    }
    leave_entry();
  }
  if (A.wave_clock && (threadIdx.x & 63) == 0) {
    A.wave_clock[4 * my_wave] = t_first; A.wave_clock[4 * my_wave + 1] = wall_clock64();
    A.wave_clock[4 * my_wave + 2] = redo_pairs; A.wave_clock[4 * my_wave + 3] = redo_ticks;
  }
}

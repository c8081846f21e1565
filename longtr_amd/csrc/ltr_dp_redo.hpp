// ltr_dp_redo.hpp -- a pair whose certificate failed, scored by the wavefront that found out, inside the same launch (included
// after ltr_dp_kernel.hpp by the translation units of the plan kernel).
//
// The certificate kernels prove "no row's band-penalised maximum is below -600" (HapAligner.cpp:283, :297-306) from one cell
// per lane and row; the pair they cannot prove it for (0.07 % of BASELINE config 3) used to be appended to a device-side list
// and scored by an exact kernel launched behind the certificate launches -- a launch that starts when the last class that can
// feed its list is through and lasts as long as its longest pair: on a GPU's share of config 4 at N = 8 (1250 loci, 33 ms) the
// plan ended in 1 - 4 ms of such launches.  Inside the plan kernel (ltr_dp_plan.hpp) the wavefront calls the exact body itself:
//   * redo_thr_call<W>: every cell against the exact threshold table (kModeThr, ltr_dp_kernel.hpp) on strips of W = 4, 8, 12,
//     16 or 20 columns -- even widths: the threshold reads stride W + 1 doubles from lane to lane and odd widths conflict in LDS;
//   * redo_generic_call: the reference's running row maximum with byte-compare emissions (kModeMax, W = kExactW, column blocks
//     through the wave's scratch strips) -- any model, any length; taken when the penalty table does not fit LDS (xlut == 0)
//     and for the rare row that only a last lane with slack columns certifies (kStatusUncertain out of kModeThr).
// Each body is a real call (its own register allocation, see class_walk_call); a call per failed pair is rare by construction.

constexpr int kRedoNote = 64;                                  // pairs a walk of the plan kernel notes per wave before it returns (>= two packed groups of 32 pairs)
constexpr int kWalkDrained = 1 << 16;                          // a walk's return value: pairs noted | kWalkDrained when the entry's counter ran out
typedef __attribute__((address_space(3))) int* LdsInts;

template <bool SYM>
__device__ __attribute__((noinline)) void redo_generic_call(int64_t kernarg_v, int pi_v) {
  const KernelArgs& A = *(const KernelArgs*)(KernArgPtr)(uintptr_t)uni64(kernarg_v);
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  double* scr = A.scratch + ((size_t)blockIdx.x * kBlockWaves + wave) * 6 * A.scratch_stride;
  const PairDesc* pp = A.pairs + uni(pi_v);
  const int n = uni(pp->n), m = uni(pp->m), hfl = uni(pp->hap_full_len);
  const int64_t out_idx = uni64(pp->out_idx);
  const double IMP = kImp;
  double r;
  if (hfl <= 60) r = IMP;                                      // HapAligner.cpp:241-244
  else if (abs(n - m) > 600) r = -700.0;                       // :249-252
  else {
    PairCtx P;
    P.hap = A.hap_bytes + uni64(pp->hap_off);
    P.hapc = A.hap_codes + uni64(pp->hap_off);
    P.read = A.read_bytes + uni64(pp->read_off);
    P.n = n; P.m = m; P.dd = n - m;
    const int h0 = uni((int)P.hap[0]), r0 = uni((int)P.read[0]);
    P.emit00 = (h0 == r0) ? (double)A.mc.match : (double)A.mc.mismatch;
    if (m == 1) r = (n == 1) ? dmax(IMP, dmax(IMP, P.emit00)) : -700.0;      // no interior column (see ltr_dp_kernel)
    else {
      P.e01 = (h0 == uni((int)P.read[1])) ? 1 : 0;
      int status = kStatusOk;
      r = align_pair<kExactW, kModeMax, SYM, false>(A, P, scr, lane, &status, nullptr, nullptr);
      if (status == kStatusAbort) r = -700.0;
    }
  }
  if (lane == 0) A.out_ll[out_idx] = r;
}

// A pure-ACGT pair with 2 <= m, n and no constant score (what the certificate bodies are given).  Returns 1 when the rare
// ambiguity is left for redo_generic_call.
template <int W, bool SYM>
__device__ __attribute__((noinline)) int redo_thr_call(int64_t kernarg_v, int pi_v, unsigned emit_lds_v, unsigned pen_lds_v) {
  const KernelArgs& A = *(const KernelArgs*)(KernArgPtr)(uintptr_t)uni64(kernarg_v);
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const double* emit_tab = (const double*)(LdsDoubles)(uintptr_t)(unsigned)uni((int)emit_lds_v);
  const double* pen_tab = (const double*)(LdsDoubles)(uintptr_t)(unsigned)uni((int)pen_lds_v);
  double* scr = A.scratch + ((size_t)blockIdx.x * kBlockWaves + wave) * 6 * A.scratch_stride;
  const PairDesc* pp = A.pairs + uni(pi_v);
  PairCtx P;
  P.hap = A.hap_bytes + uni64(pp->hap_off);
  P.hapc = A.hap_codes + uni64(pp->hap_off);
  P.read = A.read_bytes + uni64(pp->read_off);
  P.n = uni(pp->n); P.m = uni(pp->m); P.dd = P.n - P.m;
  const int64_t out_idx = uni64(pp->out_idx);
  const int h0 = uni((int)P.hap[0]), r0 = uni((int)P.read[0]);
  P.emit00 = (h0 == r0) ? (double)A.mc.match : (double)A.mc.mismatch;
  P.e01 = (h0 == uni((int)P.read[1])) ? 1 : 0;
  int status = kStatusOk;
  double r = align_pair<W, kModeThr, SYM, true>(A, P, scr, lane, &status, emit_tab, pen_tab);
  if (status == kStatusUncertain) return 1;
  if (status == kStatusAbort) r = -700.0;
  if (lane == 0) A.out_ll[out_idx] = r;
  return 0;
}

// C = the read's columns (m - 1), wave-uniform.
template <bool SYM>
__device__ __forceinline__ void redo_dispatch(const KernelArgs& A, int64_t kernarg_v, int pi, int C, unsigned emit_lds, unsigned pen_lds) {
  int again = 1;
  {
    // (statistics: how many pairs took the exact body, by the list a single-class kernel would have sent them to)
    const int cls = !A.thr_ok ? kXGeneric : ((C <= 64 * kXShortW) ? kXShort : ((C <= 64 * kXMidW) ? kXMid : ((C <= 64 * kXLongW) ? kXLong
                    : ((C <= kXWg4MaxC) ? kXWg4 : ((C <= kXWg8MaxC) ? kXWg8 : kXLong)))));
    if ((threadIdx.x & 63) == 0) atomicAdd(A.xcount + kInlineCountOff + cls, 1u);
  }
  if (A.thr_ok && C >= 1) {                                     // (either model; a one-base read has no interior column: the generic body knows that case)
    if (C <= 64 * 4) again = redo_thr_call<4, SYM>(kernarg_v, pi, emit_lds, pen_lds);
    else if (C <= 64 * 8) again = redo_thr_call<8, SYM>(kernarg_v, pi, emit_lds, pen_lds);
    else if (C <= 64 * 12) again = redo_thr_call<12, SYM>(kernarg_v, pi, emit_lds, pen_lds);
    else if (C <= 64 * 16) again = redo_thr_call<16, SYM>(kernarg_v, pi, emit_lds, pen_lds);
    else again = redo_thr_call<20, SYM>(kernarg_v, pi, emit_lds, pen_lds);      // (any length: column blocks through the scratch strips)
  }
  if (uni(again)) redo_generic_call<SYM>(kernarg_v, pi);
}

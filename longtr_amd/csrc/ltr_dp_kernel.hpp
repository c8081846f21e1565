// ltr_dp_kernel.hpp -- device side of the read-vs-haplotype DP (included by ltr_gpu.hip only).
//
// Replaces HapAligner::align_seq_to_hap (reference src/SeqAlignment/HapAligner.cpp:236-343).
//
// Geometry.  One 64-lane wavefront scores one (read, haplotype) pair.  The m-1 interior read
// columns are cut into `ncb` column blocks; inside a block lane l owns W consecutive columns
// (its "strip"), and haplotype rows stream through the lanes skewed by one row per lane:
// at step t lane l works on row t-l+1.  Cell (i,j) needs (i-1,j-1) and (i-1,j) -- the lane's
// own registers from the previous step -- and (i,j-1): the previous slot, or for slot 0 the left
// neighbour's last slot, handed over by DPP wave_shr:1 (no LDS).  A block's right boundary
// column is parked in a per-wave scratch strip (3 doubles per row, L2-resident) and is the
// next block's left boundary, so the n*m matrices of the reference never exist.
//   W = ceil(C / (64*ncb)), ncb = ceil(C / (64*WMAX));  every block but the last is Lb lanes
//   x W columns (Lb = ceil(C / (W*ncb)): balanced); the last block takes the remainder and its
//   last lane may own only Wl <= W real columns (the slack sits in ONE lane, at the very end,
//   where nothing is handed on: its dummy columns compute garbage nobody reads).
//
// Recurrence (bit-exact).  Per cell the three matrices are carried as the three max-terms the
// NEXT cells consume:  X = max(M+e, D+d, I+b) (diagonal), Y = max(M+f, I+a) (below),
// Z = max(M+g, D+c) (right); M = emit + X(i-1,j-1), I = MATCH + Y(i-1,j), D = Z(i,j-1).
// 13 FP64 add/max per cell (11 when b == d and f == g, template SYM), IEEE double, float-typed model constants promoted exactly where
// the reference promotes them; the translation unit is compiled with -ffp-contract=off.
//
// Row abort (:283,:297-306: a row whose band-penalised maximum is < -600 ends the pair with
// -700).  EXACT = true evaluates that maximum cell by cell like the reference (+4 FP64 ops and
// the int->float->double penalty per cell).  EXACT = false replaces it by a CERTIFICATE: per
// lane and row ONE cell is tested, fl(M(i,j) + pen(i,j)) >= -600; since best(i,j) >= M(i,j)
// and fl is monotone, one passing cell proves the row's maximum is >= -600.  If every row has
// a passing cell in some lane the pair cannot abort and its score is final; otherwise the
// pair is queued for the EXACT kernel.  Scores never depend on which kernel produced them.

#include <type_traits>

#include "ltr_dp_types.h"

__device__ __forceinline__ double dmax(double x, double y) { return fmax(x, y); }

// lane l <- lane l-1 (lane 0 keeps `fill`): DPP moves, no LDS.
__device__ __forceinline__ int wave_shr1_i(int v, int fill) {
  return __builtin_amdgcn_update_dpp(fill, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}
__device__ __forceinline__ int wave_shr1_zero(int v) {       // lane 0 <- 0 (bound_ctrl), no fill register
  return __builtin_amdgcn_update_dpp(0, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ double wave_shr1_nofill(double v) {   // lane 0 <- 0.0: for callers that overwrite lane 0 anyway (the packed body's segment heads)
  return __hiloint2double(wave_shr1_zero(__double2hiint(v)), wave_shr1_zero(__double2loint(v)));
}
__device__ __forceinline__ double wave_shr1(double v, double fill) {
  const int lo = wave_shr1_i(__double2loint(v), __double2loint(fill));
  const int hi = wave_shr1_i(__double2hiint(v), __double2hiint(fill));
  return __hiloint2double(hi, lo);
}

// Boundary strips are written by plain (write-through) vector stores and read back by the
// same wave one column block later: agent-scope loads (sc1) are served from L2, never from a
// stale L1 line or the scalar cache.
__device__ __forceinline__ double strip_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// Everything that describes the PAIR is wave-uniform; readfirstlane tells the compiler so
// (SGPRs, scalar branches, no waterfall loops).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// One element off a work queue for the whole wavefront.  While the queue is the launch's own (`A.queue`, a kernel
// argument) hipcc's atomic optimizer turns `atomicAdd(A.queue, lane == 0 ? 1 : 0)` into one lane's add by itself; once the
// queue is a loop-carried pointer (the classes a multi-width launch walks, KernelArgs::mk_class) it does not, and the
// wave-wide form costs 64 same-address atomics per pop -- 0.7 microseconds, measured on MI355X: every one-wave launch of
// config 3 took 2.6 x its time.  So: lane 0 adds, the others hold an opaque 0 (the empty asm keeps hipcc from
// jump-threading the phi: it once gave the lanes != 0 a copy of the loop body with q == 0, which re-ran pair 0 forever).
// The queue's address is laundered through vector registers first: an address hipcc can prove wave-uniform makes its
// atomic optimizer rewrite the add INSIDE the `if (lane == 0)` as well, and that form never left the loop on MI355X
// (ROCm 7.2) -- seen twice: in ltr_dp_kernel with `A.queue`, in ltr_dp_multi_kernel with `A.queue_base + class`.
// Round 5 tried what the advisor proposed -- the optimizer off for these translation units (-mllvm
// -amdgpu-atomic-optimizer-strategy=None) and ONE explicit form everywhere, `if (lane == 0) q = atomic..; readfirstlane(q)` without
// the laundering.  The assembly looked right (s_and_saveexec / one global_atomic_add sc0 / s_or exec / v_readfirstlane) and the
// certificate kernels and the plan kernel ran, but the exact kernels (the pair loop with `continue`s, ltr_dp_kernel<W, true, ..>)
// and the NW kernels never came back on MI355X: every launch of them ran into its time-out.  Reverted; the forms below are the
// ones that have run 10^7 launches, and tests/test_isa_budget.py pins what hipcc emits for them (one lane's atomic under a saved
// exec mask, no loop over lanes around it) so that a toolchain that changes it turns a CPU test red instead of hanging a GPU.
__device__ __forceinline__ int pop_one(uint32_t* queue, const int lane) {
  uint32_t lo = (uint32_t)(uintptr_t)queue, hi = (uint32_t)((uintptr_t)queue >> 32);
  asm volatile("" : "+v"(lo), "+v"(hi));
  queue = (uint32_t*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
  int qv = 0;
  if (lane == 0) qv = (int)__hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" : "+v"(qv));
  return __builtin_amdgcn_readfirstlane(qv);
}

// A pair the certificate could not clear: append it to the list of the exact kernel that fits it.
// Every lane issues the add (lane 0 adds 1, the rest 0): see the queue pop in ltr_dp_kernel.
__device__ __forceinline__ void push_redo(const KernelArgs& A, const int lane, const int pi, const int m) {
  int cls = kXGeneric;
  if (A.xlut) {
    const int C = m - 1;
    cls = (C <= 64 * kXShortW) ? kXShort : ((C <= 64 * kXMidW) ? kXMid : ((C <= 64 * kXLongW) ? kXLong
          : ((C <= kXWg4MaxC) ? kXWg4 : ((C <= kXWg8MaxC) ? kXWg8 : kXLong))));
  }
  const int slot = (int)atomicAdd(A.xcount + cls, lane == 0 ? 1u : 0u);
  if (lane == 0) A.xlist[cls][__builtin_amdgcn_readfirstlane(slot)] = pi;
}

struct PairCtx {                 // wave-uniform description of the pair being scored
  const uint8_t* hap;            // haplotype window
  const uint16_t* hapc;          // ... as emission-table block offsets (LUT kernels)
  const uint8_t* read;
  int n, m, dd;
  int e01;
  double emit00;
  // column-block geometry
  int ncb, Lb, Ll, Wl;
  int k600;                      // band offsets |k| >= k600 carry a penalty below -600 whatever the cell holds
};

enum { kStatusOk = 0, kStatusAbort = 1, kStatusUncertain = 2 };


template <bool V> struct BoolTag { static constexpr bool value = V; };

// One column block of one pair.  FIRST: the block starts at column 1, so its left boundary is
// the reference's first column (HapAligner.cpp:274-280), read from the model tables; otherwise
// it is the strip the previous block parked.
// LUT: the emission MATCH/MISMATCH comes from a 16-entry table in LDS indexed by (2-bit code of
// the haplotype base, 2-bit code of the read base): one integer add + one ds_read_b64 per cell
// instead of compare + two selects (the kernel is VALU-issue-bound, the LDS pipe is idle).
// Only for pairs whose bytes are all in {A,C,G,T}; anything else takes the byte-compare path.
// How a kernel decides "no row's band-penalised maximum is below -600" (HapAligner.cpp:297-306):
//   kModeCert  one cell per lane and row against a conservative threshold; a row nobody certifies sends the pair to an exact list
//   kModeMax   the reference's running row maximum of fl(best + pen), cell by cell (penalties formed per cell)
//   kModeThr   every cell against the exact threshold table (ltrp::build_threshold_table): exact like kModeMax, 13 FP64
//              operations per cell instead of 14 and no maximum handed from lane to lane; needs the LUT kernels' tables
enum { kModeCert = 0, kModeMax = 1, kModeThr = 2 };

template <int W, bool FIRST, int MODE, bool SYM, bool LUT>
__device__ __forceinline__ void column_block(const KernelArgs& A, const PairCtx& P, const int lane, const int cbi,
                                             double* scr, double* result, int* status, const double* emit_tab,
                                             const double* pen_tab) {
  constexpr bool EXACT = (MODE == kModeMax);
  constexpr bool FULL = (MODE == kModeThr);
  static_assert(!FULL || LUT, "the threshold test reads the LDS tables");
  constexpr bool PEN = FULL;                                   // per-cell thresholds from the LDS table (pen_tab)
  const int n = P.n, m = P.m;
  const uint8_t* __restrict__ hap = P.hap;
  const uint8_t* __restrict__ read = P.read;
  const double ca = A.mc.a, cb = A.mc.b, cc = A.mc.c, cd = A.mc.d, ce = A.mc.e, cf = A.mc.f, cg = A.mc.g;
  const double MATCH = A.mc.match, MISMATCH = A.mc.mismatch;
  const float c32 = A.mc.c;
  const double IMP = kImp;
  const double* __restrict__ lpc = A.lpc;
  const int sstride = A.scratch_stride;

  const bool final_block = (cbi == P.ncb - 1);
  const int L = final_block ? P.Ll : P.Lb;                     // active lanes
  const int Wl = final_block ? P.Wl : W;                       // real columns of the LAST lane
  const bool is_last_lane = (lane == L - 1);
  // Rows this block can already decide.  Every cell value is < 0 and the band penalty of column j in
  // row i is |k| * c with k = dd - i + j, so columns with k >= k600 stay below -600: they can neither
  // certify the row nor save it from the abort.  Row i is therefore settled once the blocks up to
  // this one cover j < k600 - dd + i -- long pairs that abort (or cannot be certified) leave after
  // the block that holds their diagonal instead of after the last one.
  const int i_dec = uni(final_block ? 0x7fffffff : ((cbi + 1) * P.Lb * W + 1 + P.dd - P.k600));
  // first column of my strip
  const int j0 = 1 + (cbi * P.Lb + lane) * W;
  // boundary strips: read what the previous block wrote, write for the next block
  const double* rdX = scr + (size_t)((cbi + 1) & 1) * 3 * sstride;
  const double* rdZ = rdX + sstride;
  const double* rdR = rdZ + sstride;
  double* wrX = scr + (size_t)(cbi & 1) * 3 * sstride;
  double* wrZ = wrX + sstride;
  double* wrR = wrZ + sstride;

  // ---- row 0 (HapAligner.cpp:263-272) for my columns -> X(0,j), Y(0,j) ----------------------
  double Xp[W], Yp[W];
  // LUT: one word per FOUR slots -- the byte offset of the quad's row in the emission table
  // ((r0 | r1 << 2 | r2 << 4 | r3 << 6) * 16); otherwise the read bytes themselves
  constexpr int NQ = (W + 3) / 4;
  uint32_t rc[LUT ? NQ : W];
  if (LUT) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) rc[q] = 0;
  }
  const uint32_t r0 = (uint32_t)uni((int)read[0]);
  // row-0 cell of (clamped) column jc: match_matrix[jc] and deletion_matrix[jc]
  // (every load unconditional, every condition a select: the set-up is a handful of independent
  // memory round trips per lane, not one after the other behind divergent branches)
  auto row0 = [&](const int jc, double& M0, double& D0j) __attribute__((always_inline)) {
    const double lp1 = lpc[max(jc - 1, 0)], lp = lpc[jc];
    const uint32_t hb = (uint32_t)hap[min(jc, n - 1)];
    const double D0jm1 = (jc == 1) ? IMP : (cg + lp1);         // deletion_matrix[j-1]
    D0j = cg + lp;                                             // deletion_matrix[j] = g + left_prob
    // match_matrix[j] = D[j-1] + d + emit(hap[j] vs read[0]): the reference indexes the
    // haplotype with the READ index here; past its end ('\0' / undefined) counts as a mismatch
    const bool eq = (jc < n) & (hb == r0);
    M0 = (D0jm1 + cd) + (eq ? MATCH : MISMATCH);
  };
#pragma unroll
  for (int s = 0; s < W; ++s) {
    const int jc = min(j0 + s, m - 1);                         // inactive lanes / the last lane's slack: clamp the loads
    double M0, D0j;
    row0(jc, M0, D0j);
    Xp[s] = dmax(M0 + ce, dmax(D0j + cd, IMP + cb));
    Yp[s] = dmax(M0 + cf, IMP + ca);
    if (LUT) rc[(s / 4) < (LUT ? NQ : W) ? (s / 4) : 0] |= (((uint32_t)read[jc] >> 1) & 3u) << (2 * (s % 4) + 4);
    else rc[s < (LUT ? NQ : W) ? s : 0] = (uint32_t)read[jc];
    if ((s % 4) == 3) __builtin_amdgcn_sched_barrier(0);     // (four slots of set-up loads in flight, not all 4W)
  }
  // X(0, j0-1): left neighbour's last slot; lane 0: X(0,0) or the previous block's strip
  double outX = Xp[W - 1];
  double leftX;
  {
    double fill = dmax(P.emit00 + ce, dmax(IMP + cd, IMP + cb));
    if (!FIRST) fill = strip_load(rdX);
    leftX = wave_shr1(outX, fill);
  }
  if (!final_block && is_last_lane) wrX[0] = Xp[W - 1];

  if (n == 1) {                                                // single row: the result is row 0's last cell (m-1)
    if (final_block) {
      double M0, D0j;
      row0(m - 1, M0, D0j);
      *result = dmax(D0j, dmax(IMP, M0));
    }
    return;
  }

  double outZ = IMP;
  double outR = IMP;                                           // EXACT: running row maximum
  // !EXACT: bit l = "some lane <= l certified the row lane l is on".  The certificate chain lives in
  // SGPRs: per step one v_cmp, then shift / or / bit-test on the scalar unit (no VALU, no DPP)
  // It starts as all ones: the ones sit ahead of the wavefront (lane l only ever reads what lane l-1
  // produced on a real row), so the last lane's bit reads "certified" until its first real row
  // arrives and the per-step test needs no "has the last lane started" condition.
  uint64_t fmask = ~0ull;
  const uint64_t lastbit = 1ull << (L - 1);
  const uint64_t watch = final_block ? lastbit : 0;            // final block: every row of the last lane is settled
  double certM = 0.0;                                          // M of my slot 0 in my current row
  double minR = 0.0;                                           // EXACT: smallest row maximum I finished so far
  double res_cap = 0.0, res_slot = 0.0;
  const int T = (n - 1) + (L - 1);
  // per-step inputs, loaded one step ahead
  // my row at step t is t + 1 - lane; I am active while 1 <= row <= n-1, i.e. lane <= t <= lane+n-2
  // haplotype rows stream as (uniform base + t)[per-lane constant]: scalar pointer bump, no VALU.
  // LUT kernels stream the pre-coded table-row offsets instead of the bytes.  (The buffers are
  // padded by >= 96 bytes either side on the device, so rows outside [0, n) -- read only by lanes
  // that are not active at that step -- need no clamping.)
  typedef typename std::conditional<LUT, uint16_t, uint8_t>::type hrow_t;
  const hrow_t* __restrict__ hs = (LUT ? (const hrow_t*)P.hapc : (const hrow_t*)hap) - 63;
  const uint32_t hoff = 64u - (uint32_t)lane;                  // (hs + t)[hoff] = row t + 1 - lane
  // (a BUFFER load: resource = hs in scalar registers, the lane's constant in the vector offset, the step in the scalar offset --
  // as a flat pointer hipcc kept hs + hoff in a pair of VGPRs and bumped it with a v_lshl_add_u64 every step, plus a v_and_b32 for
  // the zero extension the load has already done)
  // (not for the general model at W = 20: no scalar registers to spare for the descriptor there -- a spill inside the step loop)
  constexpr bool kBufRows = SYM || W < 20;
  const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)uni64((int64_t)(uintptr_t)hs), 0, 0x7fffffff, 0x00020000);
  const int hvo = (int)(hoff * (uint32_t)sizeof(hrow_t));
  auto hrow_at = [&](const int t) __attribute__((always_inline)) -> uint32_t {
    if constexpr (!kBufRows) return (hs + t)[hoff];
    else if constexpr (LUT) return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(hrs, hvo, t * 2, 0);
    else return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(hrs, hvo, t, 0);
  };
  uint32_t h_next = hrow_at(0);
  double bX_next, bZ_next, bR_next = IMP;                      // lane 0's boundary for ITS next row
  // FIRST: record i of the interleaved column-0 table = {X, Z}(i, 0) for my emit(hap[0], read[1]); one
  // 16-byte load per step through a pointer that just advances (rows past n-1 are never used, and
  // the table is longer than any haplotype plus 64 lanes, so no clamp).
  // (the same for the column-0 records: one 16-byte buffer load per step, the record's offset in a scalar register)
  typedef double d2v_t __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)uni64((int64_t)(uintptr_t)((const double2*)A.colXZ + P.e01)), 0, 0x7fffffff, 0x00020000);
  int bso = 2 * 2 * (int)sizeof(double2);                                         // -> record 2: lane 0's row at step 1
  if (FIRST) { const double2 b1 = ((const double2*)A.colXZ)[P.e01 + 2 * 1]; bX_next = b1.x; bZ_next = b1.y; }
  else { bX_next = strip_load(rdX + 1); bZ_next = strip_load(rdZ + 1); bR_next = strip_load(rdR + 1); }
  // !EXACT certificate threshold for my slot 0 (see below): thr(k) = -600 + |k|*|c| rounded UP
  // (|c|(1+2^-22) >= the float product's magnitude, +1e-6 >> every double rounding involved)
  double kd = (double)(P.dd - (1 - lane) + j0);                // band offset k of (row, j0); -1 per step
  const double cabs_up = fabs((double)c32) * (1.0 + 0x1p-22);
  const double thr0 = -600.0 + 1e-6;

  // One wavefront step.  FIN (the very last step of the final block: only the last lane is still
  // active, on row n-1) additionally captures the pair's result.
  auto step = [&](auto fin_tag, const int t) __attribute__((always_inline)) {
    constexpr bool FIN = decltype(fin_tag)::value;
    const int i = t + 1 - lane;
    const uint32_t h = h_next;
    const double bX = bX_next, bZ = bZ_next, bR = bR_next;
    h_next = hrow_at(t + 1);
    {
      const int ib = min(t + 2, n - 1);                        // lane 0's row at the next step
      if (FIRST) {
        d2v_t bn;
        if constexpr (kBufRows) bn = __builtin_bit_cast(d2v_t, __builtin_amdgcn_raw_buffer_load_b128(brs, 0, bso, 0));
        else { const double2 b2 = *(const double2*)((const char*)((const double2*)A.colXZ + P.e01) + bso); bn.x = b2.x; bn.y = b2.y; }
        bso += 2 * (int)sizeof(double2); bX_next = bn.x; bZ_next = bn.y;
      }
      else { bX_next = strip_load(rdX + ib); bZ_next = strip_load(rdZ + ib); bR_next = strip_load(rdR + ib); }
    }
    // hand-off from the left neighbour (its state at the end of the previous step)
    const double mX = wave_shr1(outX, bX);                     // X(i, j0-1)
    const double mZ = wave_shr1(outZ, bZ);                     // Z(i, j0-1)
    double mR = IMP;
    if (EXACT) mR = wave_shr1(outR, bR);                       // row i's running maximum over columns < j0
    const double kcur = kd;
    if (!EXACT) kd = kcur - 1.0;

    // the lanes with a row at this step are a contiguous range [t-(n-2), t] clipped to [0, L-1]:
    // the mask is formed on the scalar unit and becomes EXEC without any per-lane compare
    const int a_hi = min(t, L - 1), a_lo = max(t - (n - 2), 0);
    const uint64_t active_mask = (~0ull >> (63 - a_hi)) & (~0ull << a_lo);
    const bool active = __builtin_amdgcn_inverse_ballot_w64(active_mask);
    uint64_t bprev = 0;
    uint64_t okm = 0;                             // FULL: lanes with a cell that reaches -600 with its penalty (scalar masks)
    if (active) {
      double diag = leftX;                                     // X(i-1, j0-1)
      leftX = mX;
      double zleft = mZ;
      double rm = mR, rm_cap = IMP;
      double Iv = 0.0, Dv = 0.0;
      const int k0 = P.dd - i + j0;
      // PEN: the W penalties of this row's cells, one clamped base + constant offsets (kPenHalf)
      // (fetched four slots at a time, two quads ahead of their use: W penalties at once cost 2W registers)
      double pn[PEN ? W : 1];
      const int kc = min(max(k0, -kPenHalf), kPenHalf - W);
      const double* pp = pen_tab + (kc + kPenHalf);
      auto fetch_pen = [&](const int q) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 4 * q; k < 4 * q + 4; ++k) if (k < W) pn[k < (PEN ? W : 1) ? k : 0] = pp[k];
      };
      if (PEN) {
        fetch_pen(0);
        if (NQ > 1) fetch_pen(1);
      }
      // M of slot s+1 is formed from the OLD X of slot s before that register is overwritten, so
      // old and new X never live at once (no end-of-loop register shuffle)
      // LUT: the emissions of four slots come from ONE table row -- h is the byte offset of my
      // haplotype base's block, rc[q] the offset of the quad's row inside it: one v_add_u32 and two
      // ds_read_b128 per four cells.  Quads are fetched one quad ahead of their use.
      double em[W];
      auto fetch_quad = [&](const int q) __attribute__((always_inline)) {
        // (slots 0-1 of every row live in the first 16 KB, slots 2-3 in the second: 16-byte rows
        // spread over all 16 bank quads of the ds_read_b128 lane groups)
        const double2* row = (const double2*)((const char*)emit_tab + (h + rc[q < (LUT ? NQ : W) ? q : 0]));
        const double2 lo = row[0];
        em[4 * q] = lo.x;
        if (4 * q + 1 < W) em[(4 * q + 1) < W ? (4 * q + 1) : 0] = lo.y;
        if (4 * q + 2 < W) {
          const double2 hi = row[kEmitTabDoubles / 4];
          em[(4 * q + 2) < W ? (4 * q + 2) : 0] = hi.x;
          if (4 * q + 3 < W) em[(4 * q + 3) < W ? (4 * q + 3) : 0] = hi.y;
        }
      };
      if (LUT) {
        fetch_quad(0);
        if (NQ > 1) fetch_quad(1 < NQ ? 1 : 0);
      } else {
#pragma unroll
        for (int k = 0; k < W; ++k) em[k] = (h == rc[k < (LUT ? NQ : W) ? k : 0]) ? MATCH : MISMATCH;
      }
      certM = em[0] + diag;                                    // match_matrix[i][j], :287-289
      double Mv = certM;
#pragma unroll
      for (int s = 0; s < W; ++s) {
        double Mnext = 0.0;
        if (LUT && (s % 4) == 2 && (s / 4 + 2) < NQ) fetch_quad((s / 4 + 2) < NQ ? (s / 4 + 2) : 0);
        if (PEN && (s % 4) == 2 && (s / 4 + 2) < NQ) fetch_pen(s / 4 + 2);
        if (s + 1 < W) Mnext = em[(s + 1) < W ? (s + 1) : 0] + Xp[s];
        Iv = MATCH + Yp[s];                                    // insertion_matrix[i][j], :291-292
        Dv = zleft;                                            // deletion_matrix[i][j], :294-295
        double best = 0.0;
        const double di = dmax(Dv, Iv);
        if (EXACT || FIN || FULL) best = dmax(di, Mv);         // :297 (max is exact: any association gives the same bits)
        if (!EXACT && FIN) { if (Wl == s + 1) res_cap = best; } // (!EXACT: the pair's result, in the peeled final step)
        if (SYM) {
          // b == d and f == g (the LongTR defaults and every symmetric indel model): x -> fl(x + k)
          // is monotone, so max(fl(D+d), fl(I+d)) == fl(max(D,I) + d) bit for bit, and M+f is
          // shared by Y and Z: 11 FP64 ops per cell instead of 13
          const double t2 = di + cd;
          const double mf = Mv + cf;
          Xp[s] = dmax(Mv + ce, t2);
          Yp[s] = dmax(mf, Iv + ca);
          zleft = dmax(mf, Dv + cc);
        } else {
          Xp[s] = dmax(Mv + ce, dmax(Dv + cd, Iv + cb));
          Yp[s] = dmax(Mv + cf, Iv + ca);
          zleft = dmax(Mv + cg, Dv + cc);
        }
        // pin the schedule: hipcc otherwise defers every Y update to the end of the step (two
        // more live doubles per slot) and shuffles all new X's home with W v_mov_b64's
        if (s + 1 < W) asm volatile("" : "+v"(Xp[s]), "+v"(Yp[s]), "+v"(zleft), "+v"(Mnext));
        else asm volatile("" : "+v"(Xp[s]), "+v"(Yp[s]), "+v"(zleft));
        __builtin_amdgcn_sched_barrier(0);                     // ... and keep each slot's emission fetch in its slot
        if (FULL) {
          // (the OR of slot s's mask is issued one slot later: a scalar instruction right behind the v_cmp it reads stalls the wave)
          okm |= bprev;
          bprev = __builtin_amdgcn_ballot_w64(best >= pn[s < (PEN ? W : 1) ? s : 0]);
        }
        if (EXACT) {
          {
            const float penf = (float)abs(k0 + s) * c32;       // int*float -> float, :298
            rm = dmax(rm, best + (double)penf);
          }
          // the last lane of the final block may own fewer than W real columns: its row maximum and the
          // pair's result are picked up at its last real slot -- one scalar branch per slot (Wl is
          // wave-uniform; the empty asm keeps hipcc from turning it into per-slot selects), nothing
          // kept per slot (arrays of W running maxima cost 64 spilled VGPRs)
          if (final_block && Wl == s + 1) { asm volatile("" : "+v"(rm), "+v"(best)); rm_cap = rm; res_slot = best; }
        }
        if (s + 1 < W) Mv = Mnext;
      }
      outX = Xp[W - 1];
      outZ = zleft;
      if (FULL) okm |= bprev;
      if (EXACT) {
        outR = rm;
        if (final_block && is_last_lane) outR = rm_cap;          // (Wl == W: captured at the last slot, == rm)
        if (i <= i_dec) minR = fmin(minR, outR);               // (meaningful on the last lane: the whole row, settled)
        if (final_block && i == n - 1) res_cap = res_slot;     // :309
      }
      if (!final_block && is_last_lane) {
        const int il = t + 2 - L;                              // == i on the last lane: a scalar address
        wrX[il] = outX; wrZ[il] = outZ;
        if (EXACT) wrR[il] = outR;
      }
    }
    if (FULL) {
      // the masks were formed inside the divergent region: uniform there, but a per-lane value behind it (the lanes that skipped
      // the region hold 0) -- take them back from a lane that was in it
      okm = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(okm >> 32), a_lo) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)okm, a_lo);
    }
    if (!EXACT) {
      // certificate from ONE cell per lane and row (my slot 0, always a real column; certM keeps its
      // M): best >= M there and thr(k) >= -600 - pen(k), so M >= thr(k) proves fl(best + pen) >= -600,
      // i.e. the row's band-penalised maximum cannot be below -600.  The compare runs outside the
      // `active` region so that its lane mask stays in SGPRs.
      const uint64_t cert = (FULL ? okm : __builtin_amdgcn_ballot_w64(certM >= __builtin_fma(__builtin_fabs(kcur), cabs_up, thr0))) & active_mask;
      // row flags move one lane up (lane 0: nothing yet, or what the previous block certified)
      uint64_t in0 = 0;
      if (!FIRST) in0 = __builtin_amdgcn_ballot_w64(bR != 0.0) & 1ull;
      // FULL: the last lane of the final block may own fewer than W real columns, and its mask covers all W slots.  Its bit only
      // decides a row that no lane before it certified: that case (rare) goes to the exact kernel instead of a per-slot capture
      if (FULL && final_block && Wl != W && (cert & lastbit) != 0 && (((fmask << 1) | in0) & lastbit) == 0) return 2;
      fmask = cert | (fmask << 1) | in0;
      // a settled row nobody certified: hand the pair to the exact kernel
      if ((~fmask & watch) != 0) return 1;
      if (!final_block && t >= L - 1) {                          // the last lane has just finished a row: park its flag
        const bool row_ok = (fmask & lastbit) != 0;
        const int il = t + 2 - L;                                // the row the last lane is on
        if (is_last_lane) wrR[il] = row_ok ? 1.0 : 0.0;
        if (!row_ok && il <= i_dec) return 1;
      }
    }
    return 0;
  };
  // the pair is lost (EXACT: a row maximum below -600, :300-306) as soon as the last lane has seen
  // such a row among the rows this block settles
  auto lost = [&]() __attribute__((always_inline)) {            // EXACT only (the fast kernels test a scalar every step)
    const bool bad = EXACT && is_last_lane && (minR < -600.0);
    return __builtin_amdgcn_ballot_w64(bad) != 0;
  };
  int t = 0;
  // (the certificate bodies take their steps two to a loop trip: the values fetched a step ahead -- boundary records, the haplotype
  // row -- then alternate between two sets of registers instead of being moved home every step; 2.5 of a step's 180 vector
  // instructions at W = 15.  Measured on MI355X, config 3: 234.3 -> 231.9 ms per pass.)
  if (!EXACT && !FULL && SYM) for (; t + 1 < T - 1; t += 2) {  // (the general model's widest strips and the threshold bodies have no registers to spare for it)
    if (const int rs = step(BoolTag<false>{}, t)) { *status = (FULL && A.thr_ok && rs == 1) ? kStatusAbort : kStatusUncertain; return; }
    if (const int rs = step(BoolTag<false>{}, t + 1)) { *status = (FULL && A.thr_ok && rs == 1) ? kStatusAbort : kStatusUncertain; return; }
  }
  for (; t < T - 1; ++t) {
    if (const int rs = step(BoolTag<false>{}, t)) { *status = (FULL && A.thr_ok && rs == 1) ? kStatusAbort : kStatusUncertain; return; }
    if (EXACT && (t & 3) == 3 && lost()) { *status = kStatusAbort; return; }   // early exit, every 4th step
  }
  if (final_block && !EXACT) { if (const int rs = step(BoolTag<true>{}, T - 1)) { *status = (FULL && A.thr_ok && rs == 1) ? kStatusAbort : kStatusUncertain; return; } }
  else if (const int rs = step(BoolTag<false>{}, T - 1)) { *status = (FULL && A.thr_ok && rs == 1) ? kStatusAbort : kStatusUncertain; return; }
  if (EXACT && lost()) { *status = kStatusAbort; return; }
  if (final_block) *result = lane_bcast(res_cap, L - 1);
  else __threadfence();                                        // strip stores visible before the next block reads them
}

// One pair, one wavefront.
template <int W, int MODE, bool SYM, bool LUT>
__device__ __forceinline__ double align_pair(const KernelArgs& A, PairCtx& P, double* scr, int lane, int* status,
                                             const double* emit_tab, const double* pen_tab) {
  const int C = P.m - 1;
  P.ncb = (C + 64 * W - 1) / (64 * W);
  P.Lb = (C + W * P.ncb - 1) / (W * P.ncb);                    // lanes of every block but the last
  P.ncb = (C + P.Lb * W - 1) / (P.Lb * W);                     // (very long reads: rounding Lb up can save a block)
  const int Cl = C - (P.ncb - 1) * P.Lb * W;                   // columns of the last block (>= 1)
  P.Ll = (Cl + W - 1) / W;
  P.Wl = Cl - (P.Ll - 1) * W;                                  // real columns of its last lane, 1..W
  {
    const float cabs = fabsf(A.mc.c);
    P.k600 = (cabs * 1.0e9f > 600.0f) ? ((int)(600.0f / cabs) + 2) : 0x3fffffff;
  }
  double result = 0.0;
  *status = kStatusOk;
  column_block<W, true, MODE, SYM, LUT>(A, P, lane, 0, scr, &result, status, emit_tab, pen_tab);
  for (int cbi = 1; cbi < P.ncb && *status == kStatusOk; ++cbi)
    column_block<W, false, MODE, SYM, LUT>(A, P, lane, cbi, scr, &result, status, emit_tab, pen_tab);
  return result;
}

// (exact kernels: measured on MI355X, config 3 with every pair through the exact lists: W = 16 at three waves per
// SIMD -- 168 VGPRs, a few set-up values in scratch, none in the step loop -- 1.71e12 cells/s, at two waves 1.46e12)
template <int W, bool EXACT, bool SYM, bool LUT>
__global__ __launch_bounds__(64 * kBlockWaves, EXACT ? ((W <= 4) ? 4 : ((W <= 10) ? 3 : ((W <= 16) ? LTR_XLB_LONG : 2))) : LTR_LB) void ltr_dp_kernel(KernelArgs A) {
  // a workgroup is kBlockWaves independent wavefronts (own queue pops, own scratch strips); they
  // only share the emission table
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  // [slot pair][hap base h][read bases r0..r3 of four consecutive slots][2 slots]: 16-byte rows
  __shared__ __attribute__((aligned(16))) double s_emit[LUT ? kEmitTabDoubles : 4];
  constexpr int MODE = EXACT ? (LUT ? kModeThr : kModeMax) : kModeCert;
  __shared__ double s_pen[(MODE == kModeThr) ? kPenTabDoubles : 2];    // kModeThr: the exact thresholds, entry k + kPenHalf
  if (LUT) {
    for (int idx = threadIdx.x; idx < kEmitTabDoubles; idx += 64 * kBlockWaves) {
      const int half = idx >> 11, hcode = (idx >> 9) & 3, quad = (idx >> 1) & 255, k = 2 * half + (idx & 1);
      s_emit[idx] = (hcode == ((quad >> (2 * k)) & 3)) ? (double)A.mc.match : (double)A.mc.mismatch;
    }
    if (MODE == kModeThr) {
      // (built on the host once per parameter set -- a bisection per entry, ltrp::build_threshold_table -- and copied here)
      for (int idx = threadIdx.x; idx < kPenTabDoubles; idx += 64 * kBlockWaves) s_pen[idx] = A.thr_tab[idx];
    }
    __syncthreads();
  }
  double* scr = A.scratch + ((size_t)blockIdx.x * kBlockWaves + wave) * 6 * A.scratch_stride;
  const double IMP = kImp;
  int n_pairs = A.n_pairs;
  if (A.n_pairs_dev) n_pairs = uni((int)*A.n_pairs_dev);
  for (;;) {
    // Every lane issues the add (lane 0 adds 1, the rest 0) and the first lane's return value is broadcast: with the
    // queue a kernel argument hipcc's atomic optimizer makes that one lane's add.  (NOT pop_one here: with a constant
    // queue pointer the optimizer rewrites the add inside its `if (lane == 0)` and the loop came out wrong -- the
    // launch never ended on MI355X.  pop_one is for loop-carried queue pointers, ltr_dp_multi_kernel.)
    int q = (int)atomicAdd(A.queue, lane == 0 ? 1u : 0u);
    q = uni(q);
    if (q >= n_pairs) break;
    int pi = A.first_pair + q;
    if (A.index) pi = uni(A.index[pi]);
    const PairDesc* pp = A.pairs + pi;
    const int n = uni(pp->n), m = uni(pp->m), hfl = uni(pp->hap_full_len);
    if (EXACT && (m - 1 < A.c_lo || m - 1 > A.c_hi)) continue;   // (a list shared by two exact launches: the other one's pair)
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
        // no interior column: n == 1 -> the single cell; n > 1 -> row 1 keeps max_score_per_row
        // = IMPOSSIBLE < -600 -> abort (:283, :300-306)
        r = (n == 1) ? dmax(IMP, dmax(IMP, P.emit00)) : -700.0;
      } else {
        P.e01 = (h0 == uni((int)P.read[1])) ? 1 : 0;           // emission of the whole first column, :276
        // The exact lists are few (a launch per list is latency when a list holds a handful of pairs) and wide: a read of
        // 650 columns on W = 16 strips keeps 41 of 64 lanes busy.  The LUT exact kernels therefore carry narrower bodies
        // and pick the strip width per pair (wave-uniform): registers are those of the widest body, lanes 84-100 % busy.
        const int C = m - 1;
        if (EXACT && LUT && W == 16 && C <= 64 * 12) r = align_pair<(W == 16 ? 12 : W), MODE, SYM, LUT>(A, P, scr, lane, &status, s_emit, s_pen);
        else if (EXACT && LUT && W == 16 && C <= 64 * 14) r = align_pair<(W == 16 ? 14 : W), MODE, SYM, LUT>(A, P, scr, lane, &status, s_emit, s_pen);
        else if (EXACT && LUT && W == 10 && C <= 64 * 6) r = align_pair<(W == 10 ? 6 : W), MODE, SYM, LUT>(A, P, scr, lane, &status, s_emit, s_pen);
        else if (EXACT && LUT && W == 10 && C <= 64 * 8) r = align_pair<(W == 10 ? 8 : W), MODE, SYM, LUT>(A, P, scr, lane, &status, s_emit, s_pen);
        else r = align_pair<W, MODE, SYM, LUT>(A, P, scr, lane, &status, s_emit, s_pen);
        if (MODE == kModeThr && status == kStatusUncertain) {
          // a row that only the last lane certified, and that lane owns fewer than W real columns (its mask covers all W slots):
          // rare (a few pairs in ten thousand) -- the pair is scored again with the reference's running maximum
          status = kStatusOk;
          r = align_pair<W, (MODE == kModeThr ? kModeMax : MODE), SYM, LUT>(A, P, scr, lane, &status, s_emit, s_pen);
        }
        if (status == kStatusAbort) r = -700.0;
      }
    }
    if (!EXACT && status == kStatusUncertain) {
      push_redo(A, lane, pi, m);                               // could not prove "no row aborts": an exact kernel scores the pair
    } else if (lane == 0) {
      A.out_ll[out_idx] = r;
    }
  }
}

// A real call per CLASS: every strip width's body is register-allocated on its own, exactly like its single-class kernel
// (inlined into one function -- with the kernel arguments by value or through a pointer laundered per case -- the ten bodies
// spilled 240-250 VGPRs, hundreds of scratch accesses inside the step loops).  What the body needs crosses the call in
// registers or is re-derived behind it: the class's range by value (wave-uniform: readfirstlane'd back into SGPRs), the kernel
// arguments straight from the kernel's own argument segment (scalar loads, as in a kernel -- a `const KernelArgs&` would point
// into the caller's scratch), the emission table as its LDS address (a generic pointer would turn every ds_read into a flat
// load: measured 0.72 of the FP64 peak against 0.80).  The call is per class, not per pair: a callee saves the ~50 callee-saved
// VGPRs it uses on entry, and with a call per pair those saves were 6.2 GB of scratch write-backs per config-3 pass
// (rocprofv3 WRITE_SIZE) -- no time, but 40 x the pass's algorithmic bytes.
typedef __attribute__((address_space(3))) const double* LdsDoubles;
typedef __attribute__((address_space(4))) const KernelArgs* KernArgPtr;

template <int W, bool SYM>
__device__ __attribute__((noinline)) void class_walk_call(int64_t kernarg_v, int first_pair_v, int n_pairs_v, int cls_v, unsigned emit_lds_v) {
  // (the address of the kernel's argument segment comes as an argument: __builtin_amdgcn_kernarg_segment_ptr() behind a call
  // returned null on ROCm 7.2 / gfx950)
  const KernelArgs& A = *(const KernelArgs*)(KernArgPtr)(uintptr_t)uni64(kernarg_v);
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const double* emit_tab = (const double*)(LdsDoubles)(uintptr_t)(unsigned)uni((int)emit_lds_v);
  double* scr = A.scratch + ((size_t)blockIdx.x * kBlockWaves + wave) * 6 * A.scratch_stride;
  const int first_pair = uni(first_pair_v), n_pairs = uni(n_pairs_v);
  uint32_t* queue = A.queue_base + uni(cls_v);
  const double IMP = kImp;
  for (;;) {
    const int q = pop_one(queue, lane);
    if (q >= n_pairs) break;
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
    if (status == kStatusUncertain) push_redo(A, lane, pi, m);   // could not prove "no row aborts": an exact kernel scores the pair
    else if (lane == 0) A.out_ll[out_idx] = r;
  }
}

// The one-wave certificate kernels of strip widths kMultiMinW .. kWMax as ONE persistent launch (see KernelArgs::mk_*): the
// classes of the plan in launch order, every pair scored by the body of its own class's strip width -- same code, same bits
// as ltr_dp_kernel<W, false, SYM, true> -- and no drain between classes.
static_assert(kWMax == 20 && kMultiMinW == 11 && kWMax - kMultiMinW + 1 == kMultiMax, "the switch below names the strip widths 11 .. 20");
template <bool SYM>
__global__ __launch_bounds__(64 * kBlockWaves, 3) void ltr_dp_multi_kernel(KernelArgs A) {
  __shared__ __attribute__((aligned(16))) double s_emit[kEmitTabDoubles];
  for (int idx = threadIdx.x; idx < kEmitTabDoubles; idx += 64 * kBlockWaves) {
    const int half = idx >> 11, hcode = (idx >> 9) & 3, quad = (idx >> 1) & 255, k = 2 * half + (idx & 1);
    s_emit[idx] = (hcode == ((quad >> (2 * k)) & 3)) ? (double)A.mc.match : (double)A.mc.mismatch;
  }
  __syncthreads();
  const unsigned emit_lds = (unsigned)(uintptr_t)(LdsDoubles)s_emit;
  const int64_t kargs = (int64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int n_ranges = A.mk_n;
  for (int r = 0; r < n_ranges; ++r) {
    // (range r's parameters by a select chain: indexing the argument struct with r would put it into scratch)
    int W = A.mk_w[0], first_pair = A.mk_first[0], n_pairs = A.mk_np[0], cls = A.mk_class[0];
#pragma unroll
    for (int j = 1; j < kMultiMax; ++j) if (r == j) { W = A.mk_w[j]; first_pair = A.mk_first[j]; n_pairs = A.mk_np[j]; cls = A.mk_class[j]; }
    switch (W) {
      case 11: class_walk_call<11, SYM>(kargs, first_pair, n_pairs, cls, emit_lds); break;
      case 12: class_walk_call<12, SYM>(kargs, first_pair, n_pairs, cls, emit_lds); break;
      case 13: class_walk_call<13, SYM>(kargs, first_pair, n_pairs, cls, emit_lds); break;
      case 14: class_walk_call<14, SYM>(kargs, first_pair, n_pairs, cls, emit_lds); break;
      case 15: class_walk_call<15, SYM>(kargs, first_pair, n_pairs, cls, emit_lds); break;
      case 16: class_walk_call<16, SYM>(kargs, first_pair, n_pairs, cls, emit_lds); break;
      case 17: class_walk_call<17, SYM>(kargs, first_pair, n_pairs, cls, emit_lds); break;
      case 18: class_walk_call<18, SYM>(kargs, first_pair, n_pairs, cls, emit_lds); break;
      case 19: class_walk_call<19, SYM>(kargs, first_pair, n_pairs, cls, emit_lds); break;
      default: class_walk_call<20, SYM>(kargs, first_pair, n_pairs, cls, emit_lds); break;
    }
  }
}

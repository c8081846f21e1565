// ltr_dp_chain.hpp -- one pair per wavefront WITHOUT the fill and drain of the skew: the next pair's first rows enter the
// lanes the previous pair's last rows have left (included after ltr_dp_kernel.hpp / ltr_dp_redo.hpp by ltr_k_plan.hip).
//
// Replaces HapAligner::align_seq_to_hap (reference src/SeqAlignment/HapAligner.cpp:236-343; the loop :282-307) for the pairs of
// a one-wave class whose read fits ONE column block (C <= 64 W) -- every pair of BASELINE config 3.  Same recurrence, same
// certificate, same emission table as column_block<W, true, kModeCert, SYM, true> (ltr_dp_kernel.hpp): same bits.
//
// Why.  Lane l works on haplotype row t - l + 1 at step t: a pair of n rows on L lanes takes (n - 1) + (L - 1) steps of which
// L - 1 run with part of the wavefront idle -- 63 of ~975 for the 930-base pairs of config 3's largest class, 63 of ~735 for its
// 700-base ones: 6 - 9 % of every one-wave launch (the packed kernels exist because of it; a one-wave pair cannot be packed).
// Here the wavefront is a pipeline that never drains inside a class: when lane 0 has finished pair A's last row it starts pair B's
// row 1 in the next step, lane 1 follows one step later, ... -- at any step the lanes [0, x) are on B and the lanes [x + 1, L) on A.
//   * what describes a pair stays wave-uniform (scalar registers), now for two pairs: the one lane 0 is on (C) and the one still
//     draining in the upper lanes (O); which lanes are on which is a scalar lane mask made from the step counter, as before;
//   * what a lane must hold when it ENTERS a pair -- X(0, j), Y(0, j) of its W columns (HapAligner.cpp:263-272), its read bases as
//     table offsets -- is computed for all 64 lanes at once a few steps before lane 0 needs it (one set-up per pair, as before)
//     and parked in the wave's scratch strip (free: a one-block pair never uses it); the lane picks its 2 W + 1 doubles up in the
//     step in which it enters (loads under a one-lane exec mask: vector-memory instructions, not vector-ALU ones);
//   * the certificate chain needs nothing: the bit a lane receives is its left neighbour's of one step ago, which is the same row
//     of the same pair; a pair's result and its "some row was not certified" flag are taken when its last lane finishes;
//   * pairs that cannot enter the pipeline (constant scores, reads of one base, more than one column block, fewer than
//     kChainMinRows rows -- three pairs would be in flight) are noted for the plain body (bit 30 of the note).
// A pair whose certificate fails is noted for the exact body like in plan_class_call.

constexpr int kChainPre = 4;                                   // the next pair is popped and set up this many steps before lane 0 enters it
constexpr int kChainMinRows = 64 + kChainPre + 6;              // rows (n - 1) a pair needs to enter the pipeline: at most two pairs in flight
// STATUS (round 5): bit-identical to the plain walk and AS FAST, not faster -- off by default (ltr_ctx_set_debug "chain" = 1 turns
// it on; tests/test_gpu_scale.py keeps it honest).  Measured on MI355X, a 1250-locus shard of config 3 (tests/manual/gpu_chain_ab.py,
// profiles/r05/chain/): plain 30.2 ms per pass, chained strip widths 11 .. 20 30.4, 11 .. 15 30.2, 15 alone 30.1; per-dispatch
// counters of the plan kernel: 1.721e10 vector instructions against 1.770e10 (-2.8 %: the fill and drain steps are gone) at a
// vector-ALU issue rate of 0.95 against 0.98 -- what the pipeline saves in steps it loses in waits.  How it got there:
//   1. first version (the entering lane loads its parked row from the strip -- global memory through L2 -- in the step it needs
//      it; best-of-three + select per slot in every rotation step): 31.4 ms for the widths 11 .. 15, 35.2 for 11 .. 20;
//   2. issuing those loads from inline asm one step ahead is NOT safe: hipcc moves the carried registers to other registers
//      behind the branch and the late data landed in registers that held addresses by then (a memory fault);
//   3. the parked row through a one-row LDS ring, fetched from the strip one step ahead by 2 W + NQ + 1 lanes: no change (31.0)
//      -- the round trip was not the largest wait;
//   4. the ~64 steps around a rotation as a loop of their own: 30.2 - 30.4.  As part of the outer loop (set-up, rotation, steady
//      loop and rotation steps as alternatives of one loop body) hipcc ended every rotation step in s_waitcnt vmcnt(0) -- the
//      haplotype row codes prefetched for the NEXT step were waited for in THIS one, an L2 / HBM round trip per step.
// What is left (ISA, W = 12): a rotation step is 188 vector instructions against 149 steady (the fetch, the ring, selects of the
// two haplotype streams), 106 scalar ones against 21 - 32, the entering lane's 2 W + 1 LDS reads waited for inside the step; the
// strips of 14 columns and more keep 2 - 4 spilled table offsets in the steady loop (168 registers: three waves per SIMD).
constexpr int kChainGap = 0;
// The entering lane's parked row on its way from the strip (global memory, written and read through L2) to its registers: a
// one-row ring in LDS per wavefront.  In the step BEFORE a lane enters, the lanes 0 .. 2 W + NQ fetch one word each of that lane's
// row from the strip (one load per lane, issued ahead of the step's arithmetic, written to LDS behind it: the L2 round trip hides
// behind the step); the entering lane then reads its 2 W + 1 doubles from LDS (~100 cycles instead of ~2000).
constexpr int kChainRingDoubles = 2 * kWMax + 1 + 3;           // X/Y of W slots, the left neighbour's X(0, j0 - 1), NQ <= 6 words
constexpr int kNotePlain = 0x40000000;                         // note: score this pair with the plain one-wave body (plan_plain_pair_call)

// The lane number, formed where it is needed (v_mbcnt) instead of kept: the steady loop runs at exactly the register budget of
// three waves per SIMD (168), and a lane number or a parked-strip address alive across it meant three spilled table offsets
// reloaded every step (ISA, W = 15).  Volatile: not hoisted out of the cold paths that call it.
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <int W, bool SYM>
__device__ __forceinline__ int chain_walk(const KernelArgs& A, uint32_t* queue, const int first_pair, const int n_pairs, double* scr,
                                          const double* emit_tab, int* note, double* ring) {
  const double ca = A.mc.a, cb = A.mc.b, cc = A.mc.c, cd = A.mc.d, ce = A.mc.e, cf = A.mc.f, cg = A.mc.g;
  const double MATCH = A.mc.match, MISMATCH = A.mc.mismatch;
  const float c32 = A.mc.c;
  const double IMP = kImp;
  const double cabs_up = fabs((double)c32) * (1.0 + 0x1p-22);
  const double thr0 = -600.0 + 1e-6;
  constexpr int NQ = (W + 3) / 4;
  constexpr int NB = (W + 7) / 8;
  // the parked set-up of the pair that enters next: [2 s][lane] = X(0, j0 + s), [2 s + 1][lane] = Y(0, j0 + s), then NQ words per lane
  double* const stX = scr;
  uint32_t* const stR = (uint32_t*)(scr + 2 * W * 64);
  double* const stH = scr + 2 * W * 64 + NQ * 32;              // header: [0] = X(0,0) fill of lane 0, [1] = e01 (as a double)

  // ---- per lane ----
  double Xp[W], Yp[W];
  uint32_t rc[NQ];
#pragma unroll
  for (int s = 0; s < W; ++s) { Xp[s] = IMP; Yp[s] = IMP; }
#pragma unroll
  for (int q = 0; q < NQ; ++q) rc[q] = 0;
  double leftX = IMP, outX = IMP, outZ = IMP, certM = 0.0, kd = 0.0, res_cap = 0.0;
  uint64_t fmask = 0;
  const uint32_t hoff = 64u - (uint32_t)fresh_lane();          // (hs + t)[hoff] = row t + 1 - lane

  // ---- per pair, wave-uniform: C = the pair lane 0 is on (or about to enter), O = the pair draining in the upper lanes ----
  int nC = 2, LC = 0, WlC = 1, ddC = 0, piC = -1, GC = 0, lostC = 0;
  int nO = 2, LO = 0, WlO = 1, piO = -1, GO = 0, lostO = 0;
  int64_t outC = 0, outO = 0;
  const uint16_t* hsC = A.hap_codes - 63;
  const uint16_t* hsO = hsC;
  const double2* colC = (const double2*)A.colXZ;
  double fillC = IMP;
  bool haveC = false, haveO = false, haveN = false, stop = false;
  int piN = -1, GN = 0;
  int noted = 0, drained = 0;
  uint32_t h_next = 0;
  double bX_next = IMP, bZ_next = IMP;

  // set-up of pair `pi` for all lanes at once, parked in the strip (what ltr_dp_pack.hpp's set-up computes, by the same operations)
  auto park = [&](const int pi) __attribute__((always_inline)) {
    const int lane = fresh_lane();
    const PairDesc* pp = A.pairs + pi;
    const int n = uni(pp->n), m = uni(pp->m);
    const uint8_t* __restrict__ hap = A.hap_bytes + uni64(pp->hap_off);
    const uint8_t* __restrict__ read = A.read_bytes + uni64(pp->read_off);
    const int j0 = 1 + lane * W;
    const int js = min(j0, m - 1);
    uint64_t rw[NB], hw[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) { __builtin_memcpy(&rw[k], read + js + 8 * k, 8); __builtin_memcpy(&hw[k], hap + js + 8 * k, 8); }
    const uint32_t h0 = (uint32_t)uni((int)hap[0]), r0 = (uint32_t)uni((int)read[0]), r1 = (uint32_t)uni((int)read[1]);
    const double emit00 = (h0 == r0) ? MATCH : MISMATCH;        // match_matrix[0], :265
    const double2* __restrict__ row0 = (const double2*)A.row0XY;
    uint32_t rcn[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) rcn[q] = 0;
#pragma unroll
    for (int s = 0; s < W; ++s) {
      const uint32_t rb = (uint32_t)(rw[s / 8] >> (8 * (s % 8))) & 0xffu;
      const uint32_t hb = (uint32_t)(hw[s / 8] >> (8 * (s % 8))) & 0xffu;
      const int jc = min(js + s, m - 1);
      const uint32_t eq = ((js + s < n) & (hb == r0)) ? 1u : 0u;   // (the reference indexes the haplotype with the READ index here, :268)
      const double2 xy = row0[2 * jc + eq];
      stX[(2 * s) * 64 + lane] = xy.x;
      stX[(2 * s + 1) * 64 + lane] = xy.y;
      rcn[s / 4] |= ((rb >> 1) & 3u) << (2 * (s % 4) + 4);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) stR[q * 64 + lane] = rcn[q];
    if (lane == 0) {
      stH[0] = dmax(emit00 + ce, dmax(IMP + cd, IMP + cb));     // X(0, 0): the diagonal of (1, 1)
      stH[1] = (h0 == r1) ? 1.0 : 0.0;                          // emission of the whole first column, :276
    }
    __threadfence();                                           // parked before anybody picks it up
  };

  // One wavefront step at global step g.  Two copies, one loop each (copies as alternatives inside ONE loop make hipcc keep two
  // register sets for the 2 W carried values, ltr_dp_pack.hpp):
  //   STEADY  every lane is on C (nobody enters, nothing drains): the loop a pair spends ~93 % of its steps in;
  //   else    the ~64 steps around a rotation: lanes enter C one per step while O drains above them; this copy also computes the
  //           best-of-three of every slot so that O's result (:309) can be taken in the step its last lane finishes (res_cap).
  auto step = [&](auto steady_tag, const int g) __attribute__((always_inline)) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    constexpr bool FIN = !STEADY;
    const int tC = g - GC, tO = g - GO;
    // lanes with a row at this step: contiguous ranges of C and of O, made on the scalar unit
    uint64_t maskC = 0, maskO = 0;
    if (haveC && tC >= 0) {
      const int hi = min(tC, LC - 1), lo = max(tC - (nC - 2), 0);
      if (hi >= lo) maskC = (~0ull >> (63 - hi)) & (~0ull << lo);
    }
    if (!STEADY && haveO) {
      const int hi = min(tO, LO - 1), lo = max(tO - (nO - 2), 0);
      if (hi >= lo) maskO = (~0ull >> (63 - hi)) & (~0ull << lo);
    }
    const uint64_t active_mask = maskC | maskO;
    const uint32_t h = h_next;
    const double bX = bX_next, bZ = bZ_next;
    // the haplotype row codes of the NEXT step: the lanes that will be on C (l <= tC + 1) from C's stream, the others from O's
    {
      if (STEADY) h_next = (hsC + (tC + 1))[hoff];
      else {
        h_next = (hsO + (haveO ? tO + 1 : 0))[hoff];
        if (haveC && tC + 1 >= 0) {
          const uint64_t onC = (tC + 1 >= 63) ? ~0ull : (~0ull >> (62 - tC));
          if (__builtin_amdgcn_inverse_ballot_w64(onC)) h_next = (hsC + (tC + 1))[hoff];
        }
      }
      const double2 bn = colC[2 * min(max(tC + 2, 1), A.table_len)];   // lane 0's first-column record of its next row (C's table)
      bX_next = bn.x; bZ_next = bn.y;
    }
    const double mX = wave_shr1(outX, bX);                     // X(i, j0-1)
    const double mZ = wave_shr1(outZ, bZ);                     // Z(i, j0-1)
    const double kcur = kd;
    kd = kcur - 1.0;
    const bool active = __builtin_amdgcn_inverse_ballot_w64(active_mask);
    if (active) {
      double diag = leftX;
      leftX = mX;
      double zleft = mZ;
      double Iv = 0.0, Dv = 0.0;
      double em[W];
      auto fetch_quad = [&](const int q) __attribute__((always_inline)) {
        const double2* row = (const double2*)((const char*)emit_tab + (h + rc[q < NQ ? q : 0]));
        const double2 lo = row[0];
        em[4 * q] = lo.x;
        if (4 * q + 1 < W) em[(4 * q + 1) < W ? (4 * q + 1) : 0] = lo.y;
        if (4 * q + 2 < W) {
          const double2 hi = row[kEmitTabDoubles / 4];
          em[(4 * q + 2) < W ? (4 * q + 2) : 0] = hi.x;
          if (4 * q + 3 < W) em[(4 * q + 3) < W ? (4 * q + 3) : 0] = hi.y;
        }
      };
      fetch_quad(0);
      if (NQ > 1) fetch_quad(1 < NQ ? 1 : 0);
      certM = em[0] + diag;                                    // match_matrix[i][j], :287-289
      double Mv = certM;
#pragma unroll
      for (int s = 0; s < W; ++s) {
        double Mnext = 0.0;
        if ((s % 4) == 2 && (s / 4 + 2) < NQ) fetch_quad((s / 4 + 2) < NQ ? (s / 4 + 2) : 0);
        if (s + 1 < W) Mnext = em[(s + 1) < W ? (s + 1) : 0] + Xp[s];
        Iv = MATCH + Yp[s];                                    // insertion_matrix[i][j], :291-292
        Dv = zleft;                                            // deletion_matrix[i][j], :294-295
        double di = dmax(Dv, Iv);
        // :309 -- O's result is best-of-three of its last lane's last real slot: one scalar branch per slot (WlO is wave-uniform;
        // the empty asm keeps hipcc from turning it into a max and two selects per slot)
        if (FIN) { if (WlO == s + 1) { asm volatile("" : "+v"(di), "+v"(Mv)); res_cap = dmax(di, Mv); } }
        if (SYM) {
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
        if (s + 1 < W) asm volatile("" : "+v"(Xp[s]), "+v"(Yp[s]), "+v"(zleft), "+v"(Mnext));
        else asm volatile("" : "+v"(Xp[s]), "+v"(Yp[s]), "+v"(zleft));
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < W) Mv = Mnext;
      }
      outX = Xp[W - 1];
      outZ = zleft;
    }
    // certificate: one cell per lane and row (column_block); the chain moves one lane up per step, rows and pairs alike
    const uint64_t cert = __builtin_amdgcn_ballot_w64(certM >= __builtin_fma(__builtin_fabs(kcur), cabs_up, thr0)) & active_mask;
    fmask = cert | (fmask << 1);
    // a last lane has just finished a row nobody certified: that pair goes to the exact body
    if (haveC && LC > 0 && ((~fmask & maskC) >> (LC - 1) & 1ull)) lostC = 1;
    if (!STEADY && haveO && LO > 0 && ((~fmask & maskO) >> (LO - 1) & 1ull)) lostO = 1;
  };

  for (int g = 0;; ++g) {
    // ---- (1) the pair that enters next: popped and parked kChainPre steps before lane 0 leaves C (or at once when nothing runs) ----
    if (!haveN && !stop && (!haveC || g == GC + nC - 2 - kChainPre)) {
      for (;;) {
        if (noted >= kRedoNote - 2) { stop = true; break; }   // (the list is nearly full: drain what is in flight and leave)
        const int lane = fresh_lane();
        const int q = pop_one(queue, lane);
        if (q >= n_pairs) { stop = true; drained = kWalkDrained; break; }
        const int pi = first_pair + q;
        const PairDesc* pp = A.pairs + pi;
        const int n = uni(pp->n), m = uni(pp->m), hfl = uni(pp->hap_full_len);
        const bool chainable = hfl > 60 && abs(n - m) <= 600 && m >= 2 && (m - 1) <= 64 * W && (n - 1) >= kChainMinRows;
        if (!chainable) { if (lane == 0) note[noted] = pi | kNotePlain; ++noted; continue; }
        park(pi);
        piN = pi; GN = haveC ? (GC + nC - 1 + kChainGap) : (g + 1 + kChainGap); haveN = true;
        break;
      }
    }
    // ---- (2) rotation: in the step in which lane 0 is on C's last row (or nothing runs) the entering pair becomes C ----
    if ((haveC && g == GC + nC - 2) || (!haveC && haveN && g == GN - 1 - kChainGap)) {
      haveO = haveC; nO = nC; LO = LC; WlO = WlC; piO = piC; GO = GC; lostO = lostC; outO = outC; hsO = hsC;
      haveC = haveN; haveN = false;
      if (haveC) {
        const PairDesc* pp = A.pairs + piN;
        const int n = uni(pp->n), m = uni(pp->m);
        nC = n; ddC = n - m; piC = piN; GC = GN; lostC = 0; outC = uni64(pp->out_idx);
        const int C = m - 1;
        LC = (C + W - 1) / W; WlC = C - (LC - 1) * W;
        hsC = A.hap_codes + uni64(pp->hap_off) - 63;
        fillC = lane_bcast(strip_load(stH), 0);
        colC = (const double2*)A.colXZ + (lane_bcast(strip_load(stH + 1), 0) != 0.0 ? 1 : 0);
      } else { LC = 0; }
    }
    if (!haveC && !haveO) { if (stop) break; continue; }
    // ---- (3) steady state: every lane on C until the next event (the next pair's set-up, or the rotation) ----
    if (haveC && !haveO && g - GC >= LC) {
      const int g_end = GC + nC - 2 - ((!haveN && !stop) ? kChainPre : 0);
      for (; g < g_end; ++g) step(BoolTag<true>{}, g);
      --g;                                                     // (the for's ++g brings it back to g_end: the events of that step)
      continue;
    }
    // ---- (4) around a rotation, a loop of its own (no set-up or rotation can fall into it: the next one is >= kChainMinRows -
    // kChainPre steps after lane 0 entered C): per step the lane that enters C reads its parked row from the LDS ring, the lanes
    // 0 .. 2 W + NQ fetch the NEXT entering lane's row from the strip (written to the ring behind the step), the step, and in
    // O's last step its result ----
    {
      const int g_fin = haveO ? GO + (nO - 1) + (LO - 1) - 1 : -1;
      const int g_end = max(haveC ? GC + LC - kChainGap : g, g_fin + 1);
      uint32_t* const ringw = (uint32_t*)(ring + 2 * W + 1);
      for (; g < g_end; ++g) {
        const int le = g - GC + kChainGap;                     // the lane that enters C now
        if (haveC && le >= 0 && le < LC) {
          if (__builtin_amdgcn_inverse_ballot_w64(1ull << le)) {   // (the one lane, by a scalar mask)
            const double2* r2 = (const double2*)ring;
#pragma unroll
            for (int s = 0; s < W; ++s) { const double2 xy = r2[s]; Xp[s] = xy.x; Yp[s] = xy.y; }
#pragma unroll
            for (int q = 0; q < NQ; ++q) rc[q] = ringw[q];
            leftX = (le == 0) ? fillC : ring[2 * W];           // X(0, j0 - 1): lane 0's is X(0, 0)
            kd = (double)(ddC + le * W + kChainGap);           // band offset k of (row 1, j0) when the lane gets there; -1 per step
            // (outX / outZ stay: they are O's last row of this lane, which the right neighbour reads in this very step)
          }
        }
        // word i of the next entering lane's row: X/Y of slot i / 2 (i < 2 W), its left neighbour's X(0, j0 - 1) (i = 2 W), its
        // table offsets (2 W < i <= 2 W + NQ) -- ONE unconditional load of each kind per lane, the address picked by selects
        // (loads inside branches made hipcc wait for them at the branch's end)
        const bool fetch = haveC && le + 1 >= 0 && le + 1 < LC;
        const int ln = fetch ? le + 1 : 0;
        const int fi = fresh_lane();
        const int frow = fi < 2 * W ? fi : 2 * (W - 1);
        const int fcol = fi < 2 * W ? ln : max(ln - 1, 0);
        const double fv = strip_load(stX + frow * 64 + fcol);
        const int fq = min(max(fi - 2 * W - 1, 0), NQ - 1);
        const uint32_t fw = __hip_atomic_load(stR + fq * 64 + ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        res_cap = 0.0;                                         // (not alive across steps: only the value of O's finishing step is read)
        step(BoolTag<false>{}, g);
        if (fetch) {
          const int i = fresh_lane();
          if (i <= 2 * W) ring[i] = fv;
          else if (i <= 2 * W + NQ) ringw[i - 2 * W - 1] = fw;
        }
        if (g == g_fin) {                                      // O's last lane has just finished O's last row
          const int lane = fresh_lane();
          if (lostO) { if (lane == 0) note[noted] = piO; ++noted; }
          else if (lane == LO - 1) A.out_ll[outO] = res_cap;
          haveO = false;
        }
      }
      --g;                                                     // (the outer for's ++g brings it back to g_end)
    }
  }
  return noted | drained;
}

// The walk of one class as a real call (plan_class_call's reasons), and the plain body for the pairs it cannot take.
template <int W, bool SYM>
__device__ __attribute__((noinline)) int plan_chain_call(int64_t kernarg_v, int first_pair_v, int n_pairs_v, int cls_v, unsigned emit_lds_v, unsigned note_lds_v, unsigned ring_lds_v) {
  const KernelArgs& A = *(const KernelArgs*)(KernArgPtr)(uintptr_t)uni64(kernarg_v);
  const int wave = uni((int)(threadIdx.x >> 6));
  const double* emit_tab = (const double*)(LdsDoubles)(uintptr_t)(unsigned)uni((int)emit_lds_v);
  int* note = (int*)(LdsInts)(uintptr_t)(unsigned)uni((int)note_lds_v);
  double* scr = A.scratch + ((size_t)blockIdx.x * kBlockWaves + wave) * 6 * A.scratch_stride;
  double* ring = (double*)(__attribute__((address_space(3))) double*)(uintptr_t)(unsigned)uni((int)ring_lds_v);
  return chain_walk<W, SYM>(A, A.queue_base + uni(cls_v), uni(first_pair_v), uni(n_pairs_v), scr, emit_tab, note, ring);
}

// One pair with the plain one-wave body (constant scores, one-base reads, several column blocks, short haplotypes).  Returns 1
// when its certificate failed (the caller has the exact body score it).
template <int W, bool SYM>
__device__ __attribute__((noinline)) int plan_plain_pair_call(int64_t kernarg_v, int pi_v, unsigned emit_lds_v) {
  const KernelArgs& A = *(const KernelArgs*)(KernArgPtr)(uintptr_t)uni64(kernarg_v);
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const double* emit_tab = (const double*)(LdsDoubles)(uintptr_t)(unsigned)uni((int)emit_lds_v);
  double* scr = A.scratch + ((size_t)blockIdx.x * kBlockWaves + wave) * 6 * A.scratch_stride;
  const PairDesc* pp = A.pairs + uni(pi_v);
  const int n = uni(pp->n), m = uni(pp->m), hfl = uni(pp->hap_full_len);
  const int64_t out_idx = uni64(pp->out_idx);
  const double IMP = kImp;
  double r;
  int status = kStatusOk;
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
      r = align_pair<W, kModeCert, SYM, true>(A, P, scr, lane, &status, emit_tab, nullptr);
    }
  }
  if (status == kStatusUncertain) return 1;
  if (lane == 0) A.out_ll[out_idx] = r;
  return 0;
}

// ltr_dp_dual.hpp -- TWO pairs per wavefront (included by ltr_gpu.hip after ltr_dp_kernel.hpp).
//
// Same recurrence, certificate and emission table as ltr_dp_kernel (EXACT = false, LUT = true), for
// pairs whose read fits ONE column block of 32 lanes: lanes 0..31 score pair A, lanes 32..63 pair B,
// each lane owning W <= 32 consecutive columns.  Why: the anti-diagonal skew costs L-1 wavefront
// steps of fill and drain per pair -- 63 steps at 64 lanes, 8 % of a 700-row pair -- and every
// step carries ~12 bookkeeping instructions whatever W is.  Half the lanes with strips twice as
// wide halve both: 31 skew steps, bookkeeping shared by 2W cells.
//
// What is per pair and wave-uniform in ltr_dp_kernel (n, m, pointers, geometry) is per HALF here
// and selected by lane; what was per step and scalar stays scalar: the step-activity mask, the
// certificate chain (cut between lanes 31 and 32) and the row tests are 64-bit SGPR masks with one
// 32-bit half per pair.  Lane 32 takes its left boundary from the model table like lane 0 (four
// selects per step undo the DPP shift across the halves).  The two pairs run in lock step until the
// longer one ends; a pair the certificate cannot clear only stops contributing (its half idles).

#ifndef LTR_DUAL_QPF
#define LTR_DUAL_QPF 1
#endif

struct DualArgs {                  // scalar description of one half
  int n, m, dd, L, Wl, T;
  bool lost;                       // absent, uncertain or finished
};

template <int W, bool SYM>
__device__ __forceinline__ void dual_pairs(const KernelArgs& A, const PairCtx& PA, const PairCtx& PB, const bool haveB,
                                           const int lane, double* resA, double* resB, int* statA, int* statB,
                                           const double* emit_tab) {
  const double ca = A.mc.a, cb = A.mc.b, cc = A.mc.c, cd = A.mc.d, ce = A.mc.e, cf = A.mc.f, cg = A.mc.g;
  const double MATCH = A.mc.match, MISMATCH = A.mc.mismatch;
  const float c32 = A.mc.c;
  const double IMP = kImp;
  const double* __restrict__ lpc = A.lpc;
  const bool isB = lane >= 32;
  const int hl = lane & 31;                                    // lane inside my half
  // ---- per-half scalars, and their per-lane selections --------------------------------------
  DualArgs a, b;
  a.n = PA.n; a.m = PA.m; a.dd = PA.dd; a.lost = false;
  b.n = haveB ? PB.n : 2; b.m = haveB ? PB.m : 2; b.dd = b.n - b.m; b.lost = !haveB;
  a.L = (a.m - 1 + W - 1) / W; a.Wl = (a.m - 1) - (a.L - 1) * W; a.T = (a.n - 1) + (a.L - 1);
  b.L = (b.m - 1 + W - 1) / W; b.Wl = (b.m - 1) - (b.L - 1) * W; b.T = haveB ? (b.n - 1) + (b.L - 1) : 0;
  const int n = isB ? b.n : a.n, m = isB ? b.m : a.m, dd = isB ? b.dd : a.dd;
  const int Wl = isB ? b.Wl : a.Wl;
  const uint8_t* __restrict__ hap = isB ? PB.hap : PA.hap;
  const uint16_t* __restrict__ hapc = isB ? PB.hapc : PA.hapc;
  const uint8_t* __restrict__ read = isB ? PB.read : PA.read;
  const double emit00 = isB ? PB.emit00 : PA.emit00;
  const uint32_t e01 = (uint32_t)(isB ? PB.e01 : PA.e01);

  const int j0 = 1 + hl * W;                                   // first column of my strip
  // ---- row 0 (HapAligner.cpp:263-272) for my columns -> X(0,j), Y(0,j) ----------------------
  double Xp[W], Yp[W];
  constexpr int NQ = (W + 3) / 4;
  uint32_t rc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) rc[q] = 0;
  const uint32_t r0 = (uint32_t)read[0];
#pragma unroll
  for (int s = 0; s < W; ++s) {
    const int jc = min(j0 + s, m - 1);                         // inactive lanes / the last lane's slack: clamp the loads
    // every load unconditional, every condition a select: the set-up is a handful of independent
    // memory round trips per lane, not one after the other behind divergent branches
    const double lp1 = lpc[max(jc - 1, 0)], lp = lpc[jc];
    const uint32_t hb = (uint32_t)hap[min(jc, n - 1)];
    const uint32_t rb = (uint32_t)read[jc];
    const double D0jm1 = (jc == 1) ? IMP : (cg + lp1);         // deletion_matrix[j-1]
    const double D0j = cg + lp;                                // deletion_matrix[j] = g + left_prob
    // match_matrix[j] = D[j-1] + d + emit(hap[j] vs read[0]): the reference indexes the
    // haplotype with the READ index here; past its end ('\0' / undefined) counts as a mismatch
    const bool eq = (jc < n) & (hb == r0);
    const double M0 = (D0jm1 + cd) + (eq ? MATCH : MISMATCH);
    Xp[s] = dmax(M0 + ce, dmax(D0j + cd, IMP + cb));
    Yp[s] = dmax(M0 + cf, IMP + ca);
    rc[s / 4] |= ((rb >> 1) & 3u) << (2 * (s % 4) + 4);
    // (keep the set-up loads of four slots together: left alone, hipcc hoists all 4W of them and
    // the register allocation of the whole kernel pays for this prologue)
    if ((s % 4) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  double outX = Xp[W - 1];
  double leftX;
  {
    const double fill = dmax(emit00 + ce, dmax(IMP + cd, IMP + cb));
    leftX = wave_shr1(outX, fill);
    if (hl == 0) leftX = fill;                                 // lane 32 starts a pair too
  }
  double outZ = IMP;
  // certificate chain, one 32-bit half per pair; all ones ahead of the wavefronts (ltr_dp_kernel.hpp)
  uint64_t fmask = ~0ull;
  const uint64_t bitA = 1ull << (a.L - 1), bitB = 1ull << (32 + b.L - 1);
  uint64_t watch = bitA | (haveB ? bitB : 0);                  // last lanes of the pairs still running
  double certM = 0.0;
  double res_cap = 0.0;
  const int Tmax = max(a.T, b.T);                             // (b.T = 0 without a second pair)
  // haplotype rows: (base + t)[per-lane constant], pre-coded as emission-table block offsets
  const uint16_t* __restrict__ hs = hapc - 31;
  const uint32_t hoff = 32u - (uint32_t)hl;                    // (hs + t)[hoff] = row t + 1 - hl
  uint32_t h_next = hs[hoff];
  // left boundary of lanes 0 and 32: record i of the interleaved model table = X(i,0), Z(i,0) for
  // emit(hap[0], read[1]) = mismatch | match (HapAligner.cpp:274-280)
  const double2* __restrict__ colXZ = (const double2*)A.colXZ + e01;
  double2 b_next = colXZ[2 * 1];
  double kd = (double)(dd - (1 - hl) + j0);                    // band offset k of (my row, j0); -1 per step
  const double cabs_up = fabs((double)c32) * (1.0 + 0x1p-22);
  const double thr0 = -600.0 + 1e-6;
  const uint64_t lane32 = 1ull << 32;

  auto step = [&](auto fin_tag, const int t) __attribute__((always_inline)) {
    constexpr bool FIN = decltype(fin_tag)::value;
    const uint32_t h = h_next;
    const double2 bnd = b_next;
    h_next = (hs + (t + 1))[hoff];
    b_next = colXZ[2 * min(t + 2, A.table_len)];
    double mX = wave_shr1(outX, bnd.x);                        // X(i, j0-1)
    double mZ = wave_shr1(outZ, bnd.y);                        // Z(i, j0-1)
    if (lane == 32) { mX = bnd.x; mZ = bnd.y; }                // not lane 31's: the first column of pair B
    const double kcur = kd;
    kd = kcur - 1.0;
    // lanes with a row at this step: [t-(n-2), t] clipped to [0, L-1] in each half
    uint64_t active_mask = 0;
    if (!a.lost && t < a.T) {
      const int hi = min(t, a.L - 1), lo = max(t - (a.n - 2), 0);
      active_mask = (~0ull >> (63 - hi)) & (~0ull << lo);
    }
    if (!b.lost && t < b.T) {
      const int hi = min(t, b.L - 1), lo = max(t - (b.n - 2), 0);
      active_mask |= ((~0ull >> (63 - hi)) & (~0ull << lo)) << 32;
    }
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
      // (LTR_DUAL_QPF quads in flight ahead of the one being consumed)
#pragma unroll
      for (int q = 0; q < NQ && q <= LTR_DUAL_QPF; ++q) fetch_quad(q);
      certM = em[0] + diag;
      double Mv = certM;
#pragma unroll
      for (int s = 0; s < W; ++s) {
        double Mnext = 0.0;
        if ((s % 4) == 2 && (s / 4 + 1 + LTR_DUAL_QPF) < NQ) fetch_quad((s / 4 + 1 + LTR_DUAL_QPF) < NQ ? (s / 4 + 1 + LTR_DUAL_QPF) : 0);
        if (s + 1 < W) Mnext = em[(s + 1) < W ? (s + 1) : 0] + Xp[s];
        Iv = MATCH + Yp[s];
        Dv = zleft;
        // (FIN: the pair's result is best(n-1, m-1), :309 -- slot Wl-1 of my half's last lane)
        if (FIN) { const double best = dmax(Dv, dmax(Iv, Mv)); if (Wl == s + 1) res_cap = best; }
        if (SYM) {
          const double t2 = dmax(Dv, Iv) + cd;
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
    const uint64_t cert = __builtin_amdgcn_ballot_w64(certM >= __builtin_fma(__builtin_fabs(kcur), cabs_up, thr0)) & active_mask;
    fmask = cert | ((fmask << 1) & ~lane32);
    // a last lane has just finished a row nobody certified: that pair goes to the exact kernel
    const uint64_t miss = ~fmask & watch;
    if (miss != 0) {
      if (miss & bitA) { a.lost = true; *statA = kStatusUncertain; }
      if (miss & bitB) { b.lost = true; *statB = kStatusUncertain; }
      watch &= ~miss;
    }
  };

  // The step that finishes a pair (its last lane on row n-1) runs the FIN copy of the body.  The
  // copies are laid out one after the other -- plain loop, FIN step of the shorter pair, plain loop,
  // FIN step of the longer pair -- never as alternatives inside one loop: merging two copies at a
  // loop back-edge makes hipcc keep two register sets for the 2W carried values and shuffle them
  // every step.
  const int T1 = haveB ? min(a.T, b.T) : a.T, T2 = Tmax;
  auto fin_step = [&](const int t) __attribute__((always_inline)) {
    const bool finA = !a.lost && (t == a.T - 1), finB = !b.lost && (t == b.T - 1);
    step(BoolTag<true>{}, t);
    if (finA && !a.lost) *resA = lane_bcast(res_cap, a.L - 1);
    if (finB && !b.lost) *resB = lane_bcast(res_cap, 32 + b.L - 1);
    if (finA) watch &= ~bitA;                                    // finished: its chain bits decay from here on
    if (finB) watch &= ~bitB;
  };
  for (int t = 0; t < T1 - 1; ++t) {
    step(BoolTag<false>{}, t);
    if (a.lost && b.lost) return;
  }
  fin_step(T1 - 1);
  if (T2 > T1) {
    for (int t = T1; t < T2 - 1; ++t) {
      step(BoolTag<false>{}, t);
      if (a.lost && b.lost) return;
    }
    fin_step(T2 - 1);
  }
}

#ifndef LTR_DUAL_LB
#define LTR_DUAL_LB ((W <= 6) ? 5 : ((W <= 12) ? 4 : ((W <= 20) ? 3 : 2)))   // (2 waves per SIMD do not keep the VALU busy: see kDualWMax)
#endif
// Widest strip of the two-pairs-per-wave kernels (reads up to 32*kDualWMax+1 bases).  Measured on
// MI355X, config 3: up to W = 20 the body fits 168 VGPRs (3 waves per SIMD) and beats the 64-lane
// kernel by 5-8 %; wider strips need 2 waves per SIMD, which no longer hide the pair set-up and
// hand-off latencies, and only tie with it -- those reads stay on the 64-lane kernels.
// (kDualWMax: ltr_dp_types.h)

template <int W, bool SYM>
__global__ __launch_bounds__(64 * kBlockWaves, LTR_DUAL_LB) void ltr_dp_dual_kernel(KernelArgs A) {
  const int lane = threadIdx.x & 63;
  __shared__ __attribute__((aligned(16))) double s_emit[kEmitTabDoubles];
  for (int idx = threadIdx.x; idx < kEmitTabDoubles; idx += 64 * kBlockWaves) {
    const int half = idx >> 11, hcode = (idx >> 9) & 3, quad = (idx >> 1) & 255, k = 2 * half + (idx & 1);
    s_emit[idx] = (hcode == ((quad >> (2 * k)) & 3)) ? (double)A.mc.match : (double)A.mc.mismatch;
  }
  __syncthreads();
  const int n_pairs = A.n_pairs;
  for (;;) {
    // two pairs per pop (neighbours in the cost-sorted order: about the same length); all lanes
    // issue the add, see ltr_dp_kernel
    int q = (int)atomicAdd(A.queue, lane == 0 ? 2u : 0u);
    q = uni(q);
    if (q >= n_pairs) break;
    const bool haveB = (q + 1 < n_pairs);
    PairCtx P[2];
    int64_t out_idx[2];
    int pis[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pi = A.first_pair + min(q + k, n_pairs - 1);
      const PairDesc* pp = A.pairs + pi;
      pis[k] = pi;
      P[k].n = uni(pp->n); P[k].m = uni(pp->m); P[k].dd = P[k].n - P[k].m;
      out_idx[k] = uni64(pp->out_idx);
      P[k].hap = A.hap_bytes + uni64(pp->hap_off);
      P[k].hapc = A.hap_codes + uni64(pp->hap_off);
      P[k].read = A.read_bytes + uni64(pp->read_off);
      const int h0 = uni((int)P[k].hap[0]), r0 = uni((int)P[k].read[0]);
      P[k].emit00 = (h0 == r0) ? (double)A.mc.match : (double)A.mc.mismatch;
      P[k].e01 = (h0 == uni((int)P[k].read[1])) ? 1 : 0;
    }
    double r[2] = {0.0, 0.0};
    int st[2] = {kStatusOk, kStatusOk};
    dual_pairs<W, SYM>(A, P[0], P[1], haveB, lane, &r[0], &r[1], &st[0], &st[1], s_emit);
    auto finish = [&](const int k) __attribute__((always_inline)) {
      if (st[k] == kStatusUncertain) {
        push_redo(A, lane, pis[k], P[k].m);                      // could not prove "no row aborts": an exact kernel scores the pair
      } else if (lane == 0) {
        A.out_ll[out_idx[k]] = r[k];
      }
    };
    finish(0);
    if (haveB) finish(1);
  }
}

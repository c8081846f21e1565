// ltr_dp_wg.hpp -- ONE pair per WORKGROUP of NW wavefronts (included by ltr_gpu.hip after
// ltr_dp_kernel.hpp).  Replaces HapAligner::align_seq_to_hap (reference
// src/SeqAlignment/HapAligner.cpp:236-343) for the pairs the one-wave kernels serve badly:
//
//  * long reads (m > 1025): ltr_dp_kernel walks their column blocks one after the other on one
//    wavefront and parks every block's right boundary column in a global scratch strip (2 x 3 x n
//    doubles written and read back per block).  Here the NW column blocks of a pair run
//    CONCURRENTLY, one per wavefront of the workgroup, as one skewed pipeline 64*NW lanes long: the
//    boundary column of block w reaches block w+1 through a 128-row ring in LDS (32-byte records
//    {X, Z, certificate flag}), a few rows behind its producer.  Nothing but the pair's input bytes
//    and its 8-byte result touches HBM.
//  * small batches (a one-locus call: a few hundred pairs on 1024 SIMDs): with one wave per SIMD
//    nothing hides a global load issued one step ahead, so here NOTHING in the step loop waits on
//    global memory: haplotype rows and the first-column table are streamed in 64-row chunks, one
//    chunk ahead, into small LDS rings and read from there (NW = 1 is the latency variant of the
//    one-wave kernel).
//
// Same recurrence, same certificate, same emission table and the same bits as ltr_dp_kernel
// (EXACT = false, LUT = true); pairs the certificate cannot clear go to the exact kernels' lists.
// FULL = true is the EXACT form of the same pipeline (round 6; kModeThr of ltr_dp_kernel.hpp): every cell
// is compared with the exact threshold thr(k) -- the smallest double x with fl(x + pen(k)) >= -600,
// ltrp::build_threshold_table, a two-sided LDS table -- so "some cell of the row passes" IS the
// reference's "the row's band-penalised maximum is >= -600" (HapAligner.cpp:283, :297-306; x -> fl(x + pen)
// is monotone): a settled row nobody certifies aborts the pair (-700), nothing is handed from lane to
// lane but the certificate chain's bit.  13 FP64 operations per cell against 11; it replaces the
// running-maximum bodies of rounds 2-5 (14 operations, a third field in every ring record, 127 / 173
// spilled SGPRs).  Used (a) as ltr_dp_wgx_kernel, the redo kernel of the long pairs' exact lists, and
// (b) as the FIRST pass of the workgroup classes when the context has learnt that certificates fail here
// (ONT reads under the default model: every pair of BASELINE config 5 aborts -- ltr_plan_execute).
// Symmetric indel models only (b == d, f == g: the LongTR defaults and --alignment-params with
// f = g); other models and non-ACGT pairs stay on the one-wave kernels.
//
// Synchronisation between the waves of a pair is by progress words in LDS, no barriers inside a
// pair: LDS operations of one wavefront execute in issue order and LDS has no cache, so a record
// written before its progress word is visible to whoever reads that word.  A consumer polls only
// when the row it needs is not known to be there, and then waits for kWgLag rows more (one poll
// per ~kWgLag steps); a producer checks the consumer's progress only when it is about to lap the
// ring.  A wave that finds the pair uncertain raises a status word every other wave looks at
// when it polls and every 32 steps.

constexpr int kWgRing = 128;           // rows per boundary ring (records of 32 bytes)
constexpr int kWgLag = 8;              // extra rows a consumer waits for when it has to poll
constexpr int kWgBlock = 8;            // steps between two looks at the neighbours' progress (divides 64)
constexpr int kWgSpinLimit = 1 << 22;  // polls before a wave gives the pair up (seconds; never reached unless a partner wave died)
constexpr int kHapRing = 256;          // haplotype-row ring entries (stored twice: a 64-row window never wraps)

// A boundary record of the certificate kernels: {X, Z, -, F} (32 bytes, as since round 2; F = the certificate flag of the row).
// The threshold (FULL) kernels carry 16.5 KB of thresholds on top: their records are {X, Z} (16 bytes, one ds_read_b128) with the
// flags in an array of their own -- 20 bytes a row, so that eight waves still fit two workgroups into a CU's 160 KB.
template <bool FULL> struct WgRecT;
template <> struct __attribute__((aligned(16))) WgRecT<false> { double X, Z, unused; uint32_t F; uint32_t pad; };
template <> struct __attribute__((aligned(16))) WgRecT<true> { double X, Z; };

template <int NW, bool FULL = false>
struct WgShared {
  double emit[kEmitTabDoubles];                    // 32 KB emission table (ltr_dp_kernel.hpp)
  double pen[FULL ? kPenTabDoubles : 2];           // FULL: exact row-test thresholds, entry k + kPenHalf (16.5 KB)
  WgRecT<FULL> ring[NW][kWgRing];                  // INPUT ring of wave w: fed by wave w-1, or (w = 0) from the first-column table
  uint32_t flag[FULL ? NW : 1][FULL ? kWgRing : 1];   // FULL: ... certificate flag of the row (wave 0's stay 0: nothing to the left certifies)
  uint16_t hap[NW][2 * kHapRing];                  // haplotype rows as emission-table block offsets, per wave (own lag)
  uint32_t prod[NW];                               // prod[w]: highest row published in ring[w]
  uint32_t cons[NW];                               // cons[w]: rows <= cons[w] of ring[w] have been consumed
  uint32_t status;                                 // != 0: some wave found the pair uncertain -- everyone leaves
  int32_t pair_q;                                  // queue slot of the pair being scored
  double result;
};

// Progress words and ring entries other waves write: volatile accesses through explicit LDS pointers
// (a volatile access through a generic pointer stays a flat_load/flat_store, which also waits on vmcnt).
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;
__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) { return *(const volatile lds_u32_t*)p; }
__device__ __forceinline__ void lds_st(uint32_t* p, uint32_t v) { *(volatile lds_u32_t*)p = v; }

#ifndef LTR_WG_LB
#ifndef LTR_WG8_LB4_MAXW
#define LTR_WG8_LB4_MAXW 18
#endif
/* LDS (emission table + rings) admits 4 one-wave / 3 four-wave / 2 eight-wave workgroups per CU.  Two eight-wave workgroups are
   four waves per SIMD: only with <= 128 VGPRs -- at three per SIMD a CU holds ONE such workgroup, two waves per SIMD */
#define LTR_WG_LB ((NW == 1) ? 1 : ((NW == 8 && W <= LTR_WG8_LB4_MAXW) ? 4 : 3))
#endif
/* ... of the threshold (FULL) bodies: two quads of thresholds in flight on top of the certificate body's registers.  Their LDS
   (emission table, 16.5 KB of thresholds, rings, haplotype rows: 78 KB with eight waves) still admits two eight-wave workgroups. */
/* 1: the threshold bodies skip the test in blocks of steps whose strips lie outside the band |k| < k600 (wg_block, BAND) */
#ifndef LTR_WG_BAND
#define LTR_WG_BAND 1
#endif
#ifndef LTR_WGT_LB4_MAXW
#define LTR_WGT_LB4_MAXW 10
#endif
#ifndef LTR_WGT_LB
#define LTR_WGT_LB ((NW == 8 && W <= LTR_WGT_LB4_MAXW) ? 4 : 3)
#endif
/* ... of the exact list kernels (three strip widths in one kernel: the registers of the widest) */
#ifndef LTR_WGX_LB
#define LTR_WGX_LB ((NW == 8 && W2 <= LTR_WGT_LB4_MAXW) ? 4 : 3)
#endif

// The column block of wave `w` (lanes 0..L-1, strips of W columns) of one pair.  Returns != kWgDone when
// the wave leaves the pair early: !FULL -- the pair has to go to an exact kernel (found here or
// signalled by another wave); FULL -- a settled row has no cell that reaches -600: the pair aborts (-700).
// CAP (FULL only): this wave is the pair's last block and its last lane owns fewer than W real columns -- the
// lane's bit of the row mask is taken from the slots < Wl only (a scalar branch per slot, in this one wave).
enum { kWgDone = 0, kWgFound = 1, kWgStopped = 2 };          // wg_block: finished / found the pair uncertain (FULL: aborted) / told to stop

template <int W, int NW, bool FULL, bool SYM, bool CAP = false, bool BANDOK = true>
__device__ __forceinline__ int wg_block(const KernelArgs& A, const PairCtx& P, WgShared<NW, FULL>& S, const int lane, const int w) {
  static_assert(!CAP || FULL, "the slot capture belongs to the threshold test");
  const int n = P.n, m = P.m;
  const uint8_t* __restrict__ hap = P.hap;
  const uint8_t* __restrict__ read = P.read;
  const double ca = A.mc.a, cc = A.mc.c, cd = A.mc.d, ce = A.mc.e, cf = A.mc.f, cg = A.mc.g;
  const double cb = A.mc.b;
  const double MATCH = A.mc.match, MISMATCH = A.mc.mismatch;
  const float c32 = A.mc.c;
  const double IMP = kImp;
  const double* __restrict__ lpc = A.lpc;
  const double* emit_tab = S.emit;
  // thresholds are fetched this many quads of slots ahead of their use (the emissions: two).  The eight-wave bodies built for four
  // waves per SIMD (128 registers) take one: eight live registers less, and three more waves to hide an LDS round trip behind
  constexpr int kPenAhead = (NW == 8 && W <= LTR_WGT_LB4_MAXW) ? 1 : 2;
  constexpr int kEmAhead = (FULL && NW == 8 && W <= LTR_WGT_LB4_MAXW) ? 1 : 2;     // ... and so are their emissions
  // Band skipping (see `step`) for the eight-wave bodies of strips up to LTR_WGT_LB4_MAXW columns only.  Measured on MI355X, 1536
  // pairs of one length through the threshold first pass (profiles/r06/band_ab.log): 5 kb on eight waves x 10 columns 22.3 -> 21.1 ms,
  // BASELINE config 5 3.33 -> 3.06 ms per pass; the wider bodies lose -- two more copies of the step in one function cost them
  // spills inside the loops (7.4 kb on 16 columns: 59.0 -> 67.2 ms) -- and keep the one form.
  constexpr bool kBandSkip = FULL && BANDOK && (LTR_WG_BAND != 0) && NW == 8 && W <= LTR_WGT_LB4_MAXW;

  const bool first = (w == 0);
  const bool final_block = (w == P.ncb - 1);
  const int L = final_block ? P.Ll : P.Lb;                     // active lanes
  const int Wl = final_block ? P.Wl : W;                       // real columns of the LAST lane
  const bool is_last_lane = (lane == L - 1);
  // rows this block can already decide (see column_block in ltr_dp_kernel.hpp)
  const int i_dec = uni(final_block ? 0x7fffffff : ((w + 1) * P.Lb * W + 1 + P.dd - P.k600));
  const int j0 = 1 + (w * P.Lb + lane) * W;                    // first column of my strip

  // ---- haplotype rows: 64-row chunks -> my LDS ring, one chunk ahead of use -------------------
  // entry (row & 255) and its copy 256 entries later; at step t lane l reads row t+2-l.
  uint16_t* hring = S.hap[w];
  const uint16_t* __restrict__ hapc = P.hapc;
  auto hap_put = [&](const int chunk, const uint32_t v) __attribute__((always_inline)) {
    const int e = ((chunk * 64) & (kHapRing - 1)) + lane;
    hring[e] = (uint16_t)v; hring[e + kHapRing] = (uint16_t)v;
  };
  hap_put(0, hapc[lane]);
  hap_put(1, hapc[64 + lane]);
  uint32_t hchunk = hapc[128 + lane];                          // chunk 2, written at step 64
  // my read pointer: hp[(t+2) & 255] = row t+2-lane (entries 193..511: the copy keeps the window linear)
  const volatile lds_u16_t* hp = (const volatile lds_u16_t*)(hring + (kHapRing - lane));

  // ---- first column (HapAligner.cpp:274-280) for wave 0: table records -> my input ring ------
  typedef WgRecT<FULL> WgRec;
  WgRec* iring = S.ring[w];
  const uint32_t* iflag = S.flag[FULL ? w : 0];
  auto flag_of = [&](const int idx) __attribute__((always_inline)) -> uint32_t {
    if constexpr (FULL) return iflag[idx]; else return iring[idx].F;
  };
  const double2* __restrict__ colXZ = (const double2*)A.colXZ + P.e01;
  auto col_load = [&](const int chunk) __attribute__((always_inline)) {
    return colXZ[2 * min(chunk * 64 + lane, A.table_len)];
  };
  auto col_put = [&](const int chunk, const double2 v) __attribute__((always_inline)) {
    WgRec* r = iring + ((chunk * 64 + lane) & (kWgRing - 1));
    r->X = v.x; r->Z = v.y;                                    // (the flags of wave 0's ring stay 0: nothing to the left certifies)
  };
  double2 cchunk = make_double2(0.0, 0.0);
  if (first) {
    col_put(0, col_load(0));
    col_put(1, col_load(1));
    cchunk = col_load(2);
  }

  // ---- row 0 (HapAligner.cpp:263-272) for my columns -> X(0,j), Y(0,j) ----------------------
  double Xp[W], Yp[W];
  constexpr int NQ = (W + 3) / 4;
  // (the threshold bodies: two quads' row offsets to a register, 16 bits each -- the add takes its half by SDWA; the loop-invariant
  // offsets were the first values hipcc spilled at 128 registers, reloaded from scratch in every step)
  constexpr bool kPackRc = FULL;
  constexpr int NRC = kPackRc ? (NQ + 1) / 2 : NQ;
  uint32_t rc[NRC];
#pragma unroll
  for (int q = 0; q < NRC; ++q) rc[q] = 0;
  const uint32_t r0 = (uint32_t)uni((int)read[0]);
  auto row0 = [&](const int jc, double& M0, double& D0j) __attribute__((always_inline)) {
    const double lp1 = lpc[max(jc - 1, 0)], lp = lpc[jc];
    const uint32_t hb = (uint32_t)hap[min(jc, n - 1)];
    const double D0jm1 = (jc == 1) ? IMP : (cg + lp1);         // deletion_matrix[j-1]
    D0j = cg + lp;                                             // deletion_matrix[j] = g + left_prob
    const bool eq = (jc < n) & (hb == r0);                     // the reference indexes the haplotype with the READ index here
    M0 = (D0jm1 + cd) + (eq ? MATCH : MISMATCH);
  };
#pragma unroll
  for (int s = 0; s < W; ++s) {
    const int jc = min(j0 + s, m - 1);                         // inactive lanes / the last lane's slack: clamp the loads
    double M0, D0j;
    row0(jc, M0, D0j);
    Xp[s] = dmax(M0 + ce, dmax(D0j + cd, IMP + cb));
    Yp[s] = dmax(M0 + cf, IMP + ca);
    if (kPackRc) rc[(s / 8) < NRC ? (s / 8) : 0] |= (((uint32_t)read[jc] >> 1) & 3u) << (2 * (s % 4) + 4 + 16 * ((s / 4) & 1));
    else rc[(s / 4) < NRC ? (s / 4) : 0] |= (((uint32_t)read[jc] >> 1) & 3u) << (2 * (s % 4) + 4);
    if ((s % 4) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  // X(0, j0-1) of lane 0: column 0 for the first block, else the last column of the block to the left
  // (a row-0 cell: computed, not communicated)
  double outX = Xp[W - 1];
  double leftX;
  {
    double fill;
    if (first) fill = dmax(P.emit00 + ce, dmax(IMP + cd, IMP + cb));
    else {
      double M0, D0j;
      row0(w * P.Lb * W, M0, D0j);
      fill = dmax(M0 + ce, dmax(D0j + cd, IMP + cb));
    }
    leftX = wave_shr1(outX, fill);
  }

  // ---- progress bookkeeping ------------------------------------------------------------------
  int avail = first ? 0x7fffffff : 0;                          // rows of my input ring known to be published
  int consd = 0;                                               // rows of my OUTPUT ring known to be consumed
  WgRec* oring = S.ring[(w + 1 < NW) ? w + 1 : 0];
  uint32_t* oflag = S.flag[FULL ? ((w + 1 < NW) ? w + 1 : 0) : 0];
  uint32_t* const my_prod = &S.prod[w];
  uint32_t* const my_cons = &S.cons[w];
  uint32_t* const out_prod = &S.prod[(w + 1 < NW) ? w + 1 : 0];
  uint32_t* const out_cons = &S.cons[(w + 1 < NW) ? w + 1 : 0];
  bool stop = false;
  // rows <= r of my input ring must be there before I read row r (rows < r are consumed)
  // (my_cons is NOT advanced here: the rows up to r are only read by the steps that follow -- block_prologue
  // publishes what earlier steps have fetched.  The producer may run up to kWgRing rows ahead of that, which is
  // always >= the kWgBlock + 1 + kWgLag rows waited for: no deadlock.)
  auto need_rows = [&](const int r) __attribute__((always_inline)) {
    if (r > avail) {
      const int want = min(r + kWgLag, n - 1);
      for (int spins = 0;; ++spins) {
        avail = uni((int)lds_ld(my_prod));
        if (uni((int)lds_ld(&S.status)) != 0 || spins > kWgSpinLimit) { stop = true; break; }   // (every spin is bounded: the pair would go to the exact kernel)
        if (avail >= want) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
  };
  // the slot of row r in my output ring is free once the consumer is past row r - kWgRing
  auto need_space = [&](const int r) __attribute__((always_inline)) {
    if (r - kWgRing > consd) {
      for (int spins = 0;; ++spins) {
        consd = uni((int)lds_ld(out_cons));
        if (uni((int)lds_ld(&S.status)) != 0 || spins > kWgSpinLimit) { stop = true; break; }
        if (consd >= r - kWgRing) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
  };

  double outZ = IMP;
  // FULL: the thresholds of my strip's W cells come from S.pen at k0 + s, k0 = dd - i + j0 = kq0 - t
  const int kq0 = P.dd + lane + j0 - 1;
  // ... of the whole wave: lane 0's first slot .. the last lane's last slot (k = kq - t); a block of steps [t0, tend) meets the band
  // |k| < k600 iff its largest k is above -k600 and its smallest below k600
  const int kq_lo = P.dd + w * P.Lb * W, kq_hi = kq_lo + (L - 1) * (W + 1) + (W - 1);
  auto in_band = [&](const int t0, const int tend) __attribute__((always_inline)) {
    return (kq_hi - t0 > -P.k600) && (kq_lo - (tend - 1) < P.k600);
  };
  uint64_t fmask = ~0ull;                                      // certificate chain (SGPRs), all ones ahead of the wavefront
  const uint64_t lastbit = 1ull << (L - 1);
  const uint64_t watch = final_block ? lastbit : 0;
  double certM = 0.0;
  double res_cap = 0.0;
  const int T = (n - 1) + (L - 1);
  // per-step inputs, fetched one step ahead -- all from LDS
  need_rows(1);
  if (stop) return kWgStopped;
  asm volatile("" ::: "memory");
  uint32_t h_next = hp[1];                                     // row 1 - lane   ((t+2) with t = -1)
  double bX_next, bZ_next; uint32_t bF_next = 0;
  { const WgRec* r = iring + 1; bX_next = r->X; bZ_next = r->Z; bF_next = flag_of(1); }
  double kd = (double)(P.dd - (1 - lane) + j0);                // band offset k of (row, j0); -1 per step
  const double cabs_up = fabs((double)c32) * (1.0 + 0x1p-22);
  const double thr0 = -600.0 + 1e-6;

  // BAND (FULL only): some cell of the wave's strips at this step lies inside the band |k| < k600 -- outside it every threshold is
  // +inf (a penalty below -600 whatever the cell holds), no compare can pass, and the step is the plain recurrence (11 operations,
  // no threshold reads): a 5-kb pair's band is ~1200 of its 5000 columns, so a wave tests in about a third of its steps.
  auto step = [&](auto fin_tag, auto band_tag, const int t) __attribute__((always_inline)) {
    constexpr bool FIN = decltype(fin_tag)::value;
    constexpr bool TEST = FULL && decltype(band_tag)::value;
    const uint32_t h = h_next;
    const double bX = bX_next, bZ = bZ_next;
    const uint32_t bF = bF_next;
    {
      // next step's inputs (the block prologue made sure the rows are there)
      const int ib = min(t + 2, n - 1);                        // lane 0's row at the next step
      asm volatile("" ::: "memory");
      h_next = hp[(t + 2) & (kHapRing - 1)];
      const WgRec* r = iring + (ib & (kWgRing - 1));
      bX_next = r->X; bZ_next = r->Z;
      bF_next = flag_of(ib & (kWgRing - 1));
    }
    const int il = t + 2 - L;                                  // the row my last lane is on (>= 1 from t = L-1)

    const double mX = wave_shr1(outX, bX);                     // X(i, j0-1)
    const double mZ = wave_shr1(outZ, bZ);                     // Z(i, j0-1)
    const double kcur = kd;
    if (!FULL) kd = kcur - 1.0;
    const int a_hi = min(t, L - 1), a_lo = max(t - (n - 2), 0);
    const uint64_t active_mask = (~0ull >> (63 - a_hi)) & (~0ull << a_lo);
    const bool active = __builtin_amdgcn_inverse_ballot_w64(active_mask);
    uint64_t bprev = 0;
    uint64_t okm = 0, okc = 0;                                 // FULL: lanes with a cell that reaches -600 with its penalty (scalar masks); okc: ... among the slots < Wl
    if (active) {
      double diag = leftX;                                     // X(i-1, j0-1)
      leftX = mX;
      double zleft = mZ;
      double Iv = 0.0, Dv = 0.0;
      double em[W];
      auto fetch_quad = [&](const int q) __attribute__((always_inline)) {
        const uint32_t rq = kPackRc ? ((rc[(q / 2) < NRC ? (q / 2) : 0] >> (16 * (q & 1))) & 0xffffu) : rc[q < NRC ? q : 0];
        const double2* row = (const double2*)((const char*)emit_tab + (h + rq));
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
      if (NQ > 1 && kEmAhead > 1) fetch_quad(1 < NQ ? 1 : 0);
      // FULL: the W thresholds of this row's cells, one clamped base + constant offsets
      // (fetched four slots at a time, two quads ahead of their use, like the emissions)
      double pn[TEST ? W : 1];
      const int kc = min(max(kq0 - t, -kPenHalf), kPenHalf - W);
      const double* pp = S.pen + (TEST ? (kc + kPenHalf) : 0);
      auto fetch_pen = [&](const int q) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 4 * q; k < 4 * q + 4; ++k) if (k < W) pn[k < (TEST ? W : 1) ? k : 0] = pp[k];
      };
      if (TEST) {
        fetch_pen(0);
        if (NQ > 1 && kPenAhead > 1) fetch_pen(1);
      }
      certM = em[0] + diag;                                    // match_matrix[i][j], :287-289
      double Mv = certM;
#pragma unroll
      for (int s = 0; s < W; ++s) {
        double Mnext = 0.0;
        if ((s % 4) == 2 && (s / 4 + kEmAhead) < NQ) fetch_quad((s / 4 + kEmAhead) < NQ ? (s / 4 + kEmAhead) : 0);
        if (TEST && (s % 4) == 2 && (s / 4 + kPenAhead) < NQ) fetch_pen(s / 4 + kPenAhead);
        if (s + 1 < W) Mnext = em[(s + 1) < W ? (s + 1) : 0] + Xp[s];
        Iv = MATCH + Yp[s];                                    // insertion_matrix[i][j], :291-292
        Dv = zleft;                                            // deletion_matrix[i][j], :294-295
        const double di = dmax(Dv, Iv);
        double best = 0.0;
        if (TEST || FIN) best = dmax(di, Mv);                  // :297 (max is exact: any association gives the same bits)
        if (FIN) { if (Wl == s + 1) res_cap = best; }          // :309, the pair's result (the peeled final step)
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
        if (TEST) {
          // :297-298 as a compare: fl(best + pen(k)) >= -600  <=>  best >= thr(k)   (the OR of slot s's mask is issued one slot
          // later: a scalar instruction right behind the v_cmp it reads stalls the wave)
          okm |= bprev;
          bprev = __builtin_amdgcn_ballot_w64(best >= pn[s < (TEST ? W : 1) ? s : 0]);
          // the last lane's slack columns (final block, Wl < W) are no cells of the reference: its bit counts the slots < Wl
          // (a scalar select per slot -- s_cmp / s_cselect_b64 --, no branch: as a branch hipcc kept one condition per slot in SGPRs)
          if (CAP && s + 1 < W) okc = (Wl == s + 1) ? (okm | bprev) : okc;
        }
        if (s + 1 < W) Mv = Mnext;
      }
      outX = Xp[W - 1];
      outZ = zleft;
      if (TEST) okm |= bprev;
      if (!final_block && is_last_lane) {                      // my right boundary column, row il -> the next wave's ring
        WgRec* r = oring + (il & (kWgRing - 1));
        r->X = outX; r->Z = outZ;
      }
    }
    uint64_t cert;
    if (FULL && !TEST) cert = 0;                                 // (outside the band: no cell of these strips can reach -600)
    else if (FULL) {
      // the masks were formed inside the divergent region: uniform there, but a per-lane value behind it (the lanes that skipped
      // the region hold 0) -- take them back from a lane that was in it
      okm = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(okm >> 32), a_lo) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)okm, a_lo);
      if (CAP) {
        okc = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(okc >> 32), a_lo) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)okc, a_lo);
        okm = (okm & ~lastbit) | (okc & lastbit);
      }
      cert = okm & active_mask;
    } else {
      // certificate (see column_block): one cell per lane and row, chain in SGPRs
      cert = __builtin_amdgcn_ballot_w64(certM >= __builtin_fma(__builtin_fabs(kcur), cabs_up, thr0)) & active_mask;
    }
    uint64_t in0 = 0;
    if (!first) in0 = (uint64_t)(uni((int)bF) & 1);
    fmask = cert | (fmask << 1) | in0;
    if ((~fmask & watch) != 0) return true;                    // a settled row nobody certified (FULL: no cell of it reaches -600 -- :300-306)
    if (!final_block && il >= 1) {                             // the last lane has just finished row il: publish it with its flag
      const bool row_ok = (fmask & lastbit) != 0;
      if (is_last_lane) { if constexpr (FULL) oflag[il & (kWgRing - 1)] = row_ok ? 1u : 0u; else oring[il & (kWgRing - 1)].F = row_ok ? 1u : 0u; }
      asm volatile("" ::: "memory");
      lds_st(out_prod, (uint32_t)il);
      if (!row_ok && il <= i_dec) return true;
    }
    return false;
  };
  // Steps run in blocks of kWgBlock: everything that is not the recurrence -- the streaming of haplotype rows
  // and first-column records (every 64 steps), the progress hand-shakes with the neighbouring waves, the look
  // at the status word -- happens once per block, for the whole block; the steps themselves carry no checks.
  auto block_prologue = [&](const int t0, const int tend, const bool stream) __attribute__((always_inline)) {
    if (stream && (t0 & 63) == 0 && t0 > 0) {                  // (blocks start at multiples of kWgBlock, which divides 64)
      const int c = (t0 >> 6) + 1;                             // chunk needed from step 64c - 2 on
      hap_put(c, hchunk);
      hchunk = hapc[(c + 1) * 64 + lane];
      if (first) { col_put(c, cchunk); cchunk = col_load(c + 1); }
    }
    if (NW > 1) {
      if (!first) lds_st(my_cons, (uint32_t)t0);               // rows <= t0 were used in earlier steps
      if ((t0 & 31) == 0 && uni((int)lds_ld(&S.status)) != 0) { stop = true; return; }
      need_rows(min(tend + 1, n - 1));                         // step t fetches row t+2 for step t+1
      if (stop) return;
      const int il_max = tend + 1 - L;                         // last row my last lane finishes in this block
      if (!final_block && il_max >= 1) need_space(il_max);
    }
  };
  for (int t0 = 0; t0 < T - 1; t0 += kWgBlock) {
    const int tend = min(t0 + kWgBlock, T - 1);
    block_prologue(t0, tend, true);
    if (stop) return kWgStopped;
#ifdef LTR_WG_STRESS
    // stress builds only (tests/manual/gpu_wg_ring_stress.sh): odd waves fall behind their producers, even ones behind
    // their consumers, by up to 127 x 64 clocks per block -- every ring runs full, every poll path is taken
    if (((w + (t0 >> 9)) & 1) != 0) __builtin_amdgcn_s_sleep(LTR_WG_STRESS);
#endif
    if (kBandSkip && !in_band(t0, tend)) {
      for (int t = t0; t < tend; ++t)
        if (step(BoolTag<false>{}, BoolTag<false>{}, t)) return stop ? kWgStopped : kWgFound;
    } else {
      for (int t = t0; t < tend; ++t)
        if (step(BoolTag<false>{}, BoolTag<true>{}, t)) return stop ? kWgStopped : kWgFound;
    }
  }
  {
    block_prologue(T - 1, T, false);                           // the last step: its row still needs room in the output ring
    if (stop) return kWgStopped;
    if (final_block) { if (step(BoolTag<true>{}, BoolTag<true>{}, T - 1)) return stop ? kWgStopped : kWgFound; }
    else if (step(BoolTag<false>{}, BoolTag<true>{}, T - 1)) return stop ? kWgStopped : kWgFound;
  }
  if (final_block) {
    const double r = lane_bcast(res_cap, L - 1);
    if (lane == 0) S.result = r;
  }
  return kWgDone;
}

// Column-block geometry of a pair for NW waves of strips W columns wide (balanced over the waves).
template <int W, int NW>
__device__ __forceinline__ void wg_geometry(PairCtx& P) {
  const int C = P.m - 1;
  P.Lb = min((C + W * NW - 1) / (W * NW), 64);                 // lanes of every block but the last
  P.ncb = (C + P.Lb * W - 1) / (P.Lb * W);                     // blocks actually needed (<= NW for a read that fits 64*W*NW columns)
  const int Cl = C - (P.ncb - 1) * P.Lb * W;
  P.Ll = (Cl + W - 1) / W;
  P.Wl = Cl - (P.Ll - 1) * W;
}

// This wave's block of a pair whose geometry is set: the threshold bodies take the slot capture only where it is needed
// (wave-uniform: the pair's last block when its last lane has slack columns).
template <int W, int NW, bool FULL, bool SYM, bool BANDOK = true>
__device__ __forceinline__ int wg_run_block(const KernelArgs& A, const PairCtx& P, WgShared<NW, FULL>& S, const int lane, const int wave) {
  if (wave >= P.ncb) return (int)kWgDone;
  if constexpr (FULL && W > 1) {
    if (wave == P.ncb - 1 && P.Wl != W) return wg_block<W, NW, true, SYM, true, BANDOK>(A, P, S, lane, wave);
  }
  return wg_block<W, NW, FULL, SYM, false, BANDOK>(A, P, S, lane, wave);
}

// The pair loop of a workgroup: pop, shortcuts, the NW column blocks, result.  `blocks(P, lane, wave)`
// runs this wave's block with the strip width of the kernel (certificate kernels) or of the pair
// (exact kernel) and returns a kWg* code.
template <int NW, bool FULL, class Blocks>
__device__ __forceinline__ void wg_pair_loop(const KernelArgs& A, WgShared<NW, FULL>& S, const int lane, const int wave, Blocks blocks) {
  const double IMP = kImp;
  int n_pairs = A.n_pairs;
  if (A.n_pairs_dev) n_pairs = uni((int)*A.n_pairs_dev);      // (exact kernels: the list's length lives on the device)
  for (;;) {
    // (the previous pair's closing barrier is behind every wave: the words below are free)
    // Wave 0 pops the queue -- every lane issues the add (lane 0 adds 1, the rest 0) under wave-uniform
    // control flow.  NOT `if (threadIdx.x == 0) S.pair_q = atomicAdd(..)`: around a divergent pop at the
    // head of this loop hipcc (ROCm 7.2) structurizes the lanes != 0 into an inner loop of their own that
    // re-reads the old slot and never lets lane 0 pop again (seen on gfx950: every workgroup kernel hung).
    if (wave == 0) {
      int q0 = (int)atomicAdd(A.queue, lane == 0 ? 1u : 0u);
      q0 = uni(q0);
      if (lane == 0) { S.pair_q = q0; S.status = 0; }
      if (lane < NW) { S.prod[lane] = 0; S.cons[lane] = 0; }
    }
    __syncthreads();
    const int q = uni((int)lds_ld((const uint32_t*)&S.pair_q));
    if (q >= n_pairs) break;
    int pi = A.first_pair + q;
    if (A.index) pi = uni(A.index[pi]);
    const PairDesc* pp = A.pairs + pi;
    const int n = uni(pp->n), m = uni(pp->m), hfl = uni(pp->hap_full_len);
    bool have_result = true;
    double r = 0.0;
    const bool skip = FULL && (m - 1 < A.c_lo || m - 1 > A.c_hi);   // (a list shared by two exact launches: the other one's pair; class launches pass the whole range)
    if (skip) {}
    else if (hfl <= 60) r = IMP;                               // HapAligner.cpp:241-244
    else if (abs(n - m) > 600) r = -700.0;                     // :249-252
    else {
      PairCtx P;
      P.hap = A.hap_bytes + uni64(pp->hap_off);
      P.hapc = A.hap_codes + uni64(pp->hap_off);
      P.read = A.read_bytes + uni64(pp->read_off);
      P.n = n; P.m = m; P.dd = n - m;
      const int h0 = uni((int)P.hap[0]), r0 = uni((int)P.read[0]);
      P.emit00 = (h0 == r0) ? (double)A.mc.match : (double)A.mc.mismatch;   // match_matrix[0], :265
      if (m == 1) r = (n == 1) ? dmax(IMP, dmax(IMP, P.emit00)) : -700.0;
      else if (n == 1) {
        // single row: the result is row 0's last cell (HapAligner.cpp:267-272, :309)
        const double cg = A.mc.g, cd = A.mc.d;
        const int jc = m - 1;
        const double D0jm1 = (jc == 1) ? IMP : (cg + A.lpc[jc - 1]);
        const double D0j = cg + A.lpc[jc];
        const bool eq = (jc < n) && ((int)P.hap[0] == r0);
        const double M0 = (D0jm1 + cd) + (eq ? (double)A.mc.match : (double)A.mc.mismatch);
        r = dmax(D0j, dmax(IMP, M0));
      } else {
        P.e01 = (h0 == uni((int)P.read[1])) ? 1 : 0;           // emission of the whole first column, :276
        {
          const float cabs = fabsf(A.mc.c);
          P.k600 = (cabs * 1.0e9f > 600.0f) ? ((int)(600.0f / cabs) + 2) : 0x3fffffff;
        }
        have_result = false;
        const int code = blocks(P, lane, wave);
        // found: 1 (certificate: uncertain; FULL: abort); a wave that stopped without anyone having
        // found anything ran out of polls (never, unless a partner wave died): 2 = the pair failed
        if (code == kWgFound && lane == 0) lds_st(&S.status, 1u);
        if (code == kWgStopped && lane == 0 && lds_ld(&S.status) == 0) lds_st(&S.status, 2u);
      }
    }
    __syncthreads();                                           // every wave is done with the pair (rings, progress words, result)
    if (wave == 0 && !skip) {
      const uint32_t st = (uint32_t)uni((int)lds_ld(&S.status));
      if (have_result) { if (lane == 0) A.out_ll[pp->out_idx] = r; }
      else if (FULL) {
        // :300-306 abort -> -700; (a failed hand-shake leaves a NaN: loud, never a plausible score)
        if (lane == 0) A.out_ll[pp->out_idx] = (st == 0) ? S.result : ((st == 1) ? -700.0 : __longlong_as_double(0x7ff8000000000000ll));
      } else if (st != 0) {
        push_redo(A, lane, pi, m);                             // could not prove "no row aborts": an exact kernel scores it
      } else if (lane == 0) A.out_ll[pp->out_idx] = S.result;
      // (a class launch, not a list: what the context learns the first pass of its next plans from -- ltr_plan_execute)
      if (!have_result && !A.index && lane == 0) {
        atomicAdd(A.xcount + kWgStatOff + 1, 1u);
        if (st != 0) atomicAdd(A.xcount + kWgStatOff, 1u);
      }
    }
  }
}

template <int NW, bool FULL>
__device__ __forceinline__ void wg_init_shared(const KernelArgs& A, WgShared<NW, FULL>& S) {
  for (int idx = threadIdx.x; idx < kEmitTabDoubles; idx += 64 * NW) {
    const int half = idx >> 11, hcode = (idx >> 9) & 3, quad = (idx >> 1) & 255, k = 2 * half + (idx & 1);
    S.emit[idx] = (hcode == ((quad >> (2 * k)) & 3)) ? (double)A.mc.match : (double)A.mc.mismatch;
  }
  for (int idx = threadIdx.x; idx < NW * kWgRing; idx += 64 * NW) { if constexpr (FULL) (&S.flag[0][0])[idx] = 0; else (&S.ring[0][0] + idx)->F = 0; }
  if (FULL) {
    // (built on the host once per parameter set -- a bisection per entry, ltrp::build_threshold_table -- and copied here)
    for (int idx = threadIdx.x; idx < kPenTabDoubles; idx += 64 * NW) S.pen[idx] = A.thr_tab[idx];
  }
  __syncthreads();
}

// FULL = false: the certificate kernel of launch class (NW, W).  FULL = true: the same class scored exactly in one pass
// (even strip widths only: the threshold reads stride W + 1 doubles from lane to lane and odd widths conflict in LDS -- an odd
// class is launched with the next even width, the geometry follows the kernel's W).
template <int W, int NW, bool SYM, bool FULL = false>
__global__ __launch_bounds__(64 * NW, FULL ? LTR_WGT_LB : LTR_WG_LB) void ltr_dp_wg_kernel(KernelArgs A) {
  __shared__ WgShared<NW, FULL> S;
  const int lane = threadIdx.x & 63;
  const int wave = (NW == 1) ? 0 : uni((int)(threadIdx.x >> 6));
  wg_init_shared<NW, FULL>(A, S);
  wg_pair_loop<NW, FULL>(A, S, lane, wave, [&](PairCtx& P, const int ln, const int wv) __attribute__((always_inline)) {
    wg_geometry<W, NW>(P);
    if (P.ncb > NW) return (int)(FULL ? kWgStopped : kWgFound);  // (never for a correctly binned pair: the exact kernels take any length)
    return wg_run_block<W, NW, FULL, SYM>(A, P, S, ln, wv);
  });
}

// The exact redo kernel for long pairs (symmetric models, ACGT pairs): every pair of its list gets the
// narrowest of three (even) strip widths that covers its read with NW waves.
template <int NW, int W0, int W1, int W2>
__global__ __launch_bounds__(64 * NW, LTR_WGX_LB) void ltr_dp_wgx_kernel(KernelArgs A) {
  __shared__ WgShared<NW, true> S;
  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  wg_init_shared<NW, true>(A, S);
  // (no band skipping here: every instance holds two or three strip widths in one function and the copies cost it spills inside
  // the step loops; the eight-wave list's narrow pairs go through the first-pass kernel of 10 columns instead -- ltr_plan_execute)
  constexpr bool kBand = false;
  wg_pair_loop<NW, true>(A, S, lane, wave, [&](PairCtx& P, const int ln, const int wv) __attribute__((always_inline)) {
    const int C = P.m - 1;
    if (C <= 64 * NW * W0) {
      wg_geometry<W0, NW>(P);
      return wg_run_block<W0, NW, true, true, kBand>(A, P, S, ln, wv);
    }
    if constexpr (W1 != W2) {                                  // (W1 == W2: a kernel of two strip widths)
      if (C <= 64 * NW * W1) {
        wg_geometry<W1, NW>(P);
        return wg_run_block<W1, NW, true, true, kBand>(A, P, S, ln, wv);
      }
    }
    wg_geometry<W2, NW>(P);
    if (P.ncb > NW) return (int)kWgStopped;                    // (the plan never lists such a pair here)
    return wg_run_block<W2, NW, true, true, kBand>(A, P, S, ln, wv);
  });
}

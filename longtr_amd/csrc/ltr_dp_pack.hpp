// ltr_dp_pack.hpp -- SEVERAL pairs per wavefront (included by ltr_k_pack.hip after ltr_dp_kernel.hpp).
//
// Replaces HapAligner::align_seq_to_hap (reference src/SeqAlignment/HapAligner.cpp:236-343) for the pairs
// real tandem-repeat catalogues are made of: reads of a few dozen to a few hundred bases after trimming.
// Same recurrence, certificate and emission table as ltr_dp_kernel (EXACT = false, LUT = true) -- same bits.
//
// Geometry.  The 64 lanes are cut into 64 / LP segments of LP = 2, 4, 8, 16 or 32 lanes (a launch parameter);
// every segment scores its own pair, lane hl of a segment owning W consecutive read columns, the haplotype rows
// skewed through the segment's lanes by one row per lane.  Why: the skew costs LP - 1 steps of fill and drain
// per pair -- at 32 lanes and 40 haplotype rows that is 44 % of the steps -- and a wave of 32-lane halves idles
// every lane beyond ceil((m - 1) / W).  Narrow segments with wide strips cut both: a 40 x 40 pair on 4 lanes of
// 10 columns runs 42 steps on 4 of 4 lanes instead of 58 steps on 20 of 32.
//
// Control.  What is per pair is PER LANE here (n, m, pointers, geometry: vector registers, loaded by the lanes of
// the segment); what is per step stays on the scalar unit as 64-bit lane masks whatever LP is: the activity mask
// comes out of one vector compare, the certificate chain is the shifted mask of ltr_dp_kernel cut at the segment
// heads (`& ~head_mask`), "a last lane finished an uncertified row" and "a last lane finishes its pair in this
// step" are mask tests.  The DPP wave_shr:1 hand-off crosses segment boundaries; the head lane of every segment
// takes the first-column table instead (four v_cndmask per step under a constant SGPR mask).
// The segments run in lock step until the longest pair of the wave ends; pairs are popped 64 / LP at a time,
// neighbours in the cost-sorted launch order.
//
// One launch per strip width.  LP is read PER POPPED GROUP, not per launch: a launch scores up to five ranges of the
// sorted pair list -- the pairs that want 32, 16, 8, 4 and 2 lanes of W columns each, in that order: groups of wide
// segments last longest, they go first -- through ONE queue of groups (KernelArgs::pk_*).  A catalogue of short repeats
// used to run 33 packed launches (every LP x W with pairs), each with its own ramp and a tail as long as its longest
// group; it runs one per W now, each long enough to hide both.

#include "ltr_dp_redo.hpp"

// (Loading the next group's descriptors while the current group is scored was measured on MI355X: no gain, +10 live registers.)

// maximum over the 64 lanes: xor 1, xor 2 inside the quads, half-row and row mirrors, then one lane of each row of 16
__device__ __forceinline__ int wave_max_i(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1 /*quad_perm [1,0,3,2]*/, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E /*quad_perm [2,3,0,1]*/, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141 /*row_half_mirror*/, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140 /*row_mirror*/, 0xf, 0xf, false));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
#ifndef LTR_PACK_LB
// waves per SIMD the register allocator must leave room for
#define LTR_PACK_LB ((W <= 6) ? 5 : ((W <= 12) ? 4 : ((W <= 20) ? 3 : 2)))
#endif

// The ranges of one strip width (KernelArgs::pk_* or one PackTable of a multi-width launch), wave-uniform.
struct PackRanges { int shift[5], first[5], end[5], grp_end[5]; };

// Every group of one strip width's ranges, popped from `queue` until it is empty.
// INL (the plan kernel, ltr_dp_plan.hpp): a pair the certificate could not clear is NOTED in the wave's list `note` (LDS,
// kRedoNote entries) instead of being appended to an exact kernel's list, and the walk returns -- pairs noted, | kWalkDrained
// when the queue ran out -- as soon as the list could not take another group's worth; the kernel scores the noted pairs with the
// exact body and calls again.
template <int W, bool SYM, bool INL = false>
__device__ __forceinline__ int pack_walk(const KernelArgs& A, const PackRanges& R, uint32_t* queue, const double* emit_tab, const int lane, int* note = nullptr) {
  const double ca = A.mc.a, cb = A.mc.b, cc = A.mc.c, cd = A.mc.d, ce = A.mc.e, cf = A.mc.f, cg = A.mc.g;
  const double MATCH = A.mc.match, MISMATCH = A.mc.mismatch;
  const float c32 = A.mc.c;
  const double IMP = kImp;
  const double cabs_up = fabs((double)c32) * (1.0 + 0x1p-22);
  const double thr0 = -600.0 + 1e-6;
  constexpr int NQ = (W + 3) / 4;
  // ranges of the launch (<= 5, widest segments first): groups [grp_end[r-1], grp_end[r]) are the pairs
  // [pk_first[r], pk_end[r]) taken 64 >> pk_shift[r] at a time
  const int ge0 = R.grp_end[0], ge1 = R.grp_end[1], ge2 = R.grp_end[2], ge3 = R.grp_end[3], ge4 = R.grp_end[4];
  const int n_groups = ge4;

  struct Desc { int64_t read_off, hap_off, out_idx; int32_t m, n, hfl; };
  auto pop = [&]() __attribute__((always_inline)) {
    // one group per pop; all lanes issue the add (lane 0 adds 1, the rest 0), see ltr_dp_kernel
    const int g0 = (int)atomicAdd(queue, lane == 0 ? 1u : 0u);
    return uni(g0);
  };
  int g;
  if constexpr (INL) {
    const int held = uni(note[kRedoNote]);                       // a group the previous call of this wave had popped when it left
    if (held >= 0) { g = held; if (lane == 0) note[kRedoNote] = -1; } else g = pop();
  } else g = pop();
  int noted = 0;
  constexpr int NB = (W + 7) / 8;                                // 8-byte words of a lane's strip of bases
  for (;;) {
    if (g >= n_groups) break;
    // ---- the group's range: lanes per pair, first pair (all on the scalar unit) ----
    int lp_shift, q, q_end;
    if (g < ge0)      { lp_shift = R.shift[0]; q = R.first[0] + (g << (6 - R.shift[0])); q_end = R.end[0]; }
    else if (g < ge1) { lp_shift = R.shift[1]; q = R.first[1] + ((g - ge0) << (6 - R.shift[1])); q_end = R.end[1]; }
    else if (g < ge2) { lp_shift = R.shift[2]; q = R.first[2] + ((g - ge1) << (6 - R.shift[2])); q_end = R.end[2]; }
    else if (g < ge3) { lp_shift = R.shift[3]; q = R.first[3] + ((g - ge2) << (6 - R.shift[3])); q_end = R.end[3]; }
    else              { lp_shift = R.shift[4]; q = R.first[4] + ((g - ge3) << (6 - R.shift[4])); q_end = R.end[4]; }
    lp_shift = uni(lp_shift); q = uni(q); q_end = uni(q_end);
    const int LP = 1 << lp_shift;
    const int hl = lane & (LP - 1), seg = lane >> lp_shift;
    // bit of the first lane of every segment
    const uint64_t head_mask = lp_shift == 1 ? 0x5555555555555555ull : (lp_shift == 2 ? 0x1111111111111111ull : (lp_shift == 3 ? 0x0101010101010101ull
                               : (lp_shift == 4 ? 0x0001000100010001ull : 0x0000000100000001ull)));
    const bool is_head = __builtin_amdgcn_inverse_ballot_w64(head_mask);
    // ---- my segment's pair: everything per lane -----------------------------------------------
    const int pi = min(q + seg, q_end - 1);                          // (past the end: any valid pair, never used)
    Desc D;
    {
      const PairDesc* __restrict__ pp = A.pairs + pi;
      D.read_off = pp->read_off; D.hap_off = pp->hap_off; D.out_idx = pp->out_idx; D.m = pp->m; D.n = pp->n; D.hfl = pp->hap_full_len;
    }
    const bool have = (q + seg) < q_end;
    int n = D.n, m = D.m;
    const int hfl = D.hfl;
    const int64_t hap_off = D.hap_off, read_off = D.read_off, out_idx = D.out_idx;
    // HapAligner.cpp:241-244, :249-252: constant scores; single rows / single columns never get here (the
    // plan bins them with the one-wave kernels) -- if one does, the generic exact kernel takes it
    const bool konst = (hfl <= 60) || (abs(n - m) > 600);
    const bool odd = !konst && (n < 2 || m < 2 || (m - 1) > (W << lp_shift));
    const bool dead = !have || konst || odd;
    if (dead) { n = 2; m = 2; }
    const int dd = n - m;
    const int L = (m - 1 + W - 1) / W;                           // lanes of my segment that own real columns (<= LP)
    const int Wl = (m - 1) - (L - 1) * W;                        // real columns of the last of them
    const int T = (n - 1) + (L - 1);
    const bool is_last = !dead && (hl == L - 1);
    const uint32_t nrows = (!dead && hl < L) ? (uint32_t)(n - 1) : 0u;   // I own rows 1 .. nrows
    const int tfin = is_last ? (T - 1) : -1;                     // the step in which I finish my pair
    const uint8_t* __restrict__ hap = A.hap_bytes + hap_off;
    const uint8_t* __restrict__ read = A.read_bytes + read_off;
    const int j0 = 1 + hl * W;                                   // first column of my strip
    // ---- loads of the set-up, all issued at once: the first bases, my strip of the read and of the haplotype
    // (eight bases per load, unclamped past column m - 1: both buffers are padded, and the slots beyond the read --
    // the last lane's slack, idle lanes -- compute values nobody reads) ----
    const int js = min(j0, m - 1);
    uint64_t rw[NB], hw[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) { __builtin_memcpy(&rw[k], read + js + 8 * k, 8); __builtin_memcpy(&hw[k], hap + js + 8 * k, 8); }
    const uint32_t h0 = (uint32_t)hap[0], r0 = (uint32_t)read[0], r1 = (uint32_t)read[1];
    // ... and the next group's number
    const int g_next = pop();
    const int Tmax = wave_max_i(dead ? 0 : T);
    const double emit00 = (h0 == r0) ? MATCH : MISMATCH;          // match_matrix[0], :265
    const uint32_t e01 = (h0 == r1) ? 1u : 0u;                     // emission of the whole first column, :276

    // ---- row 0 (HapAligner.cpp:263-272) for my columns -> X(0,j), Y(0,j): one record of the model table per column,
    // picked by the row's emission test (the reference indexes the haplotype with the READ index here; past its end
    // -- '\0' / undefined -- counts as a mismatch) ----
    double Xp[W], Yp[W];
    uint32_t rc[NQ];
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) rc[qd] = 0;
    const double2* __restrict__ row0 = (const double2*)A.row0XY;
#pragma unroll
    for (int s = 0; s < W; ++s) {
      const uint32_t rb = (uint32_t)(rw[s / 8] >> (8 * (s % 8))) & 0xffu;
      const uint32_t hb = (uint32_t)(hw[s / 8] >> (8 * (s % 8))) & 0xffu;
      const int jc = min(js + s, m - 1);
      const uint32_t eq = ((js + s < n) & (hb == r0)) ? 1u : 0u;
      const double2 xy = row0[2 * jc + eq];
      Xp[s] = xy.x; Yp[s] = xy.y;
      rc[s / 4] |= ((rb >> 1) & 3u) << (2 * (s % 4) + 4);
    }
    double outX = Xp[W - 1];
    double leftX;
    {
      const double fill = dmax(emit00 + ce, dmax(IMP + cd, IMP + cb));
      leftX = wave_shr1(outX, fill);
      if (is_head) leftX = fill;
    }
    double outZ = IMP;
    // certificate chain: all ones ahead of the wavefronts (ltr_dp_kernel.hpp)
    uint64_t fmask = ~0ull;
    uint64_t watch = __builtin_amdgcn_ballot_w64(is_last);      // last lanes of the pairs still running
    uint64_t lost = 0;                                           // ... of the pairs the certificate could not clear
    double certM = 0.0;
    double res_cap = 0.0;
    // haplotype rows as emission-table block offsets: row t + 1 - hl at step t
    const uint16_t* __restrict__ hs = A.hap_codes + hap_off + (1 - hl);
    uint32_t h_next = hs[0];
    // left boundary of the head lanes: record i of the interleaved model table = X(i,0), Z(i,0) for
    // emit(hap[0], read[1]) = mismatch | match (HapAligner.cpp:274-280)
    const double2* __restrict__ colXZ = (const double2*)A.colXZ + e01;
    double2 b_next = colXZ[2 * 1];
    double kd = (double)(dd - (1 - hl) + j0);                    // band offset k of (my row, j0); -1 per step

    auto step = [&](auto fin_tag, const int t) __attribute__((always_inline)) {
      constexpr bool FIN = decltype(fin_tag)::value;
      const uint32_t h = h_next;
      const double2 bnd = b_next;
      h_next = hs[t + 1];
      b_next = colXZ[2 * min(t + 2, A.table_len)];
      // (lane 0 is a head like every first lane of a segment: no fill value for the shift -- two moves per double less a step)
      double mX = wave_shr1_nofill(outX);                        // X(i, j0-1)
      double mZ = wave_shr1_nofill(outZ);                        // Z(i, j0-1)
      if (is_head) { mX = bnd.x; mZ = bnd.y; }                   // not the previous segment's last lane: my pair's first column
      const double kcur = kd;
      kd = kcur - 1.0;
      // my row at this step is t + 1 - hl; I have one while 1 <= row <= nrows
      const uint32_t row = (uint32_t)(t + 1 - hl);
      const uint64_t active_mask = __builtin_amdgcn_ballot_w64((row - 1u) < nrows);
      const bool active = __builtin_amdgcn_inverse_ballot_w64(active_mask);
      if (active) {
        double diag = leftX;
        leftX = mX;
        double zleft = mZ;
        double Iv = 0.0, Dv = 0.0;
        double em[W];
        auto fetch_quad = [&](const int qd) __attribute__((always_inline)) {
          const double2* rowp = (const double2*)((const char*)emit_tab + (h + rc[qd < NQ ? qd : 0]));
          const double2 lo = rowp[0];
          em[4 * qd] = lo.x;
          if (4 * qd + 1 < W) em[(4 * qd + 1) < W ? (4 * qd + 1) : 0] = lo.y;
          if (4 * qd + 2 < W) {
            const double2 hi = rowp[kEmitTabDoubles / 4];
            em[(4 * qd + 2) < W ? (4 * qd + 2) : 0] = hi.x;
            if (4 * qd + 3 < W) em[(4 * qd + 3) < W ? (4 * qd + 3) : 0] = hi.y;
          }
        };
#pragma unroll
        for (int qd = 0; qd < NQ && qd <= 1; ++qd) fetch_quad(qd);
        certM = em[0] + diag;
        double Mv = certM;
#pragma unroll
        for (int s = 0; s < W; ++s) {
          double Mnext = 0.0;
          if ((s % 4) == 2 && (s / 4 + 2) < NQ) fetch_quad((s / 4 + 2) < NQ ? (s / 4 + 2) : 0);
          if (s + 1 < W) Mnext = em[(s + 1) < W ? (s + 1) : 0] + Xp[s];
          Iv = MATCH + Yp[s];
          Dv = zleft;
          // (FIN: the pair's result is best(n-1, m-1), :309 -- slot Wl-1 of my segment's last lane)
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
      fmask = cert | ((fmask << 1) & ~head_mask);
      // a last lane has just finished a row nobody certified: that pair goes to the exact kernel
      const uint64_t miss = ~fmask & watch;
      if (miss != 0) { lost |= miss; watch &= ~miss; }
    };

    // The step that finishes a pair (its last lane on row n-1) runs the FIN copy of the body.  The copies
    // follow one another -- plain steps up to the next finishing step, that step, plain steps again -- and are
    // never alternatives inside one loop: merging two copies at a loop back-edge makes hipcc keep two register
    // sets for the 2W carried values and shuffle them every step (ltr_dp_dual.hpp).
    int t = 0;
    while (t < Tmax) {
      uint64_t fin_now = __builtin_amdgcn_ballot_w64(tfin == t);
      while (fin_now == 0) {                                     // (Tmax - 1 is some pair's finishing step: the loop ends there at the latest)
        step(BoolTag<false>{}, t);
        ++t;
        fin_now = __builtin_amdgcn_ballot_w64(tfin == t);
      }
      step(BoolTag<true>{}, t);
      watch &= ~fin_now;                                         // finished: their chain bits decay from here on
      ++t;
      if (watch == 0) break;                                     // every pair of the wave finished or lost
    }

    // ---- results: every segment's last lane holds its pair's -----------------------------------
    const bool is_lost = __builtin_amdgcn_inverse_ballot_w64(lost);
    if (is_last) {
      if (is_lost) {
        if (!INL) {
          // could not prove "no row aborts": an exact kernel scores the pair (push_redo, one lane per pair)
          int cls = kXGeneric;
          if (A.xlut) {
            const int C = m - 1;
            cls = (C <= 64 * kXShortW) ? kXShort : ((C <= 64 * kXMidW) ? kXMid : ((C <= 64 * kXLongW) ? kXLong
                  : ((C <= kXWg4MaxC) ? kXWg4 : ((C <= kXWg8MaxC) ? kXWg8 : kXLong))));
          }
          const int slot = (int)atomicAdd(A.xcount + cls, 1u);
          A.xlist[cls][slot] = pi;
        }
      } else {
        A.out_ll[out_idx] = res_cap;
      }
    }
    if (have && hl == 0 && (konst || odd)) {
      if (konst) A.out_ll[out_idx] = (hfl <= 60) ? IMP : -700.0;
      else if (!INL) { const int slot = (int)atomicAdd(A.xcount + kXGeneric, 1u); A.xlist[kXGeneric][slot] = pi; }
    }
    if constexpr (INL) {
      // ... or, inside the plan kernel, noted for the exact body (bit 31: a pair this geometry cannot take -> the generic body)
      const bool mine = (is_last && is_lost) || (have && hl == 0 && odd && !konst);
      const uint64_t nm = __builtin_amdgcn_ballot_w64(mine);
      if (nm != 0) {
        if (mine) note[noted + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0u))] = (is_last && is_lost) ? pi : (pi | (int)0x80000000);
        noted += (int)__builtin_popcountll(nm);
      }
    }
    g = g_next;
    if constexpr (INL) {
      // pairs noted: leave at once (the exact body starts now, not behind the entry's last group), and hand the group that is
      // already popped to the next call
      if (noted > 0 && g < n_groups) { if (lane == 0) note[kRedoNote] = g; return noted; }
    }
  }
  return noted | kWalkDrained;
}

template <int W, bool SYM>
__global__ __launch_bounds__(64 * kBlockWaves, LTR_PACK_LB) void ltr_dp_pack_kernel(KernelArgs A) {
  const int lane = threadIdx.x & 63;
  __shared__ __attribute__((aligned(16))) double s_emit[kEmitTabDoubles];
  for (int idx = threadIdx.x; idx < kEmitTabDoubles; idx += 64 * kBlockWaves) {
    const int half = idx >> 11, hcode = (idx >> 9) & 3, quad = (idx >> 1) & 255, k = 2 * half + (idx & 1);
    s_emit[idx] = (hcode == ((quad >> (2 * k)) & 3)) ? (double)A.mc.match : (double)A.mc.mismatch;
  }
  __syncthreads();
  PackRanges R;
#pragma unroll
  for (int r = 0; r < 5; ++r) { R.shift[r] = A.pk_shift[r]; R.first[r] = A.pk_first[r]; R.end[r] = A.pk_end[r]; R.grp_end[r] = A.pk_grp_end[r]; }
  pack_walk<W, SYM>(A, R, A.queue, s_emit, lane);
}

// The packed launches of strip widths kPackMultiMinW .. kPackWMax as ONE persistent launch: table t of A.pk_tabs = one strip
// width's ranges and work counter, widest first; a wavefront that finds a width's queue empty goes on with the next width
// instead of draining (see ltr_dp_multi_kernel).  A call per strip width: every body keeps its own register allocation.
template <int W, bool SYM>
__device__ __attribute__((noinline)) void pack_walk_call(int64_t kernarg_v, int tab_v, unsigned emit_lds_v) {
  const KernelArgs& A = *(const KernelArgs*)(KernArgPtr)(uintptr_t)uni64(kernarg_v);
  const int lane = threadIdx.x & 63;
  const double* emit_tab = (const double*)(LdsDoubles)(uintptr_t)(unsigned)uni((int)emit_lds_v);
  const PackTable* __restrict__ T = A.pk_tabs + uni(tab_v);
  PackRanges R;
#pragma unroll
  for (int r = 0; r < 5; ++r) { R.shift[r] = uni(T->shift[r]); R.first[r] = uni(T->first[r]); R.end[r] = uni(T->end[r]); R.grp_end[r] = uni(T->grp_end[r]); }
  uint32_t* queue = A.queue_base + uni(T->queue_class);
  pack_walk<W, SYM>(A, R, queue, emit_tab, lane);
}

static_assert(kPackWMax == 20 && kPackMultiMinW == 13, "the switch below names the strip widths 13 .. 20");
template <bool SYM>
__global__ __launch_bounds__(64 * kBlockWaves, 3) void ltr_dp_pack_multi_kernel(KernelArgs A) {
  __shared__ __attribute__((aligned(16))) double s_emit[kEmitTabDoubles];
  for (int idx = threadIdx.x; idx < kEmitTabDoubles; idx += 64 * kBlockWaves) {
    const int half = idx >> 11, hcode = (idx >> 9) & 3, quad = (idx >> 1) & 255, k = 2 * half + (idx & 1);
    s_emit[idx] = (hcode == ((quad >> (2 * k)) & 3)) ? (double)A.mc.match : (double)A.mc.mismatch;
  }
  __syncthreads();
  const unsigned emit_lds = (unsigned)(uintptr_t)(LdsDoubles)s_emit;
  const int64_t kargs = (int64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int n_tabs = A.pk_ntabs;
  for (int t = 0; t < n_tabs; ++t) {
    const int W = uni(A.pk_tabs[t].W);
    switch (W) {
      case 13: pack_walk_call<13, SYM>(kargs, t, emit_lds); break;
      case 14: pack_walk_call<14, SYM>(kargs, t, emit_lds); break;
      case 15: pack_walk_call<15, SYM>(kargs, t, emit_lds); break;
      case 16: pack_walk_call<16, SYM>(kargs, t, emit_lds); break;
      case 17: pack_walk_call<17, SYM>(kargs, t, emit_lds); break;
      case 18: pack_walk_call<18, SYM>(kargs, t, emit_lds); break;
      case 19: pack_walk_call<19, SYM>(kargs, t, emit_lds); break;
      default: pack_walk_call<20, SYM>(kargs, t, emit_lds); break;
    }
  }
}

// ltr_short.hip -- the SHORT (seeded, stutter-aware) alignment path on the GPU, SURVEY.md row a-7.
//
// Replaces, for period-1 loci under --stutter-align-len (reference HapAligner.cpp:552):
//   HapAligner::process_read, short_ branch           src/SeqAlignment/HapAligner.cpp:855-990 (retrace_aln = false)
//   HapAligner::align_seq_to_hap_short                HapAligner.cpp:27-163
//   HapAligner::compute_aln_logprob                   HapAligner.cpp:165-233
//   StutterAlignerClass::load_read / align_*_reverse  src/SeqAlignment/StutterAlignerClass.cpp:12-166
//   fast_log_sum_exp(vector) with fasterexp/fasterlog src/mathops.cpp:98-107, src/fastonebigheader.h:207-218,349-358
// Host prep (this file, integer/libm work): calc_seed_base (:494-542), BaseQuality tables
// (base_quality.h:29-75), StutterModel::log_stutter_pmf (stutter_model.cpp:29-53),
// RepeatStutterInfo::log_prob_pcr_artifact (RepeatStutterInfo.h:53-61), upstream-match tables
// (StutterAlignerClass.h:34-41), int_log (mathops.cpp:14-22).  Every libm value (log, pow) is
// formed on the host exactly like the reference forms it; the device only adds, compares and runs
// the FP32 bit tricks, so results are bit-identical to the CPU restatement.
//
// Mapping: prep kernel (per-read tables) -> flank kernels (one wavefront per (read, haplotype, side),
// rows skewed across lanes with wave_shr:1, strip width 2/4/8 per side) -> block kernel (four-wave workgroups, stutter tables
// in LDS, one (position, artifact) walk per lane) -> final kernel (compute_aln_logprob's log-sum over the seed positions).
// The reference's seed x max_hap matrices are never materialised: only each row's last column is kept, which is all
// compute_aln_logprob reads.  The round-2 lane-per-pair kernel stays behind ltr_ctx_set_debug("short_lane_kernel") and
// for sides wider than 512 bases.  This path is minor (homopolymer loci, ~5e4 cells per pair); see DESIGN.md.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ltr_internal.h"

namespace {

constexpr double kImpS = -1000000000.0;        // IMPOSSIBLE, HapAligner.cpp:20
constexpr int kMaxIns = 6, kMaxDel = 6;        // MAX_STUTTER_REPEAT_INS / -DEL, RepeatStutterInfo.h:10-11
constexpr int kNumArt = kMaxIns + kMaxDel + 1;

struct ShortHap {            // one haplotype combination, one direction
  int64_t seq_off;           // into hap_bytes: the three blocks back to back
  int32_t len[3];            // block lengths (block 1 = repeat)
  int32_t up_off;            // into upstream: num_deletions (or 1) arrays of len[1] ints
  int32_t num_deletions;     // StutterAlignerClass::num_deletions_
  int32_t pad;
};
struct ShortRead {
  int64_t seq_off;           // read bytes (full sequence), then the reversed right part at rev_off
  int64_t rev_off;
  int64_t q_off;             // into wrong/correct: [0,seed) left part, [seed+1,len) right part REVERSED (HapAligner.cpp:889-890)
  int64_t cum_off;           // into cum: seed + 1 running sums of the left part's log_correct (left_prob before every base, then the
                             // total, HapAligner.cpp:36-44), then len - seed of the reversed right part's
  int32_t len, seed;
};
struct ShortArgs {
  const ShortRead* reads; const ShortHap* fw; const ShortHap* rv;
  const uint8_t* read_bytes; const uint8_t* hap_bytes; const int32_t* upstream;
  const double* wrong; const double* correct;
  const double* cum;         // ShortRead::cum_off
  const double* art;         // [H][kNumArt] log_prob_pcr_artifact per combination
  const double* int_log;     // log(i)
  const int32_t* pair_read; const int32_t* pair_hap; const int64_t* pair_out;
  int32_t n_pairs, period;
  double* out;
  double* scratch;           // [block][elements][64]
  int64_t scratch_per_block; // doubles
  int32_t S, HS, LP;         // max side length, max haplotype size, log_probs_ capacity of this call
  int32_t n_ilog, maxB;      // entries of int_log; longest repeat block
  int32_t chunk_first, chunk_pairs;   // the pairs this set of launches scores
  double* g_row; double* g_last;      // per (pair of the chunk, side): S doubles (the row handed from launch to launch), HS + 2 doubles (last column of every row)
  double* g_terms;                    // ... and 13 x S doubles: the block row's terms by (artifact size, read position)
  float a, b, c, d, e, f, g;
  double log_thresh;         // log(0.001), mathops.h:36
  // ltr_short_prep_kernel: base qualities -> wrong / correct / cum, on the device
  const uint8_t* qidx;       // per base (ShortRead::q_off, the order of wrong / correct): BaseQuality's table index
  const double* qtab;        // [0, 64): log_error_ by index, [64, 128): log_correct_ (base_quality.h:29-43), from the host's libm
  double* wrong_w; double* correct_w; double* cum_w;
  int32_t n_reads;
};

template <bool V> struct BoolTagS { static constexpr bool value = V; };
__device__ __forceinline__ double dmx(double x, double y) { return x < y ? y : x; }   // std::max
// fastonebigheader.h:207-218
__device__ __forceinline__ float d_fasterexp(float p) {
  p = 1.442695040f * p;
  const float clipp = (p < -126) ? -126.0f : p;
  const uint32_t i = (uint32_t)((1 << 23) * (clipp + 126.94269504f));
  return __uint_as_float(i);
}
// fastonebigheader.h:349-358
__device__ __forceinline__ float d_fasterlog(float x) {
  float y = (float)__float_as_uint(x);
  y *= 8.2629582881927490e-8f;
  return y - 87.989971088f;
}

#define LN(buf, idx) (buf)[(size_t)(idx) * 64 + lane]

// fast_log_sum_exp over lane-private values v[0..n), mathops.cpp:98-107
__device__ __forceinline__ double d_flse(const double* v, int n, int lane, double log_thresh) {
  double mx = LN(v, 0);
  for (int i = 1; i < n; i++) { const double x = LN(v, i); if (mx < x) mx = x; }
  double total = 0;
  for (int i = 0; i < n; i++) { const double diff = LN(v, i) - mx; if (diff > log_thresh) total += d_fasterexp((float)diff); }
  return mx + d_fasterlog((float)total);
}

struct Side {                // per-lane view of one alignment side
  const uint8_t* seq; int seq_len; const double* wrong; const double* correct;
  const uint8_t* hap; int len0, len1, len2; const int32_t* up; int num_del;
};

// align_seq_to_hap_short (HapAligner.cpp:27-163) with rolling rows; lastM[row] = M[row][seq_len-1].
// Returns left_prob (the sum of base_log_correct over the side, :42).
__device__ double short_side(const ShortArgs& A, const Side& sd, const double* art, double* W, int lane, double* lastM) {
  const int S = A.S;
  double* Mp = W;               double* Ip = W + (size_t)S * 64;        double* Dp = W + (size_t)2 * S * 64;
  double* Mc = W + (size_t)3 * S * 64; double* Ic = W + (size_t)4 * S * 64; double* Dc = W + (size_t)5 * S * 64;
  double* insP = W + (size_t)6 * S * 64;          // [seq_len * 6]
  double* delP = W + (size_t)12 * S * 64;         // [seq_len * num_del]
  double* matP = W + (size_t)18 * S * 64;         // [seq_len]
  double* lp = W + (size_t)19 * S * 64;           // log_probs_ scratch, A.LP elements
  const int seq_len = sd.seq_len;
  const uint8_t* seq = sd.seq; const double* wrong = sd.wrong; const double* correct = sd.correct;
  const double ca = A.a, cb = A.b, cc = A.c, cd = A.d, ce = A.e, cf = A.f, cg = A.g;
  const int period = A.period;

  double left_prob = 0.0;
  const uint8_t first_hap_base = sd.hap[0];
  for (int j = 0; j < seq_len; ++j) {                                  // :36-44
    LN(Mp, j) = (seq[j] == first_hap_base ? correct[j] : wrong[j]) + left_prob;
    LN(Ip, j) = correct[j] + left_prob;
    LN(Dp, j) = kImpS;
    left_prob += correct[j];
  }
  LN(lastM, 0) = LN(Mp, seq_len - 1);
  int hap_index = 1, stutter_R = -1;
  const int blen[3] = {sd.len0, sd.len1, sd.len2};
  int boff = 0;
  for (int bi = 0; bi < 3; bi++) {
    const uint8_t* bseq = sd.hap + boff;
    const int block_len = blen[bi];
    boff += block_len;
    if (bi == 1) {                                                     // stutter block, :64-111
      const int max_ins = kMaxIns * period, num_del = sd.num_del, max_del = -period * num_del;
      const uint8_t* blk = bseq + (block_len - 1);                     // block_seq_ points at the LAST base
      // ---- StutterAlignerClass::load_read(seq_len, seq_0+seq_len-1, ...), StutterAlignerClass.cpp:12-53
      {
        const uint8_t* bs = seq + (seq_len - 1); const double* bw = wrong + (seq_len - 1); const double* bc = correct + (seq_len - 1);
        int ins_index = 0, del_index = 0;
        for (int i = 0; i < seq_len; i++) {
          int j; double log_prob = 0.0;
          for (j = 0; j < min(seq_len - i, -max_del); j++) {
            log_prob += (bs[-i - j] == blk[-j] ? bc[-i - j] : bw[-i - j]);
            if ((j + 1) % period == 0) { LN(delP, del_index) = log_prob; del_index++; }
          }
          for (; j < -max_del; j++) if ((j + 1) % period == 0) del_index++;
          for (; j < min(seq_len - i, block_len); j++) log_prob += (bs[-i - j] == blk[-j] ? bc[-i - j] : bw[-i - j]);
          LN(matP, i) = log_prob;
          double log_ins_prob = 0.0;
          for (j = 0; j < min(max_ins, seq_len - i); j++) {
            if (j % period < block_len) log_ins_prob += (bs[-i - j] == blk[-(j % period)] ? bc[-i - j] : bw[-i - j]);
            else log_ins_prob += bc[-i - j];
            if ((j + 1) % period == 0) { LN(insP, ins_index) = log_ins_prob; ins_index++; }
          }
          for (; j < max_ins; j++) if ((j + 1) % period == 0) { LN(insP, ins_index) = log_ins_prob; ins_index++; }
        }
      }
      // Mp currently holds the row before the block (prev_row_index); the block's last row goes to Mc
      int offset = seq_len - 1;
      for (int j = 0; j < seq_len; ++j, --offset) {
        double bp[kNumArt];
        int art_idx = 0;
        for (int asz = -kMaxDel * period; asz <= max_ins; asz += period, ++art_idx) {
          const int base_len = min(block_len + asz, j + 1);
          if (base_len < 0) { bp[art_idx] = kImpS; continue; }
          const uint8_t* bs = seq + j; const double* bw = wrong + j; const double* bc = correct + j;
          double prob;
          if (asz == 0) prob = LN(matP, offset);                       // align_no_artifact_reverse, :55-57
          else if (asz > 0) {                                          // align_pcr_insertion_reverse, :59-104
            const int D = asz; int np = 0;
            const int32_t* up = sd.up + (block_len - 1);               // upstream_match_lengths_[0]
            double log_prob = -A.int_log[block_len + 1] + LN(insP, kMaxIns * offset + D / period - 1) +
                              (base_len > D ? LN(matP, offset + D) : 0);
            LN(lp, np) = log_prob; np++;
            int i = 0;
            for (; i > -min(max(0, base_len - D), block_len); i--) {
              if (-i + period < block_len) {
                if (up[i] == 0) {
                  for (int index = i - period; index >= i - D; index -= period) {
                    log_prob -= (bs[index] == blk[i] ? bc[index] : bw[index]);
                    log_prob += (bs[index] == blk[i - period] ? bc[index] : bw[index]);
                  }
                  LN(lp, np) = log_prob; np++;
                } else {
                  LN(lp, np) = A.int_log[up[i]] + log_prob; np++;
                  i -= (up[i] - 1);
                }
              } else { LN(lp, np) = log_prob; np++; }
            }
            if (i > -block_len) { LN(lp, np) = A.int_log[block_len + i] + log_prob; np++; }
            prob = d_flse(lp, np, lane, A.log_thresh);
          } else {                                                     // align_pcr_deletion_reverse, :106-154
            const int D = asz; int np = 0;
            const int32_t* up = sd.up + (size_t)(-D / period - 1) * block_len + (block_len - 1);
            double log_prob = -A.int_log[block_len + D + 1];
            if (offset + D >= 0) log_prob += LN(matP, offset + D) - LN(delP, (offset + D) * num_del - D / period - 1);
            else for (int jj = 0; jj > -base_len; jj--) log_prob += (blk[jj + D] == bs[jj] ? bc[jj] : bw[jj]);
            LN(lp, np) = log_prob; np++;
            int i;
            for (i = 0; i > -base_len; i--) {
              if (up[i] == 0) {
                log_prob -= (blk[i + D] == bs[i] ? bc[i] : bw[i]);
                log_prob += (blk[i] == bs[i] ? bc[i] : bw[i]);
                LN(lp, np) = log_prob; np++;
              } else {
                LN(lp, np) = A.int_log[up[i]] + log_prob; np++;
                i -= (up[i] - 1);
              }
            }
            if (-i < block_len + D) { LN(lp, np) = A.int_log[block_len + D + i] + log_prob; np++; }
            prob = d_flse(lp, np, lane, A.log_thresh);
          }
          const double pre_prob = (j - base_len < 0 ? 0 : LN(Mp, j - base_len));
          bp[art_idx] = art[art_idx] + prob + pre_prob;                // :91
        }
        // fast_log_sum_exp(block_probs), :103
        double mx = bp[0];
        for (int q = 1; q < kNumArt; q++) if (mx < bp[q]) mx = bp[q];
        double total = 0;
        for (int q = 0; q < kNumArt; q++) { const double diff = bp[q] - mx; if (diff > A.log_thresh) total += d_fasterexp((float)diff); }
        LN(Mc, j) = mx + d_fasterlog((float)total);
        LN(Ic, j) = kImpS; LN(Dc, j) = kImpS;
      }
      stutter_R = hap_index + block_len - 1;
      hap_index += block_len;
      LN(lastM, stutter_R) = LN(Mc, seq_len - 1);
      double* t;
      t = Mp; Mp = Mc; Mc = t; t = Ip; Ip = Ic; Ic = t; t = Dp; Dp = Dc; Dc = t;
    } else {                                                           // flank block, :112-159
      for (int coord = (bi == 0 ? 1 : 0); coord < block_len; ++coord, ++hap_index) {
        const uint8_t hap_char = bseq[coord];
        const bool after = (hap_index == stutter_R + 1);
        LN(Mc, 0) = (seq[0] == hap_char ? correct[0] : wrong[0]);
        LN(Ic, 0) = after ? kImpS : correct[0];
        LN(Dc, 0) = after ? kImpS : dmx(LN(Dp, 0) + cc, LN(Mp, 0) + cd);
        if (after) {                                                   // a stutter block must be followed by a match, :132-141
          for (int j = 1; j < seq_len; ++j) {
            const double emit = (seq[j] == hap_char ? correct[j] : wrong[j]);
            LN(Mc, j) = emit + LN(Mp, j - 1);
            LN(Ic, j) = kImpS; LN(Dc, j) = kImpS;
          }
        } else {
          double Mdiag = LN(Mp, 0), Ddiag = LN(Dp, 0), Ileft = LN(Ic, 0);
          for (int j = 1; j < seq_len; ++j) {
            const double Mup = LN(Mp, j), Dup = LN(Dp, j);
            const double p0 = Ileft + cf, p1 = Mdiag + ce, p2 = Ddiag + cg;
            const double emit = (seq[j] == hap_char ? correct[j] : wrong[j]);
            const double Mn = emit + dmx(p0, dmx(p1, p2));
            const double In = correct[j] + dmx(Mdiag + cb, Ileft + ca);
            const double Dn = dmx(Mup + cd, Dup + cc);
            LN(Mc, j) = Mn; LN(Ic, j) = In; LN(Dc, j) = Dn;
            Mdiag = Mup; Ddiag = Dup; Ileft = In;
          }
        }
        LN(lastM, hap_index) = LN(Mc, seq_len - 1);
        double* t;
        t = Mp; Mp = Mc; Mc = t; t = Ip; Ip = Ic; Ic = t; t = Dp; Dp = Dc; Dc = t;
      }
    }
  }
  return left_prob;
}

__global__ __launch_bounds__(64) void ltr_short_kernel(ShortArgs A) {
  const int lane = threadIdx.x;
  double* base = A.scratch + (size_t)blockIdx.x * A.scratch_per_block;
  double* W = base;                                        // 19*S work elements + LP log_probs_ elements, x 64
  double* lastL = base + (size_t)(19 * A.S + A.LP) * 64;
  double* lastR = lastL + (size_t)(A.HS + 2) * 64;
  for (int p = blockIdx.x * 64 + lane; p < A.n_pairs; p += gridDim.x * 64) {
    const ShortRead rd = A.reads[A.pair_read[p]];
    const int k = A.pair_hap[p];
    const ShortHap hf = A.fw[k], hr = A.rv[k];
    const double* art = A.art + (size_t)k * kNumArt;
    const int seed = rd.seed, len = rd.len, rlen = len - seed - 1;
    Side L, R;
    L.seq = A.read_bytes + rd.seq_off; L.seq_len = seed; L.wrong = A.wrong + rd.q_off; L.correct = A.correct + rd.q_off;
    L.hap = A.hap_bytes + hf.seq_off; L.len0 = hf.len[0]; L.len1 = hf.len[1]; L.len2 = hf.len[2];
    L.up = A.upstream + hf.up_off; L.num_del = hf.num_deletions;
    R.seq = A.read_bytes + rd.rev_off; R.seq_len = rlen; R.wrong = A.wrong + rd.q_off + seed + 1; R.correct = A.correct + rd.q_off + seed + 1;
    R.hap = A.hap_bytes + hr.seq_off; R.len0 = hr.len[0]; R.len1 = hr.len[1]; R.len2 = hr.len[2];
    R.up = A.upstream + hr.up_off; R.num_del = hr.num_deletions;
    const double l_prob = short_side(A, L, art, W, lane, lastL);       // :905
    const double r_prob = short_side(A, R, art, W, lane, lastR);       // :908
    // ---- compute_aln_logprob, HapAligner.cpp:165-233 ----
    const int hapsize = hf.len[0] + hf.len[1] + hf.len[2];
    const uint8_t* hs = A.hap_bytes + hf.seq_off;
    const uint8_t seed_char = (A.read_bytes + rd.seq_off)[seed];
    const double sw = A.wrong[rd.q_off + seed], sc = A.correct[rd.q_off + seed];
    const double PRIOR = -A.int_log[hf.len[0] + hf.len[2]];             // num_seeds = non-stutter bases, :175-179
    double* lp = W + (size_t)19 * A.S * 64;
    int np = 0;
    LN(lp, np) = PRIOR + (seed_char == hs[0] ? sc : sw) + l_prob + LN(lastR, hapsize - 2); np++;               // :184-185
    LN(lp, np) = PRIOR + (seed_char == hs[hapsize - 1] ? sc : sw) + r_prob + LN(lastL, hapsize - 2); np++;     // :190-191
    int lrow = 0, rrow = hapsize - 3, coord_abs = 0;
    for (int b = 0; b < 3; ++b) {
      const int bl = hf.len[b];
      if (b == 1) { lrow += bl; rrow -= bl; coord_abs += bl; continue; }
      const int c0 = (b == 0 ? 1 : 0), c1 = (b == 2 ? bl - 1 : bl);
      for (int cidx = c0; cidx < c1; ++cidx) {
        LN(lp, np) = PRIOR + (seed_char == hs[coord_abs + cidx] ? sc : sw) + LN(lastL, lrow) + LN(lastR, rrow); np++;   // :220
        lrow++; rrow--;
      }
      coord_abs += bl;
    }
    A.out[A.pair_out[p]] = d_flse(lp, np, lane, A.log_thresh);         // :230
  }
}


// =============================================================================================================
// The seeded path as THREE kinds of work, each with the mapping that suits it (five launches per set of pairs: 0. the reads'
// quality tables and prefix sums, ltr_short_prep_kernel, then 1. - 4.):
//
//  1. flank rows before the stutter block (HapAligner.cpp:112-159) -- ltr_short_flank_kernel<false>: one wavefront
//     per (pair, side).  A Viterbi recurrence with base-quality emissions whose only same-row dependency is the insertion
//     chain I(i,j-1) -> I(i,j): lane l owns W consecutive read positions and the haplotype rows stream through the lanes
//     skewed by one row per lane -- the K1 geometry of ltr_dp_kernel.hpp -- with the three values a lane needs from its
//     left neighbour (I of the same row, M and D of the row before) handed over by DPP wave_shr:1.  The previous row
//     lives in registers (2 W doubles); nothing per cell touches memory.  Out: the row before the block (one double per
//     read position) and the last column of every row.
//  2. the stutter-block row (:64-111) -- ltr_short_block_kernel: one workgroup of four wavefronts per (pair, side).
//     load_read's match table (StutterAlignerClass.cpp:12-53), one read position per thread, in LDS; then per read
//     position the log-sum over 13 artifact sizes of pmf + stutter-aligner LL + the match value `base_len` positions back
//     in the row before the block.  Every (position, size) term is independent: a row of 16 lanes takes one position,
//     lane a of the row the term of artifact size a -- the reference's walk over the block (.cpp:59-154) run twice, once
//     for the maximum and once for the sum, instead of keeping its log_probs_ list, with load_read's deletion / insertion
//     tables re-summed where they are used (same terms, same order) -- and the row's log-sum is a 16-lane reduction.
//     Block bases, upstream-match tables and integer logs come from LDS: a walk is a chain of dependent look-ups.
//  3. flank rows after the block -- ltr_short_flank_kernel<true>: as 1., starting from the block row; its first row
//     must follow the block with a match (:132-141).
//  4. compute_aln_logprob (:165-233) -- ltr_short_final_kernel: one wavefront per pair, one seed position per lane.
//
// fast_log_sum_exp (mathops.cpp:98-107) adds floats of (0.001, 1.06] into a double -- 24-bit values whose exponents
// span 11 binades: up to half a million of them add EXACTLY, so the order of the sum does not matter and the log-sums of
// 2. and 4. are lane reductions.  The FP operation order of every value the reference rounds is untouched: bit-identical
// to the lane-per-pair kernel above (and to the restatement, whose stutter-block rows are pinned to the compiled reference).
// Between the launches a (pair, side) keeps 8 bytes per read position and per haplotype row in HBM (written once, read
// once); no per-cell traffic anywhere.
// =============================================================================================================
__device__ __forceinline__ double w_shr1(double v) {          // lane l <- lane l-1 (lane 0: 0)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double w_max(double v, const int width) {     // maximum over aligned groups of `width` lanes
  for (int o = width >> 1; o > 0; o >>= 1) { const double x = __shfl_xor(v, o); v = v < x ? x : v; }
  return v;
}
__device__ __forceinline__ double w_sum(double v, const int width) {     // sum over aligned groups of `width` lanes (callers: exact sums only)
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
constexpr int kShortThreads = 256;

// (pair, side) -> the side's view of the read and of the haplotype
__device__ __forceinline__ void pair_side(const ShortArgs& A, const int p, const int side, Side* sd, const double** cum) {
  const ShortRead rd = A.reads[A.pair_read[p]];
  const int k = A.pair_hap[p];
  const int seed = rd.seed;
  if (side == 0) {
    const ShortHap hf = A.fw[k];
    sd->seq = A.read_bytes + rd.seq_off; sd->seq_len = seed; sd->wrong = A.wrong + rd.q_off; sd->correct = A.correct + rd.q_off;
    sd->hap = A.hap_bytes + hf.seq_off; sd->len0 = hf.len[0]; sd->len1 = hf.len[1]; sd->len2 = hf.len[2];
    sd->up = A.upstream + hf.up_off; sd->num_del = hf.num_deletions;
    *cum = A.cum + rd.cum_off;
  } else {
    const ShortHap hr = A.rv[k];
    sd->seq = A.read_bytes + rd.rev_off; sd->seq_len = rd.len - seed - 1; sd->wrong = A.wrong + rd.q_off + seed + 1; sd->correct = A.correct + rd.q_off + seed + 1;
    sd->hap = A.hap_bytes + hr.seq_off; sd->len0 = hr.len[0]; sd->len1 = hr.len[1]; sd->len2 = hr.len[2];
    sd->up = A.upstream + hr.up_off; sd->num_del = hr.num_deletions;
    *cum = A.cum + rd.cum_off + seed + 1;
  }
}

// 0.: one wavefront per read: BaseQuality's log tables looked up per base (base_quality.h:53-75), and left_prob of
// align_seq_to_hap_short's first row (HapAligner.cpp:36-44: left_prob += base_log_correct[j], base by base) for both
// sides -- it depends on the read alone, so it is summed once here, in the reference's order (one lane per side), not
// once per haplotype.  The host sends one byte per base instead of three doubles.
__global__ __launch_bounds__(kShortThreads) void ltr_short_prep_kernel(ShortArgs A) {
  const int lane = threadIdx.x & 63;
  const int r = (int)blockIdx.x * (kShortThreads / 64) + (int)(threadIdx.x >> 6);
  if (r >= A.n_reads) return;
  const ShortRead rd = A.reads[r];
  const uint8_t* q = A.qidx + rd.q_off;
  for (int j = lane; j < rd.len; j += 64) { const int k = q[j]; A.wrong_w[rd.q_off + j] = A.qtab[k]; A.correct_w[rd.q_off + j] = A.qtab[64 + k]; }
  if (lane < 2) {
    const int j0 = lane == 0 ? 0 : rd.seed + 1, j1 = lane == 0 ? rd.seed : rd.len;
    double* cum = A.cum_w + rd.cum_off + j0;
    double acc = 0.0;
    for (int j = j0; j < j1; j++) { *cum++ = acc; acc += A.qtab[64 + q[j]]; }
    *cum = acc;
  }
}

// 1. / 3.: one wavefront per (pair, side).  g_row: the row handed from launch to launch (the row before the block, then
// the block's row); g_last: M[row][seq_len - 1] of every haplotype row.
template <int W, bool SECOND>
__device__ __forceinline__ void short_flank_rows(const ShortArgs& A, const Side& sd, const double* cum, double* g_row, double* last, const int lane) {
  const int S = sd.seq_len;
  const double ca = A.a, cb = A.b, cc = A.c, cd = A.d, ce = A.e, cf = A.f, cg = A.g;
  const int nl = (S + W - 1) / W;                              // lanes that own read positions
  const int j0 = lane * W;
  const int own_last = (S - 1) - (nl - 1) * W;                 // slot of the last read position, in lane nl - 1
  double Mp[W], Dp[W], cw[W], ww[W];
  uint32_t sq[W];
  const int block_len = sd.len1;
#pragma unroll
  for (int s = 0; s < W; ++s) {
    const int j = min(j0 + s, S - 1);                          // (slack positions: clamped loads, values nobody reads)
    cw[s] = sd.correct[j]; ww[s] = sd.wrong[j]; sq[s] = sd.seq[j];
    Dp[s] = kImpS;
    if (SECOND) Mp[s] = g_row[j];                              // the block's row; I = D = IMPOSSIBLE there (:104-105)
    else {
      const double lp = cum[j];                                // left_prob before base j, :36-44
      Mp[s] = (sq[s] == (uint32_t)sd.hap[0] ? cw[s] : ww[s]) + lp;
    }
  }
  if (!SECOND && lane == nl - 1) {
#pragma unroll
    for (int s = 0; s < W; ++s) if (s == own_last) last[0] = Mp[s];
  }
  // what my right neighbour takes from me at its next step: I of my last position in the row I have just finished, M and
  // D of that position in the row before (to begin with: row 0, or the block's row)
  double pubI = 0.0, pubM2 = Mp[W - 1], pubD2 = Dp[W - 1];
  const uint8_t* hap_chars = SECOND ? sd.hap + sd.len0 + block_len : sd.hap + 1;
  const int nrows = SECOND ? sd.len2 : sd.len0 - 1;
  const int hap_row0 = SECOND ? sd.len0 + block_len : 1;
  const int T = nrows > 0 ? nrows + nl - 1 : 0;
  for (int t = 0; t < T; ++t) {
    const double nI = w_shr1(pubI), nM = w_shr1(pubM2), nD = w_shr1(pubD2);
    const int r = t - lane;
    const bool active = (r >= 0) & (r < nrows) & (lane < nl);
    if (active) {
      const uint32_t hap_char = hap_chars[r];
      const bool after = SECOND && r == 0;                     // a stutter block must be followed by a match, :132-141
      double Mdiag = nM, Ddiag = nD, Ileft = nI;
      const double keepM = Mp[W - 1], keepD = Dp[W - 1];
#pragma unroll
      for (int s = 0; s < W; ++s) {
        const double Mup = Mp[s], Dup = Dp[s];
        const double emit = (sq[s] == hap_char ? cw[s] : ww[s]);
        const double p0 = Ileft + cf, p1 = Mdiag + ce, p2 = Ddiag + cg;
        double Mn = emit + dmx(p0, dmx(p1, p2));
        double In = cw[s] + dmx(Mdiag + cb, Ileft + ca);
        double Dn = dmx(Mup + cd, Dup + cc);
        if (SECOND) { if (after) { Mn = emit + Mdiag; In = kImpS; Dn = kImpS; } }
        if (s == 0) {                                          // the read's first base, :124-129
          if (lane == 0) { Mn = emit; In = after ? kImpS : cw[0]; Dn = after ? kImpS : dmx(Dup + cc, Mup + cd); }
        }
        Mdiag = Mup; Ddiag = Dup; Ileft = In;
        Mp[s] = Mn; Dp[s] = Dn;
      }
      pubI = Ileft; pubM2 = keepM; pubD2 = keepD;
      if (lane == nl - 1) {
#pragma unroll
        for (int s = 0; s < W; ++s) if (s == own_last) last[hap_row0 + r] = Mp[s];
      }
    }
  }
  if (!SECOND) {
#pragma unroll
    for (int s = 0; s < W; ++s) if (j0 + s < S) g_row[j0 + s] = Mp[s];       // the row before the block
  }
}

// The strip width is picked per (pair, side), wave-uniformly: the narrowest of 2 / 4 / 8 read positions per lane that covers
// the side with 64 lanes -- a flank is ~35 haplotype rows, so a side of 230 positions on 8-wide strips would run 63 steps on
// 29 lanes, on 4-wide strips 91 steps of half the work on 58.  (Registers: those of the widest body.)
template <bool SECOND>
__global__ __launch_bounds__(kShortThreads) void ltr_short_flank_kernel(ShortArgs A) {
  const int lane = threadIdx.x & 63;
  const int ps = (int)blockIdx.x * (kShortThreads / 64) + (int)(threadIdx.x >> 6);     // (pair, side) of my wavefront
  if (ps >= 2 * A.chunk_pairs) return;
  const int p = A.chunk_first + (ps >> 1), side = ps & 1;
  Side sd; const double* cum;
  pair_side(A, p, side, &sd, &cum);
  double* g_row = A.g_row + (size_t)ps * A.S;
  double* last = A.g_last + (size_t)ps * (A.HS + 2);
  const int S = __builtin_amdgcn_readfirstlane(sd.seq_len);
  if (S <= 64 * 2) short_flank_rows<2, SECOND>(A, sd, cum, g_row, last, lane);
  else if (S <= 64 * 4) short_flank_rows<4, SECOND>(A, sd, cum, g_row, last, lane);
  else short_flank_rows<8, SECOND>(A, sd, cum, g_row, last, lane);
}

// 2.: one workgroup per (pair, side): g_row (the row before the block) -> g_row (the block's row); last[stutter_R].
// (eight wavefronts a SIMD: a walk is a chain of dependent LDS look-ups, other wavefronts are what hides their latency)
// (the repeat period as a compile-time 1 -- the only period HapAligner.cpp:552 sends down this path -- was measured: no gain,
// the walks themselves are the instructions: ~900 per (position, artifact size), VALU issue 0.92 - 0.97 by the PMC pass)
__global__ __launch_bounds__(kShortThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void ltr_short_block_kernel(ShortArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
  const int tid = threadIdx.x;
  double* s_cor = (double*)s_raw; double* s_wr = s_cor + A.S; double* s_matP = s_wr + A.S; double* s_prev = s_matP + A.S;
  double* s_ilog = s_prev + A.S;
  int32_t* s_up = (int32_t*)(s_ilog + A.n_ilog);
  uint8_t* s_seq = (uint8_t*)(s_up + (size_t)kMaxDel * A.maxB + 8); uint8_t* s_hap = s_seq + ((A.S + 7) & ~7);
  for (int i = tid; i < A.n_ilog; i += kShortThreads) s_ilog[i] = A.int_log[i];
  const int period = A.period;
  for (int ps = blockIdx.x; ps < 2 * A.chunk_pairs; ps += gridDim.x) {
    const int p = A.chunk_first + (ps >> 1), side = ps & 1;
    Side sd; const double* cum;
    pair_side(A, p, side, &sd, &cum);
    const double* art = A.art + (size_t)A.pair_hap[p] * kNumArt;
    double* g_row = A.g_row + (size_t)ps * A.S;
    const int S = sd.seq_len, block_len = sd.len1, num_del = sd.num_del;
    const int max_del = -period * num_del;
    __syncthreads();                                           // (the previous (pair, side)'s readers are done with the LDS arrays)
    for (int j = tid; j < S; j += kShortThreads) { s_seq[j] = sd.seq[j]; s_cor[j] = sd.correct[j]; s_wr[j] = sd.wrong[j]; s_prev[j] = g_row[j]; }
    {
      const int hs = sd.len0 + sd.len1 + sd.len2, nup = max(num_del, 1) * block_len;
      for (int j = tid; j < hs; j += kShortThreads) s_hap[j] = sd.hap[j];
      for (int j = tid; j < nup; j += kShortThreads) s_up[j] = sd.up[j];
    }
    __syncthreads();
    const uint8_t* seq = s_seq; const double* correct = s_cor; const double* wrong = s_wr;
    const uint8_t* blk = s_hap + sd.len0 + (block_len - 1);     // block_seq_ points at the LAST base of the block
    const uint8_t* bsE = seq + (S - 1); const double* bwE = wrong + (S - 1); const double* bcE = correct + (S - 1);   // ... like the read's arrays
    // ---- StutterAlignerClass::load_read (StutterAlignerClass.cpp:12-53): the match table, one read position per thread ----
    for (int i = tid; i < S; i += kShortThreads) {
      double log_prob = 0.0;
      const int n = min(S - i, block_len);
      for (int j = 0; j < n; j++) log_prob += (bsE[-i - j] == blk[-j] ? bcE[-i - j] : bwE[-i - j]);
      s_matP[i] = log_prob;
    }
    // del_probs_[i * num_deletions_ + q]: the match sum of read position i over the first (q + 1) * period block bases (:34-38)
    auto del_sum = [&](const int i, const int q) __attribute__((always_inline)) {
      double log_prob = 0.0;
      const int n = min(min(S - i, -max_del), (q + 1) * period);
      for (int j = 0; j < n; j++) log_prob += (bsE[-i - j] == blk[-j] ? bcE[-i - j] : bwE[-i - j]);
      return log_prob;
    };
    // ins_probs_[i * num_insertions_ + q]: (q + 1) * period inserted bases copied from the block's last repeat unit (:43-52)
    auto ins_sum = [&](const int i, const int q) __attribute__((always_inline)) {
      double log_ins_prob = 0.0;
      const int n = min(min(kMaxIns * period, S - i), (q + 1) * period);
      for (int j = 0; j < n; j++) {
        if (j % period < block_len) log_ins_prob += (bsE[-i - j] == blk[-(j % period)] ? bcE[-i - j] : bwE[-i - j]);
        else log_ins_prob += bcE[-i - j];
      }
      return log_ins_prob;
    };
    __syncthreads();
    const int32_t* up_base = s_up;                             // [max(num_del, 1)][block_len]
    const double* ilog = s_ilog;
    // VISIT(v): what fast_log_sum_exp does with one entry of the reference's log_probs_ list -- pass 0 its maximum, pass 1 its sum
#define LTR_VISIT(v) do { const double v_ = (v); if (pass == 0) { if (mx < v_) mx = v_; } else { const double diff_ = v_ - mx; if (diff_ > A.log_thresh) total += d_fasterexp((float)diff_); } } while (0)
    // item = (artifact size, read position), size-major: the lanes of a wavefront walk the same kind of artifact
    double* terms = A.g_terms + (size_t)ps * kNumArt * A.S;
    for (int item = tid; item < kNumArt * S; item += kShortThreads) {
      const int ai = item / S, j = item - ai * S;              // artifact size (ai - 6) * period at read position j
      double term;
      {
        const int asz = (ai - kMaxDel) * period;
        const int base_len = min(block_len + asz, j + 1);
        term = kImpS;
        if (base_len >= 0) {
          const int offset = S - 1 - j;
          const uint8_t* bs = seq + j; const double* bw = wrong + j; const double* bc = correct + j;
          double prob;
          if (asz == 0) prob = s_matP[offset];                 // align_no_artifact_reverse, :55-57
          else if (asz > 0) {                                  // align_pcr_insertion_reverse, :59-104
            const int D = asz;
            const int32_t* up0 = up_base + (block_len - 1);    // upstream_match_lengths_[0]
            const double first = -ilog[block_len + 1] + ins_sum(offset, D / period - 1) + (base_len > D ? s_matP[offset + D] : 0);
            double mx = first, total = 0;
            for (int pass = 0; pass < 2; ++pass) {
              double log_prob = first;
              LTR_VISIT(log_prob);
              int i = 0;
              for (; i > -min(max(0, base_len - D), block_len); i--) {
                if (-i + period < block_len) {
                  const int u = up0[i];
                  if (u == 0) {
                    for (int index = i - period; index >= i - D; index -= period) {
                      log_prob -= (bs[index] == blk[i] ? bc[index] : bw[index]);
                      log_prob += (bs[index] == blk[i - period] ? bc[index] : bw[index]);
                    }
                    LTR_VISIT(log_prob);
                  } else {
                    LTR_VISIT(ilog[u] + log_prob);
                    i -= (u - 1);
                  }
                } else LTR_VISIT(log_prob);
              }
              if (i > -block_len) LTR_VISIT(ilog[block_len + i] + log_prob);
            }
            prob = mx + d_fasterlog((float)total);             // fast_log_sum_exp(log_probs_)
          } else {                                             // align_pcr_deletion_reverse, :106-154
            const int D = asz;
            const int32_t* up = up_base + (size_t)(-D / period - 1) * block_len + (block_len - 1);
            double first = -ilog[block_len + D + 1];
            if (offset + D >= 0) first += s_matP[offset + D] - del_sum(offset + D, -D / period - 1);
            else for (int jj = 0; jj > -base_len; jj--) first += (blk[jj + D] == bs[jj] ? bc[jj] : bw[jj]);
            double mx = first, total = 0;
            for (int pass = 0; pass < 2; ++pass) {
              double log_prob = first;
              LTR_VISIT(log_prob);
              int i;
              for (i = 0; i > -base_len; i--) {
                const int u = up[i];
                if (u == 0) {
                  log_prob -= (blk[i + D] == bs[i] ? bc[i] : bw[i]);
                  log_prob += (blk[i] == bs[i] ? bc[i] : bw[i]);
                  LTR_VISIT(log_prob);
                } else {
                  LTR_VISIT(ilog[u] + log_prob);
                  i -= (u - 1);
                }
              }
              if (-i < block_len + D) LTR_VISIT(ilog[block_len + D + i] + log_prob);
            }
            prob = mx + d_fasterlog((float)total);
          }
          const double pre_prob = (j - base_len < 0 ? 0 : s_prev[j - base_len]);
          term = art[ai] + prob + pre_prob;                    // :91
        }
      }
      terms[item] = term;
    }
    __syncthreads();                                           // (workgroup scope: orders the global stores above, too)
    for (int j = tid; j < S; j += kShortThreads) {             // fast_log_sum_exp(block_probs), :103
      double mx = terms[j];
#pragma unroll
      for (int a = 1; a < kNumArt; ++a) { const double v = terms[(size_t)a * S + j]; if (mx < v) mx = v; }
      double total = 0;
#pragma unroll
      for (int a = 0; a < kNumArt; ++a) { const double diff = terms[(size_t)a * S + j] - mx; if (diff > A.log_thresh) total += d_fasterexp((float)diff); }
      const double v = mx + d_fasterlog((float)total);
      g_row[j] = v;
      if (j == S - 1) A.g_last[(size_t)ps * (A.HS + 2) + (sd.len0 - 1) + block_len] = v;       // last[stutter_R], stutter_R = len0 + block_len - 1
    }
#undef LTR_VISIT
  }
}

// 4.: compute_aln_logprob, HapAligner.cpp:165-233: one wavefront per pair, one seed position per lane
__global__ __launch_bounds__(kShortThreads) void ltr_short_final_kernel(ShortArgs A) {
  const int lane = threadIdx.x & 63;
  const int q = (int)blockIdx.x * (kShortThreads / 64) + (int)(threadIdx.x >> 6);
  if (q >= A.chunk_pairs) return;
  const int p = A.chunk_first + q;
  const ShortRead rd = A.reads[A.pair_read[p]];
  const ShortHap hf = A.fw[A.pair_hap[p]];
  const int seed = rd.seed;
  const double* lastL = A.g_last + (size_t)(2 * q) * (A.HS + 2);
  const double* lastR = lastL + (A.HS + 2);
  const double l_prob = A.cum[rd.cum_off + seed], r_prob = A.cum[rd.cum_off + seed + 1 + (rd.len - seed - 1)];     // left_prob of either side, :42
  const int hapsize = hf.len[0] + hf.len[1] + hf.len[2];
  const uint8_t* hs = A.hap_bytes + hf.seq_off;
  const uint8_t seed_char = (A.read_bytes + rd.seq_off)[seed];
  const double sw = A.wrong[rd.q_off + seed], sc = A.correct[rd.q_off + seed];
  const double PRIOR = -A.int_log[hf.len[0] + hf.len[2]];               // num_seeds = non-stutter bases, :175-179
  // entry e: 0 and 1 are the two ends (:184-185, :190-191); then the flank positions in order (:193-225):
  // block 0 coords 1 .. len0-1 (rows lrow = 0.., rrow = hapsize-3 downwards), block 2 coords 0 .. len2-2
  const int n0 = hf.len[0] - 1, n2 = hf.len[2] - 1, ne = 2 + n0 + n2;
  auto entry = [&](const int e) -> double {
    if (e == 0) return PRIOR + (seed_char == hs[0] ? sc : sw) + l_prob + lastR[hapsize - 2];
    if (e == 1) return PRIOR + (seed_char == hs[hapsize - 1] ? sc : sw) + r_prob + lastL[hapsize - 2];
    int k = e - 2, coord, lrow, rrow;
    if (k < n0) { coord = 1 + k; lrow = k; rrow = hapsize - 3 - k; }
    else { k -= n0; coord = hf.len[0] + hf.len[1] + k; lrow = n0 + hf.len[1] + k; rrow = hapsize - 3 - n0 - hf.len[1] - k; }
    return PRIOR + (seed_char == hs[coord] ? sc : sw) + lastL[lrow] + lastR[rrow];
  };
  double mx = -1.7976931348623157e308;
  for (int e = lane; e < ne; e += 64) { const double v = entry(e); if (mx < v) mx = v; }
  mx = w_max(mx, 64);
  double total = 0;
  for (int e = lane; e < ne; e += 64) { const double diff = entry(e) - mx; if (diff > A.log_thresh) total += d_fasterexp((float)diff); }
  total = w_sum(total, 64);                                             // (exact: see the header of this section)
  if (lane == 0) A.out[A.pair_out[p]] = mx + d_fasterlog((float)total); // :230
}

// ---- host prep -------------------------------------------------------------------------------

// HapAligner::calc_best_seed_position, HapAligner.cpp:467-493
void best_seed_position(const std::vector<int32_t>& rs, const std::vector<int32_t>& re, int32_t region_start, int32_t region_end,
                        int32_t* best_dist, int32_t* best_pos) {
  *best_dist = *best_pos = -1;
  int32_t pos = region_start;
  size_t ri = 0;
  while (ri < rs.size() && pos <= region_end) {
    if (pos < rs[ri]) {
      const int32_t dist = 1 + (std::min(region_end, rs[ri] - 1) - pos) / 2;
      if (dist >= *best_dist) { *best_dist = dist; *best_pos = dist - 1 + pos; }
      pos = re[ri++];
    } else if (pos < re[ri]) pos = re[ri++];
    else ri++;
  }
  if (pos <= region_end) {
    const int32_t dist = 1 + (region_end - pos) / 2;
    if (dist >= *best_dist) { *best_dist = dist; *best_pos = dist - 1 + pos; }
  }
}

// HapAligner::calc_seed_base, HapAligner.cpp:494-542; -2 = CIGAR op the reference dies on
int calc_seed_base(const ltr_alignment* aln, const ltr_haplotype_blocks* hap) {
  std::vector<int32_t> rs, re;
  for (int b = 0; b < hap->n_blocks; b++) if (hap->is_repeat[b]) { rs.push_back(hap->block_start[b]); re.push_back(hap->block_end[b]); }
  const int32_t first_start = hap->block_start[0], last_end = hap->block_end[hap->n_blocks - 1];
  int32_t pos = aln->start;
  int best_seed = -1, cur_base = 0, max_dist = 5;                      // MIN_SEED_DIST, :17
  for (int k = 0; k < aln->n_cigar; k++) {
    const int num = aln->cigar_num[k];
    switch (aln->cigar_type[k]) {
      case '=': {
        const int32_t min_region = std::max(pos, first_start), max_region = std::min(pos + num - 1, last_end - 1);
        if (min_region <= max_region) {
          int32_t distance, dist_pos;
          best_seed_position(rs, re, min_region, max_region, &distance, &dist_pos);
          if (distance >= max_dist) { max_dist = distance; best_seed = cur_base + (dist_pos - pos); }
        }
        pos += num; cur_base += num; break;
      }
      case 'I': cur_base += num; break;
      case 'X': pos += num; cur_base += num; break;
      case 'D': pos += num; break;
      default: return -2;
    }
  }
  if (best_seed < -1 || best_seed == 0 || best_seed >= aln->seq_len - 1) return -1;
  return best_seed;
}

struct StutterLogs { double in_nostep, in_step, in_up, in_down, equal, out_nostep, out_step, out_up, out_down; };
// StutterModel::log_stutter_pmf, stutter_model.cpp:29-53
double stutter_pmf(const StutterLogs& s, int motif_len, int sample_bps, int read_bps) {
  const int bp_diff = read_bps - sample_bps;
  if (bp_diff % motif_len != 0) {
    const int eff = bp_diff - (bp_diff / motif_len);
    return eff < 0 ? s.out_down + s.out_nostep + s.out_step * (-eff - 1) : s.out_up + s.out_nostep + s.out_step * (eff - 1);
  }
  const int rep = bp_diff / motif_len;
  if (rep == 0) return s.equal;
  return rep < 0 ? s.in_down + s.in_nostep + s.in_step * (-rep - 1) : s.in_up + s.in_nostep + s.in_step * (rep - 1);
}

}  // namespace

namespace ltr {

#define S_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_)); rc = LTR_ERR_HIP; goto done; } } while (0)

// Host-side accumulator of the seeded path: loci are prepared one by one (short_batch_add) and
// scored together in ONE launch (short_batch_run) -- a pair is one lane running a sequential
// recurrence, so the only parallelism is across pairs, and one locus has ~100 of them.
struct ShortBatch {
  std::vector<ShortRead> reads; std::vector<ShortHap> fw, rv;
  std::vector<uint8_t> rbytes, hbytes; std::vector<int32_t> upstream;
  std::vector<double> art;
  std::vector<uint8_t> qidx;               // per base: index into BaseQuality's tables, in the order the kernels read wrong / correct
  int64_t n_cum = 0;
  std::vector<int32_t> pread, phap;
  std::vector<double*> pdst;               // where every pair's result goes on the host
  int period = 0, maxS = 1, maxHS = 1, maxB = 1;
};
ShortBatch* short_batch_new() { return new ShortBatch(); }
// dst += src (src is left empty): loci are prepared one batch each on the host's cores and strung together here --
// offsets into the byte / table arrays and the pairs' read / haplotype indices move by what dst already holds.
int short_batch_merge(ltr_ctx* ctx, ShortBatch* dst, ShortBatch* src) {
  if (src->reads.empty() && src->pread.empty()) return LTR_OK;
  if (dst->period == 0) dst->period = src->period;
  if (src->period != 0 && dst->period != src->period) { set_error(ctx, "short path: loci of one batch must share the repeat period"); return LTR_ERR_UNSUPPORTED; }
  const int64_t rb0 = (int64_t)dst->rbytes.size(), hb0 = (int64_t)dst->hbytes.size(), q0 = (int64_t)dst->qidx.size(), c0 = dst->n_cum;
  const int64_t up0 = (int64_t)dst->upstream.size();
  const int32_t r0 = (int32_t)dst->reads.size(), h0 = (int32_t)dst->fw.size();
  if ((uint64_t)up0 + src->upstream.size() > 0x7fffff00u || (uint64_t)r0 + src->reads.size() > 0x7fffff00u) { set_error(ctx, "short path: batch too large"); return LTR_ERR_INVALID; }
  for (ShortRead sr : src->reads) { sr.seq_off += rb0; sr.rev_off += rb0; sr.q_off += q0; sr.cum_off += c0; dst->reads.push_back(sr); }
  for (ShortHap h : src->fw) { h.seq_off += hb0; h.up_off += (int32_t)up0; dst->fw.push_back(h); }
  for (ShortHap h : src->rv) { h.seq_off += hb0; h.up_off += (int32_t)up0; dst->rv.push_back(h); }
  dst->rbytes.insert(dst->rbytes.end(), src->rbytes.begin(), src->rbytes.end());
  dst->hbytes.insert(dst->hbytes.end(), src->hbytes.begin(), src->hbytes.end());
  dst->qidx.insert(dst->qidx.end(), src->qidx.begin(), src->qidx.end());
  dst->upstream.insert(dst->upstream.end(), src->upstream.begin(), src->upstream.end());
  dst->art.insert(dst->art.end(), src->art.begin(), src->art.end());
  for (int32_t v : src->pread) dst->pread.push_back(v + r0);
  for (int32_t v : src->phap) dst->phap.push_back(v + h0);
  dst->pdst.insert(dst->pdst.end(), src->pdst.begin(), src->pdst.end());
  dst->n_cum += src->n_cum;
  dst->maxS = std::max(dst->maxS, src->maxS); dst->maxHS = std::max(dst->maxHS, src->maxHS); dst->maxB = std::max(dst->maxB, src->maxB);
  *src = ShortBatch();
  return LTR_OK;
}
void short_batch_free(ShortBatch* b) { delete b; }

// HapAligner::process_reads with short_ == 1 (HapAligner.cpp:545-581), host half, for one locus:
// seeds and the all-zero rows of seedless reads are written at once, the pairs are queued.
// aln_probs must stay valid until short_batch_run.
int short_batch_add(ltr_ctx* ctx, ShortBatch* B, const ltr_haplotype_blocks* hap, const uint8_t* realign_to_hap,
                    const ltr_alignment* alns, int32_t n_alns, int32_t init_read_index,
                    const uint8_t* realign_read, double* aln_probs, int32_t* seed_positions) {
  if (hap->n_blocks != 3 || hap->is_repeat[0] || !hap->is_repeat[1] || hap->is_repeat[2]) {
    set_error(ctx, "short path: expected [flank][repeat][flank] blocks (Haplotype.cpp:8 asserts the same)");
    return LTR_ERR_UNSUPPORTED;
  }
  const ltr_stutter_params sp = ctx_stutter_params(ctx);
  const int period = hap->period[1];
  if (B->period == 0) B->period = period;
  if (B->period != period) { set_error(ctx, "short path: loci of one batch must share the repeat period"); return LTR_ERR_UNSUPPORTED; }
  std::vector<int32_t> counts; int64_t H = 0;
  int rc = haplotype_counts(hap, &counts, &H);
  if (rc != LTR_OK) return rc;

  // ---- reads: seeds, quality table indices (BaseQuality, base_quality.h:45-75; the tables themselves: short_batch_run) ----
  auto qidx = [](uint8_t q) { const char c = (char)q; return c < '!' ? 0 : (c > 'J' ? 'J' - '!' : c - '!'); };
  std::vector<int32_t> read_of_aln((size_t)n_alns, -1);
  double* prob_ptr = aln_probs + (int64_t)init_read_index * H;
  const size_t reads0 = B->reads.size();
  for (int32_t r = 0; r < n_alns; r++) {
    if (realign_read && !realign_read[r]) continue;
    const ltr_alignment& a = alns[r];
    const int seed = calc_seed_base(&a, hap);                                     // :568
    if (seed == -2) { set_error(ctx, "Unrecognized CIGAR char in calc_seed_base()"); return LTR_ERR_CIGAR; }
    seed_positions[init_read_index + r] = seed;
    if (seed == -1) { for (int64_t k = 0; k < H; k++) prob_ptr[(int64_t)r * H + k] = 0; continue; }   // :570-574
    if (!a.qual) { set_error(ctx, "short path needs base qualities (ltr_alignment.qual)"); return LTR_ERR_INVALID; }
    ShortRead sr; sr.len = a.seq_len; sr.seed = seed;
    const int len = a.seq_len;
    sr.seq_off = (int64_t)B->rbytes.size(); sr.rev_off = sr.seq_off + len;
    B->rbytes.resize(B->rbytes.size() + (size_t)len + (size_t)(len - 1 - seed));
    {
      uint8_t* fwd = B->rbytes.data() + sr.seq_off; uint8_t* rev = fwd + len;
      std::memcpy(fwd, a.seq, (size_t)len);
      for (int j = len - 1; j > seed; j--) *rev++ = a.seq[j];                     // rev_rseq, :887-888
    }
    sr.q_off = (int64_t)B->qidx.size(); sr.cum_off = B->n_cum;
    B->qidx.resize(B->qidx.size() + (size_t)len); B->n_cum += (int64_t)len + 1;
    {
      uint8_t* qi = B->qidx.data() + sr.q_off;
      for (int j = 0; j <= seed; j++) *qi++ = (uint8_t)qidx(a.qual[j]);
      for (int j = len - 1; j > seed; j--) *qi++ = (uint8_t)qidx(a.qual[j]);      // :889-890
    }
    read_of_aln[(size_t)r] = (int32_t)B->reads.size();
    B->reads.push_back(sr);
    B->maxS = std::max(B->maxS, std::max(seed, a.seq_len - seed - 1));
  }
  if (B->reads.size() == reads0) return LTR_OK;

  // ---- haplotype combinations, both directions -------------------------------------------
  StutterLogs sl;
  sl.in_step = std::log(1 - sp.in_geom); sl.in_nostep = std::log(sp.in_geom); sl.in_up = std::log(sp.in_up); sl.in_down = std::log(sp.in_down);
  sl.out_step = std::log(1 - sp.out_geom); sl.out_nostep = std::log(sp.out_geom); sl.out_up = std::log(sp.out_up); sl.out_down = std::log(sp.out_down);
  sl.equal = std::log(1 - sp.in_up - sp.in_down - sp.out_up - sp.out_down);
  const size_t hap0 = B->fw.size();
  B->fw.resize(hap0 + (size_t)H); B->rv.resize(hap0 + (size_t)H); B->art.resize((hap0 + (size_t)H) * kNumArt);
  std::vector<int32_t>& upstream = B->upstream;
  auto slot = [&](int b, int al) { int64_t k = 0; for (int q = 0; q < b; q++) k += hap->n_alleles[q]; return k + al; };
  auto add_upstream = [&](const std::vector<uint8_t>& blk, ShortHap* h) {      // StutterAlignerClass ctor, .h:45-79
    const int len = (int)blk.size();
    int nd = kMaxDel;
    while (nd * period > len) nd--;
    h->num_deletions = nd; h->up_off = (int32_t)upstream.size();
    auto one = [&](int per) {                                                    // num_upstream_matches, .h:34-41
      const size_t base = upstream.size();
      upstream.resize(base + (size_t)std::max(len, 1), 0);
      for (int i = per; i < len; i++) upstream[base + i] = (blk[i - per] != blk[i]) ? 0 : 1 + upstream[base + i - 1];
    };
    for (int i = 1; i <= nd; i++) one(i * period);
    if (nd == 0) one(period);
  };
  for (int64_t k = 0; k < H; k++) {
    std::vector<uint8_t> blk[3];
    for (int b = 0; b < 3; b++) {
      const int64_t s = slot(b, counts[(size_t)(k * 3 + b)]);
      blk[b].assign(hap->allele_bytes + hap->allele_off[s], hap->allele_bytes + hap->allele_off[s + 1]);
    }
    if (blk[0].empty() || blk[2].empty()) { set_error(ctx, "short path: empty flank block"); return LTR_ERR_INVALID; }
    ShortHap& f = B->fw[hap0 + (size_t)k]; ShortHap& r = B->rv[hap0 + (size_t)k];
    f.seq_off = (int64_t)B->hbytes.size(); f.pad = 0;
    for (int b = 0; b < 3; b++) { f.len[b] = (int32_t)blk[b].size(); B->hbytes.insert(B->hbytes.end(), blk[b].begin(), blk[b].end()); }
    add_upstream(blk[1], &f);
    r.seq_off = (int64_t)B->hbytes.size(); r.pad = 0;                           // Haplotype::reverse: blocks and bases reversed
    for (int b = 0; b < 3; b++) { std::vector<uint8_t> t(blk[2 - b].rbegin(), blk[2 - b].rend()); r.len[b] = (int32_t)t.size(); B->hbytes.insert(B->hbytes.end(), t.begin(), t.end()); if (b == 1) add_upstream(t, &r); }
    const int bl = (int)blk[1].size();
    for (int q = 0; q < kNumArt; q++) {                                          // log_prob_pcr_artifact, RepeatStutterInfo.h:53-61
      const int asz = (q - kMaxDel) * period, read_size = bl + asz;
      double v;
      if (asz == 0) v = stutter_pmf(sl, period, bl, read_size);
      else if (asz > 0) v = stutter_pmf(sl, period, bl, read_size);              // asz <= max_ins always
      else v = (read_size < 0) ? -10e6 : stutter_pmf(sl, period, bl, read_size);
      B->art[(hap0 + (size_t)k) * kNumArt + q] = v;
    }
    B->maxHS = std::max(B->maxHS, (int)(blk[0].size() + blk[1].size() + blk[2].size()));
    B->maxB = std::max(B->maxB, bl);
  }
  if (upstream.size() > 0x7fffff00u) { set_error(ctx, "short path: batch too large"); return LTR_ERR_INVALID; }

  // ---- pairs -----------------------------------------------------------------------------
  for (int32_t r = 0; r < n_alns; r++) {
    if (read_of_aln[(size_t)r] < 0) continue;
    for (int64_t k = 0; k < H; k++) {
      if (realign_to_hap && !realign_to_hap[k]) continue;                        // :896-900
      B->pread.push_back(read_of_aln[(size_t)r]); B->phap.push_back((int32_t)(hap0 + (size_t)k));
      B->pdst.push_back(prob_ptr + (int64_t)r * H + k);
    }
  }
  return LTR_OK;
}

// One launch for everything queued; results go to the queued host destinations.
int short_batch_run(ltr_ctx* ctx, ShortBatch* B) {
  const int64_t n_pairs64 = (int64_t)B->pread.size();
  if (n_pairs64 == 0) return LTR_OK;
  if (n_pairs64 > 0x7fffffff) { set_error(ctx, "short path: too many pairs in one batch"); return LTR_ERR_INVALID; }
  const int n_pairs = (int)n_pairs64;
  const ltr_align_params prm = ctx_params(ctx);
  int rc = LTR_OK;
  std::vector<double> int_log((size_t)B->maxHS + B->maxB + 16);
  int_log[0] = -1000;                                                            // mathops.cpp:17
  for (size_t i = 1; i < int_log.size(); i++) int_log[i] = std::log((double)i);
  double qtab[128];                                                               // BaseQuality's tables, base_quality.h:29-43
  {
    const int MAXQ = 'J' - '!';
    std::memset(qtab, 0, sizeof(qtab));
    qtab[64] = -100; qtab[0] = 0;
    for (int i = 1; i <= MAXQ; ++i) { qtab[64 + i] = std::log(1.0 - std::pow(10.0, i / (-10.0))); qtab[i] = std::log(std::pow(10.0, i / (-10.0) / 5.0)); }
  }
  std::vector<int64_t> pout((size_t)n_pairs);
  for (int q = 0; q < n_pairs; q++) pout[(size_t)q] = q;

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool trace = ctx_debug(ctx).trace != 0;
  const auto t_start = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (trace) std::fprintf(stderr, "[ltr] short_batch_run %8.2f ms: %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), what);
  };
  ShortArgs A; std::memset(&A, 0, sizeof(A));
  void* d[24] = {nullptr}; int nd_alloc = 0;
  std::vector<double> out((size_t)n_pairs, 0.0);
  hipStream_t st = (hipStream_t)ctx_stream(ctx);
  const int S = std::max(B->maxS, std::max(B->maxB + 2, kNumArt)) + 2, HS = B->maxHS + 4;
  const int LP = std::max(S, HS) + 8;
  const int64_t per_block = (int64_t)(19 * S + LP + 2 * (HS + 2)) * 64;
  // one wave per block; enough blocks to keep a few waves per SIMD busy with other pairs while a
  // lane waits for its work arrays (scratch capped at 8 GB)
  const int64_t cap_blocks = std::max<int64_t>(1, ((int64_t)8 << 30) / (per_block * (int64_t)sizeof(double)));
  const int grid = (int)std::min<int64_t>(std::min<int64_t>((n_pairs + 63) / 64, 4096), cap_blocks);
  (void)hipSetDevice(ctx_device(ctx));
  auto up = [&](const void* src, size_t bytes, void** dst) -> hipError_t {
    hipError_t e = (hipError_t)ctx_pool_alloc(ctx, dst, std::max<size_t>(bytes, 8) + 64);      // (the context's pool: no hipMalloc / hipFree per call)
    if (e != hipSuccess) return e;
    d[nd_alloc++] = *dst;
    return bytes ? hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess;
  };
  void *p_reads, *p_fw, *p_rv, *p_rb, *p_hb, *p_up, *p_w, *p_c, *p_art, *p_il, *p_pr, *p_ph, *p_po, *p_out, *p_scr = nullptr, *p_cum, *p_qi, *p_qt;
  // The four-launch path whenever a side fits 64 lanes x 8 read positions and the block kernel's tables fit 64 KB of LDS;
  // beyond that (reads cut wider than the reference's +-200 bp) the lane-per-pair kernel with its global work arrays.
  const int maxS = std::max(B->maxS, 1);
  const size_t lds_bytes = ((size_t)4 * S + int_log.size()) * sizeof(double) + ((size_t)kMaxDel * B->maxB + 8) * sizeof(int32_t) +
                           (size_t)((S + 7) & ~7) + (size_t)HS + 64;
  const bool wave_kernel = maxS <= 64 * 8 && lds_bytes <= 64 * 1024 && !ctx_debug(ctx).short_lane_kernel;
  // pairs per set of launches: the block row's terms (13 x S doubles per side) fit ONE GB -- a block the context's pool keeps
  // between calls (beyond 2 GB a block is a hipMalloc / hipFree per call: 15-20 ms each on MI355X, and a device-wide wait)
  const size_t terms_side_bytes = (size_t)kNumArt * S * sizeof(double);
  const int chunk_cap = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_pairs, (((size_t)1 << 30) - 64) / (2 * terms_side_bytes)));
  void *p_row = nullptr, *p_last = nullptr, *p_terms = nullptr;
  mark("start");
  S_TRY(up(B->reads.data(), B->reads.size() * sizeof(ShortRead), &p_reads));
  S_TRY(up(B->fw.data(), B->fw.size() * sizeof(ShortHap), &p_fw));
  S_TRY(up(B->rv.data(), B->rv.size() * sizeof(ShortHap), &p_rv));
  S_TRY(up(B->rbytes.data(), B->rbytes.size(), &p_rb));
  S_TRY(up(B->hbytes.data(), B->hbytes.size(), &p_hb));
  S_TRY(up(B->upstream.data(), B->upstream.size() * sizeof(int32_t), &p_up));
  mark("tables and pairs uploaded");
  S_TRY(up(B->qidx.data(), B->qidx.size(), &p_qi));
  S_TRY(up(qtab, sizeof(qtab), &p_qt));
  S_TRY((hipError_t)ctx_pool_alloc(ctx, &p_w, B->qidx.size() * sizeof(double) + 64)); d[nd_alloc++] = p_w;
  S_TRY((hipError_t)ctx_pool_alloc(ctx, &p_c, B->qidx.size() * sizeof(double) + 64)); d[nd_alloc++] = p_c;
  S_TRY((hipError_t)ctx_pool_alloc(ctx, &p_cum, (size_t)B->n_cum * sizeof(double) + 64)); d[nd_alloc++] = p_cum;
  S_TRY(up(B->art.data(), B->art.size() * sizeof(double), &p_art));
  S_TRY(up(int_log.data(), int_log.size() * sizeof(double), &p_il));
  S_TRY(up(B->pread.data(), B->pread.size() * sizeof(int32_t), &p_pr));
  S_TRY(up(B->phap.data(), B->phap.size() * sizeof(int32_t), &p_ph));
  S_TRY(up(pout.data(), pout.size() * sizeof(int64_t), &p_po));
  S_TRY((hipError_t)ctx_pool_alloc(ctx, &p_out, out.size() * sizeof(double) + 64)); d[nd_alloc++] = p_out;
  if (!wave_kernel) { S_TRY((hipError_t)ctx_pool_alloc(ctx, &p_scr, (size_t)grid * per_block * sizeof(double))); d[nd_alloc++] = p_scr; }
  mark("uploads queued (+ work arrays allocated)");
  A.reads = (const ShortRead*)p_reads; A.fw = (const ShortHap*)p_fw; A.rv = (const ShortHap*)p_rv;
  A.read_bytes = (const uint8_t*)p_rb; A.hap_bytes = (const uint8_t*)p_hb; A.upstream = (const int32_t*)p_up;
  A.wrong = (const double*)p_w; A.correct = (const double*)p_c; A.cum = (const double*)p_cum; A.art = (const double*)p_art; A.int_log = (const double*)p_il;
  A.pair_read = (const int32_t*)p_pr; A.pair_hap = (const int32_t*)p_ph; A.pair_out = (const int64_t*)p_po;
  A.n_pairs = n_pairs; A.period = B->period; A.out = (double*)p_out; A.scratch = (double*)p_scr; A.scratch_per_block = per_block;
  A.S = S; A.HS = HS; A.LP = LP; A.n_ilog = (int32_t)int_log.size(); A.maxB = B->maxB;
  A.a = prm.log_ins_to_ins; A.b = prm.log_ins_to_match; A.c = prm.log_del_to_del; A.d = prm.log_del_to_match;
  A.e = prm.log_match_to_match; A.f = prm.log_match_to_ins; A.g = prm.log_match_to_del;
  A.log_thresh = std::log(0.001);
  A.qidx = (const uint8_t*)p_qi; A.qtab = (const double*)p_qt; A.wrong_w = (double*)p_w; A.correct_w = (double*)p_c; A.cum_w = (double*)p_cum;
  A.n_reads = (int32_t)B->reads.size();
  S_TRY(hipEventCreate(&ev0)); S_TRY(hipEventCreate(&ev1));
  S_TRY(hipEventRecord(ev0, st));
  hipLaunchKernelGGL(ltr_short_prep_kernel, dim3((unsigned)((B->reads.size() + kShortThreads / 64 - 1) / (kShortThreads / 64))), dim3(kShortThreads), 0, st, A);
  if (wave_kernel) {
    S_TRY((hipError_t)ctx_pool_alloc(ctx, &p_row, (size_t)chunk_cap * 2 * S * sizeof(double) + 64)); d[nd_alloc++] = p_row;
    S_TRY((hipError_t)ctx_pool_alloc(ctx, &p_last, (size_t)chunk_cap * 2 * (HS + 2) * sizeof(double) + 64)); d[nd_alloc++] = p_last;
    S_TRY((hipError_t)ctx_pool_alloc(ctx, &p_terms, (size_t)chunk_cap * 2 * kNumArt * S * sizeof(double) + 64)); d[nd_alloc++] = p_terms;
    A.g_row = (double*)p_row; A.g_last = (double*)p_last; A.g_terms = (double*)p_terms;
    // (ltr_ctx_set_debug "short_split": events between the five launches -- the seeded path's time kernel by kernel, ltr_ctx_short_kernel_split)
    const bool split = ctx_debug(ctx).short_split != 0;
    hipEvent_t evs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (split) { for (hipEvent_t& e : evs) S_TRY(hipEventCreate(&e)); S_TRY(hipEventRecord(evs[0], st)); }
    for (int first = 0; first < n_pairs; first += chunk_cap) {
      A.chunk_first = first; A.chunk_pairs = std::min(chunk_cap, n_pairs - first);
      const unsigned side_blocks = (unsigned)((2 * (int64_t)A.chunk_pairs + kShortThreads / 64 - 1) / (kShortThreads / 64));
      const unsigned pair_blocks = (unsigned)(((int64_t)A.chunk_pairs + kShortThreads / 64 - 1) / (kShortThreads / 64));
      const unsigned block_grid = (unsigned)(2 * (int64_t)A.chunk_pairs);      // (one workgroup per (pair, side): the sides differ tenfold in length; the dispatcher balances them)
      const bool sp = split && first == 0;                         // (the first chunk: every bounded workload is one chunk)
      hipLaunchKernelGGL((ltr_short_flank_kernel<false>), dim3(side_blocks), dim3(kShortThreads), 0, st, A);
      if (sp) S_TRY(hipEventRecord(evs[1], st));
      hipLaunchKernelGGL(ltr_short_block_kernel, dim3(block_grid), dim3(kShortThreads), lds_bytes, st, A);
      if (sp) S_TRY(hipEventRecord(evs[2], st));
      hipLaunchKernelGGL((ltr_short_flank_kernel<true>), dim3(side_blocks), dim3(kShortThreads), 0, st, A);
      if (sp) S_TRY(hipEventRecord(evs[3], st));
      hipLaunchKernelGGL(ltr_short_final_kernel, dim3(pair_blocks), dim3(kShortThreads), 0, st, A);
      if (sp) S_TRY(hipEventRecord(evs[4], st));
    }
    if (split) {
      S_TRY(hipStreamSynchronize(st));
      double ms4[4] = {0, 0, 0, 0};
      for (int k = 0; k < 4; ++k) { float ms = 0.f; if (hipEventElapsedTime(&ms, evs[k], evs[k + 1]) == hipSuccess) ms4[k] = (double)ms; }
      ctx_note_short_split(ctx, ms4);                              // [prep + flank rows before the block, block row, flank rows after, seed log-sum]
      for (hipEvent_t& e : evs) if (e) (void)hipEventDestroy(e);
    }
  } else
  hipLaunchKernelGGL(ltr_short_kernel, dim3((unsigned)grid), dim3(64), 0, st, A);
  S_TRY(hipGetLastError());
  S_TRY(hipEventRecord(ev1, st));
  mark("launches queued");
  S_TRY(hipMemcpyAsync(out.data(), p_out, out.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  S_TRY(hipStreamSynchronize(st));
  mark("scores back");
  { float ms = 0.f; if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) add_time(ctx, kTimerShortKernel, 0.0, (double)ms); }
  for (int q = 0; q < n_pairs; q++) *B->pdst[(size_t)q] = out[(size_t)q];
done:
  if (rc != LTR_OK) (void)hipStreamSynchronize(st);                              // (nothing in flight may still use the blocks)
  for (int i = 0; i < nd_alloc; i++) ctx_pool_release(ctx, d[i]);
  if (ev0) (void)hipEventDestroy(ev0);
  if (ev1) (void)hipEventDestroy(ev1);
  return rc;
}

// HapAligner::process_reads with short_ == 1 (HapAligner.cpp:545-581) for one locus.
int process_reads_short(ltr_ctx* ctx, const ltr_haplotype_blocks* hap, const uint8_t* realign_to_hap,
                        const ltr_alignment* alns, int32_t n_alns, int32_t init_read_index,
                        const uint8_t* realign_read, double* aln_probs, int32_t* seed_positions) {
  ShortBatch B;
  int rc = short_batch_add(ctx, &B, hap, realign_to_hap, alns, n_alns, init_read_index, realign_read, aln_probs, seed_positions);
  if (rc == LTR_OK) rc = short_batch_run(ctx, &B);
  return rc;
}

}  // namespace ltr

// test hook (include/ltr_gpu.h): the host calc_seed_base above, for the pin against the compiled reference
extern "C" int ltr_debug_calc_seed_base(const ltr_alignment* aln, const ltr_haplotype_blocks* hap) {
  if (!aln || !hap || hap->n_blocks <= 0) return LTR_ERR_INVALID;
  return calc_seed_base(aln, hap);
}

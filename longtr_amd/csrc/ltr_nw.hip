// ltr_nw.hip -- haplotype -> reference-haplotype alignment on the GPU.
//
// Replaces the work of Haplotype::aln_haps_to_ref (reference src/SeqAlignment/Haplotype.cpp:58-86):
// for every candidate haplotype of a locus, NeedlemanWunsch::Align(ref haplotype, alt haplotype,
// use_ref_end_penalty = true) (NeedlemanWunsch.cpp:380-420: three float score matrices M / Iref /
// Iread with match 2, mismatch -2, gap open 5, gap extend 0.125 (:82-96), three trace matrices,
// end point at the last cell (:174-193), traceback (:247-338)), Haplotype::adjust_indels (:8-56)
// and the M / I / D string hap_aln_info_ is made of (:72-82).
//
// Who reads hap_aln_info_ in the reference: Haplotype::get_aln_info (Haplotype.h:65), whose only
// caller is HapAligner::process_read's retrace branch (HapAligner.cpp:969, stitch_alignment_trace --
// the short / stutter path with tracing, itself unreachable because HapAligner::retrace is gutted,
// :601-810), and Haplotype::reverse, which copies it (Haplotype.cpp:304-305).  The long-read path
// never reads it: a host that drives this library through ltr_process_reads / ltr_calc_hap_aln_probs
// simply does not need it, and a host that keeps the reference's Haplotype class can take it from
// here instead of paying (TR + 70)^2 cells per candidate on one CPU thread (SURVEY.md finding 5).
//
// Kernel: one workgroup per (reference, alternate) pair, anti-diagonal sweep -- cell (i, j) of the
// three matrices needs (i-1, j-1), (i, j-1) and (i-1, j), all on the two previous anti-diagonals,
// kept as rolling rows (LDS when the alternate is short enough, a global scratch strip otherwise);
// one barrier per anti-diagonal; one byte of trace per cell (2 bits per matrix) in global memory;
// the traceback is walked by one lane.  Every score is a sum of multiples of 0.125 far below 2^24:
// float arithmetic is exact, so the trace choices are the reference's whatever the evaluation order.
// HBM-light and latency-bound by design (it is the cold neighbour of the DP, not the hot path).

#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "ltr_internal.h"

namespace {

constexpr float kMatch = 2.0f, kMismatch = -2.0f, kGapOpen = 5.0f, kGapExtend = 0.125f, kLarge = 1000000.0f;   // NeedlemanWunsch.cpp:84-96
constexpr int kNwThreads = 256;
constexpr int kNwLdsRows = 1536;                               // alternates up to this length keep the rolling diagonals in LDS (54 KB)

struct NwTask {
  int64_t ref_off, alt_off;        // byte offsets into the sequence pool
  int64_t out_off;                 // offset of this task's alignment in the output: one byte per column, back to front (L1 + L2 bytes at most):
                                   // 0 = a reference base over an alternate base, 1 = a reference base over a gap, 2 = a gap over an alternate base
  int32_t L1, L2;                  // reference / alternate length
};

// base_to_int (NeedlemanWunsch.cpp:100-119): A C G T -> 0..3, anything else behaves like N (matches everything)
__device__ __forceinline__ int base_code(uint8_t c) {
  if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
  return c == 'A' ? 0 : (c == 'C' ? 1 : (c == 'G' ? 2 : (c == 'T' ? 3 : 4)));
}
// bestIndex (:121-143): the order of the comparisons is the tie-breaking rule
__device__ __forceinline__ float best_index(float s1, float s2, float s3, int* c) {
  if (s2 > s1) { if (s2 > s3) { *c = 1; return s2; } *c = 2; return s3; }
  if (s3 > s1) { *c = 2; return s3; }
  *c = 0; return s1;
}

// base_to_int (NeedlemanWunsch.cpp:100-119) as a 4-bit mask, for the whole pool of sequences (on the host this loop was a
// serial pass over 20 MB per 3000 loci)
__global__ void ltr_nw_mask_kernel(const uint8_t* __restrict__ seqs, uint8_t* __restrict__ masks, int64_t n) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    uint8_t c = seqs[k];
    if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
    masks[k] = c == 'A' ? 1 : (c == 'C' ? 2 : (c == 'G' ? 4 : (c == 'T' ? 8 : 15)));
  }
}

__global__ __launch_bounds__(kNwThreads) void ltr_nw_kernel(const NwTask* __restrict__ tasks, const int32_t* __restrict__ index, int n_tasks, uint32_t* queue,
                                                            const uint8_t* __restrict__ seqs, uint8_t* __restrict__ trace_pool,
                                                            int64_t trace_stride, float* __restrict__ diag_pool, int64_t diag_stride,
                                                            uint8_t* __restrict__ out, int32_t* __restrict__ out_len) {
  __shared__ float s_diag[3 * 3 * (kNwLdsRows + 1)];
  __shared__ int s_task;
  uint8_t* trace = trace_pool + (int64_t)blockIdx.x * trace_stride;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (;;) {
    __syncthreads();
    // (wave 0 pops under wave-uniform control flow, every lane issuing the add: see ltr_dp_wg.hpp)
    if (wave == 0) {
      int q = (int)atomicAdd(queue, lane == 0 ? 1u : 0u);
      q = __builtin_amdgcn_readfirstlane(q);
      if (lane == 0) *(volatile int*)&s_task = q;
    }
    __syncthreads();
    const int tq = __builtin_amdgcn_readfirstlane(*(volatile int*)&s_task);
    if (tq >= n_tasks) break;
    const int ti = index[tq];
    const NwTask T = tasks[ti];
    const int L1 = T.L1, L2 = T.L2;
    const uint8_t* ref = seqs + T.ref_off;
    const uint8_t* alt = seqs + T.alt_off;
    float* diag = (L2 <= kNwLdsRows) ? s_diag : diag_pool + (int64_t)blockIdx.x * diag_stride;
    const int rows = L2 + 1;
    // rolling anti-diagonals: value of matrix x at row i of diagonal d lives at diag[((d % 3) * 3 + x) * rows + i]
    for (int d = 0; d <= L1 + L2; ++d) {
      float* cur = diag + (size_t)((d % 3) * 3) * rows;
      const float* p1 = diag + (size_t)(((d + 2) % 3) * 3) * rows;     // diagonal d-1
      const float* p2 = diag + (size_t)(((d + 1) % 3) * 3) * rows;     // diagonal d-2
      const int i_lo = max(0, d - L1), i_hi = min(L2, d);
      for (int i = i_lo + (int)threadIdx.x; i <= i_hi; i += kNwThreads) {
        const int j = d - i;
        float m, ir, id;
        if (i == 0) {                                          // row 0 (initMatrices, :340-361; use_ref_end_penalty)
          m = (j == 0) ? 0.0f : -kLarge;
          ir = (j == 0) ? -kLarge : (-kGapOpen - (float)(j - 1) * kGapExtend);
          id = -kLarge;
        } else if (j == 0) {                                   // column 0 (:363-377)
          m = -kLarge; ir = -kLarge; id = -kGapOpen - (float)(i - 1) * kGapExtend;
        } else {                                               // nw_helper, :214-244
          int cm, cr, cd;
          const int rb = base_code(ref[j - 1]), ab = base_code(alt[i - 1]);
          const float sc = (rb == 4 || ab == 4 || rb == ab) ? kMatch : kMismatch;
          m = best_index(p2[0 * rows + i - 1], p2[1 * rows + i - 1], p2[2 * rows + i - 1], &cm) + sc;                               // (i-1, j-1)
          ir = best_index(p1[0 * rows + i] - kGapOpen, p1[1 * rows + i] - kGapExtend, p1[2 * rows + i] - kGapOpen, &cr);             // (i, j-1)
          id = best_index(p1[0 * rows + i - 1] - kGapOpen, p1[1 * rows + i - 1] - kGapOpen, p1[2 * rows + i - 1] - kGapExtend, &cd);   // (i-1, j)
          trace[(int64_t)i * (L1 + 1) + j] = (uint8_t)(cm | (cr << 2) | (cd << 4));
        }
        cur[0 * rows + i] = m; cur[1 * rows + i] = ir; cur[2 * rows + i] = id;
      }
      __syncthreads();
      if (diag != s_diag) __threadfence_block();
    }
    if (threadIdx.x == 0) {
      // findOptimalStopEndPenalty (:174-193): the alignment ends in the last cell
      const float* last = diag + (size_t)(((L1 + L2) % 3) * 3) * rows;
      int best_col = L1, best_row = L2, type = 0;
      float best = last[0 * rows + L2];
      if (last[1 * rows + L2] > best) { best = last[1 * rows + L2]; type = 1; }
      if (last[2 * rows + L2] > best) { best = last[2 * rows + L2]; type = 2; }
      // traceAlignment (:247-305), written back to front exactly like its stringstreams (the host reverses)
      uint8_t* ops = out + T.out_off;
      int n = 0;
      while (best_row > 0) {
        const uint8_t tr = (best_col > 0) ? trace[(int64_t)best_row * (L1 + 1) + best_col] : (uint8_t)(2 << 4);   // column 0: traceIread = 2 (:368)
        if (type == 0) { ops[n++] = 0; type = tr & 3; --best_row; --best_col; }
        else if (type == 1) { ops[n++] = 1; type = (tr >> 2) & 3; --best_col; }
        else { ops[n++] = 2; type = (tr >> 4) & 3; --best_row; }
      }
      for (int i = best_col; i > 0; --i) ops[n++] = 1;          // leading gaps, :307-310
      out_len[ti] = n;
    }
  }
}

// ---- wavefront kernel: references of up to 64 x 20 bases ------------------------------------------------
// One 64-lane wavefront per pair, no barrier and no LDS: lane l owns W consecutive reference columns, the
// alternate's rows stream through the lanes skewed by one row per lane (the geometry of ltr_dp_kernel.hpp):
// cell (i, j) takes (i-1, j) from the lane's own registers, (i, j-1) from the previous slot or -- slot 0 --
// from the left neighbour by DPP wave_shr:1, and (i-1, j-1) from what that hand-off delivered one step
// earlier.  bestIndex (NeedlemanWunsch.cpp:121-143) is a three-way maximum whose comparisons only decide the
// trace code: value = v_max3_f32, code from three strict compares.  Bases are 4-bit masks (A C G T = 1 2 4 8,
// anything else 15): "equal or either is N" is one AND.  The trace byte of cell (i, j) goes to
// ((t * 64 + lane) * Wp + slot) with t = i - 1 + lane: every step the wavefront stores one contiguous
// 64 * Wp-byte line.  The traceback is walked by the wavefront in step, a 16 x 16 tile of trace bytes per round of loads.  Measured on MI355X, 6976 haplotypes of 1000 config-3 loci (3.1e9
// cells): see profiles/r02/nw_rate.log (the workgroup-per-pair kernel below: 89 ms per call).
#ifndef LTR_NW_LB3_MAXW
#define LTR_NW_LB3_MAXW 16                  /* strips of up to this many columns are built for three waves per SIMD (168 registers) */
#endif
#ifndef LTR_NW_LB
#define LTR_NW_LB 2                         /* waves per SIMD the register allocator leaves room for; measured on MI355X, 28 k
                                               haplotypes of config 3: 25.3 ms of kernels at 2, 26.2 at 3, 33.4 at 4 (spills) */
#endif
constexpr int kNwWaveMaxW = 20;
constexpr int kNwWaveBlock = 4;                                 // wavefronts per workgroup (independent workers)

__device__ __forceinline__ float nw_shr1(float v, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}
// bestIndex (NeedlemanWunsch.cpp:120-141): the maximum, and which of the three it was under the reference's tie-breaking
// order -- s2 > s1 ? (s2 > s3 ? 1 : 2) : (s3 > s1 ? 2 : 0) -- computed as:
// the code already in its place in the trace word (SH = bit position); the first score wins every
// tie (code 0 whenever it IS the maximum); otherwise the second wins only if it beats the third strictly -- when
// s2 <= s1 < s3 that test is false as well.  max3 + two compares + two selects instead of three and three.
template <int SH>
__device__ __forceinline__ float nw_best_at(float s1, float s2, float s3, uint32_t* c) {
  const float best = fmaxf(s1, fmaxf(s2, s3));
  const uint32_t c12 = (s2 > s3) ? (1u << SH) : (2u << SH);
  *c = (s1 >= best) ? 0u : c12;
  return best;
}

// The W cells of one lane and one row (nw_helper, NeedlemanWunsch.cpp:214-244), slot S onwards: a compile-time recursion so
// that every trace code is formed at its final bit position (byte S & 3 of the word: M | Iref << 2 | Iread << 4).
// (rb: the 4-bit masks of my W reference bases, eight to a word -- W words of them were 16 - 20 registers of a kernel that runs at
// two waves per SIMD for want of registers; ab8: the alternate base's mask in every nibble: one AND per word finds the matches of
// eight slots, one AND + compare per slot picks its nibble -- the same two instructions per cell as `rb[S] & ab` before)
template <int W, int S>
__device__ __forceinline__ void nw_cells(const uint32_t (&rb)[(W + 7) / 8], const uint32_t ab8, float (&Mp)[W], float (&Rp)[W], float (&Dp)[W],
                                         float& gM, float& gR, float& gD, float& eM, float& eR, float& eD, uint32_t* tw, uint32_t word = 0, uint32_t hit = 0) {
  if constexpr (S < W) {
    constexpr int B = 8 * (S & 3);
    uint32_t cm, cr, cd;
    if ((S & 7) == 0) hit = rb[S >> 3] & ab8;
    const float sc = (hit & (0xFu << (4 * (S & 7)))) ? kMatch : kMismatch;
    const float m = nw_best_at<B>(gM, gR, gD, &cm) + sc;
    const float r = nw_best_at<B + 2>(eM - kGapOpen, eR - kGapExtend, eD - kGapOpen, &cr);
    const float d = nw_best_at<B + 4>(Mp[S] - kGapOpen, Rp[S] - kGapOpen, Dp[S] - kGapExtend, &cd);
    gM = Mp[S]; gR = Rp[S]; gD = Dp[S];
    Mp[S] = m; Rp[S] = r; Dp[S] = d;
    eM = m; eR = r; eD = d;
    word |= cm | cr | cd;
    if ((S & 3) == 3 || S == W - 1) { tw[S >> 2] = word; word = 0; }
    nw_cells<W, S + 1>(rb, ab8, Mp, Rp, Dp, gM, gR, gD, eM, eR, eD, tw, word, hit);
  }
}

template <int W>
__global__ __launch_bounds__(64 * kNwWaveBlock, (W <= LTR_NW_LB3_MAXW) ? 3 : LTR_NW_LB) void ltr_nw_wave_kernel(const NwTask* __restrict__ tasks, const int32_t* __restrict__ index,
                                                                       int n_tasks, uint32_t* queue, const uint8_t* __restrict__ seqs,
                                                                       const uint8_t* __restrict__ masks, uint8_t* __restrict__ trace_pool,
                                                                       int64_t trace_stride, uint8_t* __restrict__ out, int32_t* __restrict__ out_len) {
  constexpr int Wp = (W + 3) / 4 * 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* trace = trace_pool + ((int64_t)blockIdx.x * kNwWaveBlock + wave) * trace_stride;
  for (;;) {
    // (every lane issues the add, lane 0 adds 1: the pop stays under wave-uniform control flow, see ltr_dp_kernel.hpp)
    int q = (int)atomicAdd(queue, lane == 0 ? 1u : 0u);
    q = __builtin_amdgcn_readfirstlane(q);
    if (q >= n_tasks) break;
    const int ti = __builtin_amdgcn_readfirstlane(index[q]);
    const NwTask* tp = tasks + ti;
    const int L1 = __builtin_amdgcn_readfirstlane(tp->L1), L2 = __builtin_amdgcn_readfirstlane(tp->L2);
    const int64_t ref_off = tp->ref_off, alt_off = tp->alt_off, out_off = tp->out_off;
    const int lanes = (L1 + W - 1) / W;
    const int j0 = 1 + lane * W;                                 // my first column (1-based)
    uint32_t rb[(W + 7) / 8];
#pragma unroll
    for (int q = 0; q < (W + 7) / 8; ++q) rb[q] = 0;
    float Mp[W], Rp[W], Dp[W];                                   // row i-1 of my columns: M, Iref (gap in the alternate), Iread
#pragma unroll
    for (int s = 0; s < W; ++s) {
      const int j = j0 + s;
      rb[s >> 3] |= (uint32_t)masks[ref_off + min(j, L1) - 1] << (4 * (s & 7));
      Mp[s] = -kLarge; Rp[s] = -kGapOpen - (float)(j - 1) * kGapExtend; Dp[s] = -kLarge;   // row 0, initMatrices :340-361
    }
    // (i-1, j0-1): row 0 at my left border
    float dM = (lane == 0) ? 0.0f : -kLarge;
    float dR = (lane == 0) ? -kLarge : (-kGapOpen - (float)(j0 - 2) * kGapExtend);
    float dD = -kLarge;
    float oM = Mp[W - 1], oR = Rp[W - 1], oD = Dp[W - 1];        // what my right neighbour takes over
    const uint8_t* ap = masks + alt_off - lane;                  // ap[t] = mask of the alternate base of MY row at step t (the pool is padded)
    uint32_t a_next = ap[0];
    const int T = L2 + lanes - 1;
    for (int t = 0; t < T; ++t) {
      const int i = t - lane + 1;                                // my row (1-based)
      const uint32_t ab = a_next;
      a_next = ap[t + 1];
      // column 0 of row i (:363-377) for lane 0, the neighbour's last slot for the others
      const float lM = nw_shr1(oM, -kLarge), lR = nw_shr1(oR, -kLarge), lD = nw_shr1(oD, -kGapOpen - (float)(t) * kGapExtend);
      if (lane < lanes && i >= 1 && i <= L2) {
        float gM = dM, gR = dR, gD = dD;                         // (i-1, j-1)
        dM = lM; dR = lR; dD = lD;                               // ... of the next row
        float eM = lM, eR = lR, eD = lD;                         // (i, j-1)
        uint32_t* tw = (uint32_t*)(trace + ((int64_t)t * 64 + lane) * Wp);
        nw_cells<W, 0>(rb, ab * 0x11111111u, Mp, Rp, Dp, gM, gR, gD, eM, eR, eD, tw);
        oM = eM; oR = eR; oD = eD;
      }
    }
    // the last cell (L2, L1): findOptimalStopEndPenalty (:174-193)
    const int lo = (L1 - 1) / W, so = (L1 - 1) - lo * W;
    float fM = 0.f, fR = 0.f, fD = 0.f;
#pragma unroll
    for (int s = 0; s < W; ++s) if (s == so) { fM = Mp[s]; fR = Rp[s]; fD = Dp[s]; }
    fM = __shfl(fM, lo); fR = __shfl(fR, lo); fD = __shfl(fD, lo);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");        // the trace bytes of all lanes, visible to lane 0's loads
    {
      // traceAlignment (:247-305), written back to front exactly like its stringstreams (the host reverses).  The walk is a chain
      // of dependent look-ups; it is run by the whole wavefront in step (its state is wave-uniform: scalar registers), a 16 x 16
      // tile of trace bytes at a time: lane l loads the four cells (q * 64 + l) / 16 rows up and (q * 64 + l) % 16 columns left
      // of the tile's corner -- 256 independent loads, one memory latency -- and the walk reads its way through the tile by
      // v_readlane until it leaves it, 16 - 31 steps later (measured on MI355X, 21 k haplotypes of 3000 config-3 loci: one
      // dependent load per step by lane 0 was 3.6 of the kernels' 16.0 ms; with no walk at all they took 12.4).
      int type = 0;
      {
        float best = fM;
        if (fR > best) { best = fR; type = 1; }
        if (fD > best) { best = fD; type = 2; }
      }
      type = __builtin_amdgcn_readfirstlane(type);
      int best_col = L1, best_row = L2;
      uint8_t* ops = out + out_off;
      int n = 0;
      while (best_row > 0) {
        const int r0 = best_row, c0 = best_col;                  // the tile's corner
        uint32_t tile[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = q * 64 + lane, r = r0 - (e >> 4), c = c0 - (e & 15);
          tile[q] = 0;
          if (r >= 1 && c >= 1) {
            const int lc = (c - 1) / W, sl = (c - 1) - lc * W;
            tile[q] = trace[((int64_t)(r - 1 + lc) * 64 + lc) * Wp + sl];
          }
        }
        uint32_t mine = 0;                                       // lane k keeps the k-th operation of this tile's stretch
        int k = 0;
        while (best_row > 0 && r0 - best_row < 16 && c0 - best_col < 16) {
          uint32_t tr = 2u << 4;                                 // column 0: traceIread = 2 (:368)
          if (best_col > 0) {
            const int e = (r0 - best_row) * 16 + (c0 - best_col);
            const uint32_t v = (e < 128) ? ((e < 64) ? tile[0] : tile[1]) : ((e < 192) ? tile[2] : tile[3]);
            tr = (uint32_t)__builtin_amdgcn_readlane((int)v, e & 63);
          }
          uint32_t op;
          if (type == 0) { op = 0; type = (int)(tr & 3); --best_row; --best_col; }
          else if (type == 1) { op = 1; type = (int)((tr >> 2) & 3); --best_col; }
          else { op = 2; type = (int)((tr >> 4) & 3); --best_row; }
          if (lane == k) mine = op;
          ++k;
        }
        if (lane < k) ops[n + lane] = (uint8_t)mine;            // (at most 31 steps inside a tile)
        n += k;
      }
      for (int i = best_col - lane; i > 0; i -= 64) ops[n + best_col - i] = 1;   // leading gaps, :307-310
      n += best_col;
      if (lane == 0) out_len[ti] = n;
    }
  }
}

// Haplotype::adjust_indels (Haplotype.cpp:8-56): indels in the left flank slide right, into / up to the repeat block
void adjust_indels(std::string& ref_al, std::string& alt_al, int32_t ref_pos, const int32_t str_pos) {
  size_t aln = 0;
  while (aln < alt_al.size()) {
    if (alt_al[aln] == '-' && ref_pos < str_pos) {
      size_t index = aln;
      while (index < alt_al.size() && alt_al[index] == '-') ++index;
      int32_t pos = ref_pos;
      size_t del_index = aln;
      const int32_t del_size = (int32_t)(index - aln);
      while (index < alt_al.size() && pos < str_pos && ref_al[del_index] == ref_al[index]) {
        alt_al[del_index] = alt_al[index]; alt_al[index] = '-';
        ++index; ++del_index; ++pos;
      }
      aln = index; ref_pos = pos + del_size;
    } else if (ref_al[aln] == '-' && ref_pos < str_pos) {
      size_t index = aln;
      while (index < ref_al.size() && ref_al[index] == '-') ++index;
      int32_t pos = ref_pos;
      size_t ins_index = aln;
      while (index < ref_al.size() && pos < str_pos && alt_al[ins_index] == alt_al[index]) {
        ref_al[ins_index] = ref_al[index]; ref_al[index] = '-';
        ++index; ++ins_index; ++pos;
      }
      aln = index; ref_pos = pos;
    } else {
      if (ref_al[aln] != '-') ++ref_pos;
      ++aln;
    }
  }
}

}  // namespace

extern "C" {

// Upper bound of the bytes ltr_haplotype_align_to_ref writes for these loci (sum over haplotypes of
// reference length + haplotype length).
int64_t ltr_haplotype_aln_info_capacity(const ltr_haplotype_blocks* const* haps, int64_t n_loci) {
  if (!haps || n_loci < 0) return LTR_ERR_INVALID;
  int64_t cap = 0;
  for (int64_t l = 0; l < n_loci; ++l) {
    if (!haps[l]) return LTR_ERR_INVALID;
    std::vector<int32_t> counts; int64_t H = 0;
    const int rc = ltr::haplotype_counts(haps[l], &counts, &H);
    if (rc != LTR_OK) return rc;
    int64_t ref_len = 0, max_len = 0, k = 0;
    for (int b = 0; b < haps[l]->n_blocks; ++b) {
      int64_t mx = 0;
      for (int a = 0; a < haps[l]->n_alleles[b]; ++a, ++k) {
        const int64_t len = haps[l]->allele_off[k + 1] - haps[l]->allele_off[k];
        if (a == 0) ref_len += len;
        mx = std::max(mx, len);
      }
      max_len += mx;
    }
    cap += H * (ref_len + max_len);
  }
  return cap;
}

// Haplotype::aln_haps_to_ref for every haplotype of every locus in one launch.  aln_info: the M / I / D
// strings back to back, haplotype k of locus l (Haplotype::next() order, the reference haplotype first)
// at [info_off[h], info_off[h+1]) with h = hap_base[l] + k, hap_base[l] = number of haplotypes of the loci
// before l; info_off has sum_l H_l + 1 entries.
int ltr_haplotype_align_to_ref(ltr_ctx* ctx, const ltr_haplotype_blocks* const* haps, int64_t n_loci,
                               char* aln_info, int64_t cap, int64_t* info_off) {
  if (!ctx || (!haps && n_loci > 0) || n_loci < 0 || !aln_info || !info_off) return LTR_ERR_INVALID;
  ltr::TimedCall timed(ctx, ltr::kTimerHapBuild);              // total_hap_build_time_ (seq_stutter_genotyper.cpp:417,:479-480)
  LTR_GUARD_BEGIN
  if (hipSetDevice(ltr::ctx_device(ctx)) != hipSuccess) { ltr::set_error(ctx, "hipSetDevice failed"); return LTR_ERR_NO_DEVICE; }
  const bool nw_dbg = ltr::ctx_debug(ctx).trace != 0;                 // ltr_ctx_set_debug "trace": a timestamped phase profile on stderr
  const auto nw_t0 = std::chrono::steady_clock::now();
#define NW_TRACE(what) do { if (nw_dbg) std::fprintf(stderr, "[ltr] haplotype_align_to_ref %8.2f ms: %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - nw_t0).count(), what); } while (0)
  // ---- tasks: (reference haplotype, haplotype k) for every haplotype, sequences pooled.  Two passes: sizes and offsets in locus
  // order (serial, a few integers per haplotype), then the bytes on all host cores (20 MB per 3000 config-3 loci) ----
  std::vector<NwTask> tasks;
  std::vector<int32_t> ref_pos0, str_pos;                      // adjust_indels: blocks_[0]->start(), blocks_[1]->start()
  std::vector<int64_t> locus_task((size_t)n_loci + 1, 0);
  std::vector<std::vector<int32_t>> locus_counts((size_t)n_loci);
  int64_t out_bytes = 0, pool_bytes = 64;                      // (padded: the wavefront kernel streams rows without clamping)
  int32_t max_l1 = 1, max_l2 = 1;
  for (int64_t l = 0; l < n_loci; ++l) {
    const ltr_haplotype_blocks* h = haps[l];
    if (!h || h->n_blocks != 3) { ltr::set_error(ctx, "ltr_haplotype_align_to_ref: a haplotype needs three blocks (Haplotype::adjust_indels asserts it)"); return LTR_ERR_INVALID; }
    std::vector<int32_t>& counts = locus_counts[(size_t)l];
    int64_t H = 0;
    const int rc = ltr::haplotype_counts(h, &counts, &H);
    if (rc != LTR_OK) return rc;
    const int64_t ref_off = pool_bytes;
    int64_t ref_len = 0;
    for (int64_t k = 0; k < H; ++k) {
      int64_t len = 0, slot = 0;
      for (int b = 0; b < h->n_blocks; ++b) {
        const int64_t a = slot + counts[(size_t)(k * h->n_blocks + b)];
        len += h->allele_off[a + 1] - h->allele_off[a];
        slot += h->n_alleles[b];
      }
      if (k == 0) ref_len = len;
      if (ref_len < 1 || len < 1 || ref_len > (1 << 20) || len > (1 << 20)) { ltr::set_error(ctx, "empty or oversized haplotype"); return LTR_ERR_INVALID; }
      NwTask t;
      t.ref_off = ref_off; t.alt_off = pool_bytes; t.L1 = (int32_t)ref_len; t.L2 = (int32_t)len; t.out_off = out_bytes;
      pool_bytes += len;
      out_bytes += ref_len + len;
      max_l1 = std::max(max_l1, t.L1); max_l2 = std::max(max_l2, t.L2);
      tasks.push_back(t);
      ref_pos0.push_back(h->block_start[0]); str_pos.push_back(h->block_start[1]);
    }
    locus_task[(size_t)l + 1] = (int64_t)tasks.size();
  }
  std::vector<uint8_t> seqs((size_t)pool_bytes + 128, 0);
  ltr::parallel_for(n_loci, 64, [&](int64_t l) {
    const ltr_haplotype_blocks* h = haps[l];
    const std::vector<int32_t>& counts = locus_counts[(size_t)l];
    for (int64_t t = locus_task[(size_t)l]; t < locus_task[(size_t)l + 1]; ++t) {
      const int64_t k = t - locus_task[(size_t)l];
      uint8_t* dst = seqs.data() + tasks[(size_t)t].alt_off;
      int64_t slot = 0;
      for (int b = 0; b < h->n_blocks; ++b) {
        const int64_t a = slot + counts[(size_t)(k * h->n_blocks + b)];
        const int64_t len = h->allele_off[a + 1] - h->allele_off[a];
        std::memcpy(dst, h->allele_bytes + h->allele_off[a], (size_t)len);
        dst += len; slot += h->n_alleles[b];
      }
    }
  }, 16);
  NW_TRACE("sequences pooled");
  const int64_t nt = (int64_t)tasks.size();
  info_off[0] = 0;
  if (nt == 0) return LTR_OK;
  // ---- launch classes: references of up to 64 x 20 bases take the wavefront kernel (strip width 4 .. 20), longer
  // ones the workgroup-per-pair kernel ----
  constexpr int kClasses = 6;                                   // strip widths 4, 8, 12, 16, 20 + the workgroup kernel
  // (every even width 4 .. 20 -- nine classes, fuller lanes -- was tried on MI355X, same workload: kernels 10.3 ms, no change, and
  // the widths that are not multiples of four failed the parity test (trace words of a partly filled last word): not kept)
  std::vector<int32_t> cls_tasks[kClasses];
  int32_t cls_max_l2[kClasses] = {0}, wg_max_l1 = 1, wg_max_l2 = 1;
  for (int64_t k = 0; k < nt; ++k) {
    const int w = (tasks[(size_t)k].L1 + 63) / 64;
    const int c = (w <= kNwWaveMaxW) ? (std::max(w, 1) + 3) / 4 - 1 : kClasses - 1;
    cls_tasks[c].push_back((int32_t)k);
    cls_max_l2[c] = std::max(cls_max_l2[c], tasks[(size_t)k].L2);
    if (c == kClasses - 1) { wg_max_l1 = std::max(wg_max_l1, tasks[(size_t)k].L1); wg_max_l2 = std::max(wg_max_l2, tasks[(size_t)k].L2); }
  }
  int n_cu = 0;
  (void)ltr_ctx_device_info(ctx, nullptr, 0, &n_cu, nullptr);
  n_cu = std::max(n_cu, 1);
  // Every class has a region of its own in the trace block and a stream of its own: the launches run side by side, widest strips
  // (longest pairs) first, and inside a class the pairs are popped longest first -- a pair is 1 - 3 ms of one wavefront and a class
  // of a 3000-locus call a handful of pairs per resident wavefront, so a launch on its own ends in a tail as long as a pair with most
  // of the GPU idle (rocprofv3, round 3: VALU issue 0.35 - 0.65 per launch).  Measured on MI355X, 21 k haplotypes of 3000 config-3
  // loci, kernels first to last: one stream, task order 14.0 ms; profiles/r05/nw_rate.log.
  int64_t cls_grid[kClasses] = {0}, cls_stride[kClasses] = {0}, cls_off[kClasses] = {0}, trace_bytes = 256;
  std::vector<int32_t> index;
  int64_t cls_first[kClasses + 1] = {0};
  constexpr int64_t kTraceCap = (int64_t)16 << 30;              // of the context's trace block, all classes together
  // longest first inside a class: rows of the alternate, then the reference's length, ties in task order (the classes side by side
  // on the host cores, plain keys: a comparator that looks the tasks up was 1.1 ms for 21 k haplotypes)
  ltr::parallel_for(kClasses, 1, [&](int64_t c) {
    struct Key { int32_t l2, l1, task; };
    std::vector<Key> keys; keys.reserve(cls_tasks[c].size());
    for (const int32_t t : cls_tasks[c]) keys.push_back({tasks[(size_t)t].L2, tasks[(size_t)t].L1, t});
    std::sort(keys.begin(), keys.end(), [](const Key& x, const Key& y) { return x.l2 != y.l2 ? x.l2 > y.l2 : (x.l1 != y.l1 ? x.l1 > y.l1 : x.task < y.task); });
    for (size_t i = 0; i < keys.size(); ++i) cls_tasks[c][i] = keys[i].task;
  }, 1);
  for (int pass = 0; pass < 2; ++pass) {
    // (pass 1, only when the regions of pass 0 add up to more than the cap: every class with its share of the wavefronts)
    const double shrink = pass == 0 ? 1.0 : (double)kTraceCap / (double)trace_bytes * 0.98;
    if (pass == 1 && trace_bytes <= kTraceCap) break;
    trace_bytes = 256;
    for (int c = 0; c < kClasses; ++c) {
      if (pass == 0) {
        cls_first[c + 1] = cls_first[c] + (int64_t)cls_tasks[c].size();
        index.insert(index.end(), cls_tasks[c].begin(), cls_tasks[c].end());
      }
      if (cls_tasks[c].empty()) continue;
      cls_off[c] = trace_bytes;
      if (c < kClasses - 1) {
        const int W = 4 * (c + 1);
        cls_stride[c] = (((int64_t)(cls_max_l2[c] + 64) * 64 * W) + 255) / 256 * 256;       // per wavefront
        int64_t waves = std::min<int64_t>((int64_t)cls_tasks[c].size(), (int64_t)n_cu * 12);
        waves = std::max<int64_t>(1, std::min<int64_t>((int64_t)((double)waves * shrink), ((int64_t)6 << 30) / cls_stride[c]));
        cls_grid[c] = (waves + kNwWaveBlock - 1) / kNwWaveBlock;
        trace_bytes += cls_grid[c] * kNwWaveBlock * cls_stride[c];
      } else {
        cls_stride[c] = (((int64_t)(wg_max_l1 + 1) * (wg_max_l2 + 1)) + 255) / 256 * 256;   // per workgroup
        int64_t g = std::min<int64_t>((int64_t)cls_tasks[c].size(), (int64_t)n_cu * 2);    // LDS (54 KB) admits two workgroups per CU
        cls_grid[c] = std::max<int64_t>(1, std::min<int64_t>((int64_t)std::max(1.0, (double)g * shrink), ((int64_t)6 << 30) / cls_stride[c]));
        trace_bytes += cls_grid[c] * cls_stride[c];
      }
    }
  }
  NW_TRACE("classes sorted");
  const int64_t diag_stride = (wg_max_l2 > kNwLdsRows && !cls_tasks[kClasses - 1].empty()) ? (int64_t)9 * (wg_max_l2 + 1) : 0;
  uint8_t *d_seqs = nullptr, *d_masks = nullptr, *d_trace = nullptr, *d_out = nullptr;
  NwTask* d_tasks = nullptr; float* d_diag = nullptr; int32_t* d_len = nullptr; int32_t* d_index = nullptr; uint32_t* d_queue = nullptr;
  int rc = LTR_OK;
  std::unique_lock<std::mutex> big_lock;                        // (taken where the context's trace block is borrowed)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;                      // device time of the kernels (ltr_timers.nw_kernel_ms)
  hipEvent_t ev_join[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipStream_t st = (hipStream_t)ltr::ctx_stream(ctx);
  uint8_t* h_out = nullptr;                                    // (pinned staging of the context: the download goes over the DMA engines)
  std::vector<int32_t> h_len((size_t)nt);
  std::vector<void*> blocks;
#define NW_TRY(call) do { hipError_t e_ = (hipError_t)(call); if (e_ != hipSuccess) { ltr::set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_)); rc = LTR_ERR_HIP; goto done; } } while (0)
#define NW_ALLOC(ptr, bytes) do { void* p_ = nullptr; NW_TRY(ltr::ctx_pool_alloc(ctx, &p_, (size_t)(bytes))); blocks.push_back(p_); ptr = (decltype(ptr))p_; } while (0)
  NW_ALLOC(d_seqs, seqs.size());
  NW_ALLOC(d_masks, seqs.size());
  NW_ALLOC(d_tasks, (size_t)nt * sizeof(NwTask));
  NW_ALLOC(d_index, (size_t)nt * sizeof(int32_t));
  big_lock = ltr::ctx_call_lock(ctx);                           // one borrower of the context's big block at a time (a second host thread waits here)
  d_trace = (uint8_t*)ltr::ctx_big_scratch(ctx, (size_t)trace_bytes);     // (kept by the context between calls: gigabytes)
  if (!d_trace) { ltr::set_error(ctx, "out of device memory (NW trace)"); rc = LTR_ERR_NOMEM; goto done; }
  if (diag_stride) NW_ALLOC(d_diag, (size_t)(cls_grid[kClasses - 1] * diag_stride) * sizeof(float));
  NW_ALLOC(d_out, std::max<int64_t>(out_bytes, 1));
  NW_ALLOC(d_len, (size_t)nt * sizeof(int32_t));
  NW_ALLOC(d_queue, kClasses * sizeof(uint32_t));
  NW_TRACE("device blocks");
  NW_TRY(hipMemcpyAsync(d_seqs, seqs.data(), seqs.size(), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(ltr_nw_mask_kernel, dim3((unsigned)std::min<size_t>((seqs.size() + 255) / 256, (size_t)n_cu * 8)), dim3(256), 0, st, d_seqs, d_masks, (int64_t)seqs.size());
  NW_TRY(hipGetLastError());
  NW_TRY(hipMemcpyAsync(d_tasks, tasks.data(), (size_t)nt * sizeof(NwTask), hipMemcpyHostToDevice, st));
  NW_TRY(hipMemcpyAsync(d_index, index.data(), (size_t)nt * sizeof(int32_t), hipMemcpyHostToDevice, st));
  NW_TRY(hipMemsetAsync(d_queue, 0, kClasses * sizeof(uint32_t), st));
  NW_TRY(hipEventCreate(&ev0)); NW_TRY(hipEventCreate(&ev1));
  NW_TRY(hipEventRecord(ev0, st));
  for (int q = 0; q < kClasses; ++q) {
    const int c = (q < kClasses - 1) ? kClasses - 2 - q : kClasses - 1;       // strip widths 20, 16, .., 4, then the workgroup kernel
    const int n_c = (int)cls_tasks[c].size();
    if (n_c == 0) continue;
    hipStream_t sc = (hipStream_t)ltr::ctx_side_stream(ctx, q);            // (q = 0: the context's stream itself)
    if (sc != st) NW_TRY(hipStreamWaitEvent(sc, ev0, 0));
    const int32_t* idx = d_index + cls_first[c];
    uint8_t* tr = d_trace + cls_off[c];
    const dim3 g((unsigned)cls_grid[c]), wb(64 * kNwWaveBlock);
    switch (c) {
      case 0: hipLaunchKernelGGL((ltr_nw_wave_kernel<4>), g, wb, 0, sc, d_tasks, idx, n_c, d_queue + c, d_seqs, d_masks, tr, cls_stride[c], d_out, d_len); break;
      case 1: hipLaunchKernelGGL((ltr_nw_wave_kernel<8>), g, wb, 0, sc, d_tasks, idx, n_c, d_queue + c, d_seqs, d_masks, tr, cls_stride[c], d_out, d_len); break;
      case 2: hipLaunchKernelGGL((ltr_nw_wave_kernel<12>), g, wb, 0, sc, d_tasks, idx, n_c, d_queue + c, d_seqs, d_masks, tr, cls_stride[c], d_out, d_len); break;
      case 3: hipLaunchKernelGGL((ltr_nw_wave_kernel<16>), g, wb, 0, sc, d_tasks, idx, n_c, d_queue + c, d_seqs, d_masks, tr, cls_stride[c], d_out, d_len); break;
      case 4: hipLaunchKernelGGL((ltr_nw_wave_kernel<20>), g, wb, 0, sc, d_tasks, idx, n_c, d_queue + c, d_seqs, d_masks, tr, cls_stride[c], d_out, d_len); break;
      default:
        hipLaunchKernelGGL(ltr_nw_kernel, g, dim3(kNwThreads), 0, sc, d_tasks, idx, n_c, d_queue + c, d_seqs, tr, cls_stride[c],
                           d_diag, diag_stride, d_out, d_len);
    }
    NW_TRY(hipGetLastError());
    if (sc != st) {
      if (!ev_join[q]) NW_TRY(hipEventCreateWithFlags(&ev_join[q], hipEventDisableTiming));
      NW_TRY(hipEventRecord(ev_join[q], sc));
      NW_TRY(hipStreamWaitEvent(st, ev_join[q], 0));
    }
  }
  NW_TRY(hipEventRecord(ev1, st));
  NW_TRACE("launches queued");
  h_out = ltr::ctx_host_bytes(ctx, 0, (size_t)std::max<int64_t>(out_bytes, 1));       // (under the context's call lock, taken above)
  NW_TRY(hipMemcpyAsync(h_out, d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, st));
  NW_TRY(hipMemcpyAsync(h_len.data(), d_len, (size_t)nt * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  NW_TRY(hipStreamSynchronize(st));
  NW_TRACE("codes downloaded");
  { float ms = 0.f; if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) ltr::add_time(ctx, ltr::kTimerNwKernel, 0.0, (double)ms); }
  {
    // ---- host, all cores: reverse, adjust_indels, M / I / D (Haplotype.cpp:66-82) ----
    for (int64_t k = 0; k < nt; ++k) info_off[k + 1] = info_off[k] + h_len[(size_t)k];       // (adjust_indels keeps the length)
    if (info_off[nt] > cap) { ltr::set_error(ctx, "ltr_haplotype_align_to_ref: output buffer too small (ltr_haplotype_aln_info_capacity)"); rc = LTR_ERR_INVALID; goto done; }
    ltr::parallel_for(nt, 64, [&](int64_t k) {
      const int n = h_len[(size_t)k];
      // the two aligned strings (traceAlignment's stringstreams, reversed, :247-312) from the device's column codes, front to back
      const uint8_t* ops = h_out + tasks[(size_t)k].out_off;
      char* dst = aln_info + info_off[k];
      {
        // adjust_indels only ever touches a gap that starts while the reference position is still left of the repeat block
        // (`ref_pos < str_pos` in both of its branches): when the first str_pos - ref_pos0 columns are all matches it changes
        // nothing, and the M / I / D string is the column codes read backwards -- no strings built (most haplotypes differ from
        // the reference allele inside the repeat only; the strings + adjust_indels pass was 2.4 - 4.8 of the call's 15.4 ms).
        const int flank = std::min(n, std::max(0, str_pos[(size_t)k] - ref_pos0[(size_t)k]));
        bool plain = true;
        for (int i = 0; i < flank; ++i) if (ops[n - 1 - i] != 0) { plain = false; break; }
        if (plain) {
          for (int i = 0; i < n; ++i) { const uint8_t op = ops[n - 1 - i]; dst[i] = op == 0 ? 'M' : (op == 1 ? 'D' : 'I'); }
          return;
        }
      }
      const uint8_t* rs = seqs.data() + tasks[(size_t)k].ref_off;
      const uint8_t* as = seqs.data() + tasks[(size_t)k].alt_off;
      std::string ref_al((size_t)n, '-'), alt_al((size_t)n, '-');
      for (int i = 0, ri = 0, ai = 0; i < n; ++i) {
        const uint8_t op = ops[n - 1 - i];
        if (op != 2) ref_al[(size_t)i] = (char)rs[ri++];
        if (op != 1) alt_al[(size_t)i] = (char)as[ai++];
      }
      adjust_indels(ref_al, alt_al, ref_pos0[(size_t)k], str_pos[(size_t)k]);
      for (int i = 0; i < n; ++i) dst[i] = (ref_al[(size_t)i] == '-') ? 'I' : ((alt_al[(size_t)i] == '-') ? 'D' : 'M');
    }, 16);
  }
  NW_TRACE("strings rebuilt");
done:
#undef NW_TRACE
#undef NW_TRY
#undef NW_ALLOC
  if (rc != LTR_OK) for (int q = 0; q < 6; ++q) (void)hipStreamSynchronize((hipStream_t)ltr::ctx_side_stream(ctx, q));   // (nothing in flight may still use the blocks)
  for (void* p_ : blocks) ltr::ctx_pool_release(ctx, p_);
  for (hipEvent_t e : ev_join) if (e) (void)hipEventDestroy(e);
  if (ev0) (void)hipEventDestroy(ev0);
  if (ev1) (void)hipEventDestroy(ev1);
  return rc;
  LTR_GUARD_END(ctx)
}

}  // extern "C"

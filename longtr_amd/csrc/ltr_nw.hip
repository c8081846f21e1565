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

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "ltr_internal.h"

namespace {

constexpr float kMatch = 2.0f, kMismatch = -2.0f, kGapOpen = 5.0f, kGapExtend = 0.125f, kLarge = 1000000.0f;   // NeedlemanWunsch.cpp:84-96
constexpr int kNwThreads = 256;
constexpr int kNwLdsRows = 1536;                               // alternates up to this length keep the rolling diagonals in LDS (54 KB)

struct NwTask {
  int64_t ref_off, alt_off;        // byte offsets into the sequence pool
  int64_t out_off;                 // offset of this task's two aligned strings (2 x (L1 + L2) bytes, reversed) in the output
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

__global__ __launch_bounds__(kNwThreads) void ltr_nw_kernel(const NwTask* __restrict__ tasks, int n_tasks, uint32_t* queue,
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
    const int ti = __builtin_amdgcn_readfirstlane(*(volatile int*)&s_task);
    if (ti >= n_tasks) break;
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
      uint8_t* ref_al = out + T.out_off;
      uint8_t* alt_al = ref_al + (L1 + L2);
      int n = 0;
      while (best_row > 0) {
        const uint8_t tr = (best_col > 0) ? trace[(int64_t)best_row * (L1 + 1) + best_col] : (uint8_t)(2 << 4);   // column 0: traceIread = 2 (:368)
        if (type == 0) { ref_al[n] = ref[best_col - 1]; alt_al[n] = alt[best_row - 1]; ++n; type = tr & 3; --best_row; --best_col; }
        else if (type == 1) { ref_al[n] = ref[best_col - 1]; alt_al[n] = '-'; ++n; type = (tr >> 2) & 3; --best_col; }
        else { ref_al[n] = '-'; alt_al[n] = alt[best_row - 1]; ++n; type = (tr >> 4) & 3; --best_row; }
      }
      for (int i = best_col; i > 0; --i) { ref_al[n] = ref[i - 1]; alt_al[n] = '-'; ++n; }       // leading gaps, :307-310
      out_len[ti] = n;
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
  // ---- tasks: (reference haplotype, haplotype k) for every haplotype, sequences pooled ----
  std::vector<uint8_t> seqs;
  std::vector<NwTask> tasks;
  std::vector<int32_t> ref_pos0, str_pos;                      // adjust_indels: blocks_[0]->start(), blocks_[1]->start()
  int64_t out_bytes = 0;
  int32_t max_l1 = 1, max_l2 = 1;
  std::string s;
  for (int64_t l = 0; l < n_loci; ++l) {
    const ltr_haplotype_blocks* h = haps[l];
    if (!h || h->n_blocks != 3) { ltr::set_error(ctx, "ltr_haplotype_align_to_ref: a haplotype needs three blocks (Haplotype::adjust_indels asserts it)"); return LTR_ERR_INVALID; }
    std::vector<int32_t> counts; int64_t H = 0;
    const int rc = ltr::haplotype_counts(h, &counts, &H);
    if (rc != LTR_OK) return rc;
    const int64_t ref_off = (int64_t)seqs.size();
    int64_t ref_len = 0;
    for (int64_t k = 0; k < H; ++k) {
      const int64_t off = (int64_t)seqs.size();
      int64_t len = 0, slot = 0;
      for (int b = 0; b < h->n_blocks; ++b) {
        const int64_t a = slot + counts[(size_t)(k * h->n_blocks + b)];
        seqs.insert(seqs.end(), h->allele_bytes + h->allele_off[a], h->allele_bytes + h->allele_off[a + 1]);
        len += h->allele_off[a + 1] - h->allele_off[a];
        slot += h->n_alleles[b];
      }
      if (k == 0) ref_len = len;
      if (ref_len < 1 || len < 1 || ref_len > (1 << 20) || len > (1 << 20)) { ltr::set_error(ctx, "empty or oversized haplotype"); return LTR_ERR_INVALID; }
      NwTask t;
      t.ref_off = ref_off; t.alt_off = off; t.L1 = (int32_t)ref_len; t.L2 = (int32_t)len; t.out_off = out_bytes;
      out_bytes += 2 * (ref_len + len);
      max_l1 = std::max(max_l1, t.L1); max_l2 = std::max(max_l2, t.L2);
      tasks.push_back(t);
      ref_pos0.push_back(h->block_start[0]); str_pos.push_back(h->block_start[1]);
    }
  }
  const int64_t nt = (int64_t)tasks.size();
  info_off[0] = 0;
  if (nt == 0) return LTR_OK;
  // ---- device buffers: a trace matrix and (for long alternates) a diagonal strip per resident workgroup ----
  int n_cu = 0;
  (void)ltr_ctx_device_info(ctx, nullptr, 0, &n_cu, nullptr);
  const int64_t trace_stride = (((int64_t)(max_l1 + 1) * (max_l2 + 1)) + 255) / 256 * 256;
  const int64_t diag_stride = (max_l2 > kNwLdsRows) ? (int64_t)9 * (max_l2 + 1) : 0;
  int64_t grid = std::min<int64_t>(nt, (int64_t)std::max(n_cu, 1) * 2);                       // LDS (54 KB) admits two workgroups per CU
  grid = std::max<int64_t>(1, std::min<int64_t>(grid, ((int64_t)6 << 30) / std::max<int64_t>(trace_stride, 1)));
  uint8_t *d_seqs = nullptr, *d_trace = nullptr, *d_out = nullptr;
  NwTask* d_tasks = nullptr; float* d_diag = nullptr; int32_t* d_len = nullptr; uint32_t* d_queue = nullptr;
  int rc = LTR_OK;
  hipStream_t st = (hipStream_t)ltr::ctx_stream(ctx);
  std::vector<uint8_t> h_out((size_t)out_bytes);
  std::vector<int32_t> h_len((size_t)nt);
#define NW_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ltr::set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_)); rc = LTR_ERR_HIP; goto done; } } while (0)
  NW_TRY(hipMalloc((void**)&d_seqs, seqs.size()));
  NW_TRY(hipMalloc((void**)&d_tasks, (size_t)nt * sizeof(NwTask)));
  NW_TRY(hipMalloc((void**)&d_trace, (size_t)(grid * trace_stride)));
  if (diag_stride) NW_TRY(hipMalloc((void**)&d_diag, (size_t)(grid * diag_stride) * sizeof(float)));
  NW_TRY(hipMalloc((void**)&d_out, (size_t)out_bytes));
  NW_TRY(hipMalloc((void**)&d_len, (size_t)nt * sizeof(int32_t)));
  NW_TRY(hipMalloc((void**)&d_queue, sizeof(uint32_t)));
  NW_TRY(hipMemcpyAsync(d_seqs, seqs.data(), seqs.size(), hipMemcpyHostToDevice, st));
  NW_TRY(hipMemcpyAsync(d_tasks, tasks.data(), (size_t)nt * sizeof(NwTask), hipMemcpyHostToDevice, st));
  NW_TRY(hipMemsetAsync(d_queue, 0, sizeof(uint32_t), st));
  hipLaunchKernelGGL(ltr_nw_kernel, dim3((unsigned)grid), dim3(kNwThreads), 0, st, d_tasks, (int)nt, d_queue, d_seqs, d_trace, trace_stride,
                     d_diag, diag_stride, d_out, d_len);
  NW_TRY(hipGetLastError());
  NW_TRY(hipMemcpyAsync(h_out.data(), d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, st));
  NW_TRY(hipMemcpyAsync(h_len.data(), d_len, (size_t)nt * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  NW_TRY(hipStreamSynchronize(st));
  {
    // ---- host: reverse, adjust_indels, M / I / D (Haplotype.cpp:66-82) ----
    int64_t at = 0;
    std::string ref_al, alt_al;
    for (int64_t k = 0; k < nt; ++k) {
      const int n = h_len[(size_t)k];
      const uint8_t* r = h_out.data() + tasks[(size_t)k].out_off;
      const uint8_t* a = r + (tasks[(size_t)k].L1 + tasks[(size_t)k].L2);
      ref_al.assign(r, r + n); alt_al.assign(a, a + n);
      std::reverse(ref_al.begin(), ref_al.end()); std::reverse(alt_al.begin(), alt_al.end());
      adjust_indels(ref_al, alt_al, ref_pos0[(size_t)k], str_pos[(size_t)k]);
      if (at + n > cap) { ltr::set_error(ctx, "ltr_haplotype_align_to_ref: output buffer too small (ltr_haplotype_aln_info_capacity)"); rc = LTR_ERR_INVALID; goto done; }
      for (int i = 0; i < n; ++i) aln_info[at + i] = (ref_al[(size_t)i] == '-') ? 'I' : ((alt_al[(size_t)i] == '-') ? 'D' : 'M');
      at += n;
      info_off[k + 1] = at;
    }
  }
done:
#undef NW_TRY
  if (d_seqs) (void)hipFree(d_seqs);
  if (d_tasks) (void)hipFree(d_tasks);
  if (d_trace) (void)hipFree(d_trace);
  if (d_diag) (void)hipFree(d_diag);
  if (d_out) (void)hipFree(d_out);
  if (d_len) (void)hipFree(d_len);
  if (d_queue) (void)hipFree(d_queue);
  return rc;
  LTR_GUARD_END(ctx)
}

}  // extern "C"

// ltr_plan.h -- host-side planning of a batch (no HIP in here): the table of launch classes, the rule that
// gives every (read, haplotype) pair its class and launch-order key, and the sort that lays the pairs out
// class by class, longest first.  ltr_gpu.hip's ltr_plan_create strings these units together with the
// uploads; tests/test_plan_units.py exercises them on the CPU through the ltr_debug_* entry points.
#ifndef LTR_PLAN_H_
#define LTR_PLAN_H_

#include <cstdint>
#include <vector>

#include "ltr_dp_types.h"

namespace ltrp {

// Launch classes ("bins") of the certificate kernels, in this order:
//   [0, kNumBins)             one pair per wavefront, strip width W = k+1 (any read length: column blocks through scratch strips)
//   [kPackFirst, +kNumPack)   64 / LP pairs per wavefront, LP = 2 << (j / kPackWMax) lanes per pair, W = j % kPackWMax + 1
//   [kWg4First, +kNumWg4)     one pair per 4-wave workgroup, W = kWg4MinW+j (reads of 1282 .. 5121 bases), LDS hand-off
//   [kWg8First, +kNumWg8)     one pair per 8-wave workgroup, W = kWg8MinW+j (reads of 5122 .. 10241 bases)
//   [kWg1First, +kNumWg1)     one pair per 1-wave workgroup, W = j+1: the latency variant for small batches
// then the exact (redo) kernels, classes kNumFast + kXGeneric .. kXWg8.
constexpr int kNumBins = kWMax;
constexpr int kNumPack = kNumPackLp * kPackWMax;
constexpr int kPackFirst = kNumBins;
constexpr int kNumWg1 = kWg1MaxW;
constexpr int kWg4First = kPackFirst + kNumPack;
constexpr int kWg8First = kWg4First + kNumWg4;
constexpr int kWg1First = kWg8First + kNumWg8;
constexpr int kNumFast = kWg1First + kNumWg1;   // certificate kernel classes
constexpr int kNumKernels = kNumFast + kNumExact;
// control words of a plan: [0, kNumKernels + 1] work queues (the last two: the W = 20 exact launch, the eight-wave list's narrow launch), [kRedoCountSlot, +kNumExact) exact list lengths
constexpr int kCtrlWords = 256;
constexpr int kStartQueueSlot = 176;           // [kStartQueueSlot, +kNumExact): work counters of the plan kernel's entries of kind 2 (the pairs that start out in an exact list)
constexpr int kRedoCountSlot = 192;
static_assert(kNumKernels + 2 <= kStartQueueSlot && kStartQueueSlot + kNumExact <= kRedoCountSlot, "control block layout");
static_assert(kNumKernels + 1 <= kRedoCountSlot && kRedoCountSlot + kInlineCountOff + kNumExact <= kCtrlWords && kInlineCountOff >= kNumExact, "control block layout");
static_assert(kWgStatOff >= kInlineCountOff + kNumExact && kRedoCountSlot + kWgStatOff + 2 <= kCtrlWords, "control block layout");

// pairs per block of the host loops over a plan's pairs (counting sort, gather, class statistics): a 10 000-locus chunk of a
// catalogue is 235 k pairs -- four blocks of 64 k kept four of the host's cores busy
constexpr size_t kPlanBlock = 16384;
constexpr int kWg4WideMinW = 15;               // (ltr_plan.cpp, make_rules)
constexpr int kFoldRounds = 6;                 // automatic mode: classes below 6 x 4 x (pairs per wave) x CUs pairs are folded (tests/manual/gpu_fold_sweep.py)
enum { kFamOne = 0, kFamPack = 1, kFamWg = 2, kFamExact = 3 };
struct ClassInfo { int family; int W; int waves; int lp_shift; };   // waves per pair (workgroup kernels); lanes per pair = 1 << lp_shift (pack)
inline ClassInfo class_info(int k) {
  if (k < kPackFirst) return {kFamOne, k + 1, 1, 6};
  if (k < kWg4First) { const int j = k - kPackFirst; return {kFamPack, j % kPackWMax + 1, 1, kPackMinShift + j / kPackWMax}; }
  if (k < kWg8First) return {kFamWg, k - kWg4First + kWg4MinW, 4, 6};
  if (k < kWg1First) return {kFamWg, k - kWg8First + kWg8MinW, 8, 6};
  return {kFamWg, k - kWg1First + 1, 1, 6};
}
inline int pack_class(int lp_shift, int W) { return kPackFirst + (lp_shift - kPackMinShift) * kPackWMax + (W - 1); }

// W = ceil(C / (64 * ncb)), ncb = ceil(C / (64 * kWMax)): the narrowest strip that covers the read in the
// fewest column blocks of one wavefront (ltr_dp_kernel.hpp).
inline int strip_width_for(int m, int* ncb_out) {
  const int C = m - 1 > 1 ? m - 1 : 1;
  const int ncb = (C + 64 * kWMax - 1) / (64 * kWMax);
  if (ncb_out) *ncb_out = ncb;
  return (C + 64 * ncb - 1) / (64 * ncb);
}

constexpr int kLengthBuckets = 96;
inline int length_bucket(int C) {               // quarter octaves of the read's columns: 4 * floor(log2 C) + the next two bits
  if (C < 4) return C < 1 ? 0 : C;
  int e = 31 - __builtin_clz((unsigned)C);
  const int b = 4 * e + ((C >> (e - 2)) & 3);
  return b < kLengthBuckets ? b : kLengthBuckets - 1;
}

// What a batch as a whole decides (ltr_ctx_set_pair_packing mode, size of the batch, the indel model).
struct Rules {
  int mode = -1;               // ltr_ctx_set_pair_packing
  bool sym_model = true;       // ins->match == del->match and match->ins == match->del
  bool xlut = true;            // LUT / penalty-table exact kernels usable
  bool thr_lists = true;       // pairs are listed by read length for the threshold bodies: xlut, or -- any model -- under the plan kernel, whose exact bodies are calls of its own
  bool wg_long = false;        // workgroup kernels for reads longer than one wavefront's widest strips
  bool wg_wide4 = false;       // ... four-wave workgroups with strips of kWg4WideMinW columns and more: at any number of long pairs
  int64_t wide4_quota = INT64_MAX;   // ... for this many pairs of the batch; the rest of them on eight waves (ltr_plan_create moves them)
  bool wg_short = false;       // ... and their one-wave variant for every short read (mode 2)
  int wg_min_c = 64 * kWMax;
  int pack_min_shift = 7;      // fewest lanes per pair a packed class may use (7: no packed classes at all)
  int pack_force_shift = 0;    // != 0: this many lanes per pair whenever the read fits (modes 1, 5 .. 8)
  // fewest lanes per pair by read length (quarter-octave buckets of the read's columns, length_bucket()) -- only on request
  // (ltr_ctx_set_debug "pack_rule" = 3) since a packed launch is a whole strip width: pairs of a length that is rare in the
  // batch kept more lanes each, so that the launch of their (lanes, width) class still put about two wavefronts on every SIMD
  int8_t bucket_min_shift[kLengthBuckets];
  int flank = 5;               // indel_flank_len
  // |n - m| from which a pair goes straight to its exact list: a pair whose lengths differ by L carries a gap of at least L, and
  // once that gap alone costs ~520 the one-cell-per-lane certificate cannot hold the -600 line (measured on MI355X, a 1250-locus
  // shard of config 3: every one of the 119 pairs whose certificate failed had 531 <= |n - m| <= 600) -- scoring it with the
  // certificate body first is wasted work, and inside the plan kernel its exact body started late is the launch's tail
  int risky_dd_pos = 0x7fffffff, risky_dd_neg = 0x7fffffff;   // n - m / m - n from which a pair goes straight to the exact body (make_rules)
  // The plan kernel (ltr_dp_plan.hpp) scores the one-wave and packed classes of this batch in one persistent launch: automatic
  // mode, symmetric model.  Such a launch holds every wave slot until it ends, so launches beside it starve: once the batch can
  // fill the GPU, reads of up to two column blocks (2560 columns) stay with the one-wave classes (wg_min_c).
  bool plan_kernel = false;
};
// pairs_by_bucket: pairs of the batch by length_bucket(read columns), or nullptr (no per-length rule)
Rules make_rules(const ModelConsts& mc, int indel_flank_len, int mode, int n_cu, int64_t pairs_upper, int64_t n_long_pairs,
                 const int64_t* pairs_by_bucket = nullptr, int pack_rule = 0, int plan_knob = 0);

// Modelled cost of one pair in a packed class, in wave-cycles per pair: steps x (cells + per-step overhead)
// x the share of the wave the pair holds.  (Constants from the sweep of tests/manual/gpu_pack_sweep.py.)
double pack_cost(int n, int C, int lp_shift, int* W_out);

struct PairClass {
  int16_t cls = -1;            // launch class, or kNumFast + exact list for pairs that start out in an exact list
  double cost = 1.0;           // modelled launch time of the pair (wave-cycles; what the key is the logarithm of)
  int16_t key = 0;             // launch-order key: the cost in steps of 1/16 octave (4.4 %), 1 .. 511; shortcut pairs 0 (last)
  int8_t xc = kXGeneric;       // exact list the pair lands in if its certificate fails
  bool shortcut = false;       // HapAligner.cpp:241-244, :249-252: constant score, no DP
  bool x_candidate = false;    // counts towards the size of exact list xc
  bool uses_wg = false;
};
// n: window length (0 when hap_full_len <= 60), m: read length, generic: bytes outside ACGT
PairClass classify_pair(const Rules& R, int64_t n, int64_t m, int64_t hap_full_len, bool generic);

// Counting sort of the pairs by class (input order kept), every class then longest first by key.  In automatic mode
// a class whose pairs cannot fill the GPU's wave slots a few times over is folded into the next wider class of its
// family.  Out: order[i] = index of the pair that goes to sorted position i; bin_first[k] .. bin_first[k+1] = class k.
// fold_rounds: a class is folded while it holds fewer pairs than this many rounds of resident wavefronts (0: never)
// multi_launch: 1 = the one-wave widths kMultiMinW.. and the packed widths kPackMultiMinW.. share a launch each (nothing is
// folded there: every pair keeps its own strip width, and the last width below folds at most up to the first of them);
// 2 = the plan kernel takes every one-wave and packed class: nothing of the two families is folded
void sort_by_class(const int16_t* bin, const int16_t* key, int64_t n_pairs, int fold_rounds, int n_cu, int32_t* order,
                   int* bin_first /* [kNumKernels + 1] */, int* counts /* [kNumKernels] */, int multi_launch = 0);

// The exact kernels' row test (ltr_dp_kernel.hpp, column_block): the reference aborts a pair when a row's maximum of
// fl(best + pen(k)), pen(k) = (double)((float)|k| * c), is below -600 (HapAligner.cpp:297-306).  x -> fl(x + p) is monotone, so
// "some cell reaches -600" is "some cell has best >= thr(k)" with thr(k) the SMALLEST double that does.  Entry k + kPenHalf of
// out[kPenTabDoubles]; +inf where no (negative) cell value can pass: |k| >= k600, |k| > kPenKMax, thr(k) >= 0.
void build_threshold_table(float c, double* out);

}  // namespace ltrp

#endif

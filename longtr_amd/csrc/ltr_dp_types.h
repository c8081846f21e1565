// ltr_dp_types.h -- what the host side (ltr_gpu.hip) and the kernel translation units (ltr_k_*.hip) share:
// pair descriptors, kernel arguments, launch-class constants.  No device code.
#ifndef LTR_DP_TYPES_H_
#define LTR_DP_TYPES_H_

#include <cstdint>

struct PairDesc {          // one (read, haplotype) DP
  int64_t read_off;        // byte offset of the trimmed read in read_bytes
  int64_t hap_off;         // byte offset of the haplotype WINDOW (hap[35-F ...]) in hap_bytes
  int64_t out_idx;         // index into the LL buffer
  int32_t m;               // read length
  int32_t n;               // window length
  int32_t hap_full_len;    // full haplotype length (for the <= 60 shortcut)
  int32_t generic;         // 1: read or haplotype holds bytes other than A,C,G,T -> byte-compare (exact) kernel
};

struct PackTable {         // one strip width of a multi-width packed launch (device memory, built by ltr_plan_create)
  int32_t W, queue_class;  // strip width; its work counter = queue_base[queue_class]
  int32_t shift[5], first[5], end[5], grp_end[5];   // as KernelArgs::pk_*
};

// One entry of the plan kernel's walk (ltr_dp_plan.hpp): a launch class of the one-wave family or one strip width of the packed
// family (all its lanes-per-pair ranges), longest pairs first.  `limit` = the value of its work counter at which it is drained
// (pairs / groups), read before the call into the class's body so that a drained entry costs one load, not a call.
struct PlanEntry {
  int32_t kind;            // 0: one pair per wavefront (ltr_dp_kernel.hpp), 1: packed (ltr_dp_pack.hpp), 2: pairs that go straight to the exact body (ltr_dp_redo.hpp), 3: kind 0 by the chained walk (ltr_dp_chain.hpp)
  int32_t W;               // strip width = which body scores it (kind 2: 0 = the generic body, bytes outside ACGT; 1 = the threshold bodies)
  int32_t first, n_pairs;  // kind 0: the class's range of the sorted pair list
  int32_t queue_class;     // its work counter = queue_base[queue_class]
  int32_t tab;             // kind 1: its PackTable in pk_tabs
  int32_t limit;
  int32_t first_wave;      // the wavefronts [first_wave, next entry's) of the launch START here (shares in proportion to the entries' modelled work)
};

struct ModelConsts {       // float-typed like the reference; promoted on use
  float a, b, c, d, e, f, g;
  float match, mismatch;   // HapAligner.cpp:260-261
  float match_plus_f;      // MATCH + LOG_MATCH_TO_INS evaluated in float (HapAligner.cpp:277)
};

struct KernelArgs {
  const PairDesc* pairs;
  const int32_t* index;    // optional indirection (redo list); nullptr = identity
  const uint32_t* n_pairs_dev;  // optional: pair count lives on the device (redo list)
  int32_t first_pair;      // this launch handles pairs [first_pair, first_pair + n_pairs)
  int32_t n_pairs;
  uint32_t* queue;         // atomic work counter (zeroed before the launch)
  // pairs a certificate kernel could not clear go to the list of the exact kernel that fits them:
  // xlist[c] (capacity: all pairs), its length at xcount[c]
  int32_t* xlist[6];
  uint32_t* xcount;
  int32_t xlut;            // 1: the LUT / penalty-table exact kernels may be used (symmetric model, k600 <= kPenKMax)
  int32_t thr_ok;          // 1: thr_tab is valid for the model in force (k600 <= kPenKMax; either symmetry): the plan kernel's threshold bodies may be used
  const double* thr_tab;   // kPenTabDoubles exact row-test thresholds of the LUT exact kernels (ltrp::build_threshold_table)
  const uint8_t* read_bytes;
  const uint8_t* hap_bytes;
  const uint16_t* hap_codes; // same layout as hap_bytes: ((byte >> 1) & 3) << 12 = byte offset of the base's emission-table block
  double* out_ll;
  const double* lpc;       // row-0 table:  lpc[1] = 0, lpc[j+1] = lpc[j] + c           (HapAligner.cpp:267-272)
  // column 0 (HapAligner.cpp:274-280): record i = {X0(i), Z0(i), X1(i), Z1(i)}, X/Z(i,0) for
  // emit(hap[0], read[1]) = mismatch (0) / match (1); at least 80 records longer than any haplotype
  const double* colXZ;
  int32_t table_len;       // last valid record
  const double* row0XY;    // row 0: record j = {X0(j), Y0(j), X1(j), Y1(j)}: X/Y(0,j) for emit(hap[j], read[0]) = mismatch (0) / match (1)
  double* scratch;         // per-wave boundary strips: [wave][2 buffers][3 arrays][scratch_stride]
  int32_t scratch_stride;  // doubles per array (>= longest window in this launch + 1)
  ModelConsts mc;
  int32_t c_lo, c_hi;      // exact kernels: this launch scores the list's pairs with c_lo <= m - 1 <= c_hi and skips the others
  int32_t lp_shift;        // (unused by the kernels since the packed launches carry their ranges below; kept for the launch log)
  // packed kernels (ltr_dp_pack.hpp), one launch per strip width: up to five ranges of the sorted pair list, widest
  // segments first.  Range r = pairs [pk_first[r], pk_end[r]) popped 64 >> pk_shift[r] at a time as the groups
  // [pk_grp_end[r-1], pk_grp_end[r]) of the launch's queue; unused ranges: pk_grp_end = the total, pk_first = pk_end = 0.
  int32_t pk_shift[5], pk_first[5], pk_end[5], pk_grp_end[5];
  // the multi-width one-wave kernel (ltr_dp_kernel.hpp, ltr_dp_multi_kernel): ONE persistent launch walks up to kMultiMax
  // classes of the plan, widest strips first -- range r = pairs [mk_first[r], + mk_np[r]) scored with strips of mk_w[r]
  // columns, work counter queue_base[mk_class[r]] -- so that a wavefront that finds a class's queue empty goes on with the
  // next class instead of draining: a launch per class ends in ~0.3 ms during which its wave slots empty one by one
  // (measured on MI355X: pass time of a single class = 0.30 ms + 0.924 ms x rounds of resident wavefronts)
  int32_t mk_n;
  int32_t mk_w[10], mk_first[10], mk_np[10], mk_class[10];
  uint32_t* queue_base;
  // the multi-width packed launch (ltr_dp_pack.hpp, ltr_dp_pack_multi_kernel): pk_ntabs tables, widest strips first
  const PackTable* pk_tabs;
  int32_t pk_ntabs;
  // the plan kernel (ltr_dp_plan.hpp): every one-wave class and packed strip width of the plan in ONE persistent launch
  const PlanEntry* pl_entries;
  int32_t pl_n;
  unsigned long long* wave_clock;   // optional (ltr_ctx_set_debug "wave_clock"): per wavefront of the plan kernel {first, last wall clock (100 MHz), pairs it scored with the exact body, ticks spent there}
};

// __launch_bounds__ 2nd argument (waves per SIMD the register allocator must leave room for).
// Measured on MI355X, same box, config 3: more resident waves win even where the wide strips then
// spill a few registers (W 13..16 at 3 waves: 2.0e12 vs 1.77e12 cells/s at 2 waves).
#ifndef LTR_LB
#define LTR_LB ((W <= 6) ? 5 : ((W <= 10) ? 4 : 3))
#endif
// ... of the exact kernel with the widest strips (W = 16)
#ifndef LTR_XLB_LONG
#define LTR_XLB_LONG 3
#endif
#ifndef LTR_PF
#define LTR_PF 2
#endif
#ifndef LTR_WMAX
#define LTR_WMAX 20
#endif
constexpr int kWMax = LTR_WMAX;      // widest strip (1281-base reads in ONE column block: nothing parked in scratch strips); wider reads use more column blocks
constexpr int kInlineCountOff = 8;                // xcount[kInlineCountOff + c]: pairs of exact class c the plan kernel scored in line (statistics)
constexpr int kWgStatOff = 16;                    // xcount[kWgStatOff]: pairs the FIRST pass of the workgroup classes could not finish (certificate pass: sent to an exact list; threshold pass: aborted, -700); [+ 1]: pairs it scored -- what the context learns its first pass from
constexpr int kPackMultiMinW = 13;                // ... of the multi-width packed launch: 13 .. 20
constexpr int kMultiMinW = 11, kMultiMax = 10;   // strip widths of the multi-width launch: 11 .. 20 (all at three waves per SIMD)
constexpr int kBlockWaves = 4;       // wavefronts per workgroup: independent workers that share one emission table in LDS
constexpr int kEmitTabDoubles = 4 * 256 * 4;   // [hap base][4 read bases][4 emissions]: 32 KB
constexpr int kExactW = 8;           // strip width of the exact redo kernel (any read length)
static_assert(kWMax >= 1 && kWMax <= 20, "strip widths 1..20");
constexpr double kImp = -1000000000.0;   // IMPOSSIBLE, HapAligner.cpp:20

// Exact (redo) kernel classes.  Generic: byte-compare emission and per-cell penalty arithmetic, any
// model, any bytes (W = kExactW).  The others need pure-ACGT pairs, a symmetric model and a band
// penalty table that fits LDS: one wavefront per pair with W = 4 / 10 / 16 by read length (the last
// one walks column blocks for any length), or a 4- / 8-wave workgroup per pair (ltr_dp_wg.hpp).
enum { kXGeneric = 0, kXShort = 1, kXMid = 2, kXLong = 3, kXWg4 = 4, kXWg8 = 5, kNumExact = 6 };
constexpr int kXShortW = 4, kXMidW = 10, kXLongW = 16, kXWideW = 20;
constexpr int kXWg4MaxC = 4 * 64 * 14, kXWg8MaxC = 8 * 64 * 20;

// EXACT: band penalty table, entry k + kPenHalf = (double)((float)|k| * c) for |k| < k600, IMPOSSIBLE
// beyond (a cell value is < 0, so such a term can never lift a row maximum to -600); 32 guard
// entries either side so that a strip of up to 20 consecutive offsets can be read from a clamped base.
constexpr int kPenKMax = 1023;
constexpr int kPenHalf = kPenKMax + 32;
constexpr int kPenTabDoubles = 2 * kPenHalf;

// widest strip of the several-pairs-per-wave kernels (ltr_dp_pack.hpp); lanes per pair 2, 4, .. 32
constexpr int kPackWMax = 20;
constexpr int kPackMinShift = 1, kPackMaxShift = 5, kNumPackLp = kPackMaxShift - kPackMinShift + 1;
// strip widths of the workgroup-per-pair kernels (ltr_dp_wg.hpp): 4 waves W 5..14, 8 waves W 8..20, 1 wave W 1..16
// (they fit strips of up to 14 columns into 168 VGPRs = 3 waves per SIMD; wider ones spill, so a read gets the
// narrowest strips its class of workgroup allows)
constexpr int kWgWMax = 20;
#ifndef LTR_WG4_MAXW
#define LTR_WG4_MAXW 20
#endif
constexpr int kWg4MinW = 5, kWg4MaxW = LTR_WG4_MAXW, kNumWg4 = kWg4MaxW - kWg4MinW + 1;
constexpr int kWg8MinW = 8, kNumWg8 = kWgWMax - kWg8MinW + 1;
constexpr int kWg1MaxW = 16;

#endif

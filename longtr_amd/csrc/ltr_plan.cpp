// ltr_plan.cpp -- host-side planning units of a batch: launch-class rule, cost model, class sort (see ltr_plan.h).
// Pure host code: what it decides is WHICH kernel scores a pair and in what order -- never the score
// (every kernel returns the reference's bits, HapAligner.cpp:236-343).

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "ltr_internal.h"
#include "ltr_plan.h"

namespace ltrp {

Rules make_rules(const ModelConsts& mc, int indel_flank_len, int mode, int n_cu, int64_t pairs_upper, int64_t n_long_pairs,
                 const int64_t* pairs_by_bucket, int pack_rule, int plan_knob) {
  Rules R;
  R.mode = mode;
  R.flank = indel_flank_len;
  R.sym_model = (mc.b == mc.d) && (mc.f == mc.g);
  {
    const float cabs = std::fabs(mc.c);
    const int64_t k600 = (cabs * 1.0e9f > 600.0f) ? ((int64_t)(600.0f / cabs) + 2) : (int64_t)1 << 40;
    R.xlut = R.sym_model && k600 <= kPenKMax;
  }
  // Workgroup-per-pair kernels (ltr_dp_wg.hpp; symmetric indel models, ACGT pairs): for reads longer than one
  // wavefront's 1280 columns (64 lanes x the widest strip, W = 20) -- and only while the long pairs alone cannot fill
  // the GPU one wavefront each (fewer than ten per CU; measured on MI355X, 5-kb pairs, workgroup kernels against one
  // wavefront per pair with W = 20 strips: 1536 pairs 2.31e12 against 1.72e12 cells/s, 3008 pairs 2.30 against 2.40,
  // 9216 pairs 2.32 against 2.46).  Their one-wave variant (haplotype rows and first-column table through LDS) is only
  // taken on request (mode 2): a one-locus batch is bound by the instructions issued per step, not by memory latency
  // -- 0.151 ms per config-2 pass against 0.099 ms for the leaner one-wave kernel.
  // Round 3: four waves with WIDE strips (W = 15 .. 20: reads of 3586 .. 5121 bases, three workgroups a CU at three waves a
  // SIMD) against the eight-wave workgroups with their narrow strips (W = 8 .. 10, two a CU) and against column blocks on
  // one wavefront, measured on MI355X: 1536 pairs of 3.7 / 4.3 / 4.9 kb 2.51 / 2.53 / 2.70e12 cells/s against 1.95 / 2.12 /
  // 2.26e12 on eight waves; 9216 pairs of 4.9 kb 2.72e12 against 2.45e12 on one wavefront each.  A four-wave pair lasts
  // 1.26 x its eight-wave time on half the lanes, so what decides is how the pairs fill ROUNDS of 3 against 2 workgroups
  // per CU: 1868 such pairs (config5hifi) are 2.4 rounds of four-wave workgroups, 27.0 ms, or 3.6 of eight-wave ones,
  // 25.9 ms.  Taken when the rounds say so with a tenth to spare, up to 80 long pairs per CU.
  R.wg_long = R.sym_model && mode != 3 && (mode == 2 || n_long_pairs < (int64_t)10 * n_cu);
  {
    int64_t n_wide = n_long_pairs;                                        // pairs of 3585 .. 5120 columns (no histogram: every long pair)
    if (pairs_by_bucket) n_wide = pairs_by_bucket[length_bucket(4 * 64 * (kWg4WideMinW - 1) + 1)] + pairs_by_bucket[length_bucket(4096)];
    const int64_t slots4 = (int64_t)3 * n_cu, slots8 = (int64_t)2 * n_cu;
    const double rounds4 = std::ceil((double)n_wide / (double)slots4) * 1.26, rounds8 = std::ceil((double)n_wide / (double)slots8);
    // (up to 80 long pairs per CU: at 32 000 pairs the two are level -- 2.82e12 both at 4.9 kb, 2.66 against 2.83e12 at 3.7 kb --
    // and the chunked plans of ltr_calc_hap_aln_probs, 48 k and 144 k five-kb pairs each, ran 19 % slower on workgroups)
    R.wg_wide4 = R.sym_model && mode != 3 && (mode == 2 || (n_wide > 0 && n_long_pairs < (int64_t)80 * n_cu && rounds4 < 0.9 * rounds8));
    R.wide4_quota = INT64_MAX;
    // ... or BOTH: whole rounds of four-wave workgroups, the rest on eight waves (config5hifi: 1536 of its 1868 pairs as two
    // rounds of four-wave workgroups, the other 332 with the 180 longer pairs as one round of eight-wave ones).  Only while
    // the eight-wave kernels are in use at all (few long pairs), and only if the rounds say so.
    if (R.sym_model && mode < 0 && R.wg_long && n_wide > slots4) {
      const int64_t full = n_wide / slots4, rest = n_wide - full * slots4;
      const double split = (double)full * 1.26 + std::ceil((double)rest / (double)slots8);
      if (rest > 0 && split < 0.9 * rounds8 && split < rounds4) { R.wg_wide4 = true; R.wide4_quota = full * slots4; }
    }
  }
  {
    // A gap of L bases costs open + (L - 1) x extend + close, and which transitions those are depends on its direction
    // (HapAligner.cpp:285-295): the haplotype window LONGER than the read (n > m) is the insertion state -- match->ins f, ins->ins a,
    // ins->match b --, the read longer is the deletion state -- g, c, d.  A pair whose length difference alone costs more than ~520
    // of the 600 the reference allows cannot hold a one-cell-per-lane certificate: it starts with the exact body.  (Round 6: one
    // threshold per direction -- under a model with a != c the cheaper direction's threshold let the other direction's pairs through,
    // 214 of them in a 1250-locus shard, each a millisecond-long exact body met late in the launch.)
    auto first_risky = [](double open_close, double ext) -> int {
      return ext > 1e-3 ? (int)std::min(1.0e9, std::max(1.0, std::ceil((520.0 - open_close) / ext + 1.0))) : 0x7fffffff;
    };
    if (mode < 0) {
      R.risky_dd_pos = first_risky(std::fabs((double)mc.f) + std::fabs((double)mc.b), std::fabs((double)mc.a));
      R.risky_dd_neg = first_risky(std::fabs((double)mc.g) + std::fabs((double)mc.d), std::fabs((double)mc.c));
    }
  }
  R.wg_short = R.sym_model && mode == 2;
  // plan_knob (ltr_ctx_set_debug "plan_kernel"): 1 = never; 0 and -1 = the rule (every automatic-mode plan, any size, any model)
  // (Measured on MI355X against round 4's launches -- a launch per class / the multi-width launches, exact lists -- on cost shards
  // of config 3: 625 loci 15.1 against 17.8 ms per pass, 1250 loci 30.2 against 32.9, 5000 loci 118.1 against 122.3, all 10 000
  // loci 235.6 against 240.1; the catalogue: 12 500 loci 8.1 against 11.5, 50 000 loci 30.8 against 33.7, all 100 000 loci 60.7
  // against 62.7: profiles/r05/plan_kernel/.)
  // Round 6: the general model too (any seven negative transitions, HapAligner.h:111-119) -- ltr_dp_plan_kernel<false>, the 13-operation
  // cell, failed certificates by the generic exact body in line -- so that --alignment-params with ins != del keeps the one launch.
  R.plan_kernel = mode < 0 && plan_knob <= 0;
  {
    // the threshold test itself (every cell against thr(k), ltrp::build_threshold_table) depends on the band penalty only, not on the
    // model's symmetry: inside the plan kernel the general model's failed certificates and risky pairs take the threshold bodies too
    // (redo_thr_call<W, false>) instead of the byte-compare running-maximum body -- those few pairs were the tail of its small plans
    const float cabs = std::fabs(mc.c);
    const int64_t k600 = (cabs * 1.0e9f > 600.0f) ? ((int64_t)(600.0f / cabs) + 2) : (int64_t)1 << 40;
    R.thr_lists = R.xlut || (R.plan_kernel && k600 <= kPenKMax);
  }
  // (Workgroup launches beside the plan kernel wait for wave slots its persistent workgroups give back only at its end, and the
  // exact launches their failed certificates feed come after that: a 625-locus shard of config 3 -- a few dozen reads of ~1290
  // bases -- ended in them.  As two column blocks on one wavefront such reads are the plan kernel's first pairs, ~2 ms each.)
  // (a batch of fewer pairs than the GPU has wave slots keeps the workgroup kernels from 1281 columns on: nothing starves, and a
  // single locus of 2-kb reads is through sooner on four waves per pair)
  R.wg_min_c = (mode == 2) ? 64 * kWg1MaxW : ((R.plan_kernel && pairs_upper >= (int64_t)12 * n_cu) ? 2 * 64 * kWMax : 64 * kWMax);
  // Several pairs per wavefront is a throughput device: a wave of 64 / LP pairs is as long as its longest pair and a
  // batch that cannot fill the GPU's wave slots anyway (one locus at a time through ltr_process_reads: a few hundred
  // pairs) finishes sooner with one pair per wave.  Automatic mode: as many pairs per wave as still leave 16 waves
  // per CU -- two from 32 pairs per CU up (the rule of the two-per-wave kernels of round 2), 32 from 512 per CU up.
  R.pack_min_shift = 7;
  if (mode < 0) {
    const int64_t per_wave = pairs_upper / ((int64_t)16 * std::max(n_cu, 1));
    if (per_wave >= 2) {
      int np_shift = 1;
      while (np_shift < 6 - kPackMinShift && ((int64_t)2 << np_shift) <= per_wave) ++np_shift;
      R.pack_min_shift = 6 - np_shift;
      if (pack_rule == 2) R.pack_min_shift = kPackMinShift;
    }
  } else if (mode == 1 || (mode >= 5 && mode <= 8)) {
    R.pack_force_shift = (mode == 1) ? 5 : (9 - mode);           // 32 lanes per pair; 16, 8, 4, 2
    R.pack_min_shift = kPackMinShift;
  }
  for (int b = 0; b < kLengthBuckets; ++b) R.bucket_min_shift[b] = (int8_t)kPackMinShift;
  // (Round 4: the per-length floor is off by default, kept behind ltr_ctx_set_debug("pack_rule", 3).  It was there so that the
  // launch of a rare length still put two wavefronts on every SIMD -- when every (lanes per pair, strip width) class was a
  // launch of its own.  A packed launch is a whole strip width now, every lanes-per-pair range in one queue: measured on
  // MI355X, shards of config 3 with and without the floor: 625 loci 17.55 -> 17.13 ms per pass, 1250 loci 32.80 -> 32.34,
  // 2500 and 10 000 loci the same.)
  if (mode < 0 && pairs_by_bucket && pack_rule == 3) {
    const int64_t want_waves = (int64_t)2 * 4 * std::max(n_cu, 1);               // two wavefronts on every SIMD
    for (int b = 0; b < kLengthBuckets; ++b) {
      // (a launch class collects about a third of an octave of lengths: the bucket and its neighbours)
      const int64_t nb = pairs_by_bucket[b] + (b > 0 ? pairs_by_bucket[b - 1] : 0) + (b + 1 < kLengthBuckets ? pairs_by_bucket[b + 1] : 0);
      int np_shift = 0;
      while (np_shift < 6 - kPackMinShift && (nb >> (np_shift + 1)) >= want_waves) ++np_shift;
      R.bucket_min_shift[b] = (int8_t)(6 - np_shift);
    }
  }
  return R;
}

// steps x (cells of a step + what a step costs besides its cells, in cells) x share of the wave x what the register
// budget of the strip width costs in resident waves
static inline double occ_factor(int W) { return W <= 6 ? 1.0 : (W <= 12 ? 1.05 : 1.15); }
constexpr double kStepOverhead = 1.5;

double pack_cost(int n, int C, int lp_shift, int* W_out) {
  const int LP = 1 << lp_shift;
  const int W = (C + LP - 1) / LP;
  if (W_out) *W_out = W;
  if (W > kPackWMax) return 1e300;
  const int L = (C + W - 1) / W;
  return (double)(n - 1 + L - 1 + 1) * ((double)W + kStepOverhead) * occ_factor(W) * (double)LP / 64.0;     // (+ 1: the pair's set-up)
}

PairClass classify_pair(const Rules& R, int64_t n, int64_t m, int64_t hl, bool generic) {
  PairClass pc;
  pc.shortcut = (hl <= 60) || (std::llabs(n - m) > 600);
  double c = 1.0;
  int cls = -1;
  if (!pc.shortcut) {
    int ncb = 1;
    const int W = strip_width_for((int)m, &ncb);
    c = (double)ncb * (double)(n + 63) * (W + kStepOverhead);     // steps x (cells + per-step overhead)
    const int C = (int)m - 1;
    if (!generic && m >= 2 && n >= 2) {
      const bool wide = C > 4 * 64 * (kWg4WideMinW - 1) && C <= 4 * 64 * kWg4MaxW;     // four-wave strips of 15 .. 20 columns
      if (C > R.wg_min_c && C <= 4 * 64 * kWg4MaxW && (wide ? R.wg_wide4 : R.wg_long)) {   // four wavefronts on the pair
        const int Wg = std::max((C + 255) / 256, kWg4MinW);
        cls = kWg4First + Wg - kWg4MinW;
        c = (double)(n + 4 * 64) * (Wg + 2.0);
      } else if (R.wg_long && C > 4 * 64 * (kWg4WideMinW - 1) && C <= 8 * 64 * kWgWMax) {  // eight
        const int Wg = std::max((C + 511) / 512, kWg8MinW);
        cls = kWg8First + Wg - kWg8MinW;
        c = (double)(n + 8 * 64) * (Wg + 2.0);
      } else if (R.wg_short && C <= 64 * kWg1MaxW) {                             // one wavefront, inputs streamed through LDS
        const int Wg = (C + 63) / 64;
        cls = kWg1First + Wg - 1;
        c = (double)(n + 63) * (Wg + 2.0);
      } else if (R.pack_min_shift <= kPackMaxShift && C <= (kPackWMax << kPackMaxShift)) {
        // a read that fits LP lanes x kPackWMax columns can share its wavefront with 64 / LP - 1 other pairs
        int best_shift = 0, best_W = 0;
        double best = (ncb == 1 && R.pack_force_shift == 0) ? c * occ_factor(W) : 1e300;   // (one pair per wave, this wave all to itself)
        for (int s = std::max(std::max(R.pack_min_shift, (int)R.bucket_min_shift[length_bucket(C)]), kPackMinShift); s <= kPackMaxShift; ++s) {
          if (R.pack_force_shift != 0 && s < R.pack_force_shift) continue;
          int Wp = 0;
          const double cp = pack_cost((int)n, C, s, &Wp);
          if (cp < best) { best = cp; best_shift = s; best_W = Wp; }
          if (R.pack_force_shift != 0 && best_shift != 0) break;                  // forced: the narrowest segment from there up that fits
        }
        if (best_shift != 0) { cls = pack_class(best_shift, best_W); c = best; }
      }
    }
  }
  // which exact kernel scores the pair if its certificate fails (push_redo) -- or at once: pairs with bytes
  // outside ACGT (generic list) and, in mode 4, every pair
  const int64_t C = m - 1;
  int xc = kXGeneric;
  if (!generic && R.thr_lists && !pc.shortcut)
    xc = (C <= 64 * kXShortW) ? kXShort : ((C <= 64 * kXMidW) ? kXMid : ((C <= 64 * kXLongW) ? kXLong
         : ((C <= kXWg4MaxC) ? kXWg4 : ((C <= kXWg8MaxC) ? kXWg8 : kXLong))));
  pc.xc = (int8_t)xc;
  pc.x_candidate = !pc.shortcut || generic;
  // (risky pairs: only those a one-wave or packed class would take -- the workgroup classes keep theirs, their exact kernels are fed from the device)
  const bool risky = !pc.shortcut && (n - m >= R.risky_dd_pos || m - n >= R.risky_dd_neg) && (cls < 0 || cls < kWg4First);
  if (generic || (R.mode == 4 && !pc.shortcut) || risky) cls = kNumFast + xc;
  else if (cls < 0) cls = strip_width_for((int)m, nullptr) - 1;
  pc.uses_wg = (cls >= kWg4First && cls < kNumFast);
  pc.cls = (int16_t)cls;
  // (a pair that is scored always has a key >= 1 -- key 0 marks the constant-score pairs -- however small its cost)
  pc.cost = pc.shortcut ? 0.0 : c;
  pc.key = pc.shortcut ? (int16_t)0 : (int16_t)std::min(511, std::max(1, (c > 1.0 ? (int)(std::log2(c) * 16.0) : 0) - 16));
  return pc;
}

void sort_by_class(const int16_t* bin, const int16_t* key, int64_t n_pairs, int fold_rounds, int n_cu, int32_t* order,
                   int* bin_first, int* counts, int multi_launch) {
  // (counted and placed in blocks of kPlanBlock pairs on all host cores: block b's pairs of class k go behind those of
  // the blocks before it, which keeps the input order inside a class)
  const size_t np = (size_t)n_pairs;
  const int64_t n_blk = (int64_t)((np + kPlanBlock - 1) / kPlanBlock);
  std::vector<int32_t> blk_cnt((size_t)n_blk * kNumKernels, 0);
  ltr::parallel_for(n_blk, 1, [&](int64_t c) {
    int32_t* cn = blk_cnt.data() + (size_t)c * kNumKernels;
    for (size_t i = (size_t)c * kPlanBlock; i < std::min(np, ((size_t)c + 1) * kPlanBlock); ++i) cn[bin[i]]++;
  }, 1);
  for (int k = 0; k < kNumKernels; ++k) counts[k] = 0;
  for (int64_t c = 0; c < n_blk; ++c) for (int k = 0; k < kNumKernels; ++k) counts[k] += blk_cnt[(size_t)c * kNumKernels + k];
  int remap[kNumKernels];
  for (int k = 0; k < kNumKernels; ++k) remap[k] = k;
  // Small plans (the chunks of ltr_calc_hap_aln_probs, single loci): a class whose pairs cannot fill the GPU's
  // wave slots even once is folded into the next wider class of its family -- any strip width >= a pair's own
  // scores it with the same bits, only with idle slack columns -- as long as the widest strip of the group stays
  // within a third of its narrowest (or <= 4).  Measured on MI355X: a 600-locus chunk spent 9.7 ms in twenty
  // two-per-wave launches of 200-600 workgroups each, every one as long as its longest pair.  Automatic mode only:
  // the explicit packing modes keep one class per strip width.
  if (fold_rounds > 0) {
    bool any = false;
    {                                                                         // one-wave family
      const int min_fill = fold_rounds * 4 * n_cu;                            // (4 SIMDs per CU; ~3-4 resident wavefronts each: fold_rounds = 3 is one full round)
      int lo_w = 0;                                                           // narrowest strip folded into the running group
      // (the classes of strip widths kMultiMinW and up are one persistent launch when multi_launch is set, ltr_dp_multi_kernel:
      // a class of a few pairs costs nothing there, every pair keeps its own strip width)
      for (int j = 0; j + 1 < (multi_launch == 2 ? 0 : (multi_launch ? kMultiMinW : kNumBins)); ++j) {
        const int k = j, w = j + 1;
        if (counts[k] == 0) { lo_w = 0; continue; }
        if (lo_w == 0) lo_w = w;
        const bool fits = (w + 1 <= 4) || (3 * (w + 1) <= 4 * lo_w);
        // (only towards a class that has pairs of its own: a lone small class keeps its strip width)
        bool target = false;
        for (int j2 = j + 1; j2 < kNumBins && ((j2 + 1 <= 4) || (3 * (j2 + 1) <= 4 * lo_w)); ++j2) if (counts[j2] > 0) { target = true; break; }
        if (counts[k] < min_fill && fits && target) { counts[k + 1] += counts[k]; counts[k] = 0; remap[k] = k + 1; any = true; }
        else lo_w = 0;
      }
    }
    {
      // packed family: a launch is ONE strip width -- the pairs of every lanes-per-pair block of it (ltr_dp_pack.hpp) --
      // so what is folded is a whole width: every (LP, W) class of it into (LP, W + 1), while the wavefronts of the
      // width (groups of 64 / LP pairs) cannot fill the wave slots
      auto waves_of = [&](int w) {
        int64_t v = 0;
        for (int sft = kPackMinShift; sft <= kPackMaxShift; ++sft) { const int per = 64 >> sft; v += (counts[pack_class(sft, w)] + per - 1) / per; }
        return v;
      };
      const int64_t min_fill = (int64_t)fold_rounds * 4 * n_cu;
      int lo_w = 0;
      // (with multi_launch the widths kPackMultiMinW and up are one persistent launch, ltr_dp_pack_multi_kernel: nothing to fold there)
      for (int w = 1; w < (multi_launch == 2 ? 0 : (multi_launch ? kPackMultiMinW : kPackWMax)); ++w) {
        const int64_t wv = waves_of(w);
        if (wv == 0) { lo_w = 0; continue; }
        if (lo_w == 0) lo_w = w;
        const bool fits = (w + 1 <= 4) || (3 * (w + 1) <= 4 * lo_w);
        bool target = false;
        for (int w2 = w + 1; w2 <= kPackWMax && ((w2 <= 4) || (3 * w2 <= 4 * lo_w)); ++w2) if (waves_of(w2) > 0) { target = true; break; }
        if (wv < min_fill && fits && target) {
          for (int sft = kPackMinShift; sft <= kPackMaxShift; ++sft) {
            const int k = pack_class(sft, w);
            if (counts[k] == 0) continue;
            counts[k + 1] += counts[k]; counts[k] = 0; remap[k] = k + 1; any = true;
          }
        } else lo_w = 0;
      }
    }
    // ... and the workgroup families: a class of a few hundred pairs next to another one leaves both launches with a
    // partly filled last round of workgroups (measured on MI355X: 1116 + 420 five-kb pairs as W = 17 and W = 18
    // four-wave classes 2.05e12 cells/s, the neighbouring single-class lengths 2.5 - 2.7e12)
    for (int f = 0; f < 2; ++f) {
      const int first = f == 0 ? kWg4First : kWg8First, nk = f == 0 ? kNumWg4 : kNumWg8, w0 = f == 0 ? kWg4MinW : kWg8MinW;
      const int min_fill = fold_rounds * n_cu;                                // (two or three workgroups per CU: two to three rounds)
      int lo_w = 0;
      for (int j = 0; j + 1 < nk; ++j) {
        const int k = first + j, w = w0 + j;
        if (counts[k] == 0) { lo_w = 0; continue; }
        if (lo_w == 0) lo_w = w;
        const bool fits = 3 * (w + 1) <= 4 * lo_w;
        bool target = false;
        for (int j2 = j + 1; j2 < nk && 3 * (w0 + j2) <= 4 * lo_w; ++j2) if (counts[first + j2] > 0) { target = true; break; }
        if (counts[k] < min_fill && fits && target) { counts[k + 1] += counts[k]; counts[k] = 0; remap[k] = k + 1; any = true; }
        else lo_w = 0;
      }
    }
    if (any)
      for (int k = kNumKernels - 2; k >= 0; --k) if (remap[k] != k) remap[k] = remap[remap[k]];     // (chains resolve wide to narrow)
  }
  bin_first[0] = 0;
  for (int k = 0; k < kNumKernels; ++k) bin_first[k + 1] = bin_first[k] + counts[k];
  {
    // where block c starts inside every (folded) class
    std::vector<int32_t> blk_at((size_t)n_blk * kNumKernels, 0);
    int fill[kNumKernels];
    for (int k = 0; k < kNumKernels; ++k) fill[k] = bin_first[k];
    for (int64_t c = 0; c < n_blk; ++c) {
      int32_t* at = blk_at.data() + (size_t)c * kNumKernels;
      for (int k = 0; k < kNumKernels; ++k) at[k] = -1;
      for (int k = 0; k < kNumKernels; ++k) {
        const int32_t n_k = blk_cnt[(size_t)c * kNumKernels + k];
        if (n_k == 0) continue;
        const int t = remap[k];
        if (at[t] < 0) at[t] = fill[t];
        fill[t] += n_k;
      }
    }
    ltr::parallel_for(n_blk, 1, [&](int64_t c) {
      int32_t at[kNumKernels];
      std::memcpy(at, blk_at.data() + (size_t)c * kNumKernels, sizeof(at));
      for (size_t i = (size_t)c * kPlanBlock; i < std::min(np, ((size_t)c + 1) * kPlanBlock); ++i) order[(size_t)at[remap[bin[i]]]++] = (int32_t)i;
    }, 1);
  }
  {
    // Every class longest first (by the 1/16-octave key).  A class is cut into segments of <= 32 k pairs: the
    // segments of all classes are counting-sorted side by side on the host cores, then merged pairwise, level by
    // level (a catalogue of short repeats puts half a million pairs into one class: 13.5 ms on one core before this).
    auto longer = [&](int32_t x, int32_t y) { return key[(size_t)x] > key[(size_t)y]; };
    struct Seg { int32_t a, b; };
    constexpr int32_t kSeg = 32768;
    std::vector<Seg> segs;
    std::vector<std::vector<int32_t>> cuts((size_t)kNumKernels);        // per class: segment boundaries
    for (int k = 0; k < kNumKernels; ++k) {
      const int32_t a = bin_first[k], b2 = bin_first[k + 1];
      if (b2 <= a) continue;
      const int32_t ns = (b2 - a + kSeg - 1) / kSeg;
      for (int32_t i = 0; i <= ns; ++i) cuts[(size_t)k].push_back(a + (int32_t)((int64_t)(b2 - a) * i / ns));
      for (int32_t i = 0; i < ns; ++i) segs.push_back({cuts[(size_t)k][(size_t)i], cuts[(size_t)k][(size_t)i + 1]});
    }
    ltr::parallel_for((int64_t)segs.size(), 1, [&](int64_t i) {
      // counting sort of the segment by key, longest first, input order kept inside a key
      int32_t* seg = order + segs[(size_t)i].a;
      const int32_t n_seg = segs[(size_t)i].b - segs[(size_t)i].a;
      if (n_seg < 64) { std::stable_sort(seg, seg + n_seg, longer); return; }
      int32_t at[513] = {0};
      for (int32_t k = 0; k < n_seg; ++k) at[512 - (int)key[(size_t)seg[k]]]++;           // slot 1 + (511 - key)
      for (int q = 1; q <= 512; ++q) at[q] += at[q - 1];
      std::vector<int32_t> tmp(seg, seg + n_seg);
      for (int32_t k = 0; k < n_seg; ++k) seg[at[511 - (int)key[(size_t)tmp[(size_t)k]]]++] = tmp[(size_t)k];
    }, 1);
    for (;;) {                                                          // merge levels: neighbours of every class, all classes at once
      struct Mrg { int32_t a, m, b; };
      std::vector<Mrg> work;
      for (int k = 0; k < kNumKernels; ++k) {
        std::vector<int32_t>& c = cuts[(size_t)k];
        if (c.size() <= 2) continue;
        std::vector<int32_t> next;
        size_t i = 0;
        for (; i + 2 < c.size(); i += 2) { work.push_back({c[i], c[i + 1], c[i + 2]}); next.push_back(c[i]); }
        for (; i < c.size(); ++i) next.push_back(c[i]);
        if (next.back() != c.back()) next.push_back(c.back());
        c.swap(next);
      }
      if (work.empty()) break;
      ltr::parallel_for((int64_t)work.size(), 1, [&](int64_t i) {
        std::inplace_merge(order + work[(size_t)i].a, order + work[(size_t)i].m, order + work[(size_t)i].b, longer);
      }, 1);
    }
  }
}

void build_threshold_table(float c32, double* out) {
  const float cabs = std::fabs(c32);
  const int k600 = (cabs * 1.0e9f > 600.0f) ? ((int)(600.0f / cabs) + 2) : 0x3fffffff;   // as the kernels form it
  const double inf = std::numeric_limits<double>::infinity();
  auto bits = [](double v) { int64_t b; std::memcpy(&b, &v, 8); return b; };
  auto dbl = [](int64_t b) { double v; std::memcpy(&v, &b, 8); return v; };
  for (int idx = 0; idx < kPenTabDoubles; ++idx) {
    const int k = std::abs(idx - kPenHalf);
    const volatile double p = (double)((float)k * c32);        // int * float -> float, HapAligner.cpp:298 (volatile: no contraction, no excess precision)
    const double x0 = -600.0 - p;
    double thr = inf;
    if (k < k600 && k <= kPenKMax && x0 < -1e-6) {
      // bisection over the bit patterns between a double that passes and one that fails (both negative: the larger pattern is
      // further down).  Steps of one ulp from x0 do not get there: at k = 597, c = -1 x0 is -3 and the threshold half an ulp
      // of 600 = 128 ulps of 3 below it.
      int64_t lo = bits(x0 + 1e-9), hi = bits(x0 - 1e-9);
      while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        const volatile double sum = dbl(mid) + p;
        if (sum >= -600.0) lo = mid; else hi = mid;
      }
      thr = dbl(lo);
    }
    out[idx] = thr;
  }
}

}  // namespace ltrp

// ---- test hooks (include/ltr_gpu.h, "planning units"): the rule and the sort without a GPU -------------------
extern "C" {

int ltr_debug_num_classes(void) { return ltrp::kNumKernels; }

int ltr_debug_class_info(int k, int* family, int* strip_width, int* waves_per_pair, int* lanes_per_pair) {
  if (k < 0 || k >= ltrp::kNumKernels) return LTR_ERR_INVALID;
  if (k >= ltrp::kNumFast) {
    const int xc = k - ltrp::kNumFast;
    static const int kXW[kNumExact] = {kExactW, kXShortW, kXMidW, kXLongW, 0, 0};
    if (family) *family = ltrp::kFamExact;
    if (strip_width) *strip_width = kXW[xc];
    if (waves_per_pair) *waves_per_pair = (xc == kXWg4) ? 4 : ((xc == kXWg8) ? 8 : 1);
    if (lanes_per_pair) *lanes_per_pair = (xc == kXWg4) ? 256 : ((xc == kXWg8) ? 512 : 64);
    return LTR_OK;
  }
  const ltrp::ClassInfo ci = ltrp::class_info(k);
  if (family) *family = ci.family;
  if (strip_width) *strip_width = ci.W;
  if (waves_per_pair) *waves_per_pair = ci.waves;
  if (lanes_per_pair) *lanes_per_pair = ci.family == ltrp::kFamPack ? (1 << ci.lp_shift) : 64 * ci.waves;
  return LTR_OK;
}

int ltr_debug_classify(const ltr_align_params* p, int mode, int n_cu, int64_t pairs_in_batch, int64_t long_pairs_in_batch,
                       int32_t window_len, int32_t read_len, int32_t hap_full_len, int generic,
                       int* launch_class, int* order_key, int* exact_list) {
  if (!p || n_cu <= 0) return LTR_ERR_INVALID;
  ModelConsts mc;
  mc.a = p->log_ins_to_ins; mc.b = p->log_ins_to_match; mc.c = p->log_del_to_del; mc.d = p->log_del_to_match;
  mc.e = p->log_match_to_match; mc.f = p->log_match_to_ins; mc.g = p->log_match_to_del;
  mc.match = mc.mismatch = mc.match_plus_f = 0.f;
  const ltrp::Rules R = ltrp::make_rules(mc, p->indel_flank_len, mode, n_cu, pairs_in_batch, long_pairs_in_batch);
  const ltrp::PairClass pc = ltrp::classify_pair(R, window_len, read_len, hap_full_len, generic != 0);
  if (launch_class) *launch_class = pc.cls;
  if (order_key) *order_key = pc.key;
  if (exact_list) *exact_list = pc.xc;
  return LTR_OK;
}

int ltr_debug_pair_costs(const ltr_align_params* p, int mode, int n_cu, int64_t pairs_in_batch, int64_t long_pairs_in_batch, int64_t n,
                         const int32_t* window_len, const int32_t* read_len, const int32_t* hap_full_len, double* cost) {
  if (!p || n_cu <= 0 || n < 0 || (n > 0 && (!window_len || !read_len || !hap_full_len || !cost))) return LTR_ERR_INVALID;
  ModelConsts mc;
  mc.a = p->log_ins_to_ins; mc.b = p->log_ins_to_match; mc.c = p->log_del_to_del; mc.d = p->log_del_to_match;
  mc.e = p->log_match_to_match; mc.f = p->log_match_to_ins; mc.g = p->log_match_to_del;
  mc.match = mc.mismatch = mc.match_plus_f = 0.f;
  const ltrp::Rules R = ltrp::make_rules(mc, p->indel_flank_len, mode, n_cu, pairs_in_batch, long_pairs_in_batch);
  for (int64_t i = 0; i < n; ++i) cost[i] = ltrp::classify_pair(R, window_len[i], read_len[i], hap_full_len[i], false).cost;
  return LTR_OK;
}

int ltr_debug_threshold_table(float log_del_to_del, double* out, int64_t cap) {
  if (!out || cap < kPenTabDoubles || !(log_del_to_del < 0.f)) return LTR_ERR_INVALID;
  ltrp::build_threshold_table(log_del_to_del, out);
  return kPenTabDoubles;
}

int ltr_debug_sort_by_class(const int16_t* launch_class, const int16_t* order_key, int64_t n_pairs, int fold, int n_cu,
                            int32_t* order, int32_t* class_first /* [ltr_debug_num_classes() + 1] */) {
  if ((!launch_class || !order_key || !order) && n_pairs > 0) return LTR_ERR_INVALID;
  if (n_pairs < 0 || n_pairs > 0x7fffffff || !class_first || n_cu <= 0) return LTR_ERR_INVALID;
  for (int64_t i = 0; i < n_pairs; ++i) if (launch_class[i] < 0 || launch_class[i] >= ltrp::kNumKernels || order_key[i] < 0 || order_key[i] > 511) return LTR_ERR_INVALID;
  int bf[ltrp::kNumKernels + 1], counts[ltrp::kNumKernels];
  // fold: 0 = no folding; 1 = automatic mode's folding, a launch per class; 2 = ... with the multi-width launches (plans of
  // 512 .. 4096 pairs per CU before round 5, asymmetric models since); 3 = ... with the plan kernel (no folding in its families)
  try { ltrp::sort_by_class(launch_class, order_key, n_pairs, fold != 0 ? ltrp::kFoldRounds : 0, n_cu, order, bf, counts, fold >= 2 ? fold - 1 : 0); }
  catch (...) { return LTR_ERR_NOMEM; }
  for (int k = 0; k <= ltrp::kNumKernels; ++k) class_first[k] = bf[k];
  return LTR_OK;
}

}  // extern "C"

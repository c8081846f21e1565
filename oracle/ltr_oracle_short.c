/*
 * ltr_oracle_short.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See ltr_oracle.h.
 *
 * Plain-C restatement of LongTR's SHORT (seeded, stutter-aware) alignment path, SURVEY.md
 * section 8a row a-7: taken by HapAligner::process_reads only when the repeat period is 1 and
 * --stutter-align-len is set (HapAligner.cpp:552).
 *
 *   HapAligner::calc_seed_base / calc_best_seed_position   HapAligner.cpp:467-542
 *   HapAligner::process_read, short_ branch                HapAligner.cpp:855-990 (retrace=false)
 *   HapAligner::align_seq_to_hap_short                     HapAligner.cpp:27-163
 *   HapAligner::compute_aln_logprob                        HapAligner.cpp:165-233
 *   StutterAlignerClass (ctor, load_read, align_*)         StutterAlignerClass.h:45-79, .cpp:12-166
 *   RepeatStutterInfo::log_prob_pcr_artifact               RepeatStutterInfo.h:53-61
 *   StutterModel::log_stutter_pmf (+ctor logs)             stutter_model.cpp:29-53, .h:35-62
 *   BaseQuality tables                                     base_quality.h:29-75
 *   fast_log_sum_exp(vector), fasterexp, fasterlog         mathops.cpp:98-107, fastonebigheader.h:207-218,349-358
 *
 * PARITY: everything that links without htslib IS pinned to the compiled reference (oracle/_ref, ref_driver.cpp):
 * the whole stutter-block row (StutterAlignerClass ctor / load_read / align_*_reverse, RepeatStutterInfo,
 * StutterModel::log_stutter_pmf, BaseQuality, fast_log_sum_exp(vector) with its FP32 bit tricks) --
 * tests/golden/stutter_pieces.json + live -- and the outer functions compute_aln_logprob (:165-233),
 * calc_best_seed_position (:467-493), calc_seed_base (:494-542) -- tests/golden/short_outer.json + live
 * (tests/test_short_path.py).  What stays UNPINNED is the flank-row loop of align_seq_to_hap_short (:113-159): it calls
 * Haplotype::homopolymer_length (HapAligner.cpp:121-122 -> Haplotype.cpp:280), and Haplotype.cpp includes bam_io.h
 * (htslib), as does Haplotype::reverse; for those rows the only reference outputs are the two known-answer values of
 * SURVEY.md 8c (-7.8693081508, -4.3896419406).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "ltr_oracle.h"

#define IMPOSSIBLE (-1000000000.0)                 /* HapAligner.cpp:20 */
#define MIN_SEED_DIST 5                            /* HapAligner.cpp:17 */
#define MAX_STUTTER_REPEAT_INS 6                   /* RepeatStutterInfo.h:10-12 */
#define MAX_STUTTER_REPEAT_DEL (-6)
#define LARGE_NEGATIVE (-10e6)

static inline double dmax(double a, double b) { return a < b ? b : a; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static double int_log(int v) { return v == 0 ? -1000.0 : log((double)v); }   /* mathops.cpp:14-22 */

/* fastonebigheader.h:207-218 */
static inline float fasterpow2(float p) {
  float clipp = (p < -126) ? -126.0f : p;
  union { uint32_t i; float f; } v;
  v.i = (uint32_t)((1 << 23) * (clipp + 126.94269504f));
  return v.f;
}
static inline float fasterexp(float p) { return fasterpow2(1.442695040f * p); }
/* fastonebigheader.h:349-358 */
static inline float fasterlog(float x) {
  union { float f; uint32_t i; } vx;
  vx.f = x;
  float y = vx.i;
  y *= 8.2629582881927490e-8f;
  return y - 87.989971088f;
}

/* fast_log_sum_exp(const std::vector<double>&), mathops.cpp:98-107 */
static double fast_log_sum_exp(const double* v, int n) {
  const double LOG_THRESH = log(0.001);            /* mathops.h:36 */
  double max_val = v[0];
  for (int i = 1; i < n; i++) if (max_val < v[i]) max_val = v[i];
  double total = 0;
  for (int i = 0; i < n; i++) {
    double diff = v[i] - max_val;
    if (diff > LOG_THRESH) total += fasterexp(diff);
  }
  return max_val + fasterlog(total);
}

/* ---- BaseQuality (base_quality.h:29-75) ------------------------------------------------ */
typedef struct { double log_correct[256], log_error[256]; } base_quality_t;
static void base_quality_init(base_quality_t* bq) {
  const int MAX_QUAL_INDEX = 'J' - '!';
  bq->log_correct[0] = -100; bq->log_error[0] = 0;
  for (int i = 1; i <= MAX_QUAL_INDEX; ++i) {
    bq->log_correct[i] = log(1.0 - pow(10.0, i / (-10.0)));
    bq->log_error[i] = log(pow(10.0, i / (-10.0) / 5.0));
  }
}
static double bq_log_error(const base_quality_t* bq, char q) {
  if (q < '!') return bq->log_error[0];
  if (q > 'J') return bq->log_error['J' - '!'];
  return bq->log_error[q - '!'];
}
static double bq_log_correct(const base_quality_t* bq, char q) {
  if (q < '!') return bq->log_correct[0];
  if (q > 'J') return bq->log_correct['J' - '!'];
  return bq->log_correct[q - '!'];
}

/* ---- StutterModel logs (stutter_model.h:35-62) + pmf (stutter_model.cpp:29-53) ---------- */
typedef struct {
  double in_log_nostep, in_log_step, in_log_up, in_log_down, log_equal;
  double out_log_nostep, out_log_step, out_log_up, out_log_down;
  int motif_len;
} stutter_model_t;
static void stutter_model_init(stutter_model_t* sm, const ltr_stutter_params* p, int motif_len) {
  sm->in_log_step = log(1 - p->in_geom); sm->in_log_nostep = log(p->in_geom);
  sm->in_log_up = log(p->in_up); sm->in_log_down = log(p->in_down);
  sm->out_log_step = log(1 - p->out_geom); sm->out_log_nostep = log(p->out_geom);
  sm->out_log_up = log(p->out_up); sm->out_log_down = log(p->out_down);
  sm->log_equal = log(1 - p->in_up - p->in_down - p->out_up - p->out_down);
  sm->motif_len = motif_len;
}
static double log_stutter_pmf(const stutter_model_t* sm, int sample_bps, int read_bps) {
  int bp_diff = read_bps - sample_bps;
  if (bp_diff % sm->motif_len != 0) {
    int eff_diff = bp_diff - (bp_diff / sm->motif_len);
    if (eff_diff < 0) return sm->out_log_down + sm->out_log_nostep + sm->out_log_step * (-eff_diff - 1);
    return sm->out_log_up + sm->out_log_nostep + sm->out_log_step * (eff_diff - 1);
  }
  int rep_diff = bp_diff / sm->motif_len;
  if (rep_diff == 0) return sm->log_equal;
  if (rep_diff < 0) return sm->in_log_down + sm->in_log_nostep + sm->in_log_step * (-rep_diff - 1);
  return sm->in_log_up + sm->in_log_nostep + sm->in_log_step * (rep_diff - 1);
}
/* RepeatStutterInfo::log_prob_pcr_artifact, RepeatStutterInfo.h:53-61 */
static double log_prob_pcr_artifact(const stutter_model_t* sm, int allele_size, int period, int artifact_size) {
  const int max_ins = MAX_STUTTER_REPEAT_INS * period, max_del = MAX_STUTTER_REPEAT_DEL * period;
  int read_size = allele_size + artifact_size;
  if (artifact_size == 0) return log_stutter_pmf(sm, allele_size, read_size);
  if (artifact_size > 0) return artifact_size > max_ins ? LARGE_NEGATIVE : log_stutter_pmf(sm, allele_size, read_size);
  return (artifact_size < max_del || read_size < 0) ? LARGE_NEGATIVE : log_stutter_pmf(sm, allele_size, read_size);
}

/* ---- StutterAlignerClass ------------------------------------------------------------------ */
typedef struct {
  const char* block_end;   /* points at the LAST character of the block (block_seq_ after the ctor) */
  int block_len, period, left_align;
  int num_insertions, num_deletions, max_insertion, max_deletion;
  int** upstream;          /* upstream_match_lengths_ */
  int n_upstream;
  double *ins_probs, *del_probs, *match_probs;
  double* scratch;         /* log_probs_ */
} stutter_aligner_t;

static int* num_upstream_matches(const char* seq, int len, int period) {   /* StutterAlignerClass.h:34-41 */
  int* ml = (int*)malloc(sizeof(int) * (size_t)(len > 0 ? len : 1));
  for (int i = 0; i < imin(period, len); i++) ml[i] = 0;
  for (int i = period; i < len; i++) ml[i] = (seq[i - period] != seq[i] ? 0 : 1 + ml[i - 1]);
  return ml;
}
static void sa_init(stutter_aligner_t* sa, const char* seq, int len, int period, int left_align) {   /* .h:45-79 */
  memset(sa, 0, sizeof(*sa));
  sa->block_len = len; sa->period = period; sa->left_align = left_align;
  sa->block_end = len ? seq + (len - 1) : NULL;
  sa->num_insertions = (MAX_STUTTER_REPEAT_INS * period) / period;
  sa->num_deletions = -1 * ((MAX_STUTTER_REPEAT_DEL * period) / period);
  while (sa->num_deletions * period > len) sa->num_deletions--;
  sa->max_insertion = period * sa->num_insertions;
  sa->max_deletion = -period * sa->num_deletions;
  sa->upstream = (int**)malloc(sizeof(int*) * 16);
  for (int i = -period; i >= sa->max_deletion; i -= period) sa->upstream[sa->n_upstream++] = num_upstream_matches(seq, len, -i);
  if (sa->max_deletion == 0) sa->upstream[sa->n_upstream++] = len ? num_upstream_matches(seq, len, period) : NULL;
  sa->scratch = (double*)malloc(sizeof(double) * (size_t)(len + 8));
}
static void sa_free(stutter_aligner_t* sa) {
  for (int i = 0; i < sa->n_upstream; i++) free(sa->upstream[i]);
  free(sa->upstream); free(sa->ins_probs); free(sa->del_probs); free(sa->match_probs); free(sa->scratch);
}
/* StutterAlignerClass::load_read, .cpp:12-53; base_seq etc. point at the LAST element */
static void sa_load_read(stutter_aligner_t* sa, int base_seq_len, const char* base_seq,
                         const double* wrong, const double* correct) {
  free(sa->ins_probs); free(sa->del_probs); free(sa->match_probs);
  sa->ins_probs = (double*)malloc(sizeof(double) * (size_t)imax(base_seq_len * sa->num_insertions, 1));
  sa->match_probs = (double*)malloc(sizeof(double) * (size_t)imax(base_seq_len, 1));
  sa->del_probs = sa->num_deletions ? (double*)malloc(sizeof(double) * (size_t)imax(base_seq_len * sa->num_deletions, 1)) : NULL;
  const char* blk = sa->block_end;
  int ins_index = 0, del_index = 0, match_index = 0;
  for (int i = 0; i < base_seq_len; i++) {
    int j;
    double log_prob = 0.0;
    for (j = 0; j < imin(base_seq_len - i, -sa->max_deletion); j++) {
      log_prob += (base_seq[-i - j] == blk[-j] ? correct[-i - j] : wrong[-i - j]);
      if ((j + 1) % sa->period == 0) sa->del_probs[del_index++] = log_prob;
    }
    for (; j < -sa->max_deletion; j++) if ((j + 1) % sa->period == 0) del_index++;
    for (; j < imin(base_seq_len - i, sa->block_len); j++)
      log_prob += (base_seq[-i - j] == blk[-j] ? correct[-i - j] : wrong[-i - j]);
    sa->match_probs[match_index++] = log_prob;
    double log_ins_prob = 0.0;
    for (j = 0; j < imin(sa->max_insertion, base_seq_len - i); j++) {
      if (j % sa->period < sa->block_len)
        log_ins_prob += (base_seq[-i - j] == blk[-(j % sa->period)] ? correct[-i - j] : wrong[-i - j]);
      else
        log_ins_prob += correct[-i - j];
      if ((j + 1) % sa->period == 0) sa->ins_probs[ins_index++] = log_ins_prob;
    }
    for (; j < sa->max_insertion; j++) if ((j + 1) % sa->period == 0) sa->ins_probs[ins_index++] = log_ins_prob;
  }
}
/* align_pcr_insertion_reverse, .cpp:59-104 */
static double sa_insertion(stutter_aligner_t* sa, int base_seq_len, const char* base_seq, int offset,
                           const double* wrong, const double* correct, int D) {
  int np = 0; double* lp = sa->scratch;
  const char* blk = sa->block_end;
  double log_prior = -int_log(sa->block_len + 1);
  const int* up = sa->upstream[0] + sa->block_len - 1;
  double log_prob = log_prior + sa->ins_probs[sa->num_insertions * offset + D / sa->period - 1] +
                    (base_seq_len > D ? sa->match_probs[offset + D] : 0);
  lp[np++] = log_prob;
  int i = 0;
  for (; i > -imin(imax(0, base_seq_len - D), sa->block_len); i--) {
    if (-i + sa->period < sa->block_len) {
      if (up[i] == 0) {
        for (int index = i - sa->period; index >= i - D; index -= sa->period) {
          log_prob -= (base_seq[index] == blk[i] ? correct[index] : wrong[index]);
          log_prob += (base_seq[index] == blk[i - sa->period] ? correct[index] : wrong[index]);
        }
        lp[np++] = log_prob;
      } else {
        lp[np++] = int_log(up[i]) + log_prob;
        i -= (up[i] - 1);
      }
    } else
      lp[np++] = log_prob;
  }
  if (i > -sa->block_len) lp[np++] = int_log(sa->block_len + i) + log_prob;
  return fast_log_sum_exp(lp, np);
}
/* align_pcr_deletion_reverse, .cpp:106-154 */
static double sa_deletion(stutter_aligner_t* sa, int base_seq_len, const char* base_seq, int offset,
                          const double* wrong, const double* correct, int D) {
  int np = 0; double* lp = sa->scratch;
  const char* blk = sa->block_end;
  const int* up = sa->upstream[-D / sa->period - 1] + sa->block_len - 1;
  double log_prior = -int_log(sa->block_len + D + 1);
  double log_prob = log_prior;
  if (offset + D >= 0)
    log_prob += sa->match_probs[offset + D] - sa->del_probs[(offset + D) * sa->num_deletions - D / sa->period - 1];
  else
    for (int j = 0; j > -base_seq_len; j--) log_prob += (blk[j + D] == base_seq[j] ? correct[j] : wrong[j]);
  lp[np++] = log_prob;
  int i;
  for (i = 0; i > -base_seq_len; i--) {
    if (up[i] == 0) {
      log_prob -= (blk[i + D] == base_seq[i] ? correct[i] : wrong[i]);
      log_prob += (blk[i] == base_seq[i] ? correct[i] : wrong[i]);
      lp[np++] = log_prob;
    } else {
      lp[np++] = int_log(up[i]) + log_prob;
      i -= (up[i] - 1);
    }
  }
  if (-i < sa->block_len + D) lp[np++] = int_log(sa->block_len + D + i) + log_prob;
  return fast_log_sum_exp(lp, np);
}
/* align_stutter_region_reverse, .cpp:156-166 */
static double sa_region(stutter_aligner_t* sa, int base_seq_len, const char* base_seq, int offset,
                        const double* wrong, const double* correct, int D) {
  if (D == 0) return sa->match_probs[offset];
  if (D > 0) return sa_insertion(sa, base_seq_len, base_seq, offset, wrong, correct, D);
  return sa_deletion(sa, base_seq_len, base_seq, offset, wrong, correct, D);
}

/* The stutter-block rows of align_seq_to_hap_short, HapAligner.cpp:64-111: for every read position j the log-sum over
 * the artifact sizes of pmf + stutter-aligner LL + the match value base_len positions back in the row before the block. */
static void stutter_block_row(const stutter_model_t* sm, const char* block_seq, int block_len, int period, int left_align,
                              const char* seq_0, int seq_len, const double* wrong, const double* correct,
                              const double* prev_row, double* out_match) {
  const int max_ins = MAX_STUTTER_REPEAT_INS * period, max_del = MAX_STUTTER_REPEAT_DEL * period;
  stutter_aligner_t sa;
  sa_init(&sa, block_seq, block_len, period, left_align);
  sa_load_read(&sa, seq_len, seq_0 + seq_len - 1, wrong + seq_len - 1, correct + seq_len - 1);   /* :76 */
  double block_probs[16];
  int offset = seq_len - 1;
  for (int j = 0; j < seq_len; ++j, --offset) {
    int art_idx = 0;
    for (int artifact_size = max_del; artifact_size <= max_ins; artifact_size += period) {
      const int base_len = imin(block_len + artifact_size, j + 1);
      if (base_len >= 0) {
        const double prob = sa_region(&sa, base_len, seq_0 + j, offset, wrong + j, correct + j, artifact_size);
        const double pre_prob = (j - base_len < 0 ? 0 : prev_row[j - base_len]);
        block_probs[art_idx] = log_prob_pcr_artifact(sm, block_len, period, artifact_size) + prob + pre_prob;
      } else
        block_probs[art_idx] = IMPOSSIBLE;
      art_idx++;
    }
    out_match[j] = fast_log_sum_exp(block_probs, art_idx);
  }
  sa_free(&sa);
}

/* ---- the same pieces by themselves: what oracle/_ref (the compiled reference) can be asked too ---------------- */
double ltr_oracle_log_stutter_pmf(const ltr_stutter_params* sp, int32_t motif_len, int32_t sample_bps, int32_t read_bps) {
  stutter_model_t sm; stutter_model_init(&sm, sp, motif_len);
  return log_stutter_pmf(&sm, sample_bps, read_bps);
}
double ltr_oracle_log_prob_pcr_artifact(const ltr_stutter_params* sp, int32_t period, int32_t allele_size, int32_t artifact_size) {
  stutter_model_t sm; stutter_model_init(&sm, sp, period);
  return log_prob_pcr_artifact(&sm, allele_size, period, artifact_size);
}
void ltr_oracle_base_quality(int32_t quality_char, double* log_error, double* log_correct) {
  base_quality_t bq; base_quality_init(&bq);
  *log_error = bq_log_error(&bq, (char)quality_char); *log_correct = bq_log_correct(&bq, (char)quality_char);
}
double ltr_oracle_fast_log_sum_exp_vec(const double* vals, int32_t n) { return fast_log_sum_exp(vals, n); }
int ltr_oracle_stutter_block_row(const ltr_stutter_params* sp, const char* block_seq, int32_t block_len, int32_t period, int32_t left_align,
                                 const char* seq_0, const char* qual, int32_t seq_len, const double* prev_row, double* out_match) {
  if (seq_len <= 0 || period <= 0 || block_len < 0) return LTR_ERR_INVALID;
  stutter_model_t sm; stutter_model_init(&sm, sp, period);
  base_quality_t bq; base_quality_init(&bq);
  double* wrong = (double*)malloc(sizeof(double) * (size_t)seq_len * 2);
  if (!wrong) return LTR_ERR_NOMEM;
  double* correct = wrong + seq_len;
  for (int j = 0; j < seq_len; ++j) { wrong[j] = bq_log_error(&bq, qual[j]); correct[j] = bq_log_correct(&bq, qual[j]); }
  stutter_block_row(&sm, block_seq, block_len, period, left_align, seq_0, seq_len, wrong, correct, prev_row, out_match);
  free(wrong);
  return LTR_OK;
}

/* ---- one haplotype (fw or reversed), as a flat list of blocks ------------------------------- */
typedef struct {
  int n_blocks;
  const char* seq[8]; int len[8]; int is_repeat[8]; int period[8]; int option[8];
  int cur_size;
} flat_hap_t;

/* HapAligner::align_seq_to_hap_short, HapAligner.cpp:27-163 (reuse_alns == false everywhere: the
 * skipped blocks would hold identical rows) */
static void align_short(const ltr_align_params* P, const stutter_model_t* sm, const flat_hap_t* H, int left_align,
                        const char* seq_0, int seq_len, const double* wrong, const double* correct,
                        double* Mm, double* Im, double* Dm, double* left_prob_out) {
  const float a = P->log_ins_to_ins, b = P->log_ins_to_match, c = P->log_del_to_del, d = P->log_del_to_match,
              e = P->log_match_to_match, f = P->log_match_to_ins, g = P->log_match_to_del;
  double left_prob = 0.0;
  const char first_hap_base = H->seq[0][0];
  for (int j = 0; j < seq_len; ++j) {                          /* :36-44 */
    Mm[j] = (seq_0[j] == first_hap_base ? correct[j] : wrong[j]) + left_prob;
    Im[j] = correct[j] + left_prob;
    Dm[j] = IMPOSSIBLE;
    left_prob += correct[j];
  }
  int haplotype_index = 1, matrix_index = seq_len, stutter_R = -1;
  for (int bi = 0; bi < H->n_blocks; bi++) {
    const char* block_seq = H->seq[bi];
    const int block_len = H->len[bi];
    if (H->is_repeat[bi]) {                                    /* :64-111 */
      const int period = H->period[bi];
      const int prev_row_index = seq_len * (haplotype_index - 1);
      matrix_index = seq_len * (haplotype_index + block_len - 1);
      stutter_block_row(sm, block_seq, block_len, period, left_align, seq_0, seq_len, wrong, correct, Mm + prev_row_index, Mm + matrix_index);
      for (int j = 0; j < seq_len; ++j, ++matrix_index) { Im[matrix_index] = IMPOSSIBLE; Dm[matrix_index] = IMPOSSIBLE; }
      stutter_R = haplotype_index + block_len - 1;
      haplotype_index += block_len;
    } else {                                                   /* :112-159 */
      int coord_index = (bi == 0 ? 1 : 0);
      for (; coord_index < block_len; ++coord_index, ++haplotype_index) {
        const char hap_char = block_seq[coord_index];
        Mm[matrix_index] = (seq_0[0] == hap_char ? correct[0] : wrong[0]);
        Im[matrix_index] = (haplotype_index == stutter_R + 1 ? IMPOSSIBLE : correct[0]);
        Dm[matrix_index] = (haplotype_index == stutter_R + 1 ? IMPOSSIBLE
                            : dmax(Dm[matrix_index - seq_len] + c, Mm[matrix_index - seq_len] + d));
        matrix_index++;
        if (haplotype_index == stutter_R + 1) {                /* a stutter block must be followed by a match */
          int prev = matrix_index - seq_len - 1;
          for (int j = 1; j < seq_len; ++j, ++matrix_index, ++prev) {
            const double emit = (seq_0[j] == hap_char ? correct[j] : wrong[j]);
            Mm[matrix_index] = emit + Mm[prev];
            Im[matrix_index] = IMPOSSIBLE;
            Dm[matrix_index] = IMPOSSIBLE;
          }
          continue;
        }
        for (int j = 1; j < seq_len; ++j, ++matrix_index) {
          const double p0 = Im[matrix_index - 1] + f;
          const double p1 = Mm[matrix_index - seq_len - 1] + e;
          const double p2 = Dm[matrix_index - seq_len - 1] + g;
          const double emit = (seq_0[j] == hap_char ? correct[j] : wrong[j]);
          Mm[matrix_index] = emit + dmax(p0, dmax(p1, p2));
          Im[matrix_index] = correct[j] + dmax(Mm[matrix_index - seq_len - 1] + b, Im[matrix_index - 1] + a);
          Dm[matrix_index] = dmax(Mm[matrix_index - seq_len] + d, Dm[matrix_index - seq_len] + c);
        }
      }
    }
  }
  *left_prob_out = left_prob;
}

/* HapAligner::compute_aln_logprob, HapAligner.cpp:165-233 */
static double compute_aln_logprob(const flat_hap_t* H, int base_seq_len, int seed_base, char seed_char,
                                  double log_seed_wrong, double log_seed_correct,
                                  const double* lM, double l_prob, const double* rM, double r_prob) {
  const int lflank = seed_base, rflank = base_seq_len - seed_base - 1, hapsize = H->cur_size;
  int num_seeds = 0;
  for (int b = 0; b < H->n_blocks; b++) if (!H->is_repeat[b]) num_seeds += H->len[b];
  const double PRIOR = -int_log(num_seeds);
  double* lp = (double*)malloc(sizeof(double) * (size_t)(hapsize + 4));
  int np = 0;
  const char first_char = H->seq[0][0];
  const char last_char = H->seq[H->n_blocks - 1][H->len[H->n_blocks - 1] - 1];
  lp[np++] = PRIOR + (seed_char == first_char ? log_seed_correct : log_seed_wrong) + l_prob + rM[rflank * (hapsize - 1) - 1];
  lp[np++] = PRIOR + (seed_char == last_char ? log_seed_correct : log_seed_wrong) + r_prob + lM[lflank * (hapsize - 1) - 1];
  const double* lptr = lM + (lflank - 1);
  const double* rptr = rM + (rflank * (hapsize - 2) - 1);
  for (int b = 0; b < H->n_blocks; ++b) {
    if (H->is_repeat[b]) { lptr += lflank * H->len[b]; rptr -= rflank * H->len[b]; continue; }
    int coord = (b == 0 ? 1 : 0);
    const int end = (b == H->n_blocks - 1 ? H->len[b] - 1 : H->len[b]);
    for (; coord < end; ++coord) {
      lp[np++] = PRIOR + (seed_char == H->seq[b][coord] ? log_seed_correct : log_seed_wrong) + *lptr + *rptr;
      lptr += lflank; rptr -= rflank;
    }
  }
  const double total = fast_log_sum_exp(lp, np);
  free(lp);
  return total;
}

/* HapAligner::calc_best_seed_position, HapAligner.cpp:467-493 */
static void calc_best_seed_position(const int32_t* rs, const int32_t* re, int nrep, int32_t region_start, int32_t region_end,
                                    int32_t* best_dist, int32_t* best_pos) {
  *best_dist = *best_pos = -1;
  int32_t pos = region_start;
  int ri = 0;
  while (ri < nrep && pos <= region_end) {
    if (pos < rs[ri]) {
      int32_t dist = 1 + ((region_end < rs[ri] - 1 ? region_end : rs[ri] - 1) - pos) / 2;
      if (dist >= *best_dist) { *best_dist = dist; *best_pos = dist - 1 + pos; }
      pos = re[ri++];
    } else if (pos < re[ri])
      pos = re[ri++];
    else
      ri++;
  }
  if (pos <= region_end) {
    int32_t dist = 1 + (region_end - pos) / 2;
    if (dist >= *best_dist) { *best_dist = dist; *best_pos = dist - 1 + pos; }
  }
}
/* HapAligner::calc_seed_base, HapAligner.cpp:494-542.  Returns -2 for a CIGAR op it dies on. */
int ltr_oracle_calc_seed_base(const ltr_alignment* aln, const ltr_haplotype_blocks* hap) {
  int32_t rs[8], re[8]; int nrep = 0;
  for (int b = 0; b < hap->n_blocks && nrep < 8; b++) if (hap->is_repeat[b]) { rs[nrep] = hap->block_start[b]; re[nrep++] = hap->block_end[b]; }
  const int32_t first_start = hap->block_start[0], last_end = hap->block_end[hap->n_blocks - 1];
  int32_t pos = aln->start;
  int best_seed = -1, cur_base = 0, max_dist = MIN_SEED_DIST;
  for (int k = 0; k < aln->n_cigar; k++) {
    const int num = aln->cigar_num[k];
    switch (aln->cigar_type[k]) {
      case '=': {
        int32_t min_region = pos, max_region = pos + num - 1;
        if (min_region < first_start) min_region = first_start;
        if (max_region > last_end - 1) max_region = last_end - 1;
        if (min_region <= max_region) {
          int32_t distance, dist_pos;
          calc_best_seed_position(rs, re, nrep, min_region, max_region, &distance, &dist_pos);
          if (distance >= max_dist) { max_dist = distance; best_seed = cur_base + (dist_pos - pos); }
        }
        pos += num; cur_base += num;
        break;
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

static int64_t allele_slot(const ltr_haplotype_blocks* hap, int block, int allele) {
  int64_t k = 0;
  for (int b = 0; b < block; b++) k += hap->n_alleles[b];
  return k + allele;
}

/* exported for the pin against the compiled reference (oracle/ref_driver.cpp: ltr_ref_compute_aln_logprob,
 * ltr_ref_calc_best_seed_position): the restatements above on caller-supplied blocks / matrices */
double ltr_oracle_compute_aln_logprob(const ltr_haplotype_blocks* hap, const int32_t* counts, int32_t base_seq_len, int32_t seed_base,
                                      int32_t seed_char, double log_seed_wrong, double log_seed_correct,
                                      const double* lM, double l_prob, const double* rM, double r_prob) {
  flat_hap_t fw;
  if (hap->n_blocks > 8) return 1.0;
  fw.n_blocks = hap->n_blocks; fw.cur_size = 0;
  for (int b = 0; b < hap->n_blocks; b++) {
    const int64_t s = allele_slot(hap, b, counts[b]);
    fw.seq[b] = (const char*)hap->allele_bytes + hap->allele_off[s];
    fw.len[b] = (int)(hap->allele_off[s + 1] - hap->allele_off[s]);
    fw.is_repeat[b] = hap->is_repeat[b]; fw.period[b] = hap->period[b]; fw.option[b] = counts[b];
    fw.cur_size += fw.len[b];
  }
  return compute_aln_logprob(&fw, base_seq_len, seed_base, (char)seed_char, log_seed_wrong, log_seed_correct, lM, l_prob, rM, r_prob);
}
void ltr_oracle_calc_best_seed_position(const int32_t* repeat_starts, const int32_t* repeat_ends, int32_t n_repeats,
                                        int32_t region_start, int32_t region_end, int32_t* best_dist, int32_t* best_pos) {
  calc_best_seed_position(repeat_starts, repeat_ends, n_repeats, region_start, region_end, best_dist, best_pos);
}

/* HapAligner::process_reads with short_ == 1 (HapAligner.cpp:545-581) and process_read's
 * short branch (:855-990, retrace_aln == false) */
int ltr_oracle_process_reads_short(const ltr_align_params* P, const ltr_stutter_params* SP,
                                   const ltr_haplotype_blocks* hap, const uint8_t* realign_to_hap,
                                   const ltr_alignment* alns, int32_t n_alns, int32_t init_read_index,
                                   const uint8_t* realign_read, double* aln_probs, int32_t* seed_positions) {
  const int nb = hap->n_blocks;
  if (nb < 2 || nb > 8) return LTR_ERR_INVALID;
  const int64_t H = ltr_oracle_haplotype_num_combs(hap);
  base_quality_t bq; base_quality_init(&bq);
  /* per-combination allele choice, Haplotype::next() order */
  int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * (size_t)(H * nb));
  {
    int64_t factors[8]; int32_t dirs[8], cur[8]; int64_t nc = 1;
    for (int i = 0; i < nb; i++) { factors[i] = nc; nc *= hap->n_alleles[i]; dirs[i] = 1; cur[i] = 0; }
    for (int64_t cidx = 0;; cidx++) {
      for (int i = 0; i < nb; i++) counts[cidx * nb + i] = cur[i];
      if (cidx == H - 1) break;
      int64_t t = cidx + 1; int idx = -1;
      for (int j = nb - 1; j >= 0; j--) { t %= factors[j]; if (t == 0) { idx = j; break; } }
      cur[idx] += dirs[idx];
      if (cur[idx] == 0 || cur[idx] == hap->n_alleles[idx] - 1) dirs[idx] *= -1;
    }
  }
  int max_hap_size = 0;
  for (int b = 0; b < nb; b++) {
    int mx = 0;
    for (int k = 0; k < hap->n_alleles[b]; k++) { const int64_t s = allele_slot(hap, b, k); mx = imax(mx, (int)(hap->allele_off[s + 1] - hap->allele_off[s])); }
    max_hap_size += mx;
  }
  int rc = LTR_OK;
  double* prob_ptr = aln_probs + (int64_t)init_read_index * H;
  for (int32_t r = 0; r < n_alns && rc == LTR_OK; r++, prob_ptr += H) {
    if (realign_read && !realign_read[r]) continue;
    const ltr_alignment* aln = &alns[r];
    const int seed_base = ltr_oracle_calc_seed_base(aln, hap);                 /* :568 */
    if (seed_base == -2) { rc = LTR_ERR_CIGAR; break; }
    seed_positions[init_read_index + r] = seed_base;
    if (seed_base == -1) { for (int64_t k = 0; k < H; k++) prob_ptr[k] = 0; continue; }   /* :570-574 */
    const int len = aln->seq_len;
    if (!aln->qual) { rc = LTR_ERR_INVALID; break; }
    double* wrong = (double*)malloc(sizeof(double) * (size_t)len);
    double* correct = (double*)malloc(sizeof(double) * (size_t)len);
    for (int j = 0; j < len; j++) { wrong[j] = bq_log_error(&bq, (char)aln->qual[j]); correct[j] = bq_log_correct(&bq, (char)aln->qual[j]); }
    const char* base_seq = (const char*)aln->seq;
    const int rlen = len - seed_base - 1;
    double* lM = (double*)malloc(sizeof(double) * (size_t)seed_base * max_hap_size * 3);
    double* rM = (double*)malloc(sizeof(double) * (size_t)rlen * max_hap_size * 3);
    double *lI = lM + (size_t)seed_base * max_hap_size, *lD = lI + (size_t)seed_base * max_hap_size;
    double *rI = rM + (size_t)rlen * max_hap_size, *rD = rI + (size_t)rlen * max_hap_size;
    char* rev_rseq = (char*)malloc((size_t)rlen + 1);
    for (int j = 0; j < rlen; j++) rev_rseq[j] = base_seq[len - 1 - j];       /* :887-888 */
    const char seed_char = base_seq[seed_base];
    const double seed_wrong = wrong[seed_base], seed_correct = correct[seed_base];
    for (int x = seed_base + 1, y = len - 1; x < y; x++, y--) {               /* :889-890 */
      double t = wrong[x]; wrong[x] = wrong[y]; wrong[y] = t;
      t = correct[x]; correct[x] = correct[y]; correct[y] = t;
    }
    char* revbuf = (char*)malloc((size_t)max_hap_size + 8);
    for (int64_t k = 0; k < H; k++) {
      if (realign_to_hap && !realign_to_hap[k]) continue;                      /* :896-900 */
      flat_hap_t fw, rv;
      fw.n_blocks = rv.n_blocks = nb; fw.cur_size = 0;
      int roff = 0;
      for (int b = 0; b < nb; b++) {
        const int64_t s = allele_slot(hap, b, counts[k * nb + b]);
        fw.seq[b] = (const char*)hap->allele_bytes + hap->allele_off[s];
        fw.len[b] = (int)(hap->allele_off[s + 1] - hap->allele_off[s]);
        fw.is_repeat[b] = hap->is_repeat[b]; fw.period[b] = hap->period[b]; fw.option[b] = counts[k * nb + b];
        fw.cur_size += fw.len[b];
      }
      for (int b = 0; b < nb; b++) {                                           /* Haplotype::reverse: blocks reversed, bases reversed */
        const int src = nb - 1 - b;
        rv.len[b] = fw.len[src]; rv.is_repeat[b] = fw.is_repeat[src]; rv.period[b] = fw.period[src]; rv.option[b] = fw.option[src];
        for (int q = 0; q < fw.len[src]; q++) revbuf[roff + q] = fw.seq[src][fw.len[src] - 1 - q];
        rv.seq[b] = revbuf + roff; roff += fw.len[src];
      }
      rv.cur_size = fw.cur_size;
      stutter_model_t sm;
      int rep_period = 1;
      for (int b = 0; b < nb; b++) if (hap->is_repeat[b]) { rep_period = hap->period[b]; break; }
      stutter_model_init(&sm, SP, rep_period);
      double l_prob, r_prob;
      /* forward blocks: StutterAlignerClass(left_align = !reversed = true); reversed: false (RepeatBlock.h:27-33,55-63) */
      align_short(P, &sm, &fw, 1, base_seq, seed_base, wrong, correct, lM, lI, lD, &l_prob);                       /* :905 */
      align_short(P, &sm, &rv, 0, rev_rseq, rlen, wrong + seed_base + 1, correct + seed_base + 1, rM, rI, rD, &r_prob);   /* :908 */
      prob_ptr[k] = compute_aln_logprob(&fw, len, seed_base, seed_char, seed_wrong, seed_correct, lM, l_prob, rM, r_prob); /* :911 */
    }
    free(revbuf); free(rev_rseq); free(lM); free(rM); free(wrong); free(correct);
  }
  free(counts);
  return rc;
}

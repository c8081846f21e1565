/*
 * ltr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See ltr_oracle.h.
 *
 * Plain C restatement of the reference algorithm, one function per reference
 * function, each citing the file:line it follows (paths under the LongTR
 * repository root).  Written from the behaviour of the reference, not copied:
 * the reference is C++ with std::string / std::vector / class state; this is
 * flat C over byte arrays.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  All DP
 * arithmetic is IEEE double add/compare with float-typed constants promoted at
 * the same points as the reference, so results are bit-identical to it.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "ltr_oracle.h"

#define IMPOSSIBLE (-1000000000.0)          /* HapAligner.cpp:20 */
#define REF_FLANK_LEN 35                    /* HapAligner.cpp:245 */

static inline double dmax(double a, double b) { return a < b ? b : a; }   /* std::max */

/* float constants exactly as the reference forms them: double literal -> float */
static inline float f_match(void)    { return (float)(-0.000100005); }     /* HapAligner.cpp:261 */
static inline float f_mismatch(void) { return (float)(-9.0); }             /* HapAligner.cpp:260 */

/*
 * haplotype->get_seq().substr(REF_FLANK_LEN-INDEL_FLANK_LEN, size-2*(REF_FLANK_LEN-INDEL_FLANK_LEN))
 * (HapAligner.cpp:246).  std::string::substr clamps the count, and the count
 * is computed in size_t, so a "negative" count wraps and means "to the end".
 */
static int64_t hap_window(int64_t hap_len, int flank, int64_t* pos_out) {
  int64_t pos = REF_FLANK_LEN - flank;
  int64_t cnt = hap_len - 2 * pos;
  int64_t rest = hap_len - pos;
  if (cnt < 0 || cnt > rest) cnt = rest;
  *pos_out = pos;
  return cnt;
}

/* HapAligner::align_seq_to_hap, HapAligner.cpp:236-343 */
double ltr_oracle_align_long(const uint8_t* hap_full, int64_t hap_len,
                             const uint8_t* read, int64_t read_len,
                             const ltr_align_params* p, double* cells_executed) {
  if (hap_len <= 60) return IMPOSSIBLE;                       /* :241-244 */
  int64_t pos;
  const int64_t n = hap_window(hap_len, p->indel_flank_len, &pos);   /* :246-247 */
  const uint8_t* hap = hap_full + pos;
  const int64_t m = read_len;                                  /* :248 */
  /* n == 0 (only possible with indel_flank_len < 5) and m == 0 make the reference write
   * past zero-sized arrays: undefined there, rejected here (NaN) and by the library. */
  if (n <= 0 || m <= 0) return NAN;
  if (llabs(n - m) > 600) return -700.0;                       /* :249-252 */

  double* D = (double*)malloc(sizeof(double) * (size_t)(n * m));   /* :254-256 */
  double* M = (double*)malloc(sizeof(double) * (size_t)(n * m));
  double* I = (double*)malloc(sizeof(double) * (size_t)(n * m));
  if (!D || !M || !I) { free(D); free(M); free(I); return NAN; }

  const float MISMATCH = f_mismatch(), MATCH = f_match();
  const float a = p->log_ins_to_ins, b = p->log_ins_to_match, c = p->log_del_to_del,
              d = p->log_del_to_match, e = p->log_match_to_match, f = p->log_match_to_ins,
              g = p->log_match_to_del;

  D[0] = IMPOSSIBLE; I[0] = IMPOSSIBLE;                        /* :263-265 */
  M[0] = (hap[0] == read[0] ? MATCH : MISMATCH);

  /* first row, :267-272.  NB the reference indexes the HAPLOTYPE with the read
   * index here; past the end of the string it reads '\0' (index == size) or is
   * undefined (index > size): both are defined as "no match" here. */
  double left_prob = 0.0;
  for (int64_t j = 1; j < m; ++j) {
    const int eq = (j < n) && (hap[j] == read[0]);
    M[j] = D[j - 1] + d + (eq ? MATCH : MISMATCH);
    I[j] = IMPOSSIBLE;
    D[j] = g + left_prob;
    left_prob += c;
  }

  /* first column, :274-280.  read_seq[1] is '\0' when m == 1. */
  left_prob = 0.0;
  const int eq01 = (m > 1) && (hap[0] == read[1]);
  for (int64_t i = 1; i < n; ++i) {
    M[i * m] = I[(i - 1) * m] + b + (eq01 ? MATCH : MISMATCH);
    I[i * m] = MATCH + f + left_prob;      /* float+float first, then + double */
    D[i * m] = IMPOSSIBLE;
    left_prob += a;
  }

  double executed = (double)m;             /* row 0 */
  for (int64_t i = 1; i < n; i++) {        /* :282-307 */
    double max_score_per_row = IMPOSSIBLE;
    const double* Mp = M + (i - 1) * m; const double* Ip = I + (i - 1) * m; const double* Dp = D + (i - 1) * m;
    double* Mc = M + i * m; double* Ic = I + i * m; double* Dc = D + i * m;
    const uint8_t h = hap[i];
    for (int64_t j = 1; j < m; j++) {
      const double match_emit = (h == read[j] ? MATCH : MISMATCH);
      Mc[j] = match_emit + dmax(Mp[j - 1] + e, dmax(Dp[j - 1] + d, Ip[j - 1] + b));
      Ic[j] = MATCH + dmax(Mp[j] + f, Ip[j] + a);
      Dc[j] = dmax(Mc[j - 1] + g, Dc[j - 1] + c);
      const double best_value_here = dmax(Dc[j], dmax(Ic[j], Mc[j]));
      /* int * float -> float, then double + float (:298) */
      const float pen = (float)abs((int)((n - m) - (i - j))) * c;
      if (best_value_here + pen > max_score_per_row) max_score_per_row = best_value_here + pen;
    }
    executed += (double)m;
    if (max_score_per_row < -600) {        /* :300-306 */
      free(M); free(I); free(D);
      if (cells_executed) *cells_executed += executed;
      return -700.0;
    }
  }
  const double res = dmax(D[n * m - 1], dmax(I[n * m - 1], M[n * m - 1]));   /* :309 */
  free(M); free(I); free(D);
  if (cells_executed) *cells_executed += executed;
  return res;
}

/* Same recurrence, O(m) memory.  Self-check only. */
double ltr_oracle_align_long_rolling(const uint8_t* hap_full, int64_t hap_len,
                                     const uint8_t* read, int64_t read_len,
                                     const ltr_align_params* p) {
  if (hap_len <= 60) return IMPOSSIBLE;
  int64_t pos;
  const int64_t n = hap_window(hap_len, p->indel_flank_len, &pos);
  const uint8_t* hap = hap_full + pos;
  const int64_t m = read_len;
  if (n <= 0 || m <= 0) return NAN;
  if (llabs(n - m) > 600) return -700.0;
  const float MISMATCH = f_mismatch(), MATCH = f_match();
  const float a = p->log_ins_to_ins, b = p->log_ins_to_match, c = p->log_del_to_del,
              d = p->log_del_to_match, e = p->log_match_to_match, f = p->log_match_to_ins,
              g = p->log_match_to_del;
  double* buf = (double*)malloc(sizeof(double) * (size_t)m * 6);
  if (!buf) return NAN;
  double *Mp = buf, *Ip = buf + m, *Dp = buf + 2 * m, *Mc = buf + 3 * m, *Ic = buf + 4 * m, *Dc = buf + 5 * m;
  Dp[0] = IMPOSSIBLE; Ip[0] = IMPOSSIBLE; Mp[0] = (hap[0] == read[0] ? MATCH : MISMATCH);
  double lp = 0.0;
  for (int64_t j = 1; j < m; ++j) {
    const int eq = (j < n) && (hap[j] == read[0]);
    Mp[j] = Dp[j - 1] + d + (eq ? MATCH : MISMATCH);
    Ip[j] = IMPOSSIBLE; Dp[j] = g + lp; lp += c;
  }
  double lpa = 0.0;
  const int eq01 = (m > 1) && (hap[0] == read[1]);
  for (int64_t i = 1; i < n; i++) {
    Mc[0] = Ip[0] + b + (eq01 ? MATCH : MISMATCH);
    Ic[0] = MATCH + f + lpa; Dc[0] = IMPOSSIBLE; lpa += a;
    double rowmax = IMPOSSIBLE;
    const uint8_t h = hap[i];
    for (int64_t j = 1; j < m; j++) {
      const double emit = (h == read[j] ? MATCH : MISMATCH);
      Mc[j] = emit + dmax(Mp[j - 1] + e, dmax(Dp[j - 1] + d, Ip[j - 1] + b));
      Ic[j] = MATCH + dmax(Mp[j] + f, Ip[j] + a);
      Dc[j] = dmax(Mc[j - 1] + g, Dc[j - 1] + c);
      const double best = dmax(Dc[j], dmax(Ic[j], Mc[j]));
      const float pen = (float)abs((int)((n - m) - (i - j))) * c;
      if (best + pen > rowmax) rowmax = best + pen;
    }
    if (rowmax < -600) { free(buf); return -700.0; }
    double* t;
    t = Mp; Mp = Mc; Mc = t; t = Ip; Ip = Ic; Ic = t; t = Dp; Dp = Dc; Dc = t;
  }
  const double res = dmax(Dp[m - 1], dmax(Ip[m - 1], Mp[m - 1]));
  free(buf);
  return res;
}

/* HapAligner::trim_alignment, HapAligner.cpp:346-465.  The reference copies the
 * CIGAR vector and eats it from both ends one base at a time; here the same
 * walk runs over (index, remaining-count) cursors. */
int ltr_oracle_trim_alignment(const ltr_alignment* aln, int32_t repeat_start, int32_t repeat_end,
                              int32_t padding, int32_t* ltrim_out, int32_t* rtrim_out) {
  /* element lengths < 1 never reach the reference's get_num() == 1 exit (:376-379): undefined
   * there (endless walk), an error here and in the library */
  for (int32_t k = 0; k < aln->n_cigar; k++) if (aln->cigar_num[k] < 1) return LTR_ERR_CIGAR;
  const int32_t min_read_start = repeat_start - padding;      /* :349 */
  const int32_t max_read_stop  = repeat_end + padding;        /* :350 */
  int32_t start_pos = aln->start + 1;                         /* :351 */
  int32_t end_pos   = aln->stop + 1;                          /* :353 */
  int32_t ltrim = 0, rtrim = 0;
  int32_t fi = 0, bi = aln->n_cigar - 1;                      /* front / back element */
  int32_t* num = (int32_t*)malloc(sizeof(int32_t) * (size_t)(aln->n_cigar > 0 ? aln->n_cigar : 1));
  if (!num) return LTR_ERR_NOMEM;
  for (int32_t k = 0; k < aln->n_cigar; k++) num[k] = aln->cigar_num[k];
  int rc = LTR_OK;
#define NONEMPTY (fi <= bi)
#define POP_FRONT do { if (num[fi] == 1) fi++; else num[fi]--; } while (0)
#define POP_BACK  do { if (num[bi] == 1) bi--; else num[bi]--; } while (0)
  /* left region :360-382 */
  while (start_pos <= min_read_start && NONEMPTY) {
    switch (aln->cigar_type[fi]) {
      case 'M': case '=': case 'X': ltrim++; start_pos++; break;
      case 'D': start_pos++; break;
      case 'I': case 'S': ltrim++; break;
      case 'H': break;
      default: rc = LTR_ERR_CIGAR; goto done;
    }
    POP_FRONT;
  }
  /* left flank :385-408 */
  {
    int32_t mid = start_pos;
    while (mid > min_read_start && mid <= min_read_start + padding && NONEMPTY) {
      switch (aln->cigar_type[fi]) {
        case 'M': case '=': case 'X': mid++; break;
        case 'D': ltrim--; mid++; break;
        case 'I': case 'S': break;
        case 'H': break;
        default: rc = LTR_ERR_CIGAR; goto done;
      }
      POP_FRONT;
    }
  }
  /* right region :411-433 */
  while (end_pos > max_read_stop && NONEMPTY) {
    switch (aln->cigar_type[bi]) {
      case 'M': case '=': case 'X': rtrim++; end_pos--; break;
      case 'D': end_pos--; break;
      case 'I': case 'S': rtrim++; break;
      case 'H': break;
      default: rc = LTR_ERR_CIGAR; goto done;
    }
    POP_BACK;
  }
  /* right flank :436-458 */
  {
    int32_t mid = end_pos;
    while (mid > max_read_stop - padding && mid <= max_read_stop && NONEMPTY) {
      switch (aln->cigar_type[bi]) {
        case 'M': case '=': case 'X': mid--; break;
        case 'D': rtrim--; mid--; break;
        case 'I': case 'S': break;
        case 'H': break;
        default: rc = LTR_ERR_CIGAR; goto done;
      }
      POP_BACK;
    }
  }
  if (ltrim < 0) ltrim = 0;                                   /* :461-462 */
  if (rtrim < 0) rtrim = 0;
  if (ltrim + rtrim > aln->seq_len) rc = LTR_ERR_INVALID;     /* assert :463 */
done:
  free(num);
  *ltrim_out = ltrim; *rtrim_out = rtrim;
  return rc;
#undef NONEMPTY
#undef POP_FRONT
#undef POP_BACK
}

/* Haplotype::init + next, Haplotype.cpp:123-196 (forward iteration, inc_rev_ == false). */
int64_t ltr_oracle_haplotype_num_combs(const ltr_haplotype_blocks* hap) {
  int64_t n = 1;
  for (int32_t i = 0; i < hap->n_blocks; i++) n *= hap->n_alleles[i];
  return n;
}

/* counts[] after `index` calls of next() from reset(). */
static int hap_counts_at(const ltr_haplotype_blocks* hap, int64_t index, int32_t* counts) {
  const int32_t nb = hap->n_blocks;
  int64_t* factors = (int64_t*)malloc(sizeof(int64_t) * (size_t)nb);
  int32_t* dirs = (int32_t*)malloc(sizeof(int32_t) * (size_t)nb);
  if (!factors || !dirs) { free(factors); free(dirs); return LTR_ERR_NOMEM; }
  int64_t ncombs = 1;
  for (int32_t i = 0; i < nb; i++) { factors[i] = ncombs; ncombs *= hap->n_alleles[i]; dirs[i] = 1; counts[i] = 0; }
  int rc = LTR_OK;
  if (index < 0 || index >= ncombs) rc = LTR_ERR_INVALID;
  for (int64_t counter = 0; rc == LTR_OK && counter < index; counter++) {
    int64_t t = counter + 1; int32_t idx = -1;
    for (int32_t j = nb - 1; j >= 0; j--) { t %= factors[j]; if (t == 0) { idx = j; break; } }
    counts[idx] += dirs[idx];
    if (counts[idx] == 0 || counts[idx] == hap->n_alleles[idx] - 1) dirs[idx] *= -1;
  }
  free(factors); free(dirs);
  return rc;
}

static int64_t allele_index(const ltr_haplotype_blocks* hap, int32_t block, int32_t allele) {
  int64_t k = 0;
  for (int32_t b = 0; b < block; b++) k += hap->n_alleles[b];
  return k + allele;
}

int64_t ltr_oracle_haplotype_seq(const ltr_haplotype_blocks* hap, int64_t index, uint8_t* out, int64_t cap) {
  int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * (size_t)hap->n_blocks);
  if (!counts) return LTR_ERR_NOMEM;
  int rc = hap_counts_at(hap, index, counts);
  if (rc != LTR_OK) { free(counts); return rc; }
  int64_t len = 0;
  for (int32_t b = 0; b < hap->n_blocks; b++) {                /* Haplotype::get_seq(), Haplotype.h:99-104 */
    const int64_t k = allele_index(hap, b, counts[b]);
    const int64_t l = hap->allele_off[k + 1] - hap->allele_off[k];
    if (len + l > cap) { free(counts); return LTR_ERR_INVALID; }
    memcpy(out + len, hap->allele_bytes + hap->allele_off[k], (size_t)l);
    len += l;
  }
  free(counts);
  return len;
}

/* HapAligner::process_reads (:545-581) with process_read's long branch (:814-854). */
int ltr_oracle_process_reads(const ltr_align_params* p, const ltr_haplotype_blocks* hap,
                             const uint8_t* realign_to_hap,
                             const ltr_alignment* alns, int32_t n_alns, int32_t init_read_index,
                             const uint8_t* realign_read,
                             double* aln_probs, int32_t* seed_positions) {
  /* short_ = period-1 locus && SWITCH_OLD_ALIGN_LEN (:552): that branch is ltr_oracle_process_reads_short */
  if (p->use_short_path && hap->n_blocks > 1 && hap->period[1] == 1) return LTR_ERR_UNSUPPORTED;
  /* repeat_starts_[0] / repeat_ends_[0]: first block with repeat info (HapAligner.h:104-110) */
  int32_t rb = -1;
  for (int32_t b = 0; b < hap->n_blocks; b++) if (hap->is_repeat[b]) { rb = b; break; }
  if (rb < 0 || hap->n_blocks < 2) return LTR_ERR_INVALID;
  const int64_t H = ltr_oracle_haplotype_num_combs(hap);
  int64_t max_len = 0;
  for (int32_t b = 0; b < hap->n_blocks; b++) {
    int64_t mx = 0;
    for (int32_t k = 0; k < hap->n_alleles[b]; k++) {
      const int64_t ai = allele_index(hap, b, k);
      const int64_t l = hap->allele_off[ai + 1] - hap->allele_off[ai];
      if (l > mx) mx = l;
    }
    max_len += mx;
  }
  uint8_t* hseq = (uint8_t*)malloc((size_t)(max_len + 1) * (size_t)H);
  int64_t* hlen = (int64_t*)malloc(sizeof(int64_t) * (size_t)H);
  if (!hseq || !hlen) { free(hseq); free(hlen); return LTR_ERR_NOMEM; }
  for (int64_t k = 0; k < H; k++) hlen[k] = ltr_oracle_haplotype_seq(hap, k, hseq + k * (max_len + 1), max_len);

  int rc = LTR_OK;
  double* prob_ptr = aln_probs + (int64_t)init_read_index * H;   /* :550 */
  for (int32_t i = 0; i < n_alns && rc == LTR_OK; i++, prob_ptr += H) {
    if (realign_read && !realign_read[i]) continue;               /* :557-560 */
    seed_positions[init_read_index + i] = alns[i].seq_len - 1;    /* :562-563 */
    int32_t ltrim, rtrim;
    rc = ltr_oracle_trim_alignment(&alns[i], hap->block_start[rb], hap->block_end[rb],
                                   p->indel_flank_len, &ltrim, &rtrim);       /* :819 */
    if (rc != LTR_OK) break;
    const uint8_t* seq = alns[i].seq + ltrim;
    int64_t len = alns[i].seq_len - ltrim - rtrim;
    uint8_t subst[10];
    if (len == 0) {                                               /* :820-823 */
      const int64_t a0 = allele_index(hap, 0, 0), aL = allele_index(hap, hap->n_blocks - 1, 0);
      const int64_t l0 = hap->allele_off[a0 + 1] - hap->allele_off[a0];
      const int64_t lL = hap->allele_off[aL + 1] - hap->allele_off[aL];
      if (l0 < 5) { rc = LTR_ERR_INVALID; break; }                /* substr would throw */
      memcpy(subst, hap->allele_bytes + hap->allele_off[a0] + l0 - 5, 5);
      const int64_t take = lL < 5 ? lL : 5;
      memcpy(subst + 5, hap->allele_bytes + hap->allele_off[aL], (size_t)take);
      seq = subst; len = 5 + take;
    }
    for (int64_t k = 0; k < H; k++) {                             /* :840-852 */
      if (realign_to_hap && !realign_to_hap[k]) continue;
      prob_ptr[k] = ltr_oracle_align_long(hseq + k * (max_len + 1), hlen[k], seq, len, p, NULL);
    }
  }
  free(hseq); free(hlen);
  return rc;
}

int ltr_oracle_align_batch(const ltr_align_params* p, const ltr_locus_batch* batch,
                           double* out_ll, int32_t* out_seed, double* cells_executed) {
  int64_t ll_off = 0;
  for (int64_t l = 0; l < batch->n_loci; l++) {
    const int64_t r0 = batch->locus_read_off[l], r1 = batch->locus_read_off[l + 1];
    const int64_t h0 = batch->locus_hap_off[l], h1 = batch->locus_hap_off[l + 1];
    const int64_t H = h1 - h0;
    for (int64_t r = r0; r < r1; r++) {
      if (batch->realign_read && !batch->realign_read[r]) continue;
      const int64_t m = batch->read_off[r + 1] - batch->read_off[r];
      if (m <= 0) return LTR_ERR_INVALID;
      if (out_seed) out_seed[r] = (int32_t)m - 1;
      for (int64_t h = h0; h < h1; h++) {
        if (batch->realign_hap && !batch->realign_hap[h]) continue;
        out_ll[ll_off + (r - r0) * H + (h - h0)] =
            ltr_oracle_align_long(batch->hap_bytes + batch->hap_off[h], batch->hap_off[h + 1] - batch->hap_off[h],
                                  batch->read_bytes + batch->read_off[r], m, p, cells_executed);
      }
    }
    ll_off += (r1 - r0) * H;
  }
  return LTR_OK;
}

/* ReadPooler::add_alignment, read_pooler.cpp:3-20: pool id = order of first
 * occurrence of the exact sequence.  O(R^2) compare is fine for R ~ 10^2. */
int32_t ltr_oracle_pool_reads(const uint8_t* const* seqs, const int32_t* seq_lens, int32_t n_reads,
                              int32_t* pool_index) {
  int32_t n_pools = 0;
  int32_t* first = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_reads > 0 ? n_reads : 1));
  if (!first) return LTR_ERR_NOMEM;
  for (int32_t i = 0; i < n_reads; i++) {
    int32_t found = -1;
    for (int32_t q = 0; q < n_pools; q++) {
      const int32_t j = first[q];
      if (seq_lens[j] == seq_lens[i] && memcmp(seqs[j], seqs[i], (size_t)seq_lens[i]) == 0) { found = q; break; }
    }
    if (found < 0) { first[n_pools] = i; found = n_pools++; }
    pool_index[i] = found;
  }
  free(first);
  return n_pools;
}

/* SeqStutterGenotyper::calc_hap_aln_probs, seq_stutter_genotyper.cpp:526-559 */
int ltr_oracle_scatter_pool_probs(const double* log_pool_aln_probs, const int32_t* pool_seed_positions,
                                  const int32_t* pool_index, int32_t n_reads, int32_t n_alleles,
                                  const uint8_t* realign_to_hap, const uint8_t* copy_read,
                                  const uint8_t* second_mate,
                                  double* log_aln_probs, int32_t* seed_positions) {
  for (int32_t i = 0; i < n_reads; i++) {                       /* :527-538 */
    if (copy_read && !copy_read[i]) continue;
    if (seed_positions && pool_seed_positions) seed_positions[i] = pool_seed_positions[pool_index[i]];
    const double* src = log_pool_aln_probs + (int64_t)n_alleles * pool_index[i];
    double* dst = log_aln_probs + (int64_t)n_alleles * i;
    for (int32_t j = 0; j < n_alleles; j++)
      if (!realign_to_hap || realign_to_hap[j]) dst[j] = src[j];
  }
  for (int32_t i = 0; i < n_reads; i++) {                       /* :546-559 */
    if (!second_mate || !second_mate[i] || (copy_read && !copy_read[i])) continue;
    if (i == 0) return LTR_ERR_INVALID;
    double* m1 = log_aln_probs + (int64_t)(i - 1) * n_alleles;
    double* m2 = log_aln_probs + (int64_t)i * n_alleles;
    for (int32_t j = 0; j < n_alleles; j++)
      if (!realign_to_hap || realign_to_hap[j]) { const double t = m1[j] + m2[j]; m1[j] = t; m2[j] = t; }
  }
  return LTR_OK;
}

/* int_log(v) = log(v) from a table of log(i) (mathops.cpp:14-22); same value. */
static double int_log(int v) { return v == 0 ? -1000.0 : log((double)v); }

/* Genotyper::calc_log_sample_posteriors + get_optimal_haplotypes, genotyper.cpp:21-100 */
int ltr_oracle_posteriors(int32_t S, int32_t R, int32_t H,
                          double* ll, const double* log_p1, const double* log_p2,
                          const int32_t* sample_label, int32_t haploid,
                          double* post, double* sample_total_ll, int32_t* gts, double* total_ll) {
  const double LOG_ONE_HALF = log(0.5);                          /* mathops.cpp:10 */
  const double homoz = haploid ? -int_log(H) : int_log(2) - int_log(H) - int_log(H + 1);   /* :21-26 */
  const double hetz  = haploid ? -1.7976931348623157e308 / 2 : -int_log(H) - int_log(H + 1); /* :28-33 */
  const int64_t nd = (int64_t)H * H;
  for (int32_t s = 0; s < S; s++)                                /* :35-43 */
    for (int32_t j = 0; j < H; j++)
      for (int32_t k = 0; k < H; k++) post[s * nd + (int64_t)j * H + k] = (j == k ? homoz : hetz);
  for (int32_t r = 0; r < R; r++) {                              /* :52-63 */
    double* row = ll + (int64_t)r * H;
    double* sp = post + nd * sample_label[r];
    for (int32_t i1 = 0; i1 < H; i1++)
      for (int32_t i2 = 0; i2 < H; i2++, sp++) {
        if (row[i1] < -600) row[i1] = -600;
        if (row[i2] < -600) row[i2] = -600;
        *sp += log(exp(row[i1] + log_p1[r] + LOG_ONE_HALF) + exp(row[i2] + log_p2[r] + LOG_ONE_HALF));
      }
  }
  double tot = 0.0;
  for (int32_t s = 0; s < S; s++) {                              /* :67-75, log_sum_exp mathops.cpp:47-53 */
    double* sp = post + nd * s;
    double mx = sp[0];
    for (int64_t k = 1; k < nd; k++) if (mx < sp[k]) mx = sp[k];     /* std::max_element: first max */
    double t = 0.0;
    for (int64_t k = 0; k < nd; k++) t += exp(sp[k] - mx);
    const double stl = mx + log(t);
    sample_total_ll[s] = stl;
    for (int64_t k = 0; k < nd; k++) sp[k] -= stl;
    tot += stl;                                                  /* sum(), :78 */
  }
  if (total_ll) *total_ll = tot;
  if (gts)                                                       /* :85-100: strict >, row-major, first wins */
    for (int32_t s = 0; s < S; s++) {
      double best = -1.7976931348623157e308; int32_t b1 = -1, b2 = -1;
      for (int32_t i1 = 0; i1 < H; i1++)
        for (int32_t i2 = 0; i2 < H; i2++) {
          const double v = post[s * nd + (int64_t)i1 * H + i2];
          if (v > best) { best = v; b1 = i1; b2 = i2; }
        }
      gts[2 * s] = b1; gts[2 * s + 1] = b2;
    }
  return LTR_OK;
}

/*
 * ltr_oracle_vcf.c -- TEST INFRASTRUCTURE (see ltr_oracle.h): plain-C restatement of the genotyper's
 * last steps, each function citing the reference lines it follows:
 *   haps_to_alleles / get_unused_alleles   src/seq_stutter_genotyper.cpp:240-308
 *   add_and_remove_alleles bookkeeping     :317-375
 *   reorder_alleles / get_alleles          :667-785
 *   write_vcf_record (long-read path)      :894-1366
 *   ExtractCigar                           src/extract_indels.cpp:18-91
 *   condense_read_counts                   src/genotyper.h:50-63
 * PARITY UNPINNED: seq_stutter_genotyper.cpp includes htslib headers and cannot be compiled in the
 * dev container; this file pins the product code (longtr_amd/csrc/ltr_vcf.cpp) against an independent
 * re-reading of the same lines plus hand-checked cases (tests/test_vcf_record.py), nothing more.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ltr_oracle.h"

#define LOG_ONE_HALF (-0.6931471805599453094)   /* mathops.cpp:10 */
#define TOLERANCE 1e-10                          /* mathops.cpp:12 */

typedef struct { char* s; size_t n, cap; } sbuf;
static void sb_put(sbuf* b, const char* t, size_t len) {
  if (b->n + len + 1 > b->cap) { b->cap = (b->n + len + 1) * 2; b->s = (char*)realloc(b->s, b->cap); }
  memcpy(b->s + b->n, t, len); b->n += len; b->s[b->n] = 0;
}
static void sb_str(sbuf* b, const char* t) { sb_put(b, t, strlen(t)); }
static void sb_int(sbuf* b, long v) { char t[32]; snprintf(t, sizeof t, "%ld", v); sb_str(b, t); }
static void sb_f2(sbuf* b, double v) { char t[64]; snprintf(t, sizeof t, "%.2f", v); sb_str(b, t); }   /* precision(2), fixed (:898-899) */

static int64_t allele_slot(const ltr_haplotype_blocks* hap, int block, int allele) {
  int64_t k = 0;
  for (int b = 0; b < block; b++) k += hap->n_alleles[b];
  return k + allele;
}

/* haps_to_alleles, :240-248: cur_index(block) along Haplotype::next() -- recovered from the haplotype
 * strings' block allele by replaying the Gray walk through ltr_oracle_haplotype_seq is indirect; the
 * walk itself is restated here (Haplotype.cpp:151-196). */
int ltr_oracle_haps_to_alleles(const ltr_haplotype_blocks* hap, int32_t block, int32_t* out) {
  const int nb = hap->n_blocks;
  int64_t* factors = (int64_t*)malloc(sizeof(int64_t) * (size_t)nb);
  int* dirs = (int*)malloc(sizeof(int) * (size_t)nb);
  int* counts = (int*)calloc((size_t)nb, sizeof(int));
  int64_t ncombs = 1;
  for (int i = 0; i < nb; i++) { factors[i] = ncombs; ncombs *= hap->n_alleles[i]; dirs[i] = 1; }
  for (int64_t c = 0; c < ncombs; c++) {
    out[c] = counts[block];
    if (c == ncombs - 1) break;
    int64_t t = c + 1; int idx = -1;
    for (int j = nb - 1; j >= 0; j--) { t %= factors[j]; if (t == 0) { idx = j; break; } }
    counts[idx] += dirs[idx];
    if (counts[idx] == 0 || counts[idx] == hap->n_alleles[idx] - 1) dirs[idx] *= -1;
  }
  free(factors); free(dirs); free(counts);
  return LTR_OK;
}

/* get_unused_alleles(check_spanned = false, check_called = true), :250-308, one block */
int32_t ltr_oracle_unused_alleles(int32_t num_samples, const int32_t* haps, const uint8_t* aligned_read, const uint8_t* filtered,
                                  const int32_t* hap_to_allele, int32_t num_options, int32_t* out) {
  if (num_options == 1) return 0;                                       /* :272-274 */
  uint8_t* called = (uint8_t*)calloc((size_t)num_options, 1);
  for (int32_t s = 0; s < num_samples; s++)                             /* :286-293 */
    if ((!aligned_read || aligned_read[s]) && !(filtered && filtered[s])) {
      called[hap_to_allele[haps[2 * s]]] = 1;
      called[hap_to_allele[haps[2 * s + 1]]] = 1;
    }
  int32_t n = 0;
  for (int32_t a = 1; a < num_options; a++) if (!called[a]) out[n++] = a;   /* :296-303 */
  free(called);
  return n;
}

/* :322-362: old and new haplotype sequences matched by string */
int ltr_oracle_remap_haplotypes(const ltr_haplotype_blocks* old_hap, const ltr_haplotype_blocks* new_hap, int32_t* mapping, uint8_t* realign) {
  const int64_t Ho = ltr_oracle_haplotype_num_combs(old_hap), Hn = ltr_oracle_haplotype_num_combs(new_hap);
  const int64_t cap = 1 << 20;
  uint8_t* a = (uint8_t*)malloc((size_t)cap); uint8_t* b = (uint8_t*)malloc((size_t)cap);
  for (int64_t k = 0; k < Ho; k++) mapping[k] = -1;
  for (int64_t k = 0; k < Hn; k++) {
    const int64_t lb = ltr_oracle_haplotype_seq(new_hap, k, b, cap);
    int64_t match = -1;
    for (int64_t j = 0; j < Ho; j++) {                                  /* hap_indices[seq] = last index with that sequence (:328) */
      const int64_t la = ltr_oracle_haplotype_seq(old_hap, j, a, cap);
      if (la == lb && memcmp(a, b, (size_t)la) == 0) match = j;
    }
    if (match < 0) realign[k] = 1; else { realign[k] = 0; mapping[match] = (int32_t)k; }
  }
  free(a); free(b);
  return LTR_OK;
}

/* ExtractCigar, extract_indels.cpp:18-91 */
static int extract_cigar(const ltr_alignment* aln, int region_start, int region_end, int* bp_diff) {
  const int n = aln->n_cigar;
  int pos = aln->start, bp; size_t start_index = 0, last_match = 0; char t;
  int region_length = 0;
  for (int i = 0; i < n; i++) { t = aln->cigar_type[i]; if (t == 'M' || t == '=' || t == 'X' || t == 'D') region_length += aln->cigar_num[i]; }   /* :32-38 */
  if (region_start < aln->start) return 0;                              /* :41 */
  if (region_end >= aln->start + region_length) return 0;               /* :42 */
  if (n == 0) return 0;
  while (pos < region_start && start_index < (size_t)n) {               /* :44-53 */
    bp = aln->cigar_num[start_index]; t = aln->cigar_type[start_index];
    if (t == 'M' || t == '=' || t == 'X' || t == 'D') pos += bp;
    if (t == 'M' || t == '=' || t == 'X') last_match = start_index;
    start_index++;
  }
  start_index = last_match;
  if (start_index == 0) { t = aln->cigar_type[0]; if (!(t == 'M' || t == '=' || t == 'X')) return 0; }      /* :55-60 */
  size_t end_index = (size_t)n - 1;
  last_match = (size_t)n - 1;
  pos = aln->start + region_length;
  while (pos > region_end) {                                            /* :66-76 */
    bp = aln->cigar_num[end_index]; t = aln->cigar_type[end_index];
    if (t == 'M' || t == '=' || t == 'X' || t == 'D') pos -= bp;
    if (t == 'M' || t == '=' || t == 'X') last_match = end_index;
    if (end_index == 0) break;
    end_index -= 1;
  }
  end_index = last_match;
  if (end_index == (size_t)n - 1) { t = aln->cigar_type[end_index]; if (!(t == 'M' || t == '=' || t == 'X')) return 0; }   /* :78-83 */
  *bp_diff = 0;
  for (size_t i = start_index; i <= end_index; i++) {                   /* :85-90 */
    if (aln->cigar_type[i] == 'D') *bp_diff -= aln->cigar_num[i];
    else if (aln->cigar_type[i] == 'I') *bp_diff += aln->cigar_num[i];
  }
  return 1;
}

static int cmp_int(const void* a, const void* b) { const int x = *(const int*)a, y = *(const int*)b; return (x > y) - (x < y); }
/* condense_read_counts, genotyper.h:50-63 */
static void condense(sbuf* o, int* v, int n) {
  if (n == 0) { sb_str(o, "."); return; }
  qsort(v, (size_t)n, sizeof(int), cmp_int);
  for (int i = 0; i < n;) {
    int j = i;
    while (j < n && v[j] == v[i]) j++;
    if (i) sb_str(o, ";");
    sb_int(o, v[i]); sb_str(o, "|"); sb_int(o, j - i);
    i = j;
  }
}

typedef struct { char** s; int n; } strs;
static char* dup_n(const char* p, size_t n) { char* r = (char*)malloc(n + 1); memcpy(r, p, n); r[n] = 0; return r; }
static int is_del(const char* s) { return strcmp(s, "<DEL>") == 0; }
static char up(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }
static char* chrom_sub(const ltr_vcf_locus* v, int64_t pos, int64_t len) {    /* uppercase(chrom_seq.substr(pos, len)) */
  char* r = (char*)malloc((size_t)(len > 0 ? len : 0) + 1); int64_t n = 0;
  for (int64_t p = pos; p < pos + len; p++) { const int64_t k = p - v->chrom_seq_start; if (k < 0 || k >= v->chrom_seq_len) break; r[n++] = up((char)v->chrom_seq[k]); }
  r[n] = 0; return r;
}

/* get_alleles, :688-785 */
static strs get_alleles(const ltr_vcf_locus* v, int32_t* pos_out, uint8_t** inexact_out) {
  const ltr_haplotype_blocks* hap = v->hap;
  const int nopt = hap->n_alleles[v->block];
  strs A; A.n = nopt; A.s = (char**)malloc(sizeof(char*) * (size_t)nopt);
  uint8_t* inex = (uint8_t*)calloc((size_t)nopt, 1);
  int deleted = -1;
  for (int i = 0; i < nopt; i++) {                                      /* :695-706 */
    const int64_t k = allele_slot(hap, v->block, i);
    const int64_t len = hap->allele_off[k + 1] - hap->allele_off[k];
    if (len == 0) { A.s[i] = dup_n("<DEL>", 5); deleted = i; inex[i] = 0; continue; }
    A.s[i] = dup_n((const char*)hap->allele_bytes + hap->allele_off[k], (size_t)len);
    inex[i] = v->inexact_allele ? (v->inexact_allele[i] != 0) : 0;
  }
  if (deleted != -1) { char* t = A.s[1]; A.s[1] = dup_n("<DEL>", 5); free(A.s[deleted]); A.s[deleted] = t; }   /* :708-712 (the swap leaves "<DEL>" second) */
  int32_t left_trim = 0, start = hap->block_start[v->block];
  while (start + left_trim < v->region_start) {                         /* :717-729 */
    int trim = 1;
    for (int i = 0; i < A.n; i++) {
      if (is_del(A.s[i])) continue;
      if ((size_t)(left_trim + 1) >= strlen(A.s[i]) || A.s[i][left_trim] != A.s[0][left_trim]) { trim = 0; break; }
    }
    if (!trim) break;
    left_trim++;
  }
  start += left_trim;
  for (int i = 0; i < A.n; i++) if (!is_del(A.s[i])) { char* t = dup_n(A.s[i] + left_trim, strlen(A.s[i]) - (size_t)left_trim); free(A.s[i]); A.s[i] = t; }
  int32_t right_trim = 0, end = hap->block_end[v->block];
  while (end - right_trim > v->region_stop) {                           /* :738-753 */
    int trim = 1;
    const int ref_size = (int)strlen(A.s[0]);
    for (int i = 0; i < A.n; i++) {
      if (is_del(A.s[i])) continue;
      const int alt_size = (int)strlen(A.s[i]);
      if ((size_t)(right_trim + 1) >= strlen(A.s[i]) || A.s[i][alt_size - right_trim - 1] != A.s[0][ref_size - right_trim - 1]) { trim = 0; break; }
    }
    if (!trim) break;
    right_trim++;
  }
  end -= right_trim;
  for (int i = 0; i < A.n; i++) if (!is_del(A.s[i])) A.s[i][strlen(A.s[i]) - (size_t)right_trim] = 0;
  char* lf = (start >= v->region_start) ? chrom_sub(v, v->region_start, start - v->region_start) : dup_n("", 0);     /* :759-760 */
  char* rf = (end <= v->region_stop) ? chrom_sub(v, end, v->region_stop - end) : dup_n("", 0);
  int32_t pos = v->region_start < start ? v->region_start : start;      /* :761 */
  if (lf[0] == 0) {                                                     /* :764-777 */
    int pad = 0;
    for (int i = 1; i < A.n; i++) { if (is_del(A.s[i])) continue; if (A.s[i][0] == 0 || A.s[i][0] != A.s[0][0]) { pad = 1; break; } }
    if (pad) { pos -= 1; free(lf); lf = chrom_sub(v, pos, 1); }
  }
  for (int i = 0; i < A.n; i++) {                                       /* :779-783 */
    if (is_del(A.s[i])) continue;
    const size_t n = strlen(lf) + strlen(A.s[i]) + strlen(rf);
    char* t = (char*)malloc(n + 1);
    strcpy(t, lf); strcat(t, A.s[i]); strcat(t, rf);
    free(A.s[i]); A.s[i] = t;
  }
  free(lf); free(rf);
  *pos_out = pos + 1;                                                   /* :784 */
  *inexact_out = inex;
  return A;
}

int32_t ltr_oracle_get_alleles(const ltr_vcf_locus* v, int32_t* pos, char* out, int64_t cap, int64_t* off) {
  uint8_t* inex; strs A = get_alleles(v, pos, &inex);
  int64_t at = 0; off[0] = 0;
  for (int i = 0; i < A.n; i++) { const int64_t l = (int64_t)strlen(A.s[i]); if (at + l > cap) return LTR_ERR_INVALID; memcpy(out + at, A.s[i], (size_t)l); at += l; off[i + 1] = at; free(A.s[i]); }
  const int n = A.n; free(A.s); free(inex);
  return n;
}

static const strs* g_sort_alleles;
static int by_len_seq(const void* a, const void* b) {                   /* orderByLengthAndSequence, stringops.cpp:35-39 */
  const char* x = g_sort_alleles->s[*(const int*)a]; const char* y = g_sort_alleles->s[*(const int*)b];
  const size_t lx = strlen(x), ly = strlen(y);
  if (lx != ly) return lx < ly ? -1 : 1;
  return strcmp(x, y);
}

/* write_vcf_record, :894-1366, SWITCH_OLD_ALIGN_LEN == 0 */
int64_t ltr_oracle_vcf_record(const ltr_vcf_locus* v, const ltr_vcf_options* opt_in, char* out, int64_t cap, int32_t* pos_out) {
  ltr_vcf_options opt;
  if (opt_in) opt = *opt_in;
  else { opt.output_gls = opt.output_pls = opt.output_phased_gls = 0; opt.output_allreads = opt.output_mallreads = 1;      /* genotyper.cpp:339-346 */
         opt.output_filters = opt.output_haplotype_data = 0; opt.max_flank_indel_frac = 0.15f; }
  const int S = v->n_samples, R = v->n_reads, haploid = v->haploid != 0;
  const int64_t H = ltr_oracle_haplotype_num_combs(v->hap);
  int32_t pos; uint8_t* inexact;
  strs A = get_alleles(v, &pos, &inexact);                              /* :904 */
  const int V = A.n;
  int* bp = (int*)malloc(sizeof(int) * (size_t)V);                      /* :906-913 */
  for (int i = 0; i < V; i++) bp[i] = is_del(A.s[i]) ? -(int)strlen(A.s[0]) : (int)strlen(A.s[i]) - (int)strlen(A.s[0]);

  int32_t* h2a = (int32_t*)malloc(sizeof(int32_t) * (size_t)H);
  ltr_oracle_haps_to_alleles(v->hap, v->block, h2a);                    /* :922 */
  const int n_gl = haploid ? V : V * (V + 1) / 2, n_pgl = haploid ? V : V * V;
  int32_t* gts = (int32_t*)calloc((size_t)2 * S, sizeof(int32_t)); int32_t* pls = (int32_t*)calloc((size_t)S * n_gl, sizeof(int32_t));
  double *lph = (double*)calloc((size_t)S, 8), *lun = (double*)calloc((size_t)S, 8), *hph = (double*)calloc((size_t)S, 8), *hun = (double*)calloc((size_t)S, 8),
         *gld = (double*)calloc((size_t)S, 8), *gls = (double*)calloc((size_t)S * n_gl, 8), *pgl = (double*)calloc((size_t)S * n_pgl, 8);
  ltr_genotype_fields gf;
  gf.best_gts = gts; gf.log_phased_posteriors = lph; gf.log_unphased_posteriors = lun; gf.hap_log_phased_posteriors = hph;
  gf.hap_log_unphased_posteriors = hun; gf.gls = gls; gf.gl_diffs = gld; gf.pls = opt.output_pls ? pls : NULL; gf.phased_gls = opt.output_phased_gls ? pgl : NULL;
  ltr_oracle_extract_genotypes(S, (int32_t)H, V, h2a, haploid, v->log_sample_posteriors, v->sample_total_ll, v->best_haplotypes, &gf);   /* :924-927 */

  int *n_al = (int*)calloc((size_t)S, sizeof(int)), *n_snp = (int*)calloc((size_t)S, sizeof(int)), *n_fl = (int*)calloc((size_t)S, sizeof(int)),
      *s1 = (int*)calloc((size_t)S, sizeof(int)), *s2 = (int*)calloc((size_t)S, sizeof(int));
  int** bps = (int**)calloc((size_t)S, sizeof(int*)); int** mls = (int**)calloc((size_t)S, sizeof(int*));
  int *nb = (int*)calloc((size_t)S, sizeof(int)), *nm = (int*)calloc((size_t)S, sizeof(int));
  for (int s = 0; s < S; s++) { bps[s] = (int*)malloc(sizeof(int) * (size_t)(R + 1)); mls[s] = (int*)malloc(sizeof(int) * (size_t)(R + 1)); }
  for (int r = 0; r < R; r++) {                                         /* :946-1043 */
    const int s = v->sample_label[r];
    const double* ll = v->log_aln_probs + (size_t)r * (size_t)H;
    const int hap_a = v->best_haplotypes[2 * s], hap_b = v->best_haplotypes[2 * s + 1];
    int read_strand = 0;
    if (!haploid && hap_a != hap_b) {                                   /* :964-967 */
      const double v1 = v->log_p1[r] + ll[hap_a], v2 = v->log_p2[r] + ll[hap_b];
      read_strand = (v1 > v2 ? 0 : 1);
    }
    const int best_hap = (read_strand == 0 ? hap_a : hap_b);            /* :982 */
    n_al[s]++;                                                          /* :999 */
    if (fabs(v->log_p1[r] - v->log_p2[r]) > TOLERANCE) {                /* :1006-1012 */
      n_snp[s]++;
      if (v->log_p1[r] > v->log_p2[r]) s1[s]++; else s2[s]++;
    }
    if (v->alns) {                                                      /* :1016-1022 */
      if (v->aln_deleted && v->aln_deleted[r]) bps[s][nb[s]++] = -(int)strlen(A.s[0]);
      else { int d; if (extract_cigar(&v->alns[r], v->region_start - 5, v->region_stop + 5, &d)) bps[s][nb[s]++] = d; }
    }
    mls[s][nm[s]++] = bp[h2a[best_hap]];                                /* :1038-1040 */
  }

  const int n_out = v->n_out_samples > 0 ? v->n_out_samples : S;
  const char* const* out_names = v->n_out_samples > 0 ? v->out_sample_names : v->sample_names;
  int* counts = (int*)calloc((size_t)V, sizeof(int));
  int skip = 0, filt = 0, an = 0;
  for (int s = 0; s < S; s++) {                                         /* :1046-1071 */
    int wanted = 0;
    for (int i = 0; i < n_out; i++) if (strcmp(out_names[i], v->sample_names[s]) == 0) wanted = 1;
    if (!wanted) continue;
    if (n_al[s] == 0) continue;
    if (n_al[s] > 0 && n_fl[s] > opt.max_flank_indel_frac * n_al[s]) { filt++; continue; }
    if (!(v->sample_filter && v->sample_filter[s] && v->sample_filter[s][0])) {
      if (haploid) { counts[gts[2 * s]]++; an++; }
      else { counts[gts[2 * s]]++; counts[gts[2 * s + 1]]++; an += 2; }
    } else skip++;
  }
  /* reorder_alleles, :667-686 */
  int* n2o = (int*)malloc(sizeof(int) * (size_t)V); int* o2n = (int*)malloc(sizeof(int) * (size_t)V);
  for (int i = 0; i < V; i++) n2o[i] = i;
  g_sort_alleles = &A;
  if (V > 1 && is_del(A.s[1])) { if (V > 2) qsort(n2o + 2, (size_t)(V - 2), sizeof(int), by_len_seq); }
  else if (V > 1) qsort(n2o + 1, (size_t)(V - 1), sizeof(int), by_len_seq);
  for (int i = 0; i < V; i++) o2n[n2o[i]] = i;

  sbuf o = {NULL, 0, 0};
  sb_str(&o, v->chrom); sb_str(&o, "\t"); sb_int(&o, pos); sb_str(&o, "\t"); sb_str(&o, (v->name && v->name[0]) ? v->name : ".");     /* :1093 */
  sb_str(&o, "\t"); sb_str(&o, A.s[n2o[0]]); sb_str(&o, "\t");
  if (V == 1) sb_str(&o, ".");
  else { for (int i = 1; i < V - 1; i++) { sb_str(&o, A.s[n2o[i]]); sb_str(&o, ","); } sb_str(&o, A.s[n2o[V - 1]]); }
  sb_str(&o, "\t.\t.");                                                 /* :1106 */
  sb_str(&o, "\tSTART="); sb_int(&o, v->region_start + 1); sb_str(&o, ";END="); sb_int(&o, v->region_stop); sb_str(&o, ";MOTIF="); sb_str(&o, v->motif ? v->motif : "");
  sb_str(&o, ";PERIOD="); sb_str(&o, v->period_str ? v->period_str : ""); sb_str(&o, ";NSKIP="); sb_int(&o, skip); sb_str(&o, ";NFILT="); sb_int(&o, filt);
  sb_str(&o, ";INEXACT_ALLELE=");                                       /* :1084-1090 */
  if (V == 1) sb_str(&o, ".");
  else for (int i = 1; i < V; i++) { if (i > 1) sb_str(&o, ","); sb_str(&o, inexact[n2o[i]] ? "1" : "0"); }
  sb_str(&o, ";");
  if (V > 1) { sb_str(&o, "BPDIFFS="); for (int i = 1; i < V; i++) { if (i > 1) sb_str(&o, ","); sb_int(&o, bp[n2o[i]]); } sb_str(&o, ";"); }
  int dp = 0, dsnp = 0, dfl = 0;                                        /* :1135-1153 */
  for (int i = 0; i < n_out; i++) {
    int s = -1;
    for (int k = 0; k < S; k++) if (strcmp(out_names[i], v->sample_names[k]) == 0) s = k;
    if (s < 0) continue;
    if (v->sample_filter && v->sample_filter[s] && v->sample_filter[s][0]) continue;
    if (n_al[s] > 0 && n_fl[s] > n_al[s] * opt.max_flank_indel_frac) continue;
    dp += n_al[s]; dsnp += n_snp[s]; dfl += n_fl[s];
  }
  sb_str(&o, "DP="); sb_int(&o, dp); sb_str(&o, ";DSNP="); sb_int(&o, dsnp); sb_str(&o, ";DFLANKINDEL="); sb_int(&o, dfl); sb_str(&o, ";");
  sb_str(&o, "AN="); sb_int(&o, an); sb_str(&o, ";REFAC="); sb_int(&o, counts[0]);
  if (V > 1) { sb_str(&o, ";AC="); for (int i = 1; i < V; i++) { if (i > 1) sb_str(&o, ","); sb_int(&o, counts[n2o[i]]); } }

  int num_fields;                                                       /* :1170-1197 */
  if (!haploid) { sb_str(&o, "\tGT:GB:Q:PQ:DP:DSNP:DFLANKINDEL:PDP:PSNP:GLDIFF"); num_fields = 10; }
  else { sb_str(&o, "\tGT:GB:Q:DP:DFLANKINDEL:GLDIFF"); num_fields = 6; }
  if (opt.output_allreads) sb_str(&o, ":ALLREADS");
  if (opt.output_mallreads) sb_str(&o, ":MALLREADS");
  if (opt.output_gls) sb_str(&o, ":GL");
  if (opt.output_pls) sb_str(&o, ":PL");
  if (!haploid && opt.output_phased_gls) sb_str(&o, ":PHASEDGL");
  if (opt.output_haplotype_data) sb_str(&o, ":HQ:PHQ");
  if (opt.output_filters) sb_str(&o, ":FILTER");
  num_fields += (!haploid && opt.output_phased_gls) ? 1 : 0;
  num_fields += (opt.output_allreads != 0) + (opt.output_mallreads != 0) + (opt.output_gls != 0) + (opt.output_pls != 0) + 2 * (opt.output_haplotype_data != 0);

  for (int i = 0; i < n_out; i++) {                                     /* :1201-1362 */
    sb_str(&o, "\t");
    int s = -1;
    for (int k = 0; k < S; k++) if (strcmp(out_names[i], v->sample_names[k]) == 0) s = k;
    const char* reason = NULL;
    if (s < 0 || n_al[s] == 0) reason = "NO_READS";
    else if (v->sample_filter && v->sample_filter[s] && v->sample_filter[s][0]) reason = v->sample_filter[s];
    else if (n_al[s] > 0 && n_fl[s] > n_al[s] * opt.max_flank_indel_frac) reason = "FLANK_INDEL_FRAC";
    if (reason) {
      if (!opt.output_filters) sb_str(&o, ".");
      else { for (int k = 0; k < num_fields; k++) sb_str(&o, ".:"); sb_str(&o, reason); }
      continue;
    }
    const int g1 = gts[2 * s], g2 = gts[2 * s + 1];
    if (!haploid) {                                                     /* :1255-1271 */
      sb_int(&o, o2n[g1]); sb_str(&o, "|"); sb_int(&o, o2n[g2]); sb_str(&o, ":"); sb_int(&o, bp[g1]); sb_str(&o, "|"); sb_int(&o, bp[g2]);
      sb_str(&o, ":"); sb_f2(&o, exp(lun[s])); sb_str(&o, ":"); sb_f2(&o, exp(lph[s]));
      sb_str(&o, ":"); sb_int(&o, n_al[s]); sb_str(&o, ":"); sb_int(&o, n_snp[s]); sb_str(&o, ":"); sb_int(&o, n_fl[s]);
      sb_str(&o, ":"); sb_int(&o, v->n_p1s ? v->n_p1s[s] : 0); sb_str(&o, "|"); sb_int(&o, v->n_p2s ? v->n_p2s[s] : 0);
      sb_str(&o, ":"); sb_int(&o, s1[s]); sb_str(&o, "|"); sb_int(&o, s2[s]);
    } else {                                                            /* :1273-1284 */
      sb_int(&o, o2n[g1]); sb_str(&o, ":"); sb_int(&o, bp[g1]); sb_str(&o, ":"); sb_f2(&o, exp(lun[s])); sb_str(&o, ":"); sb_int(&o, n_al[s]); sb_str(&o, ":"); sb_int(&o, n_fl[s]);
    }
    sb_str(&o, ":"); if (V == 1) sb_str(&o, "."); else sb_f2(&o, gld[s]);
    if (opt.output_allreads) { sb_str(&o, ":"); condense(&o, bps[s], nb[s]); }                 /* :1304-1305 */
    if (opt.output_mallreads) { sb_str(&o, ":"); condense(&o, mls[s], nm[s]); }                /* :1308-1309 */
    const double* gl = gls + (size_t)s * n_gl; const int32_t* pl = pls + (size_t)s * n_gl; const double* pg = pgl + (size_t)s * n_pgl;
    if (haploid) {                                                      /* :1312-1324 */
      if (opt.output_gls) { sb_str(&o, ":"); sb_f2(&o, gl[0]); for (int a = 1; a < V; a++) { sb_str(&o, ","); sb_f2(&o, gl[n2o[a]]); } }
      if (opt.output_pls) { sb_str(&o, ":"); sb_int(&o, pl[0]); for (int a = 1; a < V; a++) { sb_str(&o, ","); sb_int(&o, pl[n2o[a]]); } }
    } else {                                                            /* :1326-1358 */
      if (opt.output_gls) { sb_str(&o, ":"); sb_f2(&o, gl[0]);
        for (int a = 1; a < V; a++) for (int b = 0; b <= a; b++) { const int ia = n2o[a] < n2o[b] ? n2o[a] : n2o[b], ib = n2o[a] < n2o[b] ? n2o[b] : n2o[a]; sb_str(&o, ","); sb_f2(&o, gl[ib * (ib + 1) / 2 + ia]); } }
      if (opt.output_pls) { sb_str(&o, ":"); sb_int(&o, pl[0]);
        for (int a = 1; a < V; a++) for (int b = 0; b <= a; b++) { const int ia = n2o[a] < n2o[b] ? n2o[a] : n2o[b], ib = n2o[a] < n2o[b] ? n2o[b] : n2o[a]; sb_str(&o, ","); sb_int(&o, pl[ib * (ib + 1) / 2 + ia]); } }
      if (opt.output_phased_gls) { sb_str(&o, ":"); sb_f2(&o, pg[0]);
        for (int a = 0; a < V; a++) for (int b = 0; b < V; b++) { if (a == 0 && b == 0) continue; sb_str(&o, ","); sb_f2(&o, pg[n2o[a] * V + n2o[b]]); } }
    }
    if (opt.output_haplotype_data) { sb_str(&o, ":"); sb_f2(&o, exp(hun[s])); sb_str(&o, ":"); sb_f2(&o, exp(hph[s])); }
    if (opt.output_filters) sb_str(&o, ":PASS");
  }
  int64_t len = (int64_t)o.n;
  if (len + 1 > cap) len = LTR_ERR_INVALID; else memcpy(out, o.s, (size_t)len + 1);
  if (pos_out) *pos_out = pos;
  for (int i = 0; i < V; i++) free(A.s[i]);
  for (int s = 0; s < S; s++) { free(bps[s]); free(mls[s]); }
  free(A.s); free(inexact); free(bp); free(h2a); free(gts); free(pls); free(lph); free(lun); free(hph); free(hun); free(gld); free(gls); free(pgl);
  free(n_al); free(n_snp); free(n_fl); free(s1); free(s2); free(bps); free(mls); free(nb); free(nm); free(counts); free(n2o); free(o2n); free(o.s);
  return len;
}


/* Genotyper::get_vcf_header, genotyper.cpp:258-336 (restated line by line; UNPINNED: genotyper.cpp needs htslib through
 * fasta_reader.h).  contig_lines stands for FastaReader::write_all_contigs_to_vcf. */
int64_t ltr_oracle_vcf_header(const char* fasta_path, const char* full_command, const char* contig_lines, const ltr_vcf_options* opt_in,
                              const char* const* sample_names, int32_t n_samples, char* out, int64_t cap) {
  ltr_vcf_options opt;
  if (opt_in) opt = *opt_in; else { memset(&opt, 0, sizeof(opt)); opt.output_allreads = 1; opt.output_mallreads = 1; opt.max_flank_indel_frac = 0.15f; }   /* genotyper.cpp:339-346 */
  int64_t at = 0;
#define PUT(...) do { int w_ = snprintf(out + at, (size_t)(cap - at), __VA_ARGS__); if (w_ < 0 || at + w_ >= cap) return LTR_ERR_INVALID; at += w_; } while (0)
#define INFO(id, num, type, desc) PUT("##INFO=<ID=%s,Number=%s,Type=%s,Description=\"%s\">\n", id, num, type, desc)
#define FMT(id, num, type, desc) PUT("##FORMAT=<ID=%s,Number=%s,Type=%s,Description=\"%s\">\n", id, num, type, desc)
  PUT("##fileformat=VCFv4.1\n##command=%s\n##reference=%s\n", full_command, fasta_path);           /* :259-262 */
  if (contig_lines) PUT("%s", contig_lines);                                                          /* :264-265 */
  INFO("START", "1", "Integer", "Inclusive start coodinate for the repetitive portion of the reference allele");   /* :275-288 */
  INFO("END", "1", "Integer", "Inclusive end coordinate for the repetitive portion of the reference allele");
  INFO("MOTIF", ".", "String", "TR motif(s)");
  INFO("PERIOD", ".", "Integer", "Length of TR motif(s)");
  INFO("NSKIP", "1", "Integer", "Number of samples not genotyped due to various issues");
  INFO("NFILT", "1", "Integer", "Number of samples whose genotypes were filtered due to various issues");
  INFO("INEXACT_ALLELE", "A", "Integer", "Boolean showing if each alternate allele is exact or approximated by POA, 0 for exact 1 for approximated.");
  INFO("BPDIFFS", "A", "Integer", "Base pair difference of each alternate allele from the reference allele");
  INFO("DP", "1", "Integer", "Total number of valid reads used to genotype all samples");
  INFO("DSNP", "1", "Integer", "Total number of reads with SNP phasing information");
  INFO("DFLANKINDEL", "1", "Integer", "Total number of reads with an indel in the regions flanking the STR");
  INFO("AN", "1", "Integer", "Total number of alleles in called genotypes");
  INFO("REFAC", "1", "Integer", "Reference allele count");
  INFO("AC", "A", "Integer", "Alternate allele counts");
  FMT("GT", "1", "String", "Genotype");                                                                /* :293-301 */
  FMT("GB", "1", "String", "Base pair differences of genotype from reference");
  FMT("Q", "1", "Float", "Posterior probability of unphased genotype");
  FMT("PQ", "1", "Float", "Posterior probability of phased genotype");
  FMT("DP", "1", "Integer", "Number of valid reads used for sample's genotype");
  FMT("DSNP", "1", "Integer", "Number of reads with SNP phasing information");
  FMT("PSNP", "1", "String", "Number of reads with SNPs supporting each haploid genotype");
  FMT("PDP", "1", "String", "Fractional reads supporting each haploid genotype");
  FMT("GLDIFF", "1", "Float", "Difference in likelihood between the reported and next best genotypes");
  if (opt.output_haplotype_data == 1) {                                                                /* :310-312 */
    FMT("HQ", "1", "Float", "Posterior probability of unphased haplotypes");
    FMT("PHQ", "1", "Float", "Posterior probability of phased haplotypes");
  }
  if (opt.output_allreads == 1) FMT("ALLREADS", "1", "String", "Base pair difference observed in each read's Needleman-Wunsch alignment");
  if (opt.output_mallreads == 1) FMT("MALLREADS", "1", "String", "Maximum likelihood bp diff in each read based on haplotype alignments for reads that span the repeat region by at least 5 base pairs");
  if (opt.output_gls == 1) FMT("GL", "G", "Float", "log10 genotype likelihoods");
  if (opt.output_pls == 1) FMT("PL", "G", "Integer", "Phred-scaled genotype likelihoods");
  if (opt.output_phased_gls == 1) FMT("PHASEDGL", ".", "Float", "log10 genotype likelihood for each phased genotype. Value for phased genotype X|Y is stored at a 0-based index of X*A + Y, where A is the number of alleles. Not applicable to haploid genotypes");
  if (opt.output_filters == 1) FMT("FILTER", "1", "String", "Reason for filtering the current call, or PASS if the call was not filtered");
  PUT("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT");                                       /* :329-333 */
  for (int32_t i = 0; i < n_samples; i++) PUT("\t%s", sample_names[i]);
  PUT("\n");
#undef PUT
#undef INFO
#undef FMT
  return at;
}

/*
 * ltr_oracle_nw.c -- TEST INFRASTRUCTURE (see ltr_oracle.h): plain-C restatement of
 * Haplotype::aln_haps_to_ref (src/SeqAlignment/Haplotype.cpp:58-86): NeedlemanWunsch::Align with
 * use_ref_end_penalty = true (NeedlemanWunsch.cpp:82-96 scores, :121-143 bestIndex, :195-246 nw_helper,
 * :340-378 initMatrices, :174-193 findOptimalStopEndPenalty, :247-338 traceAlignment, :380-420 Align),
 * Haplotype::adjust_indels (Haplotype.cpp:8-56) and the M / I / D string (:72-82).  Full matrices, row by row,
 * like the reference.  PARITY UNPINNED: NeedlemanWunsch.h includes bam_io.h -> htslib, so the reference's
 * translation unit cannot be compiled in the dev container.
 */
#include <stdlib.h>
#include <string.h>

#include "ltr_oracle.h"

static const float NW_A = 2.0f, NW_B = -2.0f, GAPOPEN = 5.0f, GAPEXTEND = 0.125f, LARGE = 1000000.0f;

static int base_to_int(char c) {                                /* :100-119 */
  if (c >= 'a' && c <= 'z') c = (char)(c - 32);
  switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 4; }
}
static float score(int r, int q) { return (r == 4 || q == 4 || r == q) ? NW_A : NW_B; }   /* s[5][5], :88-92 */
static float bestIndex(float s1, float s2, float s3, int* ptr) {   /* :121-143 */
  if (s2 > s1) { if (s2 > s3) { *ptr = 1; return s2; } else { *ptr = 2; return s3; } }
  else { if (s3 > s1) { *ptr = 2; return s3; } else { *ptr = 0; return s1; } }
}

/* Align + adjust_indels + aln_info.  Returns the alignment length (info written to out, capacity L1 + L2). */
int64_t ltr_oracle_nw_aln_info(const uint8_t* refseq, int32_t L1, const uint8_t* readseq, int32_t L2,
                               int32_t ref_pos, int32_t str_pos, char* out) {
  const size_t W = (size_t)L1 + 1, sz = W * ((size_t)L2 + 1);
  float *M = (float*)malloc(sz * 4), *Iref = (float*)malloc(sz * 4), *Iread = (float*)malloc(sz * 4);
  int8_t *tM = (int8_t*)malloc(sz), *tIref = (int8_t*)malloc(sz), *tIread = (int8_t*)malloc(sz);
  if (!M || !Iref || !Iread || !tM || !tIref || !tIread) return LTR_ERR_NOMEM;
  M[0] = 0.0f; Iref[0] = -LARGE; Iread[0] = -LARGE;              /* initMatrices, :343-345 */
  for (int i = 1; i < L1 + 1; i++) {                             /* :348-361 */
    Iref[i] = -GAPOPEN - (float)(i - 1) * GAPEXTEND; tIref[i] = 1;
    Iread[i] = -LARGE; tIread[i] = -1; M[i] = -LARGE; tM[i] = -1;
  }
  for (int i = 1; i < L2 + 1; i++) {                             /* :364-377 */
    const size_t index = (size_t)i * W;
    Iread[index] = -GAPOPEN - (float)(i - 1) * GAPEXTEND; tIread[index] = 2;
    Iref[index] = -LARGE; tIref[index] = -1; M[index] = -LARGE; tM[index] = -1;
  }
  for (int i = 1; i <= L2; i++) {                                /* nw_helper, :214-244 */
    const int read_base = base_to_int((char)readseq[i - 1]);
    for (int j = 1; j <= L1; j++) {
      const size_t nindex = (size_t)i * W + (size_t)j;
      const int ref_base = base_to_int((char)refseq[j - 1]);
      int c;
      size_t oindex = (size_t)(i - 1) * W + (size_t)(j - 1);
      M[nindex] = bestIndex(M[oindex], Iref[oindex], Iread[oindex], &c) + score(ref_base, read_base); tM[nindex] = (int8_t)c;
      oindex = (size_t)i * W + (size_t)(j - 1);
      Iref[nindex] = bestIndex(M[oindex] - GAPOPEN, Iref[oindex] - GAPEXTEND, Iread[oindex] - GAPOPEN, &c); tIref[nindex] = (int8_t)c;
      oindex = (size_t)(i - 1) * W + (size_t)j;
      Iread[nindex] = bestIndex(M[oindex] - GAPOPEN, Iref[oindex] - GAPOPEN, Iread[oindex] - GAPEXTEND, &c); tIread[nindex] = (int8_t)c;
    }
  }
  const size_t last = sz - 1;                                    /* findOptimalStopEndPenalty, :174-193 */
  int best_col = L1, best_type = 0; float best_val = M[last];
  if (Iref[last] > best_val) { best_val = Iref[last]; best_type = 1; }
  if (Iread[last] > best_val) { best_val = Iread[last]; best_type = 2; }
  char* ref_al = (char*)malloc((size_t)(L1 + L2) + 1); char* read_al = (char*)malloc((size_t)(L1 + L2) + 1);
  int n = 0, best_row = L2;                                      /* traceAlignment, :262-310 (best_col == L1: no trailing gaps) */
  while (best_row > 0) {
    const size_t index = (size_t)best_row * W + (size_t)best_col;
    if (best_type == 0) { ref_al[n] = (char)refseq[best_col - 1]; read_al[n] = (char)readseq[best_row - 1]; n++; best_type = tM[index]; best_row--; best_col--; }
    else if (best_type == 1) { ref_al[n] = (char)refseq[best_col - 1]; read_al[n] = '-'; n++; best_type = tIref[index]; best_col--; }
    else if (best_type == 2) { ref_al[n] = '-'; read_al[n] = (char)readseq[best_row - 1]; n++; best_type = tIread[index]; best_row--; }
    else { n = -1; break; }
  }
  if (n >= 0) for (int i = best_col; i > 0; i--) { ref_al[n] = (char)refseq[i - 1]; read_al[n] = '-'; n++; }
  free(M); free(Iref); free(Iread); free(tM); free(tIref); free(tIread);
  if (n < 0) { free(ref_al); free(read_al); return LTR_ERR_INVALID; }
  for (int i = 0; i < n / 2; i++) { char t = ref_al[i]; ref_al[i] = ref_al[n - 1 - i]; ref_al[n - 1 - i] = t; t = read_al[i]; read_al[i] = read_al[n - 1 - i]; read_al[n - 1 - i] = t; }   /* :316-317 */
  /* Haplotype::adjust_indels, Haplotype.cpp:8-56 (ref_hap_al = ref_al, alt_hap_al = read_al) */
  int aln_index = 0;
  while (aln_index < n) {
    if (read_al[aln_index] == '-' && ref_pos < str_pos) {
      int index = aln_index;
      while (index < n && read_al[index] == '-') index++;
      int pos = ref_pos, del_index = aln_index; const int del_size = index - aln_index;
      while (index < n && pos < str_pos && ref_al[del_index] == ref_al[index]) { read_al[del_index] = read_al[index]; read_al[index] = '-'; index++; del_index++; pos++; }
      aln_index = index; ref_pos = pos + del_size;
    } else if (ref_al[aln_index] == '-' && ref_pos < str_pos) {
      int index = aln_index;
      while (index < n && ref_al[index] == '-') index++;
      int pos = ref_pos, ins_index = aln_index;
      while (index < n && pos < str_pos && read_al[ins_index] == read_al[index]) { ref_al[ins_index] = ref_al[index]; ref_al[index] = '-'; index++; ins_index++; pos++; }
      aln_index = index; ref_pos = pos;
    } else {
      if (ref_al[aln_index] != '-') ref_pos++;
      aln_index++;
    }
  }
  for (int i = 0; i < n; i++) out[i] = (ref_al[i] == '-') ? 'I' : ((read_al[i] == '-') ? 'D' : 'M');   /* Haplotype.cpp:72-81 */
  free(ref_al); free(read_al);
  return n;
}

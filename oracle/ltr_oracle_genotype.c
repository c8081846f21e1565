/*
 * ltr_oracle_genotype.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ltr_oracle.h).
 *
 * CPU restatement of Genotyper::extract_genotypes_and_likelihoods (reference
 * src/genotyper.cpp:132-256), calc_PLs (:102-107) and calc_gl_diff (:109-130), SURVEY.md
 * section 8f next-2.
 *
 * PARITY: genotyper.cpp includes htslib headers this image lacks, so the function itself is
 * PARITY UNPINNED by a reference build.  What it is made of IS pinned: the mathops.cpp helpers
 * (fast_log_sum_exp(a,b), log_sum_exp(a,b), update/finish_streaming_log_sum_exp, int_log and the
 * constants) are compiled from the reference into oracle/_ref and the restatements below match
 * them bit for bit (tests/test_genotype_fields.py, tests/golden/mathops.json).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "ltr_oracle.h"

/* fastonebigheader.h:189-198 */
static float o_fastpow2(float p) {
  float offset = (p < 0) ? 1.0f : 0.0f;
  float clipp = (p < -126) ? -126.0f : p;
  int w = (int)clipp;
  float z = clipp - w + offset;
  union { uint32_t i; float f; } v;
  v.i = (uint32_t)((1 << 23) * (clipp + 121.2740575f + 27.7280233f / (4.84252568f - z) - 1.49012907f * z));
  return v.f;
}
/* fastonebigheader.h:200-204 */
static float o_fastexp(float p) { return o_fastpow2(1.442695040f * p); }
/* fastonebigheader.h:320-331 */
static float o_fastlog2(float x) {
  union { float f; uint32_t i; } vx;
  union { uint32_t i; float f; } mx;
  vx.f = x;
  mx.i = (vx.i & 0x007FFFFF) | 0x3f000000;
  float y = (float)vx.i;
  y *= 1.1920928955078125e-7f;
  return y - 124.22551499f - 1.498030302f * mx.f - 1.72587999f / (0.3520887068f + mx.f);
}
/* fastonebigheader.h:333-337 */
static float o_fastlog(float x) { return 0.69314718f * o_fastlog2(x); }

/* mathops.cpp:87-96 (LOG_THRESH: mathops.h:36) */
double ltr_oracle_fast_log_sum_exp2(double log_v1, double log_v2) {
  const double LOG_THRESH = log(0.001);
  if (log_v1 > log_v2) {
    double diff = log_v2 - log_v1;
    return diff < LOG_THRESH ? log_v1 : log_v1 + o_fastlog(1 + o_fastexp((float)diff));
  } else {
    double diff = log_v1 - log_v2;
    return diff < LOG_THRESH ? log_v2 : log_v2 + o_fastlog(1 + o_fastexp((float)diff));
  }
}
/* mathops.cpp:55-60 */
double ltr_oracle_log_sum_exp2(double log_v1, double log_v2) {
  if (log_v1 > log_v2) return log_v1 + log(1 + exp(log_v2 - log_v1));
  else return log_v2 + log(1 + exp(log_v1 - log_v2));
}
/* mathops.cpp:70-79 */
static void o_update_streaming(double log_val, double* max_val, double* total) {
  if (log_val <= *max_val) *total += exp(log_val - *max_val);
  else { *total *= exp(*max_val - log_val); *total += 1.0; *max_val = log_val; }
}
/* mathops.cpp:70-85 with the genotyper's initial state (genotyper.cpp:153-154) */
double ltr_oracle_streaming_log_sum_exp(const double* vals, int32_t n) {
  double mx = -DBL_MAX / 2, tot = 0.0;
  for (int32_t i = 0; i < n; i++) o_update_streaming(vals[i], &mx, &tot);
  return mx + log(tot);
}
/* mathops.cpp:16-22 */
double ltr_oracle_int_log(int32_t v) { return v == 0 ? -1000.0 : log((double)(unsigned int)v); }

/* genotyper.cpp:132-256; argument meaning as ltr_extract_genotypes. */
int ltr_oracle_extract_genotypes(int32_t num_samples, int32_t num_alleles, int32_t num_variants,
                                 const int32_t* hap_to_allele, int32_t haploid,
                                 const double* log_sample_posteriors, const double* sample_total_LLs,
                                 const int32_t* best_haplotypes, const ltr_genotype_fields* out) {
  const double LOG_E_BASE_10 = 0.4342944819, TOLERANCE = 1e-10;          /* mathops.cpp:11-12 */
  const int nv2 = num_variants * num_variants;
  double* max_lp = (double*)malloc(sizeof(double) * (size_t)num_samples * nv2);
  double* tot_lp = (double*)malloc(sizeof(double) * (size_t)num_samples * nv2);
  int32_t* best_gts = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)num_samples);
  for (int i = 0; i < num_samples * nv2; i++) { max_lp[i] = -DBL_MAX / 2; tot_lp[i] = 0.0; }
  for (int s = 0; s < num_samples; s++) {                                  /* :147-150 */
    best_gts[2 * s] = hap_to_allele[best_haplotypes[2 * s]];
    best_gts[2 * s + 1] = hap_to_allele[best_haplotypes[2 * s + 1]];
  }
  const double* ptr = log_sample_posteriors;                              /* :152-163 */
  for (int s = 0; s < num_samples; s++)
    for (int i1 = 0; i1 < num_alleles; i1++)
      for (int i2 = 0; i2 < num_alleles; i2++, ptr++) {
        int gt_index = num_variants * hap_to_allele[i1] + hap_to_allele[i2];
        o_update_streaming(*ptr, &max_lp[s * nv2 + gt_index], &tot_lp[s * nv2 + gt_index]);
      }
  for (int g = 0; g < nv2; g++)                                            /* :164-172 */
    for (int s = 0; s < num_samples; s++) tot_lp[s * nv2 + g] = max_lp[s * nv2 + g] + log(tot_lp[s * nv2 + g]);

  ptr = log_sample_posteriors;                                            /* :174-185 */
  for (int s = 0; s < num_samples; s++) {
    int a = best_haplotypes[2 * s] * num_alleles + best_haplotypes[2 * s + 1];
    int b = best_haplotypes[2 * s + 1] * num_alleles + best_haplotypes[2 * s];
    if (out->hap_log_phased_posteriors) out->hap_log_phased_posteriors[s] = ptr[a];
    if (out->hap_log_unphased_posteriors)
      out->hap_log_unphased_posteriors[s] = (a != b) ? ltr_oracle_fast_log_sum_exp2(ptr[a], ptr[b]) : ptr[a];
    ptr += num_alleles * num_alleles;
  }
  for (int s = 0; s < num_samples; s++) {                                  /* :187-198 */
    int gt_a = best_gts[2 * s], gt_b = best_gts[2 * s + 1];
    double lp = tot_lp[s * nv2 + num_variants * gt_a + gt_b];
    if (out->log_phased_posteriors) out->log_phased_posteriors[s] = lp;
    if (out->log_unphased_posteriors)
      out->log_unphased_posteriors[s] = (gt_a == gt_b) ? lp : ltr_oracle_log_sum_exp2(lp, tot_lp[s * nv2 + num_variants * gt_b + gt_a]);
  }
  if (out->best_gts) memcpy(out->best_gts, best_gts, sizeof(int32_t) * 2 * (size_t)num_samples);

  if (out->gls || out->pls || out->phased_gls || out->gl_diffs) {          /* :204-255 */
    const int n_gl = haploid ? num_variants : num_variants * (num_variants + 1) / 2;
    const int n_pgl = haploid ? num_variants : nv2;
    double* gls = (double*)malloc(sizeof(double) * (size_t)num_samples * n_gl);
    int* ngl = (int*)calloc((size_t)num_samples, sizeof(int));
    int* npgl = (int*)calloc((size_t)num_samples, sizeof(int));
    /* log_homozygous_prior / log_heterozygous_prior, :21-33, :209-210 */
    double hom_ll_correction = haploid ? -ltr_oracle_int_log(num_alleles)
                                       : ltr_oracle_int_log(2) - ltr_oracle_int_log(num_alleles) - ltr_oracle_int_log(num_alleles + 1);
    double het_ll_correction = haploid ? 0 : -ltr_oracle_int_log(num_alleles) - ltr_oracle_int_log(num_alleles + 1);
    double gl_nconfig_corr, pgl_nconfig_corr;                             /* :213-221 */
    if (haploid) {
      gl_nconfig_corr = ltr_oracle_int_log(2) + ltr_oracle_int_log(num_alleles) - ltr_oracle_int_log(num_variants);
      pgl_nconfig_corr = ltr_oracle_int_log(num_alleles) - ltr_oracle_int_log(num_variants);
    } else {
      gl_nconfig_corr = ltr_oracle_int_log(2) + 2 * (ltr_oracle_int_log(num_alleles) - ltr_oracle_int_log(num_variants));
      pgl_nconfig_corr = 2 * (ltr_oracle_int_log(num_alleles) - ltr_oracle_int_log(num_variants));
    }
    int gt_index = 0;                                                     /* :224-241 */
    for (int i1 = 0; i1 < num_variants; i1++)
      for (int i2 = 0; i2 < num_variants; i2++, gt_index++) {
        int alt = i2 * num_variants + i1;
        double gl_ll_corr = (i1 == i2 ? hom_ll_correction : het_ll_correction) + gl_nconfig_corr;
        double pgl_ll_corr = (i1 == i2 ? hom_ll_correction : het_ll_correction) + pgl_nconfig_corr;
        for (int s = 0; s < num_samples; s++) {
          if ((i2 <= i1) && (!haploid || (i1 == i2))) {
            double gl_base_e = sample_total_LLs[s] - gl_ll_corr
                               + ltr_oracle_fast_log_sum_exp2(tot_lp[s * nv2 + gt_index], tot_lp[s * nv2 + alt]);
            gls[s * n_gl + ngl[s]++] = gl_base_e * LOG_E_BASE_10;
          }
          if (out->phased_gls && (!haploid || (i1 == i2)))
            out->phased_gls[s * n_pgl + npgl[s]++] = (sample_total_LLs[s] - pgl_ll_corr + tot_lp[s * nv2 + gt_index]) * LOG_E_BASE_10;
        }
      }
    for (int s = 0; s < num_samples; s++) {
      const double* g = gls + s * n_gl;
      double max_gl = g[0];
      for (int i = 1; i < n_gl; i++) if (max_gl < g[i]) max_gl = g[i];    /* std::max_element */
      if (out->gl_diffs) {                                                /* calc_gl_diff, :109-130 */
        if (num_alleles == 1) out->gl_diffs[s] = -1000;
        else {
          double second_gl = -DBL_MAX;
          for (int i = 0; i < n_gl; i++) if (g[i] < max_gl) second_gl = (second_gl > g[i] ? second_gl : g[i]);
          if (second_gl == -DBL_MAX) second_gl = max_gl;
          int gt_a = best_gts[2 * s], gt_b = best_gts[2 * s + 1], gl_index;
          if (haploid) gl_index = gt_a;
          else {
            int min_gt = gt_a < gt_b ? gt_a : gt_b, max_gt = gt_a < gt_b ? gt_b : gt_a;
            gl_index = max_gt * (max_gt + 1) / 2 + min_gt;
          }
          out->gl_diffs[s] = (fabs(max_gl - g[gl_index]) < TOLERANCE) ? (max_gl - second_gl) : g[gl_index] - max_gl;
        }
      }
      if (out->pls)                                                       /* calc_PLs, :102-107 */
        for (int i = 0; i < n_gl; i++) {
          int v = (int)(-10 * (g[i] - max_gl));
          out->pls[s * n_gl + i] = v < 999 ? v : 999;
        }
    }
    if (out->gls) memcpy(out->gls, gls, sizeof(double) * (size_t)num_samples * n_gl);
    free(gls); free(ngl); free(npgl);
  }
  free(max_lp); free(tot_lp); free(best_gts);
  return 0;
}

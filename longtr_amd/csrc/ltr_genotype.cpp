// ltr_genotype.cpp -- host side of the genotype consumer.
//
// Replaces Genotyper::extract_genotypes_and_likelihoods (reference src/genotyper.cpp:132-256),
// calc_PLs (:102-107) and calc_gl_diff (:109-130).  O(S * H^2) on matrices the posterior kernels
// already produced; the reference runs it on the host with libm, and so does this file, in the
// same evaluation order, so the printed Q / PQ / GL values agree to the last bit whenever the
// posterior matrix does.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/ltr_gpu.h"

namespace {

// fast_log_sum_exp(double, double) (mathops.cpp:87-96) goes through the FP32 approximations
// fastexp / fastlog of fastonebigheader.h:189-204, :321-337 (Mineiro's fastapprox): restated
// here operation by operation in float (the translation unit is built with -ffp-contract=off).
inline float bits_to_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t float_to_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

inline float approx_pow2(float p) {
  const float offset = (p < 0) ? 1.0f : 0.0f;
  const float clipped = (p < -126) ? -126.0f : p;
  const int whole = (int)clipped;                                // truncation toward zero
  const float frac = clipped - (float)whole + offset;
  const float scaled = (float)(1 << 23) *
      (clipped + 121.2740575f + 27.7280233f / (4.84252568f - frac) - 1.49012907f * frac);
  return bits_to_float((uint32_t)scaled);
}
inline float approx_exp(float p) { return approx_pow2(1.442695040f * p); }
inline float approx_log2(float x) {
  const uint32_t xi = float_to_bits(x);
  const float mant = bits_to_float((xi & 0x007FFFFFu) | 0x3f000000u);
  float y = (float)xi;
  y *= 1.1920928955078125e-7f;
  return y - 124.22551499f - 1.498030302f * mant - 1.72587999f / (0.3520887068f + mant);
}
inline float approx_log(float x) { return 0.69314718f * approx_log2(x); }

const double kLogThresh = std::log(0.001);                        // mathops.h:36
const double kLogEBase10 = 0.4342944819;                          // mathops.cpp:12
const double kTolerance = 1e-10;                                  // mathops.cpp:11

inline double fast_lse2(double a, double b) {                     // mathops.cpp:87-96
  const double hi = (a > b) ? a : b;
  const double diff = (a > b) ? (b - a) : (a - b);
  if (diff < kLogThresh) return hi;
  return hi + approx_log(1 + approx_exp((float)diff));
}
inline double lse2(double a, double b) {                          // mathops.cpp:55-60
  return (a > b) ? a + std::log(1 + std::exp(b - a)) : b + std::log(1 + std::exp(a - b));
}
inline double int_log(int v) { return v == 0 ? -1000.0 : std::log((double)v); }   // mathops.cpp:16-22

}  // namespace

extern "C" int ltr_extract_genotypes(int32_t S, int32_t H, int32_t V, const int32_t* hap_to_allele, int32_t haploid,
                                     const double* post, const double* sample_total_ll,
                                     const int32_t* best_haplotypes, const ltr_genotype_fields* out) {
  if (S < 0 || H < 1 || V < 1 || !hap_to_allele || !post || !best_haplotypes || !out) return LTR_ERR_INVALID;
  for (int32_t h = 0; h < H; ++h)
    if (hap_to_allele[h] < 0 || hap_to_allele[h] >= V) return LTR_ERR_INVALID;
  for (int32_t s = 0; s < 2 * S; ++s)
    if (best_haplotypes[s] < 0 || best_haplotypes[s] >= H) return LTR_ERR_INVALID;
  const bool want_gl = out->gls || out->pls || out->phased_gls || out->gl_diffs;
  if (want_gl && !sample_total_ll) return LTR_ERR_INVALID;
  const int64_t HH = (int64_t)H * H, VV = (int64_t)V * V;

  // genotype posteriors: streaming log-sum-exp over the haplotype pairs that map to each
  // (allele, allele) cell, in haplotype-pair order (:152-172, mathops.cpp:70-85)
  std::vector<double> cell_max((size_t)(S * VV), -DBL_MAX / 2), cell_tot((size_t)(S * VV), 0.0);
  for (int32_t s = 0; s < S; ++s) {
    double* mx = cell_max.data() + s * VV;
    double* tot = cell_tot.data() + s * VV;
    const double* p = post + s * HH;
    for (int32_t h1 = 0; h1 < H; ++h1)
      for (int32_t h2 = 0; h2 < H; ++h2, ++p) {
        const int64_t g = (int64_t)V * hap_to_allele[h1] + hap_to_allele[h2];
        if (*p <= mx[g]) tot[g] += std::exp(*p - mx[g]);
        else { tot[g] *= std::exp(mx[g] - *p); tot[g] += 1.0; mx[g] = *p; }
      }
    for (int64_t g = 0; g < VV; ++g) tot[g] = mx[g] + std::log(tot[g]);
  }
  const double* gt_post = cell_tot.data();                        // [S x V x V] log phased genotype posteriors

  std::vector<int32_t> gts((size_t)(2 * S));
  for (int32_t s = 0; s < S; ++s) {
    const int32_t ha = best_haplotypes[2 * s], hb = best_haplotypes[2 * s + 1];
    const int32_t ga = hap_to_allele[ha], gb = hap_to_allele[hb];
    gts[2 * s] = ga; gts[2 * s + 1] = gb;                        // :147-150
    const double* p = post + s * HH;
    const int64_t ia = (int64_t)ha * H + hb, ib = (int64_t)hb * H + ha;
    if (out->hap_log_phased_posteriors) out->hap_log_phased_posteriors[s] = p[ia];          // :174-185
    if (out->hap_log_unphased_posteriors) out->hap_log_unphased_posteriors[s] = (ia != ib) ? fast_lse2(p[ia], p[ib]) : p[ia];
    const double phased = gt_post[s * VV + (int64_t)V * ga + gb];                           // :187-198
    if (out->log_phased_posteriors) out->log_phased_posteriors[s] = phased;
    if (out->log_unphased_posteriors)
      out->log_unphased_posteriors[s] = (ga == gb) ? phased : lse2(phased, gt_post[s * VV + (int64_t)V * gb + ga]);
  }
  if (out->best_gts && S > 0) std::memcpy(out->best_gts, gts.data(), gts.size() * sizeof(int32_t));
  if (!want_gl || S == 0) return LTR_OK;

  // likelihoods: posteriors minus the priors that went in, averaged over the haplotype
  // configurations behind each genotype (:204-241)
  const double hom_prior = haploid ? -int_log(H) : int_log(2) - int_log(H) - int_log(H + 1);     // :21-26
  const double het_prior = haploid ? 0.0 : -int_log(H) - int_log(H + 1);                          // :28-33, :210
  const double gl_cfg = haploid ? int_log(2) + int_log(H) - int_log(V) : int_log(2) + 2 * (int_log(H) - int_log(V));
  const double pgl_cfg = haploid ? int_log(H) - int_log(V) : 2 * (int_log(H) - int_log(V));
  const int64_t n_gl = haploid ? V : (int64_t)V * (V + 1) / 2;
  const int64_t n_pgl = haploid ? V : VV;
  std::vector<double> gls((size_t)(S * n_gl));
  for (int32_t s = 0; s < S; ++s) {
    int64_t kg = 0, kp = 0;
    for (int32_t a = 0; a < V; ++a)
      for (int32_t b = 0; b < V; ++b) {
        if (haploid && a != b) continue;
        const double prior = (a == b) ? hom_prior : het_prior;
        const int64_t g = (int64_t)a * V + b, galt = (int64_t)b * V + a;
        if (b <= a) {
          const double ln_gl = sample_total_ll[s] - (prior + gl_cfg) + fast_lse2(gt_post[s * VV + g], gt_post[s * VV + galt]);
          gls[(size_t)(s * n_gl + kg++)] = ln_gl * kLogEBase10;   // ln -> log10, :234
        }
        if (out->phased_gls)
          out->phased_gls[s * n_pgl + kp++] = (sample_total_ll[s] - (prior + pgl_cfg) + gt_post[s * VV + g]) * kLogEBase10;
      }
  }
  for (int32_t s = 0; s < S; ++s) {
    const double* g = gls.data() + s * n_gl;
    double max_gl = g[0];
    for (int64_t k = 1; k < n_gl; ++k) if (max_gl < g[k]) max_gl = g[k];
    if (out->gl_diffs) {                                          // calc_gl_diff, :109-130
      double d;
      if (H == 1) d = -1000;
      else {
        double second = -DBL_MAX;
        for (int64_t k = 0; k < n_gl; ++k) if (g[k] < max_gl && g[k] > second) second = g[k];
        if (second == -DBL_MAX) second = max_gl;
        const int32_t ga = gts[2 * s], gb = gts[2 * s + 1];
        const int64_t idx = haploid ? ga : (int64_t)(ga > gb ? ga : gb) * ((ga > gb ? ga : gb) + 1) / 2 + (ga < gb ? ga : gb);
        d = (std::fabs(max_gl - g[idx]) < kTolerance) ? (max_gl - second) : g[idx] - max_gl;
      }
      out->gl_diffs[s] = d;
    }
    if (out->pls)                                                 // calc_PLs, :102-107
      for (int64_t k = 0; k < n_gl; ++k) {
        const int v = (int)(-10 * (g[k] - max_gl));
        out->pls[s * n_gl + k] = v < 999 ? v : 999;
      }
  }
  if (out->gls) std::memcpy(out->gls, gls.data(), gls.size() * sizeof(double));
  return LTR_OK;
}

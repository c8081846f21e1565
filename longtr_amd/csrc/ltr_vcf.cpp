// ltr_vcf.cpp -- the genotyper's last steps after the alignment probabilities and posteriors
// (SURVEY.md 8f next-2): unused-allele pruning, the haplotype re-mapping of add_and_remove_alleles,
// allele extraction for the VCF, and the VCF record itself -- GT:GB:Q:PQ:DP:DSNP:DFLANKINDEL:PDP:
// PSNP:GLDIFF[:ALLREADS][:MALLREADS][:GL][:PL][:PHASEDGL] -- for the long-read path
// (SWITCH_OLD_ALIGN_LEN == 0: no alignment traces, so DFLANKINDEL / DSTUTTER are 0).
// Host code, flat arrays in, text out; citations are to the LongTR reference:
//   src/seq_stutter_genotyper.cpp :240-308 (haps_to_alleles, get_unused_alleles), :317-414
//   (add_and_remove_alleles, remove_alleles), :667-785 (reorder_alleles, get_alleles), :894-1366
//   (write_vcf_record), src/extract_indels.cpp:18-91 (ExtractCigar), src/genotyper.h:50-63.

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "ltr_internal.h"

namespace {

const double kLogOneHalf = -0.6931471805599453094;   // LOG_ONE_HALF, mathops.cpp:10
const double kTolerance = 1e-10;                      // TOLERANCE, mathops.cpp:12

std::string allele_of(const ltr_haplotype_blocks* hap, int block, int allele) {
  int64_t k = 0;
  for (int b = 0; b < block; ++b) k += hap->n_alleles[b];
  k += allele;
  return std::string(reinterpret_cast<const char*>(hap->allele_bytes) + hap->allele_off[k],
                     (size_t)(hap->allele_off[k + 1] - hap->allele_off[k]));
}

// log_sum_exp over a vector (mathops.cpp:40-53): max, then sum of exp(x - max) in index order
double log_sum_exp(const std::vector<double>& v) {
  double mx = v[0];
  for (size_t i = 1; i < v.size(); ++i) mx = std::max(mx, v[i]);
  double tot = 0.0;
  for (size_t i = 0; i < v.size(); ++i) tot += std::exp(v[i] - mx);
  return mx + std::log(tot);
}

// Genotyper::condense_read_counts (genotyper.h:50-63): key|count pairs in key order, ';' separated
std::string condense(const std::vector<int>& diffs) {
  if (diffs.empty()) return ".";
  std::map<int, int> counts;
  for (int d : diffs) counts[d]++;
  std::ostringstream res;
  bool first = true;
  for (const auto& kv : counts) { if (!first) res << ";"; first = false; res << kv.first << "|" << kv.second; }
  return res.str();
}

// ExtractCigar (extract_indels.cpp:18-91): net indel size of a read inside [region_start, region_end]
bool extract_cigar(const ltr_alignment& a, int region_start, int region_end, int* bp_diff) {
  const int n = a.n_cigar;
  auto is_m = [](char t) { return t == 'M' || t == '=' || t == 'X'; };
  auto is_ref = [&](char t) { return is_m(t) || t == 'D'; };
  int span = 0;
  for (int i = 0; i < n; ++i) if (is_ref(a.cigar_type[i])) span += a.cigar_num[i];
  if (region_start < a.start) return false;
  if (region_end >= a.start + span) return false;
  if (n == 0) return false;                                    // (the reference would index an empty vector here)
  int pos = a.start;
  size_t first = 0, last_match = 0;
  while (pos < region_start && first < (size_t)n) {
    if (is_ref(a.cigar_type[first])) pos += a.cigar_num[first];
    if (is_m(a.cigar_type[first])) last_match = first;
    ++first;
  }
  first = last_match;
  if (first == 0 && !is_m(a.cigar_type[0])) return false;
  size_t end = (size_t)n - 1;
  last_match = (size_t)n - 1;
  pos = a.start + span;
  while (pos > region_end) {
    if (is_ref(a.cigar_type[end])) pos -= a.cigar_num[end];
    if (is_m(a.cigar_type[end])) last_match = end;
    if (end == 0) break;
    --end;
  }
  end = last_match;
  if (end == (size_t)n - 1 && !is_m(a.cigar_type[end])) return false;
  int diff = 0;
  for (size_t i = first; i <= end; ++i) {
    if (a.cigar_type[i] == 'D') diff -= a.cigar_num[i];
    else if (a.cigar_type[i] == 'I') diff += a.cigar_num[i];
  }
  *bp_diff = diff;
  return true;
}

char upper(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

// chrom_seq.substr(pos, len) of a window that starts at coordinate `off` (std::string::substr clamps at the end)
std::string window(const ltr_vcf_locus* v, int64_t pos, int64_t len) {
  std::string s;
  for (int64_t p = pos; p < pos + len; ++p) {
    const int64_t k = p - v->chrom_seq_start;
    if (k < 0 || k >= v->chrom_seq_len) break;
    s.push_back(upper((char)v->chrom_seq[k]));
  }
  return s;
}

// get_alleles (:688-785): the block's alleles trimmed to the region and padded with reference bases
void get_alleles(const ltr_vcf_locus* v, int32_t* pos_out, std::vector<std::string>* alleles, std::vector<bool>* inexact) {
  const ltr_haplotype_blocks* hap = v->hap;
  const int nopt = hap->n_alleles[v->block];
  int deleted = -1;
  for (int i = 0; i < nopt; ++i) {
    const std::string s = allele_of(hap, v->block, i);
    if (s.empty()) { alleles->push_back("<DEL>"); deleted = i; inexact->push_back(false); continue; }
    alleles->push_back(s);
    inexact->push_back(v->inexact_allele ? v->inexact_allele[i] != 0 : false);
  }
  if (deleted != -1) { std::string t = (*alleles)[1]; (*alleles)[1] = "<DEL>"; (*alleles)[(size_t)deleted] = t; }   // :708-712
  auto is_del = [&](size_t i) { return (*alleles)[i] == "<DEL>"; };
  const std::string ref0 = (*alleles)[0];
  int32_t left = 0, start = hap->block_start[v->block];
  while (start + left < v->region_start) {                     // :715-728
    bool trim = true;
    for (size_t i = 0; i < alleles->size(); ++i) {
      if (is_del(i)) continue;
      const std::string& a = (*alleles)[i];
      if ((size_t)(left + 1) >= a.size() || a[(size_t)left] != (*alleles)[0][(size_t)left]) { trim = false; break; }
    }
    if (!trim) break;
    ++left;
  }
  start += left;
  for (size_t i = 0; i < alleles->size(); ++i) if (!is_del(i)) (*alleles)[i] = (*alleles)[i].substr((size_t)left);
  int32_t right = 0, end = hap->block_end[v->block];
  while (end - right > v->region_stop) {                       // :736-752
    bool trim = true;
    const int ref_size = (int)(*alleles)[0].size();
    for (size_t i = 0; i < alleles->size(); ++i) {
      if (is_del(i)) continue;
      const std::string& a = (*alleles)[i];
      const int alt_size = (int)a.size();
      if ((size_t)(right + 1) >= a.size() || a[(size_t)(alt_size - right - 1)] != (*alleles)[0][(size_t)(ref_size - right - 1)]) { trim = false; break; }
    }
    if (!trim) break;
    ++right;
  }
  end -= right;
  for (size_t i = 0; i < alleles->size(); ++i) if (!is_del(i)) (*alleles)[i] = (*alleles)[i].substr(0, (*alleles)[i].size() - (size_t)right);
  std::string lflank = (start >= v->region_start) ? window(v, v->region_start, start - v->region_start) : "";      // :759-760
  const std::string rflank = (end <= v->region_stop) ? window(v, end, v->region_stop - end) : "";
  int32_t pos = std::min(v->region_start, start);
  if (lflank.empty()) {                                        // :764-777: 1 bp on the left so that every allele starts like the reference
    bool pad = false;
    for (size_t i = 1; i < alleles->size(); ++i) {
      if (is_del(i)) continue;
      if ((*alleles)[i].empty() || (*alleles)[i][0] != (*alleles)[0][0]) { pad = true; break; }
    }
    if (pad) { pos -= 1; lflank = window(v, pos, 1); }
  }
  for (size_t i = 0; i < alleles->size(); ++i) if (!is_del(i)) (*alleles)[i] = lflank + (*alleles)[i] + rflank;
  *pos_out = pos + 1;                                          // :784
}

// reorder_alleles (:667-686): alternates sorted by length then sequence ("<DEL>" stays second)
void reorder(const std::vector<std::string>& alleles, std::vector<int>* old_to_new, std::vector<int>* new_to_old) {
  std::map<std::string, int> old_index;
  for (size_t i = 0; i < alleles.size(); ++i) old_index[alleles[i]] = (int)i;
  std::vector<std::string> sorted = alleles;
  auto by_len = [](const std::string& x, const std::string& y) { return x.size() != y.size() ? x.size() < y.size() : x.compare(y) < 0; };
  if (alleles.size() > 1 && alleles[1] == "<DEL>") std::sort(sorted.begin() + 2, sorted.end(), by_len);
  else if (alleles.size() > 1) std::sort(sorted.begin() + 1, sorted.end(), by_len);
  old_to_new->assign(alleles.size(), -1);
  for (size_t i = 0; i < sorted.size(); ++i) {
    const int o = old_index[sorted[i]];
    new_to_old->push_back(o);
    (*old_to_new)[(size_t)o] = (int)i;
  }
}

}  // namespace

extern "C" {

void ltr_default_vcf_options(ltr_vcf_options* o) {
  if (!o) return;
  // Genotyper's static defaults (genotyper.cpp:339-346)
  o->output_gls = 0; o->output_pls = 0; o->output_phased_gls = 0; o->output_allreads = 1; o->output_mallreads = 1;
  o->output_filters = 0; o->output_haplotype_data = 0; o->max_flank_indel_frac = 0.15f;
}

// SeqStutterGenotyper::haps_to_alleles (:240-248): allele of block `block` in every haplotype, Haplotype::next() order
int ltr_haps_to_alleles(const ltr_haplotype_blocks* hap, int32_t block, int32_t* hap_to_allele) {
  if (!hap || !hap_to_allele || block < 0 || block >= hap->n_blocks) return LTR_ERR_INVALID;
  std::vector<int32_t> counts; int64_t H = 0;
  const int rc = ltr::haplotype_counts(hap, &counts, &H);
  if (rc != LTR_OK) return rc;
  for (int64_t k = 0; k < H; ++k) hap_to_allele[k] = counts[(size_t)(k * hap->n_blocks + block)];
  return LTR_OK;
}

// SeqStutterGenotyper::get_unused_alleles(check_spanned = false, check_called = true) (:250-308) for one block:
// the non-reference alleles no called sample with an aligned read carries in its optimal haplotype pair.
int32_t ltr_unused_alleles(int32_t n_samples, const int32_t* best_haplotypes, const uint8_t* sample_has_aligned_read,
                           const uint8_t* sample_filtered, int32_t n_haplotypes, const int32_t* hap_to_allele,
                           int32_t n_block_alleles, int32_t* unused) {
  if (n_samples < 0 || !best_haplotypes || !hap_to_allele || n_block_alleles < 1 || !unused) return LTR_ERR_INVALID;
  if (n_block_alleles == 1) return 0;                          // :272-274
  std::vector<bool> called((size_t)n_block_alleles, false);
  for (int32_t s = 0; s < n_samples; ++s) {
    if (sample_has_aligned_read && !sample_has_aligned_read[s]) continue;
    if (sample_filtered && sample_filtered[s]) continue;       // call_sample_[s] not empty
    const int32_t a = best_haplotypes[2 * s], b = best_haplotypes[2 * s + 1];
    if (a < 0 || a >= n_haplotypes || b < 0 || b >= n_haplotypes) return LTR_ERR_INVALID;
    const int32_t x = hap_to_allele[a], y = hap_to_allele[b];
    if (x < 0 || x >= n_block_alleles || y < 0 || y >= n_block_alleles) return LTR_ERR_INVALID;
    called[(size_t)x] = true; called[(size_t)y] = true;
  }
  int32_t n = 0;
  for (int32_t k = 1; k < n_block_alleles; ++k) if (!called[(size_t)k]) unused[n++] = k;
  return n;
}

// add_and_remove_alleles (:317-409), the bookkeeping around the re-alignment: haplotype sequences of the
// old and the updated block list are matched by sequence -- allele_mapping[old] = new index or -1,
// realign_to_hap[new] = 1 for sequences that did not exist before -- and the alignment probabilities of
// surviving haplotypes move to their new columns (everything else -100000, :367).
int ltr_remap_haplotypes(const ltr_haplotype_blocks* old_hap, const ltr_haplotype_blocks* new_hap,
                         int32_t* allele_mapping, uint8_t* realign_to_hap) {
  if (!old_hap || !new_hap || !allele_mapping || !realign_to_hap) return LTR_ERR_INVALID;
  const int64_t Ho = ltr_haplotype_num_combs(old_hap), Hn = ltr_haplotype_num_combs(new_hap);
  if (Ho <= 0 || Hn <= 0) return LTR_ERR_INVALID;
  int64_t cap = 1;
  for (const ltr_haplotype_blocks* h : {old_hap, new_hap}) {
    int64_t k = 0;
    for (int b = 0; b < h->n_blocks; ++b) {
      int64_t mx = 0;
      for (int a = 0; a < h->n_alleles[b]; ++a, ++k) mx = std::max(mx, h->allele_off[k + 1] - h->allele_off[k]);
      cap += mx;
    }
  }
  std::vector<uint8_t> buf((size_t)cap);
  std::map<std::string, int32_t> index;                        // (a repeated sequence keeps its LAST index, like the reference's map assignment)
  for (int64_t k = 0; k < Ho; ++k) {
    const int64_t len = ltr_haplotype_seq(old_hap, k, buf.data(), cap);
    if (len < 0) return (int)len;
    index[std::string(buf.begin(), buf.begin() + len)] = (int32_t)k;
    allele_mapping[k] = -1;
  }
  for (int64_t k = 0; k < Hn; ++k) {
    const int64_t len = ltr_haplotype_seq(new_hap, k, buf.data(), cap);
    if (len < 0) return (int)len;
    auto it = index.find(std::string(buf.begin(), buf.begin() + len));
    if (it == index.end()) realign_to_hap[k] = 1;
    else { realign_to_hap[k] = 0; allele_mapping[it->second] = (int32_t)k; }
  }
  return LTR_OK;
}

int ltr_remap_aln_probs(const double* old_ll, int32_t n_reads, int32_t h_old, const int32_t* allele_mapping, int32_t h_new, double* new_ll) {
  if (!old_ll || !allele_mapping || !new_ll || n_reads < 0 || h_old <= 0 || h_new <= 0) return LTR_ERR_INVALID;
  std::fill_n(new_ll, (size_t)n_reads * (size_t)h_new, -100000.0);                 // :367
  for (int32_t i = 0; i < n_reads; ++i)
    for (int32_t j = 0; j < h_old; ++j)
      if (allele_mapping[j] != -1) {
        if (allele_mapping[j] < 0 || allele_mapping[j] >= h_new) return LTR_ERR_INVALID;
        new_ll[(size_t)i * h_new + allele_mapping[j]] = old_ll[(size_t)i * h_old + j];
      }
  return LTR_OK;
}

// get_alleles (:688-785): returns the number of alleles; their text goes to `out` back to back, allele i at
// [allele_off[i], allele_off[i+1]); *pos = 1-based VCF position.
int32_t ltr_get_alleles(const ltr_vcf_locus* v, int32_t* pos, char* out, int64_t cap, int64_t* allele_off) {
  if (!v || !v->hap || !pos || !out || !allele_off || v->block < 0 || v->block >= v->hap->n_blocks) return LTR_ERR_INVALID;
  try {
    std::vector<std::string> alleles; std::vector<bool> inexact;
    get_alleles(v, pos, &alleles, &inexact);
    int64_t at = 0;
    allele_off[0] = 0;
    for (size_t i = 0; i < alleles.size(); ++i) {
      if (at + (int64_t)alleles[i].size() > cap) return LTR_ERR_INVALID;
      std::memcpy(out + at, alleles[i].data(), alleles[i].size());
      at += (int64_t)alleles[i].size();
      allele_off[i + 1] = at;
    }
    return (int32_t)alleles.size();
  } catch (...) { return LTR_ERR_INVALID; }
}

// write_vcf_record (:894-1366) for one repeat block of one locus: CHROM .. FORMAT and one column per
// requested sample.  Returns the length of the text written to `out` (no trailing newline) or < 0.
int64_t ltr_vcf_record(const ltr_vcf_locus* v, const ltr_vcf_options* opt_in, char* out, int64_t cap, int32_t* pos_out) {
  if (!v || !v->hap || !out || v->n_reads < 0 || v->n_samples <= 0 || v->block < 0 || v->block >= v->hap->n_blocks) return LTR_ERR_INVALID;
  if (!v->log_aln_probs || !v->log_p1 || !v->log_p2 || !v->sample_label || !v->log_sample_posteriors || !v->sample_total_ll ||
      !v->best_haplotypes || !v->chrom || !v->sample_names) return LTR_ERR_INVALID;
  try {
    ltr_vcf_options opt;
    if (opt_in) opt = *opt_in; else ltr_default_vcf_options(&opt);
    const bool haploid = v->haploid != 0;
    const int S = v->n_samples, R = v->n_reads;
    const int64_t H = ltr_haplotype_num_combs(v->hap);
    if (H <= 0) return LTR_ERR_INVALID;
    std::ostringstream o;
    o.precision(2);
    o.setf(std::ios::fixed, std::ios::floatfield);             // :898-899

    int32_t pos = 0;
    std::vector<std::string> alleles; std::vector<bool> inexact;
    get_alleles(v, &pos, &alleles, &inexact);
    const int V = (int)alleles.size();
    std::vector<int> bp_diffs;                                 // :906-913
    for (int i = 0; i < V; ++i)
      bp_diffs.push_back(alleles[(size_t)i] == "<DEL>" ? -(int)alleles[0].size() : (int)alleles[(size_t)i].size() - (int)alleles[0].size());

    // genotypes and likelihoods (:916-927) -- ltr_extract_genotypes is extract_genotypes_and_likelihoods
    std::vector<int32_t> h2a((size_t)H);
    int rc = ltr_haps_to_alleles(v->hap, v->block, h2a.data());
    if (rc != LTR_OK) return rc;
    const int n_gl = haploid ? V : V * (V + 1) / 2, n_pgl = haploid ? V : V * V;
    std::vector<int32_t> gts((size_t)2 * S), pls((size_t)S * n_gl);
    std::vector<double> lphased((size_t)S), lunphased((size_t)S), hphased((size_t)S), hunphased((size_t)S), gldiff((size_t)S),
        gls((size_t)S * n_gl), pgls((size_t)S * n_pgl);
    ltr_genotype_fields gf;
    gf.best_gts = gts.data(); gf.log_phased_posteriors = lphased.data(); gf.log_unphased_posteriors = lunphased.data();
    gf.hap_log_phased_posteriors = hphased.data(); gf.hap_log_unphased_posteriors = hunphased.data();
    gf.gls = gls.data(); gf.gl_diffs = gldiff.data(); gf.pls = opt.output_pls ? pls.data() : nullptr;
    gf.phased_gls = opt.output_phased_gls ? pgls.data() : nullptr;
    rc = ltr_extract_genotypes(S, (int32_t)H, V, h2a.data(), haploid ? 1 : 0, v->log_sample_posteriors, v->sample_total_ll, v->best_haplotypes, &gf);
    if (rc != LTR_OK) return rc;

    // per-read bookkeeping (:929-1043), long path: no traces
    std::vector<int> n_aligned((size_t)S, 0), n_snp((size_t)S, 0), n_flank((size_t)S, 0), n_s1((size_t)S, 0), n_s2((size_t)S, 0);
    std::vector<std::vector<int>> bps((size_t)S), ml_bps((size_t)S);
    std::vector<std::vector<double>> phases((size_t)S);
    for (int r = 0; r < R; ++r) {
      const int s = v->sample_label[r];
      if (s < 0 || s >= S) return LTR_ERR_INVALID;
      const double* ll = v->log_aln_probs + (size_t)r * H;
      const int ha = v->best_haplotypes[2 * s], hb = v->best_haplotypes[2 * s + 1];
      if (ha < 0 || ha >= H || hb < 0 || hb >= H) return LTR_ERR_INVALID;
      const double tot = std::log(std::exp(ll[ha] + v->log_p1[r] + kLogOneHalf) + std::exp(ll[hb] + v->log_p2[r] + kLogOneHalf));   // :958
      phases[(size_t)s].push_back(kLogOneHalf + v->log_p1[r] + ll[ha] - tot);
      int strand = 0;
      if (!haploid && ha != hb) strand = (v->log_p1[r] + ll[ha] > v->log_p2[r] + ll[hb]) ? 0 : 1;          // :965-967
      const int best_hap = strand == 0 ? ha : hb;
      n_aligned[(size_t)s]++;
      if (std::fabs(v->log_p1[r] - v->log_p2[r]) > kTolerance) {                                           // :1006-1012
        n_snp[(size_t)s]++;
        if (v->log_p1[r] > v->log_p2[r]) n_s1[(size_t)s]++; else n_s2[(size_t)s]++;
      }
      if (v->alns) {                                                                                       // :1015-1022
        if (v->aln_deleted && v->aln_deleted[r]) bps[(size_t)s].push_back(-(int)alleles[0].size());
        else { int d = 0; if (extract_cigar(v->alns[r], v->region_start - 5, v->region_stop + 5, &d)) bps[(size_t)s].push_back(d); }
      }
      ml_bps[(size_t)s].push_back(bp_diffs[(size_t)h2a[(size_t)best_hap]]);                               // :1038-1040
    }

    // allele counts over the requested samples (:1045-1071); the requested samples are looked up by name
    std::map<std::string, int> sample_index;
    for (int s = 0; s < S; ++s) sample_index[v->sample_names[s]] = s;
    const int n_out = v->n_out_samples > 0 ? v->n_out_samples : S;
    auto out_name = [&](int i) { return std::string(v->n_out_samples > 0 ? v->out_sample_names[i] : v->sample_names[i]); };
    std::set<std::string> wanted;
    for (int i = 0; i < n_out; ++i) wanted.insert(out_name(i));
    auto filtered = [&](int s) { return v->sample_filter && v->sample_filter[s] && v->sample_filter[s][0] != '\0'; };
    std::vector<int> counts((size_t)V, 0);
    int skip = 0, filt = 0, an = 0;
    for (int s = 0; s < S; ++s) {
      if (!wanted.count(v->sample_names[s]) || n_aligned[(size_t)s] == 0) continue;
      if (n_flank[(size_t)s] > opt.max_flank_indel_frac * n_aligned[(size_t)s]) { ++filt; continue; }
      if (!filtered(s)) {
        if (haploid) { counts[(size_t)gts[2 * (size_t)s]]++; an += 1; }
        else { counts[(size_t)gts[2 * (size_t)s]]++; counts[(size_t)gts[2 * (size_t)s + 1]]++; an += 2; }
      } else ++skip;
    }
    std::vector<int> o2n, n2o;
    reorder(alleles, &o2n, &n2o);
    std::string inexact_seq = V == 1 ? "." : (inexact[(size_t)n2o[1]] ? "1" : "0");                       // :1084-1090
    for (int i = 2; i < V; ++i) { inexact_seq += ","; inexact_seq += inexact[(size_t)n2o[(size_t)i]] ? "1" : "0"; }

    o << v->chrom << "\t" << pos << "\t" << ((v->name && v->name[0]) ? v->name : ".");                  // :1093
    o << "\t" << alleles[(size_t)n2o[0]] << "\t";
    if (V == 1) o << ".";
    else { for (int i = 1; i < V - 1; ++i) o << alleles[(size_t)n2o[(size_t)i]] << ","; o << alleles[(size_t)n2o.back()]; }
    o << "\t.\t.";
    o << "\t" << "START=" << v->region_start + 1 << ";" << "END=" << v->region_stop << ";" << "MOTIF=" << (v->motif ? v->motif : "") << ";"
      << "PERIOD=" << (v->period_str ? v->period_str : "") << ";" << "NSKIP=" << skip << ";" << "NFILT=" << filt << ";"
      << "INEXACT_ALLELE=" << inexact_seq << ";";
    if (V > 1) { o << "BPDIFFS=" << bp_diffs[(size_t)n2o[1]]; for (int i = 2; i < V; ++i) o << "," << bp_diffs[(size_t)n2o[(size_t)i]]; o << ";"; }
    int tot_dp = 0, tot_dsnp = 0, tot_dfl = 0;                  // :1135-1153
    for (int i = 0; i < n_out; ++i) {
      auto it = sample_index.find(out_name(i));
      if (it == sample_index.end() || filtered(it->second)) continue;
      const int s = it->second;
      if (n_aligned[(size_t)s] > 0 && n_flank[(size_t)s] > n_aligned[(size_t)s] * opt.max_flank_indel_frac) continue;
      tot_dp += n_aligned[(size_t)s]; tot_dsnp += n_snp[(size_t)s]; tot_dfl += n_flank[(size_t)s];
    }
    o << "DP=" << tot_dp << ";" << "DSNP=" << tot_dsnp << ";" << "DFLANKINDEL=" << tot_dfl << ";";
    o << "AN=" << an << ";" << "REFAC=" << counts[0];
    if (V > 1) { o << ";AC="; for (int i = 1; i < V - 1; ++i) o << counts[(size_t)n2o[(size_t)i]] << ","; o << counts[(size_t)n2o.back()]; }

    int num_fields;                                             // :1170-1197
    if (!haploid) { o << "\tGT:GB:Q:PQ:DP:DSNP:DFLANKINDEL:PDP:PSNP:GLDIFF"; num_fields = 10; }
    else { o << "\tGT:GB:Q:DP:DFLANKINDEL:GLDIFF"; num_fields = 6; }
    if (opt.output_allreads) o << ":ALLREADS";
    if (opt.output_mallreads) o << ":MALLREADS";
    if (opt.output_gls) o << ":GL";
    if (opt.output_pls) o << ":PL";
    if (!haploid && opt.output_phased_gls) o << ":PHASEDGL";
    if (opt.output_haplotype_data) o << ":HQ:PHQ";
    if (opt.output_filters) o << ":FILTER";
    num_fields += (!haploid && opt.output_phased_gls) ? 1 : 0;
    num_fields += (opt.output_allreads ? 1 : 0) + (opt.output_mallreads ? 1 : 0) + (opt.output_gls ? 1 : 0) + (opt.output_pls ? 1 : 0) +
                  2 * (opt.output_haplotype_data ? 1 : 0);
    std::string empty;
    for (int k = 0; k < num_fields; ++k) empty += ".:";

    for (int i = 0; i < n_out; ++i) {                           // :1201-1366
      o << "\t";
      auto it = sample_index.find(out_name(i));
      if (it == sample_index.end() || n_aligned[(size_t)it->second] == 0) { o << (opt.output_filters ? empty + "NO_READS" : std::string(".")); continue; }
      const int s = it->second;
      if (filtered(s)) { o << (opt.output_filters ? empty + v->sample_filter[s] : std::string(".")); continue; }
      if (n_flank[(size_t)s] > n_aligned[(size_t)s] * opt.max_flank_indel_frac) { o << (opt.output_filters ? empty + "FLANK_INDEL_FRAC" : std::string(".")); continue; }
      const int g1 = gts[2 * (size_t)s], g2 = gts[2 * (size_t)s + 1];
      if (!haploid) {
        o << o2n[(size_t)g1] << "|" << o2n[(size_t)g2] << ":" << bp_diffs[(size_t)g1] << "|" << bp_diffs[(size_t)g2]
          << ":" << std::exp(lunphased[(size_t)s]) << ":" << std::exp(lphased[(size_t)s])
          << ":" << n_aligned[(size_t)s] << ":" << n_snp[(size_t)s] << ":" << n_flank[(size_t)s]
          << ":" << (v->n_p1s ? v->n_p1s[s] : 0) << "|" << (v->n_p2s ? v->n_p2s[s] : 0)
          << ":" << n_s1[(size_t)s] << "|" << n_s2[(size_t)s];
      } else {
        o << o2n[(size_t)g1] << ":" << bp_diffs[(size_t)g1] << ":" << std::exp(lunphased[(size_t)s]) << ":" << n_aligned[(size_t)s] << ":" << n_flank[(size_t)s];
      }
      if (V == 1) o << ":" << "."; else o << ":" << gldiff[(size_t)s];
      if (opt.output_allreads) o << ":" << condense(bps[(size_t)s]);
      if (opt.output_mallreads) o << ":" << condense(ml_bps[(size_t)s]);
      const double* gl = gls.data() + (size_t)s * n_gl;
      const int32_t* pl = pls.data() + (size_t)s * n_gl;
      if (haploid) {
        if (opt.output_gls) { o << ":" << gl[0]; for (int a = 1; a < V; ++a) o << "," << gl[n2o[(size_t)a]]; }
        if (opt.output_pls) { o << ":" << pl[0]; for (int a = 1; a < V; ++a) o << "," << pl[n2o[(size_t)a]]; }
      } else {
        auto tri = [&](int a, int b) { const int lo = std::min(n2o[(size_t)a], n2o[(size_t)b]), hi = std::max(n2o[(size_t)a], n2o[(size_t)b]); return hi * (hi + 1) / 2 + lo; };
        if (opt.output_gls) { o << ":" << gl[0]; for (int a = 1; a < V; ++a) for (int b = 0; b <= a; ++b) o << "," << gl[tri(a, b)]; }
        if (opt.output_pls) { o << ":" << pl[0]; for (int a = 1; a < V; ++a) for (int b = 0; b <= a; ++b) o << "," << pl[tri(a, b)]; }
        if (opt.output_phased_gls) {
          const double* pg = pgls.data() + (size_t)s * n_pgl;
          o << ":" << pg[0];
          for (int a = 0; a < V; ++a) for (int b = 0; b < V; ++b) { if (a == 0 && b == 0) continue; o << "," << pg[n2o[(size_t)a] * V + n2o[(size_t)b]]; }
        }
      }
      if (opt.output_haplotype_data) o << ":" << std::exp(hunphased[(size_t)s]) << ":" << std::exp(hphased[(size_t)s]);
      if (opt.output_filters) o << ":PASS";
    }
    const std::string text = o.str();
    if ((int64_t)text.size() + 1 > cap) return LTR_ERR_INVALID;
    std::memcpy(out, text.data(), text.size());
    out[text.size()] = '\0';
    if (pos_out) *pos_out = pos;
    (void)phases; (void)log_sum_exp;                            // (phase1/phase2 read counts feed only the disabled allele-bias fields, :1232-1233)
    return (int64_t)text.size();
  } catch (const std::bad_alloc&) { return LTR_ERR_NOMEM; } catch (...) { return LTR_ERR_INVALID; }
}

// Genotyper::get_vcf_header (genotyper.cpp:258-336): file format, command, reference, the FASTA's ##contig lines
// (FastaReader::write_all_contigs_to_vcf = ltr_fasta_contig_lines), the INFO and FORMAT definitions of the fields
// ltr_vcf_record writes, and the #CHROM line with the sample names.  Table-driven: one row per field, in the
// reference's order, with its texts (typos included: a consumer may match on them).
int64_t ltr_vcf_header(const char* fasta_path, const char* full_command, const char* contig_lines, const ltr_vcf_options* opt_in,
                       const char* const* sample_names, int32_t n_samples, char* out, int64_t cap) {
  if (!fasta_path || !full_command || n_samples < 0 || (n_samples > 0 && !sample_names) || !out || cap <= 0) return LTR_ERR_INVALID;
  ltr_vcf_options opt;
  if (opt_in) opt = *opt_in; else ltr_default_vcf_options(&opt);
  struct Field { const char* id; const char* number; const char* type; const char* desc; int on; };
  const Field info[] = {
    {"START", "1", "Integer", "Inclusive start coodinate for the repetitive portion of the reference allele", 1},
    {"END", "1", "Integer", "Inclusive end coordinate for the repetitive portion of the reference allele", 1},
    {"MOTIF", ".", "String", "TR motif(s)", 1},
    {"PERIOD", ".", "Integer", "Length of TR motif(s)", 1},
    {"NSKIP", "1", "Integer", "Number of samples not genotyped due to various issues", 1},
    {"NFILT", "1", "Integer", "Number of samples whose genotypes were filtered due to various issues", 1},
    {"INEXACT_ALLELE", "A", "Integer", "Boolean showing if each alternate allele is exact or approximated by POA, 0 for exact 1 for approximated.", 1},
    {"BPDIFFS", "A", "Integer", "Base pair difference of each alternate allele from the reference allele", 1},
    {"DP", "1", "Integer", "Total number of valid reads used to genotype all samples", 1},
    {"DSNP", "1", "Integer", "Total number of reads with SNP phasing information", 1},
    {"DFLANKINDEL", "1", "Integer", "Total number of reads with an indel in the regions flanking the STR", 1},
    {"AN", "1", "Integer", "Total number of alleles in called genotypes", 1},
    {"REFAC", "1", "Integer", "Reference allele count", 1},
    {"AC", "A", "Integer", "Alternate allele counts", 1},
  };
  const Field format[] = {
    {"GT", "1", "String", "Genotype", 1},
    {"GB", "1", "String", "Base pair differences of genotype from reference", 1},
    {"Q", "1", "Float", "Posterior probability of unphased genotype", 1},
    {"PQ", "1", "Float", "Posterior probability of phased genotype", 1},
    {"DP", "1", "Integer", "Number of valid reads used for sample's genotype", 1},
    {"DSNP", "1", "Integer", "Number of reads with SNP phasing information", 1},
    {"PSNP", "1", "String", "Number of reads with SNPs supporting each haploid genotype", 1},
    {"PDP", "1", "String", "Fractional reads supporting each haploid genotype", 1},
    {"GLDIFF", "1", "Float", "Difference in likelihood between the reported and next best genotypes", 1},
    {"HQ", "1", "Float", "Posterior probability of unphased haplotypes", opt.output_haplotype_data == 1},
    {"PHQ", "1", "Float", "Posterior probability of phased haplotypes", opt.output_haplotype_data == 1},
    {"ALLREADS", "1", "String", "Base pair difference observed in each read's Needleman-Wunsch alignment", opt.output_allreads == 1},
    {"MALLREADS", "1", "String", "Maximum likelihood bp diff in each read based on haplotype alignments for reads that span the repeat region by at least 5 base pairs", opt.output_mallreads == 1},
    {"GL", "G", "Float", "log10 genotype likelihoods", opt.output_gls == 1},
    {"PL", "G", "Integer", "Phred-scaled genotype likelihoods", opt.output_pls == 1},
    {"PHASEDGL", ".", "Float", "log10 genotype likelihood for each phased genotype. Value for phased genotype X|Y is stored at a 0-based index of X*A + Y, where A is the number of alleles. Not applicable to haploid genotypes", opt.output_phased_gls == 1},
    {"FILTER", "1", "String", "Reason for filtering the current call, or PASS if the call was not filtered", opt.output_filters == 1},
  };
  try {
    std::string h = "##fileformat=VCFv4.1\n";
    h += "##command="; h += full_command; h += "\n##reference="; h += fasta_path; h += "\n";
    if (contig_lines) h += contig_lines;
    auto rows = [&](const char* kind, const Field* f, size_t n) {
      for (size_t i = 0; i < n; ++i) {
        if (!f[i].on) continue;
        h += "##"; h += kind; h += "=<ID="; h += f[i].id; h += ",Number="; h += f[i].number; h += ",Type="; h += f[i].type;
        h += ",Description=\""; h += f[i].desc; h += "\">\n";
      }
    };
    rows("INFO", info, sizeof(info) / sizeof(info[0]));
    rows("FORMAT", format, sizeof(format) / sizeof(format[0]));
    h += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT";
    for (int32_t i = 0; i < n_samples; ++i) { if (!sample_names[i]) return LTR_ERR_INVALID; h += "\t"; h += sample_names[i]; }
    h += "\n";
    if ((int64_t)h.size() + 1 > cap) return LTR_ERR_INVALID;
    std::memcpy(out, h.c_str(), h.size() + 1);
    return (int64_t)h.size();
  } catch (...) { return LTR_ERR_NOMEM; }
}

}  // extern "C"

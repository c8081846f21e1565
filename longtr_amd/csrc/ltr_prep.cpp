// ltr_prep.cpp -- raw-read preparation and candidate haplotypes: what happens to a
// locus' reads between the BAM record and the alignment DP, so that raw long reads (8-50 kb, any CIGAR)
// enter this library the way they enter the reference's genotyper.  Host code; citations are to the LongTR
// reference:
//   BamAlignment::TrimAlignment                 src/bam_io.cpp:267-372      (ltr_left_align_reads, step 1)
//   GenotyperBamProcessor::left_align_reads     src/genotyper_bam_processor.cpp:38-168
//   HaplotypeGenerator::extract_sequence        src/SeqAlignment/HaplotypeGenerator.cpp:84-165
//   HaplotypeGenerator::gen_candidate_seqs      :295-373 (exact alleles) + :474-480 (sort, trim)
//   HaplotypeGenerator::trim                    :14-82
//   HaplotypeGenerator::add_haplotype_block     :530-578
//   HaplotypeGenerator::fuse_haplotype_blocks   :580-607
//   SeqStutterGenotyper::build_haplotype        src/seq_stutter_genotyper.cpp:416-482
// NOT here: the partial-order-alignment clustering branch of gen_candidate_seqs (:376-472: spoa, an
// un-vendored dependency, with std::random_device subsampling) -- ltr_build_haplotype reports how many reads
// the reference would have handed to it.

#include <algorithm>
#include <climits>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ltr_internal.h"

namespace {

const int kFlankSize = 200;                       // FLANK_SIZE, bam_io.h:28
char upper(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }

struct Cig { char t; int32_t n; };

struct PreparedRead {                             // class Alignment (AlignmentData.h:28-140), the fields the path reads
  int32_t start = 0, stop = 0;
  bool deleted = false, use_for_hap_gen = true;
  std::string seq, qual, aln;                     // sequence_, base_qualities_, alignment_ (read bases, '-' for deleted reference bases)
  std::vector<Cig> cigar;
  int32_t source = -1, sample = 0;
  std::string ctype; std::vector<int32_t> cnum;   // flattened CIGAR for the C view
};

// chrom_seq[pos] of a window that starts at coordinate `off`; '\0' outside (never equal to a read base)
struct Chrom {
  const uint8_t* p; int64_t off, len;
  char at(int64_t pos) const { const int64_t k = pos - off; return (k < 0 || k >= len) ? '\0' : (char)p[k]; }
  std::string sub_upper(int64_t pos, int64_t n) const {
    std::string s;
    for (int64_t q = pos; q < pos + n; ++q) { const char c = at(q); if (!c) break; s.push_back(upper(c)); }
    return s;
  }
};

// BamAlignment::TrimAlignment (bam_io.cpp:267-372).  Returns false on a CIGAR operation it does not know.
bool trim_alignment(std::vector<Cig>& cigar, int32_t& pos, int32_t& end_pos, std::string& bases, std::string& quals,
                    int32_t min_read_start, int32_t max_read_stop, bool& deleted) {
  int ltrim = 0;
  int32_t start_pos = pos;
  size_t fi = 0;
  auto front_pop = [&](std::vector<Cig>& c, size_t& i) { if (c[i].n == 1) ++i; else --c[i].n; };
  while (start_pos < min_read_start && fi < cigar.size()) {    // before the boundary, :274-299
    switch (cigar[fi].t) {
      case 'M': case '=': case 'X': ++ltrim; ++start_pos; break;
      case 'D': ++start_pos; break;
      case 'I': case 'S': ++ltrim; break;
      case 'H': break;
      default: return false;
    }
    front_pop(cigar, fi);
  }
  cigar.erase(cigar.begin(), cigar.begin() + (long)fi);
  // from the boundary to the repeat's end: is the whole repeat deleted?  (:302-339)
  int32_t rp = start_pos;
  const int32_t repeat_start = min_read_start + kFlankSize, repeat_end = max_read_stop - kFlankSize;
  int32_t deletion = 0;
  {
    std::vector<Cig> tmp = cigar;
    size_t ti = 0;
    while (rp >= min_read_start && rp < repeat_end && ti < tmp.size()) {
      switch (tmp[ti].t) {
        case 'M': case '=': case 'X': ++rp; break;
        case 'D': if (rp >= repeat_start) ++deletion; ++rp; break;
        case 'I': case 'S': case 'H': break;
        default: return false;
      }
      front_pop(tmp, ti);
    }
  }
  if (deletion >= repeat_end - repeat_start) deleted = true;
  int rtrim = 0;
  int32_t ep = end_pos;
  while (ep > max_read_stop && !cigar.empty()) {               // after the boundary, :342-366
    switch (cigar.back().t) {
      case 'M': case '=': case 'X': ++rtrim; --ep; break;
      case 'D': --ep; break;
      case 'I': case 'S': ++rtrim; break;
      case 'H': break;
      default: return false;
    }
    if (cigar.back().n == 1) cigar.pop_back(); else --cigar.back().n;
  }
  if ((size_t)(ltrim + rtrim) > bases.size()) return false;    // assert, :368
  bases = bases.substr((size_t)ltrim, bases.size() - (size_t)ltrim - (size_t)rtrim);
  if (!quals.empty()) quals = quals.substr((size_t)ltrim, quals.size() - (size_t)ltrim - (size_t)rtrim);
  pos = start_pos; end_pos = ep;
  return true;
}

// HaplotypeGenerator::extract_sequence (:84-165): the read's bases between two reference coordinates
bool extract_sequence(const PreparedRead& a, int32_t region_start, int32_t region_end, std::string& seq, bool* bad) {
  if (a.deleted) { seq = ""; return true; }
  if (a.start >= region_start) return false;
  if (a.stop <= region_end) return false;
  size_t ai = 0;                                                // index into the alignment string
  int32_t ci = 0, pos = a.start;                                // index inside the current CIGAR element
  size_t it = 0;
  std::string reg;
  auto done = [&]() { seq = reg; for (char& c : seq) c = upper(c); return true; };
  while (it < a.cigar.size()) {
    const Cig& e = a.cigar[it];
    if (ci == e.n) { ++it; ci = 0; }
    else if (pos > region_end) return done();
    else if (pos == region_end) {
      if (e.t == 'I') { reg += a.aln.substr(std::min(ai, a.aln.size()), (size_t)e.n); ai += (size_t)e.n; ci = 0; ++it; }
      else return done();
    } else if (pos >= region_start) {
      int32_t nb = std::min(region_end - pos, e.n - ci);
      switch (e.t) {
        case 'I': nb = e.n; reg += a.aln.substr(std::min(ai, a.aln.size()), (size_t)nb); break;
        case '=': case 'X': case 'M': reg += a.aln.substr(std::min(ai, a.aln.size()), (size_t)nb); pos += nb; break;
        case 'D': pos += nb; break;
        default: *bad = true; return false;                    // printErrorAndDie, :141
      }
      ai += (size_t)nb; ci += nb;
    } else {
      int32_t nb;
      if (e.t == 'I') nb = e.n - ci;
      else { nb = std::min(region_start - pos, e.n - ci); pos += nb; }
      ai += (size_t)nb; ci += nb;
    }
  }
  *bad = true;                                                  // "Logical error in extract_sequence", :163
  return false;
}

bool by_len_seq(const std::string& x, const std::string& y) { return x.size() != y.size() ? x.size() < y.size() : x.compare(y) < 0; }

// HaplotypeGenerator::trim (:14-82): clip what all candidates share at either end, down to ideal_min_length
void trim_candidates(int ideal_min_length, int left_pad, int right_pad, int32_t& region_start, int32_t& region_end, std::vector<std::string>& seqs) {
  int min_len = INT_MAX;
  for (const std::string& s : seqs) min_len = std::min(min_len, (int)s.size());
  if (min_len <= ideal_min_length) return;
  int max_left = 0, max_right = 0;
  while (max_left < min_len - ideal_min_length) {
    size_t j = 1;
    while (j < seqs.size() && seqs[j][(size_t)max_left] == seqs[j - 1][(size_t)max_left]) ++j;
    if (j != seqs.size()) break;
    ++max_left;
  }
  while (max_right < min_len - ideal_min_length) {
    const char c = seqs[0][seqs[0].size() - 1 - (size_t)max_right];
    size_t j = 1;
    while (j < seqs.size() && seqs[j][seqs[j].size() - 1 - (size_t)max_right] == c) ++j;
    if (j != seqs.size()) break;
    ++max_right;
  }
  max_left = std::min(left_pad, max_left);                      // :47-52
  max_right = std::min(right_pad, max_right);
  max_left = std::max(0, std::min(min_len - right_pad, max_left));
  max_right = std::max(0, std::min(min_len - left_pad, max_right));
  int lt, rt;
  if (min_len - 2 * std::min(max_left, max_right) <= ideal_min_length) {   // :57-65
    lt = rt = std::min(max_left, max_right);
    while (min_len - lt - rt < ideal_min_length) { if (lt > rt) --lt; else --rt; }
  } else if (max_left > max_right) { rt = max_right; lt = std::min(max_left, min_len - ideal_min_length - max_right); }
  else { lt = max_left; rt = std::min(max_right, min_len - ideal_min_length - max_left); }
  for (std::string& s : seqs) s = s.substr((size_t)lt, s.size() - (size_t)lt - (size_t)rt);
  region_start += lt; region_end -= rt;
}

}  // namespace

struct ltr_read_set {
  std::vector<PreparedRead> reads;
  std::vector<ltr_alignment> view;
  std::vector<const char*> aln_strings;
  std::vector<uint8_t> deleted;
  std::vector<int32_t> source, sample, n_p1s, n_p2s;
  int32_t fail_count = 0;
};

struct ltr_hap_result {
  std::vector<int32_t> bstart, bend, period, nall;
  std::vector<uint8_t> is_rep, bytes, inexact;
  std::vector<int64_t> off;
  ltr_haplotype_blocks view;
  std::string failure;
  int32_t unplaced = 0, needs_clustering = 0;
};

extern "C" {

// GenotyperBamProcessor::left_align_reads (genotyper_bam_processor.cpp:38-168) for one locus.
int ltr_left_align_reads(const ltr_raw_alignment* raw, int32_t n_raw, int32_t n_samples, int32_t region_start, int32_t region_stop,
                         const uint8_t* chrom_seq, int64_t chrom_seq_start, int64_t chrom_seq_len, ltr_read_set** out) {
  if ((!raw && n_raw > 0) || n_raw < 0 || n_samples <= 0 || !chrom_seq || !out || region_stop < region_start) return LTR_ERR_INVALID;
  *out = nullptr;
  try {
    const Chrom chrom{chrom_seq, chrom_seq_start, chrom_seq_len};
    std::unique_ptr<ltr_read_set> owner(new ltr_read_set());     // released into *out on the LTR_OK path only
    ltr_read_set* rs = owner.get();
    rs->n_p1s.assign((size_t)n_samples, 0); rs->n_p2s.assign((size_t)n_samples, 0);
    for (int32_t k = 0; k < n_raw; ++k) {
      const ltr_raw_alignment& r = raw[k];
      if (r.sample < 0 || r.sample >= n_samples || r.length < 0 || r.n_cigar < 0 || (r.length > 0 && !r.bases) ||
          (r.n_cigar > 0 && (!r.cigar_type || !r.cigar_num))) { return LTR_ERR_INVALID; }
      if (r.pos > region_start || r.end_pos < region_stop) { rs->fail_count++; continue; }            // :56-59 not spanning
      std::vector<Cig> cigar;
      for (int32_t c = 0; c < r.n_cigar; ++c) { if (r.cigar_num[c] < 1) { return LTR_ERR_CIGAR; } cigar.push_back({r.cigar_type[c], r.cigar_num[c]}); }
      std::string bases((const char*)r.bases, (size_t)r.length), quals = r.quals ? std::string((const char*)r.quals, (size_t)r.length) : std::string();
      int32_t pos = r.pos, end_pos = r.end_pos;
      bool deleted = false;
      if (!trim_alignment(cigar, pos, end_pos, bases, quals, region_start > kFlankSize ? region_start - kFlankSize : 1, region_stop + kFlankSize, deleted)) {
        return LTR_ERR_CIGAR;                                   // printErrorAndDie("Invalid CIGAR option encountered in TrimAlignment")
      }
      PreparedRead pr;
      pr.source = k; pr.sample = r.sample; pr.use_for_hap_gen = r.use_for_hap_generation != 0;
      if (bases.empty()) {                                      // :62-71 the repeat is deleted in this read
        pr.start = region_start; pr.stop = region_stop; pr.deleted = true; pr.use_for_hap_gen = true;
        rs->reads.push_back(pr);
        continue;
      }
      pr.start = pos; pr.stop = end_pos - 1; pr.deleted = deleted; pr.qual = quals;
      for (char& c : bases) c = upper(c);
      pr.seq = bases;
      size_t si = 0; int32_t ri = pos;
      bool soft = false;
      auto add = [&](char t, int32_t n) { pr.cigar.push_back({t, n}); };
      for (const Cig& e : cigar) {                              // :80-136: M/=/X re-derived against the reference
        switch (e.t) {
          case 'H': break;
          case 'S': add('S', e.n); si += (size_t)e.n; soft = true; break;
          case 'I': add('I', e.n); pr.aln += bases.substr(std::min(si, bases.size()), (size_t)e.n); si += (size_t)e.n; break;
          case 'D': add('D', e.n); pr.aln += std::string((size_t)e.n, '-'); ri += e.n; break;
          case 'M': case '=': case 'X': {
            char prev = '='; int32_t num = 0;
            for (int32_t c = 0; c < e.n; ++c, ++ri, ++si) {
              const char b = si < bases.size() ? bases[si] : '\0';
              const char t = (b == upper(chrom.at(ri))) ? '=' : 'X';
              if (t == prev) ++num; else { if (num) add(prev, num); prev = t; num = 1; }
              pr.aln.push_back(b);
            }
            if (num) add(prev, num);
            break;
          }
          default: return LTR_ERR_CIGAR;                        // "Invalid CIGAR option encountered in convertAlignment"
        }
      }
      if (soft) { rs->fail_count++; continue; }                 // :137-140
      int64_t qlen = 0;                                         // check_CIGAR_string (AlignmentData.h:77-90)
      for (const Cig& e : pr.cigar) if (e.t != 'D' && e.t != 'H') qlen += e.n;
      if (qlen != (int64_t)pr.seq.size()) { return LTR_ERR_CIGAR; }
      if (r.haplotype_tag == 1) rs->n_p1s[(size_t)r.sample]++;  // :145-150
      if (r.haplotype_tag == 2) rs->n_p2s[(size_t)r.sample]++;
      rs->reads.push_back(pr);
    }
    for (PreparedRead& p : rs->reads) { for (const Cig& e : p.cigar) { p.ctype.push_back(e.t); p.cnum.push_back(e.n); } if (p.cnum.empty()) p.cnum.push_back(0); }
    for (PreparedRead& p : rs->reads) {
      ltr_alignment a;
      a.start = p.start; a.stop = p.stop; a.seq = (const uint8_t*)p.seq.data(); a.seq_len = (int32_t)p.seq.size();
      a.n_cigar = (int32_t)p.cigar.size(); a.cigar_type = p.ctype.c_str(); a.cigar_num = p.cnum.data();
      a.qual = p.qual.empty() ? nullptr : (const uint8_t*)p.qual.data();
      rs->view.push_back(a); rs->aln_strings.push_back(p.aln.c_str()); rs->deleted.push_back(p.deleted ? 1 : 0);
      rs->source.push_back(p.source); rs->sample.push_back(p.sample);
    }
    *out = owner.release();
    return LTR_OK;
  } catch (const std::bad_alloc&) { return LTR_ERR_NOMEM; } catch (...) { return LTR_ERR_INVALID; }
}

int32_t ltr_read_set_size(const ltr_read_set* rs) { return rs ? (int32_t)rs->reads.size() : 0; }
const ltr_alignment* ltr_read_set_alignments(const ltr_read_set* rs) { return rs && !rs->view.empty() ? rs->view.data() : nullptr; }
const char* const* ltr_read_set_alignment_strings(const ltr_read_set* rs) { return rs && !rs->aln_strings.empty() ? rs->aln_strings.data() : nullptr; }
const uint8_t* ltr_read_set_deleted(const ltr_read_set* rs) { return rs && !rs->deleted.empty() ? rs->deleted.data() : nullptr; }
const int32_t* ltr_read_set_source(const ltr_read_set* rs) { return rs && !rs->source.empty() ? rs->source.data() : nullptr; }
const int32_t* ltr_read_set_sample(const ltr_read_set* rs) { return rs && !rs->sample.empty() ? rs->sample.data() : nullptr; }
const int32_t* ltr_read_set_n_p1s(const ltr_read_set* rs) { return rs ? rs->n_p1s.data() : nullptr; }
const int32_t* ltr_read_set_n_p2s(const ltr_read_set* rs) { return rs ? rs->n_p2s.data() : nullptr; }
int32_t ltr_read_set_fail_count(const ltr_read_set* rs) { return rs ? rs->fail_count : 0; }
void ltr_read_set_free(ltr_read_set* rs) { delete rs; }

// HaplotypeGenerator::extract_sequence for read i of a set: length of the sequence (>= 0), -1 "does not span",
// < -1 a status.  out may be NULL to ask for the length only.
int64_t ltr_extract_sequence(const ltr_read_set* rs, int32_t i, int32_t region_start, int32_t region_end, uint8_t* out, int64_t cap) {
  if (!rs || i < 0 || i >= (int32_t)rs->reads.size()) return LTR_ERR_INVALID * 2;
  try {
    std::string s; bool bad = false;
    if (!extract_sequence(rs->reads[(size_t)i], region_start, region_end, s, &bad)) return bad ? LTR_ERR_CIGAR * 2 : -1;
    if (out) { if ((int64_t)s.size() > cap) return LTR_ERR_INVALID * 2; std::memcpy(out, s.data(), s.size()); }
    return (int64_t)s.size();
  } catch (...) { return LTR_ERR_NOMEM * 2; }
}

// SeqStutterGenotyper::build_haplotype (seq_stutter_genotyper.cpp:416-482) for one region, alleles from the reads:
// add_haplotype_block (gen_candidate_seqs' exact-allele rules, trim) + fuse_haplotype_blocks -> three blocks
// [reference flank][repeat block with candidate alleles][reference flank].
int ltr_build_haplotype(ltr_ctx* ctx, const ltr_read_set* rs, int32_t n_samples, int32_t region_start, int32_t region_stop, int32_t period,
                        const uint8_t* chrom_seq, int64_t chrom_seq_start, int64_t chrom_seq_len, int64_t chrom_len,
                        int32_t indel_flank_len, ltr_hap_result** out) {
  if (!rs || n_samples <= 0 || !chrom_seq || !out || period < 1 || indel_flank_len < 0) return LTR_ERR_INVALID;
  *out = nullptr;
  ltr::TimedCall timed(ctx, ltr::kTimerHapBuild);              // total_hap_build_time_ (:417, :479-480); ctx may be NULL
  try {
    const Chrom chrom{chrom_seq, chrom_seq_start, chrom_seq_len};
    const int kRefFlank = 35, kMinFracReads_x100 = 5;          // HaplotypeGenerator.h:64-75
    const double MIN_FRAC_READS = 0.05, MIN_FRAC_SAMPLES = 0.05, MIN_FRAC_STRONG_SAMPLE = 0.2, MIN_READS_STRONG_SAMPLE = 2, MIN_STRONG_SAMPLES = 1;
    (void)kMinFracReads_x100;
    const int LEFT_PAD = indel_flank_len, RIGHT_PAD = indel_flank_len;
    std::unique_ptr<ltr_hap_result> owner(new ltr_hap_result());   // released into *out on the LTR_OK paths only: an error or an exception leaves *out NULL
    ltr_hap_result* res = owner.get();
    auto fail = [&](const char* msg) { res->failure = msg; std::memset(&res->view, 0, sizeof(res->view)); *out = owner.release(); return LTR_OK; };
    int32_t min_aln_start = INT_MAX, max_aln_stop = INT_MIN;    // :421-426 over ALL reads
    for (const PreparedRead& p : rs->reads) { min_aln_start = std::min(min_aln_start, p.start); max_aln_stop = std::max(max_aln_stop, p.stop); }
    // add_haplotype_block, :530-578
    if (region_start < kRefFlank + LEFT_PAD || (int64_t)region_stop + kRefFlank + RIGHT_PAD > chrom_len) return fail("Haplotype blocks are too near to the chromosome ends");
    int32_t gmin = INT_MAX, gmax = INT_MIN;                     // get_aln_bounds over the reads used for haplotype generation
    for (const PreparedRead& p : rs->reads) if (p.use_for_hap_gen) { gmin = std::min(gmin, p.start); gmax = std::max(gmax, p.stop); }
    int32_t rstart = region_start - LEFT_PAD, rend = region_stop + RIGHT_PAD;
    const std::string ref_seq = chrom.sub_upper(rstart, rend - rstart);
    if ((int64_t)gmin + 5 >= rstart || (int64_t)gmax - 5 <= rend) return fail("No spanning alignments");
    const int ideal_min_length = 3 * period;
    // gen_candidate_seqs, :295-373
    std::map<std::string, double> sample_counts;
    std::map<std::string, int> read_counts, must_inc;
    int tot_reads = 0, tot_samples = 0;
    std::vector<std::vector<std::string>> per_sample((size_t)n_samples);     // extracted sequences per sample (reused below)
    for (const PreparedRead& p : rs->reads) {
      if (!p.use_for_hap_gen) continue;
      if (p.sample < 0 || p.sample >= n_samples) return LTR_ERR_INVALID;
      std::string sub; bool bad = false;
      if (extract_sequence(p, rstart, rend, sub, &bad)) per_sample[(size_t)p.sample].push_back(sub);
      else if (bad) return LTR_ERR_CIGAR;
    }
    for (int s = 0; s < n_samples; ++s) {
      std::map<std::string, int> counts;
      const int samp_reads = (int)per_sample[(size_t)s].size();
      for (const std::string& sub : per_sample[(size_t)s]) { read_counts[sub] += 1; counts[sub] += 1; ++tot_reads; }
      for (const auto& kv : counts) {                           // :318-322
        if (kv.second >= MIN_READS_STRONG_SAMPLE && kv.second >= MIN_FRAC_STRONG_SAMPLE * samp_reads) must_inc[kv.first] += 1;
        sample_counts[kv.first] += kv.second * 1.0 / samp_reads;
      }
      if (samp_reads > 0) ++tot_samples;
    }
    std::vector<std::string> seqs;
    int ref_index = -1;
    for (const auto& kv : must_inc) {                           // :345-356 alleles with strong support in some sample
      if (kv.second >= MIN_STRONG_SAMPLES) {
        sample_counts.erase(kv.first); read_counts.erase(kv.first);
        seqs.push_back(kv.first);
        if (kv.first == ref_seq) ref_index = (int)seqs.size() - 1;
      }
    }
    for (const auto& kv : sample_counts) {                      // :359-365 alleles above the global thresholds
      if (kv.second > MIN_FRAC_SAMPLES * tot_samples * 2 || read_counts[kv.first] > MIN_FRAC_READS * tot_reads * 2) {
        seqs.push_back(kv.first);
        if (ref_index == -1 && kv.first == ref_seq) ref_index = (int)seqs.size() - 1;
      }
    }
    if (ref_index == -1) seqs.insert(seqs.begin(), ref_seq);    // :368-373 reference first
    else { seqs[(size_t)ref_index] = seqs[0]; seqs[0] = ref_seq; }
    // :376-400: reads without a candidate -- the reference clusters them (POA); reported, not done
    for (int s = 0; s < n_samples; ++s) {
      int ignored = 0;
      for (const std::string& sub : per_sample[(size_t)s]) if (std::find(seqs.begin(), seqs.end(), sub) == seqs.end()) ++ignored;
      res->unplaced += ignored;
      if (ignored > (int)per_sample[(size_t)s].size() * 0.25) res->needs_clustering++;
    }
    std::sort(seqs.begin() + 1, seqs.end(), by_len_seq);        // :475
    trim_candidates(ideal_min_length, LEFT_PAD, RIGHT_PAD, rstart, rend, seqs);   // :480
    // fuse_haplotype_blocks, :580-607
    if (rstart < kRefFlank || (int64_t)rend + kRefFlank > chrom_len) return fail("Haplotype blocks are too near to the chromosome ends");
    const int32_t min_start = std::min(rstart - 10, std::max(rstart - kRefFlank, min_aln_start));
    const int32_t max_stop = std::max(rend + 10, std::min(rend + kRefFlank, max_aln_stop));
    const std::string lflank = chrom.sub_upper(min_start, rstart - min_start), rflank = chrom.sub_upper(rend, max_stop - rend);
    res->bstart = {min_start, rstart, rend}; res->bend = {rstart, rend, max_stop};
    res->is_rep = {0, 1, 0}; res->period = {0, period, 0}; res->nall = {1, (int32_t)seqs.size(), 1};
    res->off.push_back(0);
    auto put = [&](const std::string& s) { res->bytes.insert(res->bytes.end(), s.begin(), s.end()); res->off.push_back((int64_t)res->bytes.size()); };
    put(lflank);
    for (const std::string& s : seqs) put(s);
    put(rflank);
    if (res->bytes.empty()) res->bytes.push_back(0);
    res->inexact.assign(seqs.size(), 0);
    res->view.n_blocks = 3; res->view.block_start = res->bstart.data(); res->view.block_end = res->bend.data();
    res->view.is_repeat = res->is_rep.data(); res->view.period = res->period.data(); res->view.n_alleles = res->nall.data();
    res->view.allele_bytes = res->bytes.data(); res->view.allele_off = res->off.data();
    *out = owner.release();
    return LTR_OK;
  } catch (const std::bad_alloc&) { return LTR_ERR_NOMEM; } catch (...) { return LTR_ERR_INVALID; }
}

const ltr_haplotype_blocks* ltr_hap_result_blocks(const ltr_hap_result* r) { return (r && r->failure.empty()) ? &r->view : nullptr; }
const char* ltr_hap_result_failure(const ltr_hap_result* r) { return r ? r->failure.c_str() : "null"; }
int32_t ltr_hap_result_unplaced_reads(const ltr_hap_result* r) { return r ? r->unplaced : 0; }
int32_t ltr_hap_result_samples_needing_clustering(const ltr_hap_result* r) { return r ? r->needs_clustering : 0; }
void ltr_hap_result_free(ltr_hap_result* r) { delete r; }

// SNPBamProcessor::process_phased_reads (snp_bam_processor.cpp:141-226, --phased-bam) for unpaired reads (long reads have no
// mates): per read group in order -- running totals over the groups, and the "not enough phased reads" verdict stays once it
// has fallen, both as the reference has them -- a read with an HP tag of 1 / 2 gets FROM_HAP_LL / OTHER_HAP_LL
// (snp_bam_processor.h:17-18), every other read 0 / 0.
int ltr_phasing_priors(int32_t n_reads, const int32_t* sample_of_read, const int32_t* haplotype, int32_t n_samples,
                       double* log_p1, double* log_p2, int32_t* phased_reads) {
  if (n_reads < 0 || n_samples < 0 || (n_reads > 0 && (!sample_of_read || !haplotype || !log_p1 || !log_p2))) return LTR_ERR_INVALID;
  for (int32_t r = 0; r < n_reads; ++r) {
    if (sample_of_read[r] < 0 || sample_of_read[r] >= n_samples) return LTR_ERR_INVALID;
    if (haplotype[r] != -1 && haplotype[r] != 1 && haplotype[r] != 2) return LTR_ERR_INVALID;      // assert(haplotype == 1 || haplotype == 2), :132
  }
  constexpr double FROM_HAP_LL = -0.000001, OTHER_HAP_LL = -1000.0;
  int32_t phased = 0, total = 0, h1 = 0, h2 = 0;
  bool not_enough = false;
  for (int32_t s = 0; s < n_samples; ++s) {
    for (int32_t r = 0; r < n_reads; ++r) {
      if (sample_of_read[r] != s) continue;
      ++total;                                                                                      // :177-183
      if (haplotype[r] == 1) ++h1; else if (haplotype[r] == 2) ++h2;
    }
    const double unphased_frac = (double)(total - (h1 + h2)) / (double)total;                       // 0 / 0 = NaN for a first group without reads: compares false, :187
    if (unphased_frac > 0.2 || h2 <= 1 || h1 <= 1) not_enough = true;                               // :190
    for (int32_t r = 0; r < n_reads; ++r) {
      if (sample_of_read[r] != s) continue;
      if (haplotype[r] != -1 && !not_enough) {                                                      // :218-227
        ++phased;
        log_p1[r] = haplotype[r] == 1 ? FROM_HAP_LL : OTHER_HAP_LL;
        log_p2[r] = haplotype[r] == 2 ? FROM_HAP_LL : OTHER_HAP_LL;
      } else { log_p1[r] = 0.0; log_p2[r] = 0.0; }
    }
  }
  if (phased_reads) *phased_reads = phased;
  return LTR_OK;
}

}  // extern "C"

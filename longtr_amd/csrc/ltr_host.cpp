// ltr_host.cpp -- host-side mirror of the reference objects either side of the DP:
// Haplotype iteration, HapAligner::trim_alignment, HapAligner::process_reads (long branch),
// ReadPooler, and the pool->read scatter of SeqStutterGenotyper::calc_hap_aln_probs.
// Integer / string work only; every DP cell is scored on the GPU through ltr_align_batch.
// Citations are to the LongTR reference (paths under its repository root).

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <map>
#include <string>
#include <vector>

#include "ltr_internal.h"

namespace ltr {

// Haplotype::init()/next(), forward direction (Haplotype.cpp:123-196): a mixed-radix
// reflected Gray walk.  factors_[i] = prod_{k<i} nopts_[k]; the block that moves at step
// `counter` is the LAST j with ((counter+1) mod factors_[j..]) == 0 taken right to left;
// it moves by dirs_[j], which flips at either end.
int haplotype_counts(const ltr_haplotype_blocks* hap, std::vector<int32_t>* counts_out, int64_t* ncombs_out) {
  if (!hap || hap->n_blocks <= 0) return LTR_ERR_INVALID;
  const int nb = hap->n_blocks;
  std::vector<int64_t> factors(nb);
  std::vector<int32_t> dirs(nb, 1), counts(nb, 0);
  int64_t ncombs = 1;
  for (int i = 0; i < nb; ++i) {
    if (hap->n_alleles[i] <= 0) return LTR_ERR_INVALID;
    factors[i] = ncombs;
    ncombs *= hap->n_alleles[i];
    if (ncombs > (1 << 24)) return LTR_ERR_INVALID;
  }
  counts_out->assign((size_t)(ncombs * nb), 0);
  for (int64_t c = 0;; ++c) {
    for (int i = 0; i < nb; ++i) (*counts_out)[(size_t)(c * nb + i)] = counts[i];
    if (c == ncombs - 1) break;
    int64_t t = c + 1;
    int idx = -1;
    for (int j = nb - 1; j >= 0; --j) { t %= factors[j]; if (t == 0) { idx = j; break; } }
    counts[idx] += dirs[idx];
    if (counts[idx] == 0 || counts[idx] == hap->n_alleles[idx] - 1) dirs[idx] *= -1;
  }
  *ncombs_out = ncombs;
  return LTR_OK;
}

static int64_t allele_slot(const ltr_haplotype_blocks* hap, int block, int allele) {
  int64_t k = 0;
  for (int b = 0; b < block; ++b) k += hap->n_alleles[b];
  return k + allele;
}

static void hap_string(const ltr_haplotype_blocks* hap, const int32_t* counts, std::string* out) {
  out->clear();
  for (int b = 0; b < hap->n_blocks; ++b) {                    // Haplotype::get_seq(), Haplotype.h:99-104
    const int64_t k = allele_slot(hap, b, counts[b]);
    out->append(reinterpret_cast<const char*>(hap->allele_bytes) + hap->allele_off[k],
                (size_t)(hap->allele_off[k + 1] - hap->allele_off[k]));
  }
}

// HapAligner::trim_alignment (HapAligner.cpp:346-465).  The reference copies the CIGAR and
// consumes it one base at a time from the front (left region, then left flank) and from the
// back (right region, right flank); two cursors with remaining counts do the same walk.
static int trim_alignment(const ltr_alignment* aln, int32_t repeat_start, int32_t repeat_end, int32_t padding,
                          int32_t* ltrim_out, int32_t* rtrim_out) {
  // A CIGAR element of length < 1 never runs out in the reference's one-base-at-a-time walk
  // (get_num() == 1 is its only exit, :376-379): rejected here instead of looping.
  for (int32_t k = 0; k < aln->n_cigar; ++k) if (aln->cigar_num[k] < 1) return LTR_ERR_CIGAR;
  const int32_t lo = repeat_start - padding, hi = repeat_end + padding;      // :349-350
  int32_t start_pos = aln->start + 1, end_pos = aln->stop + 1;               // :351,:353
  int32_t ltrim = 0, rtrim = 0;
  int fi = 0, bi = aln->n_cigar - 1;
  std::vector<int32_t> left(aln->cigar_num, aln->cigar_num + (aln->n_cigar > 0 ? aln->n_cigar : 0));
  auto pop_front = [&]() { if (left[fi] == 1) ++fi; else --left[fi]; };
  auto pop_back = [&]() { if (left[bi] == 1) --bi; else --left[bi]; };
  auto klass = [](char t) -> int {                             // 0: M/=/X  1: D  2: I/S  3: H  -1: invalid
    switch (t) { case 'M': case '=': case 'X': return 0; case 'D': return 1; case 'I': case 'S': return 2; case 'H': return 3; default: return -1; }
  };
  while (start_pos <= lo && fi <= bi) {                        // left region, :360-382
    const int k = klass(aln->cigar_type[fi]);
    if (k < 0) return LTR_ERR_CIGAR;
    if (k == 0) { ++ltrim; ++start_pos; } else if (k == 1) ++start_pos; else if (k == 2) ++ltrim;
    pop_front();
  }
  for (int32_t mid = start_pos; mid > lo && mid <= lo + padding && fi <= bi;) {   // left flank, :385-408
    const int k = klass(aln->cigar_type[fi]);
    if (k < 0) return LTR_ERR_CIGAR;
    if (k == 0) ++mid; else if (k == 1) { --ltrim; ++mid; }
    pop_front();
  }
  while (end_pos > hi && fi <= bi) {                           // right region, :411-433
    const int k = klass(aln->cigar_type[bi]);
    if (k < 0) return LTR_ERR_CIGAR;
    if (k == 0) { ++rtrim; --end_pos; } else if (k == 1) --end_pos; else if (k == 2) ++rtrim;
    pop_back();
  }
  for (int32_t mid = end_pos; mid > hi - padding && mid <= hi && fi <= bi;) {     // right flank, :436-458
    const int k = klass(aln->cigar_type[bi]);
    if (k < 0) return LTR_ERR_CIGAR;
    if (k == 0) --mid; else if (k == 1) { --rtrim; --mid; }
    pop_back();
  }
  if (ltrim < 0) ltrim = 0;                                    // :461-462
  if (rtrim < 0) rtrim = 0;
  *ltrim_out = ltrim; *rtrim_out = rtrim;
  return (ltrim + rtrim <= aln->seq_len) ? LTR_OK : LTR_ERR_INVALID;         // assert, :463
}

// trimmed read of one alignment appended to a byte pool: trim_alignment (:819) and, for an empty
// trim, the last 5 bp of the first block's reference allele + the first 5 bp of the last block's
// (HapAligner.cpp:820-823)
// (error text goes to *err: the per-locus preparation runs on several host threads)
static int append_trimmed(std::string* err, const ltr_haplotype_blocks* hap, int rb, const ltr_alignment* aln, int32_t padding,
                          std::vector<uint8_t>* read_bytes, std::vector<int64_t>* read_off) {
  int32_t lt = 0, rt = 0;
  const int rc = trim_alignment(aln, hap->block_start[rb], hap->block_end[rb], padding, &lt, &rt);
  if (rc != LTR_OK) {
    *err = rc == LTR_ERR_CIGAR ? "Invalid CIGAR option encountered in trim_alignment" : "trim_alignment: ltrim+rtrim exceeds the read length";
    return rc;
  }
  const int64_t len = (int64_t)aln->seq_len - lt - rt;
  if (len > 0) {
    read_bytes->insert(read_bytes->end(), aln->seq + lt, aln->seq + lt + len);
  } else {
    const int64_t a0 = 0, aL = allele_slot(hap, hap->n_blocks - 1, 0);
    const int64_t l0 = hap->allele_off[a0 + 1] - hap->allele_off[a0];
    const int64_t lL = hap->allele_off[aL + 1] - hap->allele_off[aL];
    if (l0 < 5) { *err = "left flank shorter than 5 bp (std::string::substr would throw in the reference)"; return LTR_ERR_INVALID; }
    const uint8_t* f0 = hap->allele_bytes + hap->allele_off[a0];
    const uint8_t* fL = hap->allele_bytes + hap->allele_off[aL];
    read_bytes->insert(read_bytes->end(), f0 + l0 - 5, f0 + l0);
    read_bytes->insert(read_bytes->end(), fL, fL + (lL < 5 ? lL : 5));
  }
  read_off->push_back((int64_t)read_bytes->size());
  return LTR_OK;
}

// haplotype strings in Haplotype::next() order appended to a byte pool; returns H or < 0
static int64_t append_haplotypes(const ltr_haplotype_blocks* hap, std::vector<uint8_t>* hap_bytes, std::vector<int64_t>* hap_off) {
  std::vector<int32_t> counts; int64_t H = 0;
  const int rc = haplotype_counts(hap, &counts, &H);
  if (rc != LTR_OK) return rc;
  for (int64_t k = 0; k < H; ++k) {
    const int32_t* ck = counts.data() + k * hap->n_blocks;
    for (int b = 0; b < hap->n_blocks; ++b) {                    // Haplotype::get_seq(), Haplotype.h:99-104
      const int64_t a = allele_slot(hap, b, ck[b]);
      hap_bytes->insert(hap_bytes->end(), hap->allele_bytes + hap->allele_off[a], hap->allele_bytes + hap->allele_off[a + 1]);
    }
    hap_off->push_back((int64_t)hap_bytes->size());
  }
  return H;
}

}  // namespace ltr

extern "C" {

int64_t ltr_haplotype_num_combs(const ltr_haplotype_blocks* hap) {
  if (!hap || hap->n_blocks <= 0) return LTR_ERR_INVALID;
  int64_t n = 1;
  for (int i = 0; i < hap->n_blocks; ++i) n *= hap->n_alleles[i];
  return n;
}

int64_t ltr_haplotype_seq(const ltr_haplotype_blocks* hap, int64_t index, uint8_t* out, int64_t cap) {
  std::vector<int32_t> counts; int64_t ncombs = 0;
  const int rc = ltr::haplotype_counts(hap, &counts, &ncombs);
  if (rc != LTR_OK) return rc;
  if (index < 0 || index >= ncombs) return LTR_ERR_INVALID;
  std::string s;
  ltr::hap_string(hap, counts.data() + index * hap->n_blocks, &s);
  if ((int64_t)s.size() > cap) return LTR_ERR_INVALID;
  std::memcpy(out, s.data(), s.size());
  return (int64_t)s.size();
}

int ltr_trim_alignment(const ltr_alignment* aln, int32_t repeat_start, int32_t repeat_end,
                       int32_t indel_flank_len, int32_t* ltrim, int32_t* rtrim) {
  if (!aln || !ltrim || !rtrim) return LTR_ERR_INVALID;
  return ltr::trim_alignment(aln, repeat_start, repeat_end, indel_flank_len, ltrim, rtrim);
}

// HapAligner::process_reads (HapAligner.cpp:545-581) + process_read's long branch (:814-854).
int ltr_process_reads(ltr_ctx* ctx, const ltr_haplotype_blocks* hap, const uint8_t* realign_to_hap,
                      const ltr_alignment* alns, int32_t n_alns, int32_t init_read_index,
                      const uint8_t* realign_read, double* aln_probs, int32_t* seed_positions) {
  if (!ctx || !hap || (!alns && n_alns > 0) || n_alns < 0 || !aln_probs || !seed_positions) return LTR_ERR_INVALID;
  for (int32_t i = 0; i < n_alns; ++i)
    if (alns[i].seq_len < 0 || (alns[i].seq_len > 0 && !alns[i].seq) || alns[i].n_cigar < 0 ||
        (alns[i].n_cigar > 0 && (!alns[i].cigar_type || !alns[i].cigar_num))) {
      ltr::set_error(ctx, "alignment with a negative length or a null sequence / CIGAR pointer"); return LTR_ERR_INVALID;
    }
  ltr::TimedCall timed(ctx, ltr::kTimerHapAln);                // total_hap_aln_time_, seq_stutter_genotyper.cpp:515,:561-562
  LTR_GUARD_BEGIN
  // repeat_starts_[0] / repeat_ends_[0]: the first block that carries repeat info (HapAligner.h:103-109)
  int rb = -1;
  for (int b = 0; b < hap->n_blocks; ++b) if (hap->is_repeat[b]) { rb = b; break; }
  if (rb < 0) { ltr::set_error(ctx, "haplotype has no repeat block"); return LTR_ERR_INVALID; }
  // short_ = (block 1 period == 1 && SWITCH_OLD_ALIGN_LEN), :552: the seeded stutter path
  if (ltr::ctx_params(ctx).use_short_path && hap->n_blocks > 1 && hap->period[1] == 1)
    return ltr::process_reads_short(ctx, hap, realign_to_hap, alns, n_alns, init_read_index, realign_read,
                                    aln_probs, seed_positions);
  // haplotype strings in Haplotype::next() order
  std::vector<uint8_t> hap_bytes; std::vector<int64_t> hap_off(1, 0);
  const int64_t H = ltr::append_haplotypes(hap, &hap_bytes, &hap_off);
  if (H < 0) { ltr::set_error(ctx, "bad haplotype block structure"); return (int)H; }
  int rc = LTR_OK;
  std::vector<uint8_t> read_bytes; std::vector<int64_t> read_off(1, 0);
  std::vector<uint8_t> mask_r((size_t)n_alns, 1);
  const int32_t padding = ltr::ctx_params(ctx).indel_flank_len;
  for (int32_t i = 0; i < n_alns; ++i) {
    if (realign_read && !realign_read[i]) { mask_r[(size_t)i] = 0; read_bytes.push_back('N'); read_off.push_back((int64_t)read_bytes.size()); continue; }
    std::string err;
    if ((rc = ltr::append_trimmed(&err, hap, rb, &alns[i], padding, &read_bytes, &read_off)) != LTR_OK) { ltr::set_error(ctx, err); return rc; }
  }
  ltr_locus_batch b;
  std::memset(&b, 0, sizeof(b));
  const int64_t lro[2] = {0, n_alns}, lho[2] = {0, H};
  b.n_loci = 1; b.locus_read_off = lro; b.locus_hap_off = lho;
  b.n_reads = n_alns; b.read_bytes = read_bytes.data(); b.read_off = read_off.data();
  b.n_haps = H; b.hap_bytes = hap_bytes.data(); b.hap_off = hap_off.data();
  b.realign_read = mask_r.data(); b.realign_hap = realign_to_hap;
  double* prob_ptr = aln_probs + (int64_t)init_read_index * H;                // :550
  rc = ltr_align_batch(ctx, &b, prob_ptr, nullptr);
  if (rc != LTR_OK) return rc;
  for (int32_t i = 0; i < n_alns; ++i)
    if (mask_r[(size_t)i]) seed_positions[init_read_index + i] = alns[i].seq_len - 1;   // :562-563 (UNtrimmed length - 1)
  return LTR_OK;
  LTR_GUARD_END(ctx)
}

// ReadPooler::add_alignment (read_pooler.cpp:3-20): pools keyed by the exact sequence,
// numbered by first occurrence.
int32_t ltr_pool_reads(const uint8_t* const* seqs, const int32_t* seq_lens, int32_t n_reads, int32_t* pool_index) {
  if (n_reads < 0 || ((!seqs || !seq_lens || !pool_index) && n_reads > 0)) return LTR_ERR_INVALID;
  for (int32_t i = 0; i < n_reads; ++i) if (seq_lens[i] < 0 || (seq_lens[i] > 0 && !seqs[i])) return LTR_ERR_INVALID;
  // (the reference keys a std::map by the sequence; the same pools, in the same order, come out of an open-addressing
  // table of 64-bit hashes with the byte comparison only on a hash match -- no key copies, no tree of kilobase strings)
  auto hash_seq = [](const uint8_t* p, int32_t len) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)len;
    int32_t k = 0;
    for (; k + 8 <= len; k += 8) { uint64_t w; std::memcpy(&w, p + k, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
    uint64_t w = 0;
    if (k < len) std::memcpy(&w, p + k, (size_t)(len - k));
    h = (h ^ w) * 0xC4CEB9FE1A85EC53ull; h ^= h >> 29;
    return h;
  };
  size_t cap = 16;
  while (cap < (size_t)n_reads * 2) cap <<= 1;
  std::vector<int32_t> slot(cap, -1);                           // -> first read of the pool stored there
  std::vector<uint64_t> hashes((size_t)n_reads);
  int32_t n_pools = 0;
  for (int32_t i = 0; i < n_reads; ++i) {
    const uint64_t h = hashes[(size_t)i] = hash_seq(seqs[i], seq_lens[i]);
    size_t at = (size_t)h & (cap - 1);
    for (;; at = (at + 1) & (cap - 1)) {
      const int32_t f = slot[at];
      if (f < 0) { slot[at] = i; pool_index[i] = n_pools++; break; }
      if (hashes[(size_t)f] == h && seq_lens[f] == seq_lens[i] && (seq_lens[i] == 0 || std::memcmp(seqs[f], seqs[i], (size_t)seq_lens[i]) == 0)) {
        pool_index[i] = pool_index[f]; break;
      }
    }
  }
  return n_pools;
}

// SeqStutterGenotyper::calc_hap_aln_probs, the part after process_reads
// (seq_stutter_genotyper.cpp:526-559).
int ltr_scatter_pool_probs(const double* log_pool_aln_probs, const int32_t* pool_seed_positions,
                           const int32_t* pool_index, int32_t n_reads, int32_t n_alleles,
                           const uint8_t* realign_to_hap, const uint8_t* copy_read, const uint8_t* second_mate,
                           double* log_aln_probs, int32_t* seed_positions) {
  if (!log_pool_aln_probs || !pool_index || !log_aln_probs || n_reads < 0 || n_alleles <= 0) return LTR_ERR_INVALID;
  for (int32_t i = 0; i < n_reads; ++i) {                      // :527-538
    if (copy_read && !copy_read[i]) continue;
    if (seed_positions && pool_seed_positions) seed_positions[i] = pool_seed_positions[pool_index[i]];
    const double* src = log_pool_aln_probs + (int64_t)n_alleles * pool_index[i];
    double* dst = log_aln_probs + (int64_t)n_alleles * i;
    for (int32_t j = 0; j < n_alleles; ++j) if (!realign_to_hap || realign_to_hap[j]) dst[j] = src[j];
  }
  for (int32_t i = 0; i < n_reads; ++i) {                      // mate pairs share one row sum, :546-559
    if (!second_mate || !second_mate[i] || (copy_read && !copy_read[i])) continue;
    if (i == 0) return LTR_ERR_INVALID;
    double* m1 = log_aln_probs + (int64_t)(i - 1) * n_alleles;
    double* m2 = log_aln_probs + (int64_t)i * n_alleles;
    for (int32_t j = 0; j < n_alleles; ++j)
      if (!realign_to_hap || realign_to_hap[j]) { const double tot = m1[j] + m2[j]; m1[j] = tot; m2[j] = tot; }
  }
  return LTR_OK;
}

// BaseQuality::median_base_qualities (base_quality.cpp:11-28): per position, the upper median
static std::vector<uint8_t> median_qualities(const std::vector<const ltr_alignment*>& members) {
  const int32_t len = members[0]->seq_len;
  std::vector<uint8_t> out((size_t)len, 'N'), col;
  for (int32_t i = 0; i < len; ++i) {
    col.clear();
    for (const ltr_alignment* m : members) col.push_back((uint8_t)(char)m->qual[i]);
    std::sort(col.begin(), col.end(), [](uint8_t x, uint8_t y) { return (char)x < (char)y; });
    out[(size_t)i] = col[col.size() / 2];
  }
  return out;
}

// SeqStutterGenotyper::calc_hap_aln_probs (seq_stutter_genotyper.cpp:514-563) for MANY loci in
// one GPU pass: pool the reads of each locus (ReadPooler, read_pooler.cpp:3-20: exact sequence,
// the pool keeps the FIRST read's start/stop/CIGAR), trim each pool (HapAligner::trim_alignment),
// score every pool x haplotype pair of every locus in a single plan, then fan the pool rows out
// to the reads and sum mate-pair rows (:526-559).  Period-1 loci under --stutter-align-len take
// the short path with the pools' median base qualities (ReadPooler::pool, read_pooler.h:42-48).
int ltr_calc_hap_aln_probs(ltr_ctx* ctx, const ltr_locus* loci, int64_t n_loci,
                           double* const* log_aln_probs, int32_t* const* seed_positions) {
  if (!ctx || (!loci && n_loci > 0) || n_loci < 0 || !log_aln_probs || !seed_positions) return LTR_ERR_INVALID;
  ltr::TimedCall timed(ctx, ltr::kTimerHapAln);                        // total_hap_aln_time_, seq_stutter_genotyper.cpp:515,:561-562
  LTR_GUARD_BEGIN
  const ltr_align_params prm = ltr::ctx_params(ctx);
  const bool dbg = std::getenv("LTR_DEBUG") != nullptr || ltr::ctx_debug(ctx).trace != 0;
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
  struct ExitStamp {                                                  // (declared first: reports after every buffer of the call is freed)
    bool on; std::chrono::steady_clock::time_point t0;
    ~ExitStamp() { if (on) std::fprintf(stderr, "[ltr] calc_hap_aln_probs: returning at %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
  } exit_stamp{dbg, t_start};
  std::vector<std::vector<int32_t>> pool_index((size_t)n_loci);
  std::vector<std::vector<int32_t>> pool_first((size_t)n_loci);      // first read of every pool
  struct ShortLocus { int64_t locus = 0, H = 0; std::vector<double> pool_probs; std::vector<int32_t> pool_seeds; };
  std::deque<ShortLocus> short_loci;                                  // (deque: the queued result pointers stay valid)
  struct ShortBatchDel { void operator()(ltr::ShortBatch* p) const { ltr::short_batch_free(p); } };
  std::unique_ptr<ltr::ShortBatch, ShortBatchDel> short_batch;

  bool any_mask = false;
  for (int64_t l = 0; l < n_loci; ++l) {
    const ltr_locus& L = loci[l];
    if (!L.hap || (!L.alns && L.n_alns > 0) || L.n_alns < 0 || !log_aln_probs[l] || !seed_positions[l]) return LTR_ERR_INVALID;
    for (int32_t i = 0; i < L.n_alns; ++i)
      if (L.alns[i].seq_len < 0 || (L.alns[i].seq_len > 0 && !L.alns[i].seq) || L.alns[i].n_cigar < 0 ||
          (L.alns[i].n_cigar > 0 && (!L.alns[i].cigar_type || !L.alns[i].cigar_num))) {
        ltr::set_error(ctx, "alignment with a negative length or a null sequence / CIGAR pointer"); return LTR_ERR_INVALID;
      }
    any_mask |= (L.realign_to_hap != nullptr) || (L.realign_pool != nullptr);
  }

  // ---- per locus, on all host cores: pools, trimmed pool sequences, haplotype strings --------
  struct LocusPrep {
    int rc = LTR_OK; std::string err;
    int rb = -1; int32_t P = 0; int64_t H = 0; bool short_path = false;
    std::vector<uint8_t> rbytes, hbytes; std::vector<int64_t> roff, hoff;      // offsets local to the locus
  };
  auto prepare = [&](int64_t l, LocusPrep& R) {
    const ltr_locus& L = loci[l];
    for (int b = 0; b < L.hap->n_blocks; ++b) if (L.hap->is_repeat[b]) { R.rb = b; break; }
    if (R.rb < 0) { R.err = "haplotype has no repeat block"; R.rc = LTR_ERR_INVALID; return; }
    std::vector<const uint8_t*> seqs((size_t)L.n_alns); std::vector<int32_t> lens((size_t)L.n_alns);
    for (int32_t i = 0; i < L.n_alns; ++i) { seqs[(size_t)i] = L.alns[i].seq; lens[(size_t)i] = L.alns[i].seq_len; }
    pool_index[(size_t)l].assign((size_t)L.n_alns, 0);
    R.P = ltr_pool_reads(seqs.data(), lens.data(), L.n_alns, pool_index[(size_t)l].data());
    if (R.P < 0) { R.rc = R.P; return; }
    pool_first[(size_t)l].assign((size_t)R.P, -1);
    for (int32_t i = 0; i < L.n_alns; ++i) { int32_t& f = pool_first[(size_t)l][(size_t)pool_index[(size_t)l][(size_t)i]]; if (f < 0) f = i; }
    R.short_path = prm.use_short_path && L.hap->n_blocks > 1 && L.hap->period[1] == 1;      // HapAligner.cpp:552
    if (R.short_path) return;                                        // prepared serially (one shared accumulator)
    R.roff.push_back(0); R.hoff.push_back(0);
    R.H = ltr::append_haplotypes(L.hap, &R.hbytes, &R.hoff);
    if (R.H < 0) { R.err = "bad haplotype block structure"; R.rc = (int)R.H; return; }
    {
      size_t upper = 0;                                                 // (one allocation: the trimmed reads are at most this long)
      for (int32_t q = 0; q < R.P; ++q) upper += (size_t)L.alns[pool_first[(size_t)l][(size_t)q]].seq_len + 10;
      R.rbytes.reserve(upper); R.roff.reserve((size_t)R.P + 1);
    }
    for (int32_t q = 0; q < R.P; ++q) {
      if (L.realign_pool && !L.realign_pool[q]) {                      // not realigned: a placeholder keeps the pool's row in place
        R.rbytes.push_back('N'); R.roff.push_back((int64_t)R.rbytes.size());
        continue;
      }
      const int rc = ltr::append_trimmed(&R.err, L.hap, R.rb, &L.alns[pool_first[(size_t)l][(size_t)q]], prm.indel_flank_len, &R.rbytes, &R.roff);
      if (rc != LTR_OK) { R.rc = rc; return; }
    }
  };

  // ---- chunks of loci: while the GPU scores chunk c the host prepares chunk c+1 (pooling, trimming,
  // haplotype strings on all cores; then validation, pair descriptors, sort and upload of its plan) ----
  struct Chunk {
    int64_t l0 = 0, l1 = 0;                     // loci [l0, l1)
    uint8_t* read_bytes = nullptr; uint8_t* hap_bytes = nullptr;   // (the context's staging arrays: uploaded by ltr_plan_create before the next chunk reuses them)
    std::vector<uint8_t> mask_r, mask_h;
    std::vector<int64_t> read_off, hap_off, lro, lho;
    std::vector<int64_t> slot_locus, locus_H;   // long-path loci of the chunk, in order
    ltr_plan* plan = nullptr;
    std::vector<double> ll;
  };
  // Two chunks, 1 : 3 -- the GPU starts on the first quarter while the host cores prepare the rest; the plans
  // run on two streams, so the tail of the first plan's launches overlaps the head of the second's.
  // Measured on MI355X, 6000 raw config-3 loci (4.86e11 cells: 177 ms of DP at the resident rate), best of 4
  // calls, same box: one plan 210.5 ms per call; 1 : 1 202.7; 1 : 2 197.6; 1 : 3 194.9; three chunks 1 : 2 : 3
  // 202.9; eight chunks 1 : .. : 8 on three streams 233 (every plan is a chain of ~17 launches, each at least
  // as long as its longest pair: small plans leave the GPU part empty).  1000 loci: one plan 45.6 ms, two 46.5.
  // (ltr_ctx_set_debug "chunks" / "chunk_streams" / "chunk_growth" override the rule: tests/manual/gpu_chunk_sweep.py.)
  int64_t n_chunks = n_loci >= 1500 ? 2 : 1;
  int n_streams = 2;
  const ltr::DebugKnobs knobs = ltr::ctx_debug(ctx);
  if (knobs.chunks > 0) n_chunks = std::max<int64_t>(1, std::min<int64_t>(knobs.chunks, std::max<int64_t>(n_loci, 1)));
  if (knobs.chunk_streams > 0) n_streams = knobs.chunk_streams;
  std::vector<Chunk> chunks((size_t)n_chunks);
  std::vector<double> cum((size_t)n_chunks + 1, 0.0);                 // cumulative chunk weights
  {
    double growth = 3.0;                                              // 0: weights 1, 2, 3, ...; g > 0: 1, g, g^2, ...; g < 0: 1, 2, .., k, k, .., 2, 1
    if (knobs.chunk_growth_set) growth = knobs.chunk_growth;
    double w = 1.0;
    for (int64_t c = 0; c < n_chunks; ++c) {
      const double wc = growth > 0.0 ? w : (growth < 0.0 ? (double)(std::min(c, n_chunks - 1 - c) + 1) : (double)(c + 1));
      cum[(size_t)c + 1] = cum[(size_t)c] + wc; w *= growth;
    }
  }
  int rc = LTR_OK;
  auto cleanup = [&]() { for (Chunk& C : chunks) if (C.plan) { ltr_plan_destroy(C.plan); C.plan = nullptr; } };
  for (int64_t c = 0; c < n_chunks && rc == LTR_OK; ++c) {
    Chunk& C = chunks[(size_t)c];
    C.l0 = (int64_t)((double)n_loci * cum[(size_t)c] / cum[(size_t)n_chunks]);
    C.l1 = (c + 1 == n_chunks) ? n_loci : (int64_t)((double)n_loci * cum[(size_t)c + 1] / cum[(size_t)n_chunks]);
    std::vector<LocusPrep> prep((size_t)(C.l1 - C.l0));
    ltr::parallel_for(C.l1 - C.l0, 64, [&](int64_t k) { prepare(C.l0 + k, prep[(size_t)k]); });
    if (dbg) std::fprintf(stderr, "[ltr] calc_hap_aln_probs: chunk %ld prepared at %.1f ms\n", (long)c, since());
    C.read_off.push_back(0); C.hap_off.push_back(0); C.lro.push_back(0); C.lho.push_back(0);
    struct Place { int64_t k, r0, h0; };                                         // where locus k's bytes go in the chunk's buffers
    std::vector<Place> place;
    int64_t n_rbytes = 0, n_hbytes = 0;
    // in locus order: first error wins; short-path loci queue up; the rest is concatenated
    for (int64_t l = C.l0; l < C.l1 && rc == LTR_OK; ++l) {
      const ltr_locus& L = loci[l];
      LocusPrep& R = prep[(size_t)(l - C.l0)];
      if (R.rc != LTR_OK) { if (!R.err.empty()) ltr::set_error(ctx, R.err); rc = R.rc; break; }
      const int32_t P = R.P;
      if (R.short_path) {
        // per-locus short path on the pooled alignments (median qualities)
        std::vector<ltr_alignment> pooled((size_t)P);
        std::vector<std::vector<uint8_t>> quals((size_t)P);
        for (int32_t q = 0; q < P && rc == LTR_OK; ++q) {
          pooled[(size_t)q] = L.alns[pool_first[(size_t)l][(size_t)q]];
          std::vector<const ltr_alignment*> members;
          for (int32_t i = 0; i < L.n_alns; ++i) if (pool_index[(size_t)l][(size_t)i] == q) members.push_back(&L.alns[i]);
          for (const ltr_alignment* m : members) if (!m->qual) { ltr::set_error(ctx, "short path needs base qualities"); rc = LTR_ERR_INVALID; break; }
          if (rc != LTR_OK) break;
          quals[(size_t)q] = median_qualities(members);
          pooled[(size_t)q].qual = quals[(size_t)q].data();
        }
        if (rc != LTR_OK) break;
        const int64_t H = ltr_haplotype_num_combs(L.hap);
        // queued: every short-path locus of the call is scored in ONE launch after the chunks are on their way
        if (!short_batch) short_batch.reset(ltr::short_batch_new());
        short_loci.emplace_back();
        ShortLocus& SLc = short_loci.back();
        SLc.locus = l; SLc.H = H;
        SLc.pool_probs.assign((size_t)P * (size_t)H, 0.0); SLc.pool_seeds.assign((size_t)P, 0);
        rc = ltr::short_batch_add(ctx, short_batch.get(), L.hap, L.realign_to_hap, pooled.data(), P, 0, L.realign_pool,
                                  SLc.pool_probs.data(), SLc.pool_seeds.data());
        continue;
      }
      const int64_t r0 = n_rbytes, h0 = n_hbytes;
      n_rbytes += (int64_t)R.rbytes.size(); n_hbytes += (int64_t)R.hbytes.size();
      place.push_back({l - C.l0, r0, h0});
      for (size_t k = 1; k < R.roff.size(); ++k) C.read_off.push_back(r0 + R.roff[k]);
      for (size_t k = 1; k < R.hoff.size(); ++k) C.hap_off.push_back(h0 + R.hoff[k]);
      C.lro.push_back((int64_t)C.read_off.size() - 1); C.lho.push_back((int64_t)C.hap_off.size() - 1);
      if (any_mask) {
        for (int32_t q = 0; q < P; ++q) C.mask_r.push_back((L.realign_pool && !L.realign_pool[q]) ? 0 : 1);
        for (int64_t h = 0; h < R.H; ++h) C.mask_h.push_back((L.realign_to_hap && !L.realign_to_hap[h]) ? 0 : 1);
      }
      C.slot_locus.push_back(l); C.locus_H.push_back(R.H);
    }
    if (rc != LTR_OK || C.slot_locus.empty()) continue;
    C.read_bytes = ltr::ctx_host_bytes(ctx, 0, (size_t)std::max<int64_t>(n_rbytes, 1));
    C.hap_bytes = ltr::ctx_host_bytes(ctx, 1, (size_t)std::max<int64_t>(n_hbytes, 1));
    ltr::parallel_for((int64_t)place.size(), 16, [&](int64_t i) {
      const Place& pl = place[(size_t)i];
      LocusPrep& R = prep[(size_t)pl.k];
      if (!R.rbytes.empty()) std::memcpy(C.read_bytes + pl.r0, R.rbytes.data(), R.rbytes.size());
      if (!R.hbytes.empty()) std::memcpy(C.hap_bytes + pl.h0, R.hbytes.data(), R.hbytes.size());
      std::vector<uint8_t>().swap(R.rbytes); std::vector<uint8_t>().swap(R.hbytes);      // (freed here, on the worker threads)
    });
    ltr_locus_batch b;
    std::memset(&b, 0, sizeof(b));
    b.n_loci = (int64_t)C.slot_locus.size(); b.locus_read_off = C.lro.data(); b.locus_hap_off = C.lho.data();
    b.n_reads = (int64_t)C.read_off.size() - 1; b.read_bytes = C.read_bytes; b.read_off = C.read_off.data();
    b.n_haps = (int64_t)C.hap_off.size() - 1; b.hap_bytes = C.hap_bytes; b.hap_off = C.hap_off.data();
    if (any_mask) { b.realign_read = C.mask_r.data(); b.realign_hap = C.mask_h.data(); }
    if (dbg) std::fprintf(stderr, "[ltr] calc_hap_aln_probs: chunk %ld concatenated at %.1f ms\n", (long)c, since());
    rc = ltr_plan_create(ctx, &b, &C.plan);
    if (dbg) std::fprintf(stderr, "[ltr] calc_hap_aln_probs: chunk %ld planned at %.1f ms\n", (long)c, since());
    // asynchronous: returns once the launches are queued.  Chunks alternate between two streams: the first
    // kernels of chunk c+1 run next to the exact kernels and the tail of chunk c.
    if (rc == LTR_OK) rc = ltr_plan_execute(C.plan, nullptr, ltr::ctx_side_stream(ctx, (int)(c % n_streams)));
    if (dbg) std::fprintf(stderr, "[ltr] calc_hap_aln_probs: chunk %ld (%ld loci) queued at %.1f ms\n", (long)c, (long)(C.l1 - C.l0), since());
  }
  if (rc != LTR_OK) { cleanup(); return rc; }
  if (short_batch) {
    rc = ltr::short_batch_run(ctx, short_batch.get());
    for (ShortLocus& SLc : short_loci) {
      if (rc != LTR_OK) break;
      const ltr_locus& L = loci[SLc.locus];
      rc = ltr_scatter_pool_probs(SLc.pool_probs.data(), SLc.pool_seeds.data(), pool_index[(size_t)SLc.locus].data(), L.n_alns,
                                  (int32_t)SLc.H, L.realign_to_hap, L.copy_read, L.second_mate, log_aln_probs[SLc.locus], seed_positions[SLc.locus]);
    }
    if (rc != LTR_OK) { cleanup(); return rc; }
  }
  // ---- in chunk order: results of chunk c are fanned out to its reads while the later chunks still run ----
  for (Chunk& C : chunks) {
    if (!C.plan) continue;
    C.ll.resize((size_t)std::max<int64_t>(ltr_plan_ll_size(C.plan), 1));
    rc = ltr_plan_fetch(C.plan, C.ll.data(), nullptr);                         // waits for THIS plan's kernels only
    if (dbg) std::fprintf(stderr, "[ltr] calc_hap_aln_probs: chunk fetched at %.1f ms\n", since());
    ltr_plan_destroy(C.plan); C.plan = nullptr;
    if (rc != LTR_OK) break;
    std::vector<int64_t> offs(C.slot_locus.size() + 1, 0);
    for (size_t k = 0; k < C.slot_locus.size(); ++k) offs[k + 1] = offs[k] + (C.lro[k + 1] - C.lro[k]) * C.locus_H[k];
    std::atomic<int> first_rc(LTR_OK);
    ltr::parallel_for((int64_t)C.slot_locus.size(), 128, [&](int64_t k) {
      const int64_t l = C.slot_locus[(size_t)k];
      const ltr_locus& L = loci[l];
      const int64_t P = C.lro[(size_t)k + 1] - C.lro[(size_t)k], H = C.locus_H[(size_t)k];
      std::vector<int32_t> pool_seeds((size_t)P);
      for (int64_t q = 0; q < P; ++q) pool_seeds[(size_t)q] = L.alns[pool_first[(size_t)l][(size_t)q]].seq_len - 1;   // HapAligner.cpp:562-563
      const int r2 = ltr_scatter_pool_probs(C.ll.data() + offs[(size_t)k], pool_seeds.data(), pool_index[(size_t)l].data(), L.n_alns, (int32_t)H,
                                            L.realign_to_hap, L.copy_read, L.second_mate, log_aln_probs[l], seed_positions[l]);
      if (r2 != LTR_OK) { int expect = LTR_OK; first_rc.compare_exchange_strong(expect, r2); }
    });
    rc = first_rc.load();
    if (rc != LTR_OK) break;
  }
  cleanup();
  if (dbg) std::fprintf(stderr, "[ltr] calc_hap_aln_probs: scatter done at %.1f ms\n", since());
  return rc;
  LTR_GUARD_END(ctx)
}

}  // extern "C"

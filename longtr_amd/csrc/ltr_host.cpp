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
#include <cmath>
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

int64_t allele_slot(const ltr_haplotype_blocks* hap, int block, int allele) {
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

// HapAligner::trim_alignment (HapAligner.cpp:346-465).  The reference copies the CIGAR and consumes it ONE BASE at a
// time from the front (left region :360-382, then left flank :385-408) and from the back (right region :411-433,
// right flank :436-458).  Inside one CIGAR element every base does the same thing, so each of the four walks is
// done element by element here: an element gives up min(its remaining bases, the bases the walk still wants).
// `rem` (scratch, n_cigar ints) holds what is left of every element; front and back cursors share it like the
// reference's list.  Same results, O(#elements) instead of O(read length) (checked against the compiled
// reference's trims: tests/golden/process_locus.json, and against the base-by-base restatement on random CIGARs).
static inline int cigar_klass(char t) {                          // 0: M/=/X  1: D  2: I/S  3: H  -1: invalid
  switch (t) { case 'M': case '=': case 'X': return 0; case 'D': return 1; case 'I': case 'S': return 2; case 'H': return 3; default: return -1; }
}
int trim_alignment_into(const ltr_alignment* aln, int32_t repeat_start, int32_t repeat_end, int32_t padding,
                               int32_t* rem, int32_t* ltrim_out, int32_t* rtrim_out) {
  // A CIGAR element of length < 1 never runs out in the reference's one-base-at-a-time walk
  // (get_num() == 1 is its only exit, :376-379): rejected here instead of looping.
  for (int32_t k = 0; k < aln->n_cigar; ++k) { if (aln->cigar_num[k] < 1) return LTR_ERR_CIGAR; rem[k] = aln->cigar_num[k]; }
  const int64_t lo = (int64_t)repeat_start - padding, hi = (int64_t)repeat_end + padding;      // :349-350
  int64_t start_pos = (int64_t)aln->start + 1, end_pos = (int64_t)aln->stop + 1;               // :351,:353
  int64_t ltrim = 0, rtrim = 0;
  int fi = 0, bi = aln->n_cigar - 1;
  while (start_pos <= lo && fi <= bi) {                        // left region, :360-382
    const int k = cigar_klass(aln->cigar_type[fi]);
    if (k < 0) return LTR_ERR_CIGAR;
    int64_t take = rem[fi];
    if (k <= 1) { take = std::min<int64_t>(take, lo - start_pos + 1); start_pos += take; }
    if (k == 0 || k == 2) ltrim += take;
    if ((rem[fi] -= (int32_t)take) == 0) ++fi;
  }
  for (int64_t mid = start_pos; mid > lo && mid <= lo + padding && fi <= bi;) {   // left flank, :385-408
    const int k = cigar_klass(aln->cigar_type[fi]);
    if (k < 0) return LTR_ERR_CIGAR;
    int64_t take = rem[fi];
    if (k <= 1) { take = std::min<int64_t>(take, lo + padding - mid + 1); mid += take; }
    if (k == 1) ltrim -= take;
    if ((rem[fi] -= (int32_t)take) == 0) ++fi;
  }
  while (end_pos > hi && fi <= bi) {                           // right region, :411-433
    const int k = cigar_klass(aln->cigar_type[bi]);
    if (k < 0) return LTR_ERR_CIGAR;
    int64_t take = rem[bi];
    if (k <= 1) { take = std::min<int64_t>(take, end_pos - hi); end_pos -= take; }
    if (k == 0 || k == 2) rtrim += take;
    if ((rem[bi] -= (int32_t)take) == 0) --bi;
  }
  for (int64_t mid = end_pos; mid > hi - padding && mid <= hi && fi <= bi;) {     // right flank, :436-458
    const int k = cigar_klass(aln->cigar_type[bi]);
    if (k < 0) return LTR_ERR_CIGAR;
    int64_t take = rem[bi];
    if (k <= 1) { take = std::min<int64_t>(take, mid - (hi - padding)); mid -= take; }
    if (k == 1) rtrim -= take;
    if ((rem[bi] -= (int32_t)take) == 0) --bi;
  }
  if (ltrim < 0) ltrim = 0;                                    // :461-462
  if (rtrim < 0) rtrim = 0;
  *ltrim_out = (int32_t)ltrim; *rtrim_out = (int32_t)rtrim;
  return (ltrim + rtrim <= aln->seq_len) ? LTR_OK : LTR_ERR_INVALID;         // assert, :463
}
static int trim_alignment(const ltr_alignment* aln, int32_t repeat_start, int32_t repeat_end, int32_t padding,
                          int32_t* ltrim_out, int32_t* rtrim_out) {
  std::vector<int32_t> rem((size_t)std::max(aln->n_cigar, 1));
  return trim_alignment_into(aln, repeat_start, repeat_end, padding, rem.data(), ltrim_out, rtrim_out);
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
  if (members.size() == 2) {                                          // upper median of two: the larger
    for (int32_t i = 0; i < len; ++i) { const char x = (char)members[0]->qual[i], y = (char)members[1]->qual[i]; out[(size_t)i] = (uint8_t)(x < y ? y : x); }
    return out;
  }
  for (int32_t i = 0; i < len; ++i) {
    col.clear();
    for (const ltr_alignment* m : members) col.push_back((uint8_t)(char)m->qual[i]);
    std::sort(col.begin(), col.end(), [](uint8_t x, uint8_t y) { return (char)x < (char)y; });
    out[(size_t)i] = col[col.size() / 2];
  }
  return out;
}

// SeqStutterGenotyper::calc_hap_aln_probs (seq_stutter_genotyper.cpp:514-563) for MANY loci in one GPU pass.
//
// Per locus, like the reference: pool the reads (ReadPooler, read_pooler.cpp:3-20: exact sequence, the pool keeps the
// FIRST read's start/stop/CIGAR), trim each pool (HapAligner::trim_alignment), score every pool x haplotype pair, fan
// the pool rows out to the reads and sum mate-pair rows (:526-559).  Period-1 loci under --stutter-align-len take the
// short path with the pools' median base qualities (ReadPooler::pool, read_pooler.h:42-48).
//
// What is added here -- none of it changes a bit of the result:
//  * the long-path score of a pair is a function of the TRIMMED read's bytes and the haplotype's alone
//    (HapAligner.cpp:236-343), and pools that differ only outside the trimmed window (a sequencing error in the
//    +-200 bp of flank a HiFi read carries) trim to the same bytes: the pools of a locus are de-duplicated by their
//    trimmed bytes, each distinct trimmed read is scored once, and its row is copied to every pool that shares it
//    (30x HiFi over a 20-bp repeat: ~18 pools, ~5 distinct trimmed reads);
//  * no per-locus heap traffic: the per-read / per-pool results of a chunk live in flat arrays indexed by the prefix sum
//    of the loci's read counts, hash tables and CIGAR scratch are per worker thread;
//  * chunks of loci: while the GPU scores chunk c the host cores prepare chunk c+1.
namespace {

// Keys of the pooling tables: equal bytes -> equal hash is all that is needed (a hit is confirmed by memcmp).  Four
// independent multiply-xor lanes over 32-byte blocks: one lane's chain (load, xor, 64-bit multiply, shift-xor) is ~6 cycles
// per 8 bytes, and hashing the 390 MB of raw reads of a 30 000-locus call with ONE chain was the longest host phase of
// ltr_calc_hap_aln_probs (3.3 ms of 9 per 10 000-locus chunk on 16 cores).
inline uint64_t hash_bytes(const uint8_t* p, int64_t len) {
  constexpr uint64_t k0 = 0xFF51AFD7ED558CCDull, k1 = 0xC4CEB9FE1A85EC53ull, k2 = 0x9E3779B97F4A7C15ull, k3 = 0xD6E8FEB86659FD93ull;
  uint64_t h0 = k2 ^ (uint64_t)len, h1 = k0, h2 = k1, h3 = k3;
  int64_t k = 0;
  for (; k + 32 <= len; k += 32) {
    uint64_t w0, w1, w2, w3;
    std::memcpy(&w0, p + k, 8); std::memcpy(&w1, p + k + 8, 8); std::memcpy(&w2, p + k + 16, 8); std::memcpy(&w3, p + k + 24, 8);
    h0 = (h0 ^ w0) * k0; h0 ^= h0 >> 32;
    h1 = (h1 ^ w1) * k1; h1 ^= h1 >> 32;
    h2 = (h2 ^ w2) * k2; h2 ^= h2 >> 32;
    h3 = (h3 ^ w3) * k3; h3 ^= h3 >> 32;
  }
  uint64_t h = ((h0 * k1) ^ (h1 >> 29)) + ((h2 * k3) ^ (h3 >> 31)) + (h1 << 17) + h3;
  for (; k + 8 <= len; k += 8) { uint64_t w; std::memcpy(&w, p + k, 8); h = (h ^ w) * k0; h ^= h >> 32; }
  uint64_t w = 0;
  if (k < len) std::memcpy(&w, p + k, (size_t)(len - k));
  h = (h ^ w) * k1; h ^= h >> 29;
  return h;
}

struct WorkerScratch {                     // one per host thread, kept between loci and calls
  std::vector<int32_t> slot;               // open-addressing table: -> item index, -1 empty
  std::vector<int32_t> used;               // slots written for the current locus (reset list)
  std::vector<uint64_t> hashes;
  std::vector<int32_t> cigar_rem;
  std::vector<int32_t> counts;             // haplotype_counts
  void table(size_t n_items) {
    size_t cap = 64;
    while (cap < n_items * 2) cap <<= 1;
    if (slot.size() < cap) slot.assign(cap, -1);
    if (hashes.size() < n_items) hashes.resize(n_items);
  }
  void reset() { for (int32_t at : used) slot[(size_t)at] = -1; used.clear(); }
};

struct LocusInfo {
  int32_t rc = LTR_OK; const char* err = nullptr;
  int32_t rb = -1, P = 0, U = 0; int64_t H = 0;
  bool short_path = false, simple_hap = false;
  int64_t rbytes = 0, hbytes = 0;          // trimmed bytes of the distinct reads / haplotype string bytes
  int64_t ubase = 0, hbase = 0, rbyte0 = 0, hbyte0 = 0, ll0 = 0;   // prefix sums inside the chunk (long-path loci only)
};

}  // namespace

int ltr_calc_hap_aln_probs(ltr_ctx* ctx, const ltr_locus* loci, int64_t n_loci,
                           double* const* log_aln_probs, int32_t* const* seed_positions) {
  if (!ctx || (!loci && n_loci > 0) || n_loci < 0 || !log_aln_probs || !seed_positions) return LTR_ERR_INVALID;
  ltr::TimedCall timed(ctx, ltr::kTimerHapAln);                        // total_hap_aln_time_, seq_stutter_genotyper.cpp:515,:561-562
  LTR_GUARD_BEGIN
  // one call at a time per context (the chunks are staged in the context's host arrays): a second host thread waits here
  const std::unique_lock<std::mutex> call_lock = ltr::ctx_call_lock(ctx);
  const ltr_align_params prm = ltr::ctx_params(ctx);
  const ltr::DebugKnobs knobs = ltr::ctx_debug(ctx);
  const bool dbg = knobs.trace != 0;
  const auto t_start = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
#define LTR_TRACE(...) do { if (dbg) { std::fprintf(stderr, "[ltr] calc_hap_aln_probs %8.2f ms: ", since()); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } } while (0)

  // ---- validation: the loci here, their alignment records where they are first read (prepare(), on all host cores) ----
  std::vector<int64_t> read_base((size_t)n_loci + 1, 0);              // prefix sum of the loci's read counts
  for (int64_t l = 0; l < n_loci; ++l) {
    const ltr_locus& L = loci[l];
    if (!L.hap || (!L.alns && L.n_alns > 0) || L.n_alns < 0 || !log_aln_probs[l] || !seed_positions[l]) return LTR_ERR_INVALID;
    read_base[(size_t)l + 1] = read_base[(size_t)l] + L.n_alns;
  }
  const int64_t R_total = read_base[(size_t)n_loci];
  // per read: its pool; per pool (stored at the locus' read base + pool number): first read, trim, distinct trimmed read
  std::unique_ptr<int32_t[]> pool_index(new int32_t[(size_t)std::max<int64_t>(R_total, 1)]);
  std::unique_ptr<int32_t[]> pool_first(new int32_t[(size_t)std::max<int64_t>(R_total, 1)]);
  std::unique_ptr<int32_t[]> pool_lt(new int32_t[(size_t)std::max<int64_t>(R_total, 1)]);      // ltrim; -1: empty trim -> the 5 + 5 flank bases (:820-823)
  std::unique_ptr<int32_t[]> pool_len(new int32_t[(size_t)std::max<int64_t>(R_total, 1)]);     // trimmed length
  std::unique_ptr<int32_t[]> pool_uniq(new int32_t[(size_t)std::max<int64_t>(R_total, 1)]);    // distinct trimmed read of the pool, -1: not realigned
  std::unique_ptr<int32_t[]> uniq_pool(new int32_t[(size_t)std::max<int64_t>(R_total, 1)]);    // first pool of every distinct trimmed read
  std::vector<LocusInfo> info((size_t)n_loci);
  LTR_TRACE("validated %ld loci, %ld reads", (long)n_loci, (long)R_total);

  // a period-1 locus under use_short_path: prepared like the others on the host's cores -- pools' median qualities, its own
  // little batch of the seeded path (ltr_short.hip) -- and strung onto the call's batch in locus order
  struct ShortBatchDel { void operator()(ltr::ShortBatch* p) const { ltr::short_batch_free(p); } };
  struct ShortLocus {
    int64_t locus = 0, H = 0; std::vector<double> pool_probs; std::vector<int32_t> pool_seeds;
    std::unique_ptr<ltr::ShortBatch, ShortBatchDel> batch;
  };
  std::vector<std::unique_ptr<ShortLocus>> short_of((size_t)n_loci);  // (heap objects: the queued result pointers stay valid)
  std::vector<ShortLocus*> short_loci;                                // ... in locus order
  std::unique_ptr<ltr::ShortBatch, ShortBatchDel> short_batch;

  // ---- per locus, on all host cores: pools, trims, distinct trimmed reads, sizes ---------------------
  auto prepare = [&](int64_t l) {
    static thread_local WorkerScratch W;
    const ltr_locus& L = loci[l];
    LocusInfo& I = info[(size_t)l];
    const int64_t rb0 = read_base[(size_t)l];
    // (checked here, chunk by chunk, not in a pass of its own over the 900 000 records of a 30 000-locus call before anything else
    // starts -- 1.2 ms with the GPU idle; nothing is written to the caller's matrices before every chunk is through here, except
    // the rows of short-path loci, as with every other error prepare() finds)
    for (int32_t i = 0; i < L.n_alns; ++i)
      if (L.alns[i].seq_len < 0 || (L.alns[i].seq_len > 0 && !L.alns[i].seq) || L.alns[i].n_cigar < 0 ||
          (L.alns[i].n_cigar > 0 && (!L.alns[i].cigar_type || !L.alns[i].cigar_num))) {
        I.err = "alignment with a negative length or a null sequence / CIGAR pointer"; I.rc = LTR_ERR_INVALID; return;
      }
    for (int b = 0; b < L.hap->n_blocks; ++b) if (L.hap->is_repeat[b]) { I.rb = b; break; }
    if (L.hap->n_blocks <= 0 || I.rb < 0) { I.err = "haplotype has no repeat block"; I.rc = LTR_ERR_INVALID; return; }
    // ReadPooler::add_alignment: pools keyed by the exact sequence, numbered by first occurrence
    W.table((size_t)std::max(L.n_alns, 1));
    int32_t P = 0;
    const size_t mask = W.slot.size() - 1;
    for (int32_t i = 0; i < L.n_alns; ++i) {
      const ltr_alignment& A = L.alns[i];
      const uint64_t h = W.hashes[(size_t)i] = hash_bytes(A.seq, A.seq_len);
      for (size_t at = (size_t)h & mask;; at = (at + 1) & mask) {
        const int32_t f = W.slot[at];
        if (f < 0) { W.slot[at] = i; W.used.push_back((int32_t)at); pool_first[(size_t)(rb0 + P)] = i; pool_index[(size_t)(rb0 + i)] = P++; break; }
        if (W.hashes[(size_t)f] == h && L.alns[f].seq_len == A.seq_len && (A.seq_len == 0 || std::memcmp(L.alns[f].seq, A.seq, (size_t)A.seq_len) == 0)) {
          pool_index[(size_t)(rb0 + i)] = pool_index[(size_t)(rb0 + f)]; break;
        }
      }
    }
    W.reset();
    I.P = P;
    I.short_path = prm.use_short_path && L.hap->n_blocks > 1 && L.hap->period[1] == 1;      // HapAligner.cpp:552
    if (I.short_path) {
      // per-locus short path on the pooled alignments (ReadPooler::pool: the pool's reads with their median qualities, read_pooler.h:42-48)
      std::vector<ltr_alignment> pooled((size_t)P);
      std::vector<std::vector<uint8_t>> quals((size_t)P);
      for (int32_t q = 0; q < P; ++q) {
        pooled[(size_t)q] = L.alns[pool_first[(size_t)(rb0 + q)]];
        std::vector<const ltr_alignment*> members;
        for (int32_t i = 0; i < L.n_alns; ++i) if (pool_index[(size_t)(rb0 + i)] == q) members.push_back(&L.alns[i]);
        for (const ltr_alignment* m : members) if (!m->qual) { I.err = "short path needs base qualities"; I.rc = LTR_ERR_INVALID; return; }
        if (members.size() == 1) continue;                            // (a pool of one read: its own qualities, already in place)
        quals[(size_t)q] = median_qualities(members);
        pooled[(size_t)q].qual = quals[(size_t)q].data();
      }
      std::unique_ptr<ShortLocus> SL(new ShortLocus());
      SL->locus = l; SL->H = ltr_haplotype_num_combs(L.hap);
      SL->pool_probs.assign((size_t)P * (size_t)SL->H, 0.0); SL->pool_seeds.assign((size_t)P, 0);
      SL->batch.reset(ltr::short_batch_new());
      // queued: every short-path locus of the call is scored in ONE set of launches after the chunks are on their way
      const int rc2 = ltr::short_batch_add(ctx, SL->batch.get(), L.hap, L.realign_to_hap, pooled.data(), P, 0, L.realign_pool,
                                           SL->pool_probs.data(), SL->pool_seeds.data());
      if (rc2 != LTR_OK) { I.rc = rc2; I.err = nullptr; return; }       // (the message is the one short_batch_add left in the context)
      short_of[(size_t)l] = std::move(SL);
      return;
    }
    // haplotypes: count and total length.  One multi-allele block (every locus the genotyper builds: [flank][repeat][flank])
    // means haplotype k == allele k of that block (Haplotype.cpp:151-206)
    {
      int multi = 0; int64_t H = 1, fixed = 0;
      for (int b = 0; b < L.hap->n_blocks; ++b) {
        if (L.hap->n_alleles[b] <= 0) { I.err = "bad haplotype block structure"; I.rc = LTR_ERR_INVALID; return; }
        H *= L.hap->n_alleles[b];
        if (H > (1 << 24)) { I.err = "bad haplotype block structure"; I.rc = LTR_ERR_INVALID; return; }
        if (L.hap->n_alleles[b] > 1) ++multi;
      }
      I.H = H; I.simple_hap = (multi <= 1);
      if (I.simple_hap) {
        int64_t k = 0, var = 0;
        for (int b = 0; b < L.hap->n_blocks; ++b) {
          const int na = L.hap->n_alleles[b];
          if (na == 1) fixed += L.hap->allele_off[k + 1] - L.hap->allele_off[k];
          else var = L.hap->allele_off[k + na] - L.hap->allele_off[k];
          k += na;
        }
        I.hbytes = fixed * H + var;
      } else {
        int64_t nc = 0;
        if (ltr::haplotype_counts(L.hap, &W.counts, &nc) != LTR_OK) { I.err = "bad haplotype block structure"; I.rc = LTR_ERR_INVALID; return; }
        int64_t tot = 0;
        for (int64_t c = 0; c < nc; ++c)
          for (int b = 0; b < L.hap->n_blocks; ++b) { const int64_t a = ltr::allele_slot(L.hap, b, W.counts[(size_t)(c * L.hap->n_blocks + b)]); tot += L.hap->allele_off[a + 1] - L.hap->allele_off[a]; }
        I.hbytes = tot;
      }
    }
    // trims (HapAligner.cpp:819), then the distinct trimmed reads among the realigned pools
    const int64_t aL = ltr::allele_slot(L.hap, L.hap->n_blocks - 1, 0);
    const int64_t l0 = L.hap->allele_off[1] - L.hap->allele_off[0], lL = L.hap->allele_off[aL + 1] - L.hap->allele_off[aL];
    const int32_t sub_len = (int32_t)(5 + std::min<int64_t>(lL, 5));
    W.table((size_t)std::max(P, 1));
    int32_t U = 0, sub_uniq = -1;
    int64_t rbytes = 0;
    for (int32_t q = 0; q < P; ++q) {
      const size_t qa = (size_t)(rb0 + q);
      pool_uniq[qa] = -1; pool_lt[qa] = 0; pool_len[qa] = 0;
      if (L.realign_pool && !L.realign_pool[q]) continue;              // not realigned: no pair, its rows stay as they are
      const ltr_alignment& A = L.alns[pool_first[qa]];
      if ((size_t)std::max(A.n_cigar, 1) > W.cigar_rem.size()) W.cigar_rem.resize((size_t)A.n_cigar * 2);
      int32_t lt = 0, rt = 0;
      const int rc = ltr::trim_alignment_into(&A, L.hap->block_start[I.rb], L.hap->block_end[I.rb], prm.indel_flank_len, W.cigar_rem.data(), &lt, &rt);
      if (rc != LTR_OK) {
        I.err = rc == LTR_ERR_CIGAR ? "Invalid CIGAR option encountered in trim_alignment" : "trim_alignment: ltrim+rtrim exceeds the read length";
        I.rc = rc; W.reset(); return;
      }
      const int32_t len = A.seq_len - lt - rt;
      if (len <= 0) {
        // empty trim: the last 5 bp of the first block's reference allele + the first 5 bp of the last block's (:820-823)
        if (l0 < 5) { I.err = "left flank shorter than 5 bp (std::string::substr would throw in the reference)"; I.rc = LTR_ERR_INVALID; W.reset(); return; }
        pool_lt[qa] = -1; pool_len[qa] = sub_len;
        if (sub_uniq < 0) { sub_uniq = U; uniq_pool[(size_t)(rb0 + U)] = q; ++U; rbytes += sub_len; }
        pool_uniq[qa] = sub_uniq;
        continue;
      }
      pool_lt[qa] = lt; pool_len[qa] = len;
      const uint8_t* tb = A.seq + lt;
      const uint64_t h = W.hashes[(size_t)q] = hash_bytes(tb, len);
      for (size_t at = (size_t)h & mask;; at = (at + 1) & mask) {
        const int32_t f = W.slot[at];                                   // -> a pool whose trimmed read is distinct so far
        if (f < 0) { W.slot[at] = q; W.used.push_back((int32_t)at); uniq_pool[(size_t)(rb0 + U)] = q; pool_uniq[qa] = U++; rbytes += len; break; }
        const size_t fa = (size_t)(rb0 + f);
        if (W.hashes[(size_t)f] == h && pool_len[fa] == len && std::memcmp(L.alns[pool_first[fa]].seq + pool_lt[fa], tb, (size_t)len) == 0) { pool_uniq[qa] = pool_uniq[fa]; break; }
      }
    }
    W.reset();
    I.U = U; I.rbytes = rbytes;
  };

  // ---- chunks of loci: while the GPU scores chunk c the host prepares chunk c+1 ----
  struct Chunk {
    int64_t l0 = 0, l1 = 0;                     // loci [l0, l1)
    std::vector<int64_t> read_off, hap_off, lro, lho;
    std::vector<uint8_t> mask_h;
    std::vector<int64_t> slot_locus;            // long-path loci of the chunk, in order
    ltr_plan* plan = nullptr;
    std::unique_ptr<double[]> ll;
    // what staging the chunk leaves for the calling thread
    int64_t n_u = 0, n_h = 0, n_rb = 0, n_hb = 0;
    bool any_mask = false;
    uint8_t* read_bytes = nullptr; uint8_t* hap_bytes = nullptr;
    std::vector<int64_t> short_l;               // short-path loci of the chunk before its first error, in order
    int rc = LTR_OK; const char* err = nullptr; // the chunk's first error in locus order
  };
  // Two chunks, 1 : 3 -- the GPU starts on the first quarter while the host cores prepare the rest; the plans
  // run on two streams, so the tail of the first plan's launches overlaps the head of the second's.
  // Measured on MI355X (round 2), 6000 raw config-3 loci: one plan 210.5 ms per call; 1 : 1 202.7; 1 : 2 197.6; 1 : 3
  // 194.9; three chunks 1 : 2 : 3 202.9; eight chunks on three streams 233 (every plan is a chain of launches, each at
  // least as long as its longest pair: small plans leave the GPU part empty).  1000 loci: one plan 45.6 ms, two 46.5.
  // (ltr_ctx_set_debug "chunks" / "chunk_streams" / "chunk_growth" override the rule: tests/manual/gpu_chunk_sweep.py.)
  // (When the call is host-bound -- a catalogue of short repeats: ~1 microsecond of host work per locus, a few hundred
  // nanoseconds of DP -- three equal chunks keep the GPU fed: 30 000 catalogue loci 42.8 ms as 1 : 3, 37.8 ms as 1 : 1 : 1;
  // config 3 / config3skew, where the DP is the longer side, lose 4 - 8 % that way.  The split is decided on an estimate of
  // both sides from what is known before any read is touched: reads, alleles and allele lengths.)
  int64_t n_chunks = n_loci >= 1500 ? 2 : 1;
  double growth_rule = 3.0;
  if (n_loci >= 6000) {
    double cells = 0.0;
    for (int64_t l = 0; l < n_loci; l += 16) {                        // every 16th locus
      const ltr_haplotype_blocks* hb = loci[l].hap;
      if (!hb || hb->n_blocks <= 0) continue;
      int64_t hap_len = 0, H = 1, k = 0;
      bool ok = hb->n_alleles && hb->allele_off;                      // (an estimate made before prepare() validates the blocks: a malformed locus is skipped here and rejected there)
      for (int b = 0; ok && b < hb->n_blocks; ++b) {
        const int na = hb->n_alleles[b];
        if (na <= 0 || na > (1 << 24) || H > (1 << 24)) { ok = false; break; }
        hap_len += hb->allele_off[k + 1] - hb->allele_off[k]; H *= na; k += na;
      }
      if (!ok) continue;
      const double side = (double)std::max<int64_t>(hap_len - 60, 1);
      cells += 16.0 * (double)std::max(loci[l].n_alns, 1) / 3.0 * (double)std::min<int64_t>(H, 1 << 20) * side * side;   // (about a third of the reads survive pooling + trimming)
    }
    const double gpu_s = cells / 2.5e12, host_s = (double)n_loci * 0.8e-6;
    if (gpu_s < 1.5 * host_s) {
      n_chunks = 3; growth_rule = 1.0;
      // (Round 5: with the next chunk staged ahead by a thread of its own, DMA-only uploads and a cheaper plan creation the host
      // side of a chunk is SHORTER than its GPU side -- 0.45 + 0.27 microseconds per locus on two threads against 0.65 -- and the
      // lead-in, the first chunk's staging + planning with the GPU idle, is what is left to shorten: a first chunk of ~2400 loci,
      // every next one 1.3 x longer (the growth at which staging + planning chunk c + 1 still fits under chunk c's launch).
      // Measured on MI355X, 30 000 catalogue loci, tests/manual/gpu_chunk_sweep_ahead.py: three equal chunks 30.3 - 31.5 ms per
      // call; 4 / 5 / 6 / 8 chunks at 1.3: 26.5 - 27.8 / 26.2 - 26.4 / 25.7 - 26.0 / 27.2 - 27.6; growth 1.5 - 1.6: 27.9 - 30.2;
      // without the thread three equal chunks stay the best, 30.1 - 33.4 against 33.7 - 34.3 for 4 - 5 chunks at 1.3.)
      if (knobs.prep_ahead > 0 || (knobs.prep_ahead == 0 && ltr::host_thread_budget() >= ltr::kPrepAheadMinThreads)) {
        growth_rule = 1.3;
        while (n_chunks < 8 && 2400.0 * (std::pow(1.3, (double)n_chunks) - 1.0) / 0.3 < (double)n_loci) ++n_chunks;
      }
    }
  }
  int n_streams = 2;
  if (knobs.chunks > 0) n_chunks = std::max<int64_t>(1, std::min<int64_t>(knobs.chunks, std::max<int64_t>(n_loci, 1)));
  if (knobs.chunk_streams > 0) n_streams = knobs.chunk_streams;
  std::vector<Chunk> chunks((size_t)n_chunks);
  std::vector<double> cum((size_t)n_chunks + 1, 0.0);                 // cumulative chunk weights
  {
    double growth = growth_rule;                                      // 0: weights 1, 2, 3, ...; g > 0: 1, g, g^2, ...; g < 0: 1, 2, .., k, k, .., 2, 1
    if (knobs.chunk_growth_set) growth = knobs.chunk_growth;
    double w = 1.0;
    for (int64_t c = 0; c < n_chunks; ++c) {
      const double wc = growth > 0.0 ? w : (growth < 0.0 ? (double)(std::min(c, n_chunks - 1 - c) + 1) : (double)(c + 1));
      cum[(size_t)c + 1] = cum[(size_t)c] + wc; w *= growth;
    }
  }
  int rc = LTR_OK;
  auto cleanup = [&]() { for (Chunk& C : chunks) if (C.plan) { ltr_plan_destroy(C.plan); C.plan = nullptr; } };
  for (int64_t c = 0; c < n_chunks; ++c) {
    Chunk& C = chunks[(size_t)c];
    C.l0 = (int64_t)((double)n_loci * cum[(size_t)c] / cum[(size_t)n_chunks]);
    C.l1 = (c + 1 == n_chunks) ? n_loci : (int64_t)((double)n_loci * cum[(size_t)c + 1] / cum[(size_t)n_chunks]);
  }
  // Chunk c + 1 is pooled, trimmed and laid out by a thread of its own (with the second worker pool, into the second pair of staging
  // arrays) WHILE the calling thread plans and launches chunk c: planning has serial stretches (prefix sums, the sort's merge, the
  // uploads) that leave the host cores idle, and on a catalogue of short repeats the host, not the GPU, is the longer side of every
  // chunk.  Measured on MI355X, 30 000 catalogue loci (tests/manual/gpu_prep_ahead_ab.py): profiles/r05/e2e_prep_ahead.log.
  // Round 6: only from a host-thread budget of 12 up (ltr_ctx_set_host_threads; rule: affinity mask, cgroup quota, ranks on this
  // host) -- two thread teams on four or eight cores are slower than one (same log: 55.8 / 35.0 ms against 30.2 with the helper off).
  const int budget = ltr::host_thread_budget();
  const bool prep_ahead = n_chunks > 1 && (knobs.prep_ahead > 0 || (knobs.prep_ahead == 0 && budget >= ltr::kPrepAheadMinThreads));
  const int ahead_threads = knobs.prep_ahead > 0 ? knobs.prep_ahead : budget;
  // (set when the call is on its way out with an error: a chunk being staged ahead stops pooling and trimming loci nobody will score)
  std::atomic<bool> cancel(false);
  auto stage_chunk = [&](Chunk& C, const int64_t c, const int pool, const int threads) {
    ltr::parallel_for(C.l1 - C.l0, 64, [&](int64_t k) { if (!cancel.load(std::memory_order_relaxed)) prepare(C.l0 + k); }, 32, pool, threads);
    if (cancel.load(std::memory_order_relaxed)) { C.rc = LTR_ERR_INVALID; return; }
    LTR_TRACE("chunk %ld: %ld loci pooled + trimmed", (long)c, (long)(C.l1 - C.l0));
    // in locus order: the first error ends the chunk; short-path loci are noted for the calling thread; prefix sums place the rest
    int64_t n_u = 0, n_h = 0, n_rb = 0, n_hb = 0, n_ll = 0;
    bool any_mask = false;
    for (int64_t l = C.l0; l < C.l1; ++l) {
      const ltr_locus& L = loci[l];
      LocusInfo& I = info[(size_t)l];
      if (I.rc != LTR_OK) { C.err = I.err; C.rc = I.rc; return; }
      if (I.short_path) { C.short_l.push_back(l); continue; }
      I.ubase = n_u; I.hbase = n_h; I.rbyte0 = n_rb; I.hbyte0 = n_hb; I.ll0 = n_ll;
      n_u += I.U; n_h += I.H; n_rb += I.rbytes; n_hb += I.hbytes; n_ll += (int64_t)I.U * I.H;
      any_mask |= (L.realign_to_hap != nullptr);
      C.slot_locus.push_back(l);
    }
    C.n_u = n_u; C.n_h = n_h; C.n_rb = n_rb; C.n_hb = n_hb; C.any_mask = any_mask;
    if (C.slot_locus.empty()) return;
    // ---- the chunk's batch: bytes and offsets written in place, all cores ----
    uint8_t* read_bytes = C.read_bytes = ltr::ctx_host_bytes(ctx, 2 * (int)(c & 1), (size_t)std::max<int64_t>(n_rb, 1));
    uint8_t* hap_bytes = C.hap_bytes = ltr::ctx_host_bytes(ctx, 2 * (int)(c & 1) + 1, (size_t)std::max<int64_t>(n_hb, 1));
    const int64_t n_slots = (int64_t)C.slot_locus.size();
    C.read_off.resize((size_t)n_u + 1); C.hap_off.resize((size_t)n_h + 1); C.lro.resize((size_t)n_slots + 1); C.lho.resize((size_t)n_slots + 1);
    if (any_mask) C.mask_h.assign((size_t)n_h, 1);
    C.read_off[(size_t)n_u] = n_rb; C.hap_off[(size_t)n_h] = n_hb; C.lro[(size_t)n_slots] = n_u; C.lho[(size_t)n_slots] = n_h;
    ltr::parallel_for(n_slots, 64, [&](int64_t k) {
      static thread_local WorkerScratch W;
      const int64_t l = C.slot_locus[(size_t)k];
      const ltr_locus& L = loci[l];
      const LocusInfo& I = info[(size_t)l];
      const int64_t rb0 = read_base[(size_t)l];
      C.lro[(size_t)k] = I.ubase; C.lho[(size_t)k] = I.hbase;
      int64_t at = I.rbyte0;
      for (int32_t u = 0; u < I.U; ++u) {
        const size_t qa = (size_t)(rb0 + uniq_pool[(size_t)(rb0 + u)]);
        C.read_off[(size_t)(I.ubase + u)] = at;
        if (pool_lt[qa] >= 0) std::memcpy(read_bytes + at, L.alns[pool_first[qa]].seq + pool_lt[qa], (size_t)pool_len[qa]);
        else {
          const int64_t aL = ltr::allele_slot(L.hap, L.hap->n_blocks - 1, 0);
          const int64_t l0 = L.hap->allele_off[1] - L.hap->allele_off[0];
          std::memcpy(read_bytes + at, L.hap->allele_bytes + L.hap->allele_off[0] + l0 - 5, 5);
          std::memcpy(read_bytes + at + 5, L.hap->allele_bytes + L.hap->allele_off[aL], (size_t)(pool_len[qa] - 5));
        }
        at += pool_len[qa];
      }
      // haplotype strings in Haplotype::next() order (Haplotype::get_seq(), Haplotype.h:99-104)
      int64_t hat = I.hbyte0;
      const int nb = L.hap->n_blocks;
      if (!I.simple_hap) { int64_t nc = 0; (void)ltr::haplotype_counts(L.hap, &W.counts, &nc); }
      for (int64_t h = 0; h < I.H; ++h) {
        C.hap_off[(size_t)(I.hbase + h)] = hat;
        int64_t slot0 = 0;
        for (int b = 0; b < nb; ++b) {
          const int na = L.hap->n_alleles[b];
          const int a = I.simple_hap ? (na > 1 ? (int)h : 0) : W.counts[(size_t)(h * nb + b)];
          const int64_t s0 = L.hap->allele_off[slot0 + a], s1 = L.hap->allele_off[slot0 + a + 1];
          std::memcpy(hap_bytes + hat, L.hap->allele_bytes + s0, (size_t)(s1 - s0));
          hat += s1 - s0; slot0 += na;
        }
        if (any_mask && L.realign_to_hap && !L.realign_to_hap[h]) C.mask_h[(size_t)(I.hbase + h)] = 0;
      }
    }, 32, pool, threads);
    LTR_TRACE("chunk %ld: batch of %ld distinct trimmed reads (%ld B), %ld haplotypes (%ld B) laid out", (long)c, (long)n_u, (long)n_rb, (long)n_h, (long)n_hb);
  };
  // the chunks' plans go whichever way the call ends (return, error, an exception out of the helper thread or a worker)
  struct PlanGuard { decltype(cleanup)& fn; ~PlanGuard() { fn(); } } plan_guard{cleanup};
  struct Ahead {
    std::thread th; std::exception_ptr err; std::atomic<bool>* cancel = nullptr; bool done = false;
    void join() { if (th.joinable()) th.join(); }
    ~Ahead() { if (!done && cancel) cancel->store(true); join(); }   // (an early exit: the thread stops at its next locus)
  } ahead;                                                           // (declared last: joined before anything its thread uses goes away)
  ahead.cancel = &cancel;
  if (prep_ahead) stage_chunk(chunks[0], 0, 0, budget);
  for (int64_t c = 0; c < n_chunks && rc == LTR_OK; ++c) {
    Chunk& C = chunks[(size_t)c];
    if (prep_ahead) {
      ahead.join();
      if (ahead.err) std::rethrow_exception(ahead.err);
      if (c + 1 < n_chunks)
        ahead.th = std::thread([&, c]() { try { stage_chunk(chunks[(size_t)c + 1], c + 1, 1, ahead_threads); } catch (...) { ahead.err = std::current_exception(); } });
    } else stage_chunk(C, c, 0, budget);
    // in locus order: the short-path loci before the chunk's first error queue up, then the error, if any
    for (const int64_t l : C.short_l) {
      if (!short_batch) short_batch.reset(ltr::short_batch_new());
      ShortLocus* SL = short_of[(size_t)l].get();
      rc = ltr::short_batch_merge(ctx, short_batch.get(), SL->batch.get());
      SL->batch.reset();
      short_loci.push_back(SL);
      if (rc != LTR_OK) break;
    }
    if (rc == LTR_OK && C.rc != LTR_OK) { if (C.err) ltr::set_error(ctx, C.err); rc = C.rc; }
    if (rc != LTR_OK || C.slot_locus.empty()) continue;
    const int64_t n_slots = (int64_t)C.slot_locus.size(), n_u = C.n_u, n_h = C.n_h;
    uint8_t* read_bytes = C.read_bytes; uint8_t* hap_bytes = C.hap_bytes;
    const bool any_mask = C.any_mask;
    ltr_locus_batch b;
    std::memset(&b, 0, sizeof(b));
    b.n_loci = n_slots; b.locus_read_off = C.lro.data(); b.locus_hap_off = C.lho.data();
    b.n_reads = n_u; b.read_bytes = read_bytes; b.read_off = C.read_off.data();
    b.n_haps = n_h; b.hap_bytes = hap_bytes; b.hap_off = C.hap_off.data();
    if (any_mask) b.realign_hap = C.mask_h.data();
    rc = ltr_plan_create(ctx, &b, &C.plan);
    LTR_TRACE("chunk %ld: planned (%ld pairs)", (long)c, C.plan ? (long)ltr_plan_num_pairs(C.plan) : 0L);
    // asynchronous: returns once the launches are queued.  Chunks alternate between two streams: the first
    // kernels of chunk c+1 run next to the exact kernels and the tail of chunk c.
    if (rc == LTR_OK) rc = ltr_plan_execute(C.plan, nullptr, ltr::ctx_side_stream(ctx, (int)(c % n_streams)));
    LTR_TRACE("chunk %ld: launches queued", (long)c);
  }
  if (rc != LTR_OK) return rc;                                       // (Ahead's destructor cancels and joins the helper, PlanGuard destroys the plans)
  ahead.join(); ahead.done = true;
  if (short_batch) {
    LTR_TRACE("short path: %ld loci queued", (long)short_loci.size());
    rc = ltr::short_batch_run(ctx, short_batch.get());
    LTR_TRACE("short path: scored");
    for (ShortLocus* SLp : short_loci) {
      ShortLocus& SLc = *SLp;
      if (rc != LTR_OK) break;
      const ltr_locus& L = loci[SLc.locus];
      rc = ltr_scatter_pool_probs(SLc.pool_probs.data(), SLc.pool_seeds.data(), pool_index.get() + read_base[(size_t)SLc.locus], L.n_alns,
                                  (int32_t)SLc.H, L.realign_to_hap, L.copy_read, L.second_mate, log_aln_probs[SLc.locus], seed_positions[SLc.locus]);
    }
    if (rc != LTR_OK) { cleanup(); return rc; }
    LTR_TRACE("short path: rows fanned out");
  }
  // ---- in chunk order: rows of chunk c are fanned out to its reads while the later chunks still run ----
  for (Chunk& C : chunks) {
    if (!C.plan) continue;
    C.ll.reset(new double[(size_t)std::max<int64_t>(ltr_plan_ll_size(C.plan), 1)]);
    rc = ltr_plan_fetch(C.plan, C.ll.get(), nullptr);                          // waits for THIS plan's kernels only
    LTR_TRACE("a chunk's rows fetched");
    // (the plan is destroyed with the others at the end: releasing its buffers waits for the streams it ran on, and a later
    // chunk shares its stream -- the rows of chunk c would be fanned out only after chunk c + 2 has finished on the GPU)
    if (rc != LTR_OK) break;
    std::atomic<int> first_rc(LTR_OK);
    ltr::parallel_for((int64_t)C.slot_locus.size(), 128, [&](int64_t k) {
      const int64_t l = C.slot_locus[(size_t)k];
      const ltr_locus& L = loci[l];
      const LocusInfo& I = info[(size_t)l];
      const int64_t rb0 = read_base[(size_t)l], H = I.H;
      const double* rows = C.ll.get() + I.ll0;                                 // [U x H]
      double* out = log_aln_probs[l];
      int32_t* seeds = seed_positions[l];
      for (int32_t i = 0; i < L.n_alns; ++i) {                                 // seq_stutter_genotyper.cpp:527-538
        if (L.copy_read && !L.copy_read[i]) continue;
        const int32_t q = pool_index[(size_t)(rb0 + i)];
        seeds[i] = L.alns[pool_first[(size_t)(rb0 + q)]].seq_len - 1;          // pool_seed_positions: HapAligner.cpp:562-563
        const int32_t u = pool_uniq[(size_t)(rb0 + q)];
        double* dst = out + (int64_t)H * i;
        if (u >= 0) {
          const double* src = rows + (int64_t)H * u;
          if (!L.realign_to_hap) std::memcpy(dst, src, (size_t)H * sizeof(double));
          else for (int64_t j = 0; j < H; ++j) if (L.realign_to_hap[j]) dst[j] = src[j];
        } else {                                                               // a pool that was not realigned: the reference copies an unwritten row, here zeros
          for (int64_t j = 0; j < H; ++j) if (!L.realign_to_hap || L.realign_to_hap[j]) dst[j] = 0.0;
        }
      }
      if (L.second_mate)                                                       // mate pairs share one row sum, :546-559
        for (int32_t i = 0; i < L.n_alns; ++i) {
          if (!L.second_mate[i] || (L.copy_read && !L.copy_read[i])) continue;
          if (i == 0) { int expect = LTR_OK; first_rc.compare_exchange_strong(expect, LTR_ERR_INVALID); return; }
          double* m1 = out + (int64_t)(i - 1) * H;
          double* m2 = out + (int64_t)i * H;
          for (int64_t j = 0; j < H; ++j)
            if (!L.realign_to_hap || L.realign_to_hap[j]) { const double tot = m1[j] + m2[j]; m1[j] = tot; m2[j] = tot; }
        }
    }, 32);
    rc = first_rc.load();
    if (rc != LTR_OK) break;
    C.ll.reset();
  }
  cleanup();
  LTR_TRACE("rows fanned out to the reads");
#undef LTR_TRACE
  return rc;
  LTR_GUARD_END(ctx)
}

// ---- host-thread budget (ltr_internal.h; reference README.md:78-82: one thread per process, N processes per node) ----
int ltr_ctx_set_host_threads(ltr_ctx* ctx, int n) {
  if (!ctx || n < 0 || n > (1 << 16)) { if (ctx) ltr::set_error(ctx, "ltr_ctx_set_host_threads: n >= 1 (the budget) or 0 (the rule)"); return LTR_ERR_INVALID; }
  ltr::host_thread_setting().store(n, std::memory_order_relaxed);
  return LTR_OK;
}
int ltr_ctx_host_threads(const ltr_ctx* ctx) { return ctx ? ltr::host_thread_budget() : LTR_ERR_INVALID; }
int ltr_host_threads_rule(int local_world_size) { return ltr::host_threads_rule(local_world_size); }

// test hooks (no GPU): the number of distinct threads a parallel_for of n items really ran on under budget `n_threads`
// (0: the rule), and whether ltr_calc_hap_aln_probs would use its helper thread under that budget
int ltr_debug_parallel_threads(int n_threads, int64_t n_items, int which_pool) {
  if (n_threads < 0 || n_items < 0) return LTR_ERR_INVALID;
  const int held = ltr::host_thread_setting().exchange(n_threads);
  std::mutex mu;
  std::vector<std::thread::id> seen;
  int out = LTR_ERR_INVALID;
  try {
    ltr::parallel_for(n_items, 1, [&](int64_t) {
      const std::thread::id me = std::this_thread::get_id();
      {
        std::lock_guard<std::mutex> lk(mu);
        if (std::find(seen.begin(), seen.end(), me) == seen.end()) seen.push_back(me);
      }
      std::this_thread::sleep_for(std::chrono::microseconds(200));       // (long enough for every thread of the team to take a share)
    }, 1, which_pool);
    out = (int)seen.size();
  } catch (...) { out = LTR_ERR_NOMEM; }
  ltr::host_thread_setting().store(held);
  return out;
}
int ltr_debug_prep_ahead_rule(int n_threads) {
  const int b = n_threads > 0 ? std::min(n_threads, ltr::kMaxHostThreads) : ltr::host_thread_budget();
  return b >= ltr::kPrepAheadMinThreads ? 1 : 0;
}

}  // extern "C"

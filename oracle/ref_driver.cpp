/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE.  Harness around the REAL reference code.
 *
 * oracle/Makefile compiles the reference's own translation units from where
 * they lie under $(LONGTR_REF)/src -- HapAligner.cpp (align_seq_to_hap,
 * trim_alignment), HapBlock.cpp, StutterAlignerClass.cpp, stutter_model.cpp,
 * mathops.cpp, base_quality.cpp, read_pooler.cpp, stringops.cpp, region.cpp,
 * error.cpp -- unmodified, with the reference Makefile's flags, and links them
 * with this file into oracle/_ref/libltr_ref.so.  Nothing from the reference is
 * copied into this repository and no header/library stand-ins are written:
 * the translation units that need htslib (Haplotype.cpp, NeedlemanWunsch.cpp,
 * genotyper.cpp, seq_stutter_genotyper.cpp ...) are simply NOT part of this
 * build, and the functions that depend on them (HapAligner::process_read's
 * do/while over Haplotype::next(), the HapAligner constructor's
 * Haplotype::reverse()) are dropped by --gc-sections because nothing here
 * calls them.
 *
 * What runs from the reference, bit for bit:
 *   - HapAligner::align_seq_to_hap   (src/SeqAlignment/HapAligner.cpp:236-343)
 *   - HapAligner::trim_alignment     (src/SeqAlignment/HapAligner.cpp:346-465)
 *   - ReadPooler::add_alignment      (src/read_pooler.cpp:3-20)
 *   - HapBlock / RepeatBlock containers, Haplotype::get_seq() (inline, Haplotype.h:99-104)
 *   - of the short (stutter) path: StutterAlignerClass (load_read, align_*_reverse), RepeatStutterInfo,
 *     StutterModel::log_stutter_pmf, BaseQuality, fast_log_sum_exp(vector) -- see ltr_ref_stutter_block_row;
 *     HapAligner::compute_aln_logprob (:165-233), calc_best_seed_position (:467-493), calc_seed_base (:494-542)
 * What this harness does itself (because Haplotype.cpp cannot be built):
 *   - builds the Haplotype / HapAligner OBJECTS by filling their fields
 *     directly instead of running constructors that call into Haplotype.cpp;
 *   - replays process_read's long-branch loop (HapAligner.cpp:818-852: trim,
 *     10-bp substitute for an empty trim, one align_seq_to_hap per allele) with
 *     the allele index set by hand; with one multi-allele block, haplotype k ==
 *     allele k (Haplotype.cpp:151-196).
 */
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <new>
#include <set>
#include <sstream>
#include <string>
#include <vector>

/* The two functions under test are private members; the harness needs to call
 * them directly.  Access control does not change object layout. */
#define private public
#define protected public
#include "SeqAlignment/HapAligner.h"
#include "SeqAlignment/RepeatBlock.h"
#include "read_pooler.h"
#include "mathops.h"
#include "stutter_model.h"
#include "base_quality.h"
#include "SeqAlignment/StutterAlignerClass.h"
#undef private
#undef protected

#define EXPORT extern "C" __attribute__((visibility("default")))

namespace {

struct RefLocus {
  std::vector<HapBlock*> blocks;
  StutterModel* sm = nullptr;
  Haplotype* hap = nullptr;       // raw storage, fields filled by hand
  HapAligner* aligner = nullptr;  // raw storage, fields filled by hand
  int n_alleles = 0;
};

/* Build [left flank][repeat block with alleles][right flank] exactly the way
 * SeqStutterGenotyper::build_haplotype lays blocks out (three blocks, the
 * middle one a RepeatBlock): the block classes are the reference's. */
RefLocus* make_locus(int32_t start, const std::string& lflank, const std::vector<std::string>& alleles,
                     const std::string& rflank, int period, const float* params7, int indel_flank_len) {
  RefLocus* L = new RefLocus();
  L->sm = new StutterModel(0.95, 0.05, 0.05, 0.95, 0.01, 0.01, std::string(period, 'A'));  // hipstr_main.cpp:362-363 defaults
  const int32_t s1 = start + (int32_t)lflank.size();
  const int32_t e1 = s1 + (int32_t)alleles[0].size();
  L->blocks.push_back(new HapBlock(start, s1, lflank));
  RepeatBlock* rb = new RepeatBlock(s1, e1, alleles[0], period, L->sm);
  for (size_t k = 1; k < alleles.size(); k++) rb->add_alternate(std::make_pair(alleles[k], false));
  L->blocks.push_back(rb);
  L->blocks.push_back(new HapBlock(e1, e1 + (int32_t)rflank.size(), rflank));
  L->n_alleles = (int)alleles.size();

  /* Haplotype: only blocks_ and counts_ are read by the inline accessors that
   * align_seq_to_hap / trim_alignment use (Haplotype.h:64-104). */
  void* hmem = ::operator new(sizeof(Haplotype));
  std::memset(hmem, 0, sizeof(Haplotype));
  Haplotype* H = reinterpret_cast<Haplotype*>(hmem);
  new (&H->blocks_) std::vector<HapBlock*>(L->blocks);
  new (&H->nopts_) std::vector<int>();
  new (&H->dirs_) std::vector<int>();
  new (&H->factors_) std::vector<int>();
  new (&H->counts_) std::vector<int>(L->blocks.size(), 0);
  new (&H->nchanges_) std::vector<int>();
  new (&H->hap_aln_info_) std::vector<std::string>();
  H->ncombs_ = L->n_alleles; H->counter_ = 0; H->last_changed_ = -1; H->fixed_ = false; H->inc_rev_ = false;
  L->hap = H;

  /* HapAligner: fields the two functions read (HapAligner.h:39-52, ctor :94-120). */
  void* amem = ::operator new(sizeof(HapAligner));
  std::memset(amem, 0, sizeof(HapAligner));
  HapAligner* A = reinterpret_cast<HapAligner*>(amem);
  A->fw_haplotype_ = H; A->rev_haplotype_ = nullptr;
  new (&A->realign_to_hap_) std::vector<bool>(L->n_alleles, true);
  new (&A->rev_blocks_) std::vector<HapBlock*>();
  new (&A->repeat_starts_) std::vector<int32_t>();
  new (&A->repeat_ends_) std::vector<int32_t>();
  A->INDEL_FLANK_LEN = indel_flank_len;
  A->SWITCH_OLD_ALIGN_LEN = 0;
  for (int i = 0; i < H->num_blocks(); i++) {                  // HapAligner.h:103-109
    HapBlock* block = H->get_block(i);
    if (block->get_repeat_info() != NULL) { A->repeat_starts_.push_back(block->start()); A->repeat_ends_.push_back(block->end()); }
  }
  A->AlnModel = new AlignmentModel(10, params7[0], params7[1], params7[2], params7[3], params7[4], params7[5], params7[6]);
  L->aligner = A;
  return L;
}

void free_locus(RefLocus* L) {
  if (!L) return;
  delete L->aligner->AlnModel;
  L->aligner->realign_to_hap_.~vector(); L->aligner->rev_blocks_.~vector();
  L->aligner->repeat_starts_.~vector(); L->aligner->repeat_ends_.~vector();
  ::operator delete(L->aligner);
  L->hap->blocks_.~vector(); L->hap->nopts_.~vector(); L->hap->dirs_.~vector(); L->hap->factors_.~vector();
  L->hap->counts_.~vector(); L->hap->nchanges_.~vector(); L->hap->hap_aln_info_.~vector();
  ::operator delete(L->hap);
  for (HapBlock* b : L->blocks) delete b;
  delete L->sm;
  delete L;
}

Alignment make_alignment(int32_t start, int32_t stop, const char* seq, int32_t seq_len,
                         const char* ctype, const int32_t* cnum, int32_t n_cigar) {
  std::string s(seq, seq + seq_len);
  Alignment aln(start, stop, false, false, "r", std::string(s.size(), 'I'), s, "");
  for (int32_t k = 0; k < n_cigar; k++) aln.add_cigar_element(CigarElement(ctype[k], cnum[k]));
  return aln;
}

}  // namespace

/* The AlignmentModel default parameter set of HapAligner.h:118, as floats. */
EXPORT void ltr_ref_default_params(float* p7) {
  AlignmentModel m(10, -1.0, -0.458675, -1.0, -0.458675, -0.00005800168, -10.448214728, -10.448214728);
  p7[0] = m.LOG_INS_TO_INS; p7[1] = m.LOG_INS_TO_MATCH; p7[2] = m.LOG_DEL_TO_DEL; p7[3] = m.LOG_DEL_TO_MATCH;
  p7[4] = m.LOG_MATCH_TO_MATCH; p7[5] = m.LOG_MATCH_TO_INS; p7[6] = m.LOG_MATCH_TO_DEL;
}

/*
 * One locus through the reference: blocks = lflank | alleles[0..H) | rflank at
 * reference coordinate `start`; R alignments (start, stop, seq, CIGAR).
 * For each alignment: reference trim_alignment -> (empty => 10-bp substitute,
 * HapAligner.cpp:820-823) -> reference align_seq_to_hap against every allele.
 * Outputs: ll[R*H], trimmed-read [ltrim_out, len_out] per alignment.
 * Returns seconds spent inside align_seq_to_hap (std::chrono), <0 on error.
 */
EXPORT double ltr_ref_process_locus(int32_t start, const char* lflank, int32_t lflank_len,
                                    const char* allele_bytes, const int64_t* allele_off, int32_t H,
                                    const char* rflank, int32_t rflank_len, int32_t period,
                                    const float* params7, int32_t indel_flank_len,
                                    int32_t R, const int32_t* aln_start, const int32_t* aln_stop,
                                    const char* seq_bytes, const int64_t* seq_off,
                                    const char* cigar_type, const int32_t* cigar_num, const int64_t* cigar_off,
                                    double* ll, int32_t* trim_off_out, int32_t* trim_len_out) {
  std::vector<std::string> alleles;
  for (int32_t k = 0; k < H; k++) alleles.push_back(std::string(allele_bytes + allele_off[k], allele_bytes + allele_off[k + 1]));
  RefLocus* L = make_locus(start, std::string(lflank, lflank + lflank_len), alleles,
                           std::string(rflank, rflank + rflank_len), period, params7, indel_flank_len);
  double secs = 0.0;
  for (int32_t r = 0; r < R; r++) {
    Alignment aln = make_alignment(aln_start[r], aln_stop[r], seq_bytes + seq_off[r], (int32_t)(seq_off[r + 1] - seq_off[r]),
                                   cigar_type + cigar_off[r], cigar_num + cigar_off[r], (int32_t)(cigar_off[r + 1] - cigar_off[r]));
    std::string base_seq_str;
    L->aligner->trim_alignment(aln, base_seq_str);                                  // HapAligner.cpp:819
    /* recover ltrim for reporting: trimmed = seq.substr(ltrim, ...) */
    int32_t ltrim = -1;
    if (trim_off_out) {
      const std::string& full = aln.get_sequence();
      /* the reference does not expose ltrim; report it only when unambiguous */
      size_t pos = full.find(base_seq_str);
      ltrim = (base_seq_str.empty() || pos == std::string::npos) ? -1 : (int32_t)pos;
      trim_off_out[r] = ltrim;
    }
    if (base_seq_str.size() == 0) {                                                  // :820-823
      base_seq_str += L->hap->get_first_block()->get_seq(0).substr(L->hap->get_first_block()->get_seq(0).size() - 5, 5);
      base_seq_str += L->hap->get_last_block()->get_seq(0).substr(0, 5);
    }
    if (trim_len_out) trim_len_out[r] = (int32_t)base_seq_str.size();
    const char* base_seq = base_seq_str.c_str();
    const int seed_base = (int)base_seq_str.size() - 1;                              // :826
    for (int32_t k = 0; k < H; k++) {                                                // :840-852 with next() replayed by hand
      L->hap->counts_[1] = k;
      double prob;
      auto t0 = std::chrono::steady_clock::now();
      L->aligner->align_seq_to_hap(L->hap, false, base_seq, seed_base, prob);       // :848
      secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      ll[(int64_t)r * H + k] = prob;
    }
    L->hap->counts_[1] = 0;
  }
  free_locus(L);
  return secs;
}

/*
 * Pre-trimmed pairs straight into the reference's align_seq_to_hap: for locus
 * l (flattened exactly like ltr_locus_batch) every read against every FULL
 * haplotype string.  The haplotype is presented as a one-block Haplotype whose
 * get_seq() returns the string.  Returns seconds inside align_seq_to_hap.
 */
EXPORT double ltr_ref_align_batch(const float* params7, int32_t indel_flank_len,
                                  int64_t n_loci, const int64_t* locus_read_off, const int64_t* locus_hap_off,
                                  const char* read_bytes, const int64_t* read_off,
                                  const char* hap_bytes, const int64_t* hap_off, double* out_ll) {
  double secs = 0.0; int64_t ll_off = 0;
  for (int64_t l = 0; l < n_loci; l++) {
    const int64_t r0 = locus_read_off[l], r1 = locus_read_off[l + 1], h0 = locus_hap_off[l], h1 = locus_hap_off[l + 1];
    const int64_t Hn = h1 - h0;
    for (int64_t h = h0; h < h1; h++) {
      /* one-block haplotype holding the full string: flank split is irrelevant to align_seq_to_hap */
      std::string hs(hap_bytes + hap_off[h], hap_bytes + hap_off[h + 1]);
      std::vector<std::string> alleles(1, hs.size() > 2 ? hs.substr(1, hs.size() - 2) : std::string());
      const std::string lf = hs.empty() ? std::string() : hs.substr(0, 1);
      const std::string rf = hs.size() > 1 ? hs.substr(hs.size() - 1) : std::string();
      RefLocus* L = make_locus(1000, lf, alleles, rf, 1, params7, indel_flank_len);
      for (int64_t r = r0; r < r1; r++) {
        std::string rs(read_bytes + read_off[r], read_bytes + read_off[r + 1]);
        double prob;
        auto t0 = std::chrono::steady_clock::now();
        L->aligner->align_seq_to_hap(L->hap, false, rs.c_str(), (int)rs.size() - 1, prob);
        secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        out_ll[ll_off + (r - r0) * Hn + (h - h0)] = prob;
      }
      free_locus(L);
    }
    ll_off += (r1 - r0) * Hn;
  }
  return secs;
}

/* ReadPooler::add_alignment over a read list (read_pooler.cpp:3-20). */
EXPORT int32_t ltr_ref_pool_reads(const char* seq_bytes, const int64_t* seq_off, int32_t n_reads, int32_t* pool_index) {
  ReadPooler pooler;
  for (int32_t i = 0; i < n_reads; i++) {
    std::string s(seq_bytes + seq_off[i], seq_bytes + seq_off[i + 1]);
    Alignment aln(0, (int32_t)s.size() - 1, false, false, "r", std::string(s.size(), 'I'), s, "");
    pool_index[i] = pooler.add_alignment(aln);
  }
  return pooler.num_pools();
}

/* The reference's own mathops.cpp helpers that Genotyper::extract_genotypes_and_likelihoods
 * (genotyper.cpp:132-256, not buildable here: htslib) is made of -- used to pin the restatement
 * of that function helper by helper. */
EXPORT double ltr_ref_fast_log_sum_exp2(double a, double b) { return fast_log_sum_exp(a, b); }        /* mathops.cpp:87-96 */
EXPORT double ltr_ref_log_sum_exp2(double a, double b) { return log_sum_exp(a, b); }                  /* mathops.cpp:55-60 */
EXPORT double ltr_ref_streaming_log_sum_exp(const double* vals, int32_t n) {                           /* mathops.cpp:70-85 */
  double mx = -DBL_MAX / 2, tot = 0.0;                                                                  /* genotyper.cpp:153-154 */
  for (int32_t i = 0; i < n; i++) update_streaming_log_sum_exp(vals[i], mx, tot);
  return finish_streaming_log_sum_exp(mx, tot);
}
EXPORT double ltr_ref_int_log(int32_t v) {                                                              /* mathops.cpp:16-22 */
  static bool ready = false;
  if (!ready) { precompute_integer_logs(); ready = true; }
  return int_log(v);
}
EXPORT void ltr_ref_math_consts(double* out3) { out3[0] = LOG_THRESH; out3[1] = LOG_E_BASE_10; out3[2] = TOLERANCE; }

/* ---- the pieces of the SHORT (stutter) path that compile here -------------------------
 * StutterAlignerClass.cpp, stutter_model.cpp, base_quality.h, mathops.cpp and RepeatStutterInfo.h need nothing from
 * htslib; what does (and keeps HapAligner::align_seq_to_hap_short as a whole out of this build) is
 * Haplotype::homopolymer_length, HapAligner.cpp:121-122 -> Haplotype.cpp:280 -> bam_io.h. */

/* StutterModel::log_stutter_pmf (stutter_model.cpp:29-53); s6 = in_geom, in_up, in_down, out_geom, out_up, out_down */
EXPORT double ltr_ref_log_stutter_pmf(const double* s6, int32_t motif_len, int32_t sample_bps, int32_t read_bps) {
  StutterModel sm(s6[0], s6[1], s6[2], s6[3], s6[4], s6[5], std::string((size_t)motif_len, 'A'));
  return sm.log_stutter_pmf(sample_bps, read_bps);
}
/* RepeatStutterInfo::log_prob_pcr_artifact (RepeatStutterInfo.h:53-61) for the reference allele of a block */
EXPORT double ltr_ref_log_prob_pcr_artifact(const double* s6, int32_t period, int32_t allele_size, int32_t artifact_size) {
  StutterModel sm(s6[0], s6[1], s6[2], s6[3], s6[4], s6[5], std::string((size_t)period, 'A'));
  RepeatStutterInfo info(period, std::string((size_t)allele_size, 'A'), &sm);
  return info.log_prob_pcr_artifact(0, artifact_size);
}
/* BaseQuality::log_prob_error / log_prob_correct (base_quality.h:29-75) */
EXPORT void ltr_ref_base_quality(int32_t quality_char, double* log_error, double* log_correct) {
  static BaseQuality bq;
  *log_error = bq.log_prob_error((char)quality_char);
  *log_correct = bq.log_prob_correct((char)quality_char);
}
/* fast_log_sum_exp(const std::vector<double>&) (mathops.cpp:98-107: fasterexp / fasterlog bit tricks) */
EXPORT double ltr_ref_fast_log_sum_exp_vec(const double* vals, int32_t n) {
  return fast_log_sum_exp(std::vector<double>(vals, vals + n));
}
/*
 * The stutter-block rows of HapAligner::align_seq_to_hap_short (HapAligner.cpp:64-111) with the reference's own
 * StutterAlignerClass (ctor StutterAlignerClass.h:45-79, load_read / align_stutter_region_reverse .cpp:12-166),
 * RepeatStutterInfo, StutterModel, BaseQuality and fast_log_sum_exp: for every read position j the log-sum over the
 * artifact sizes -6p .. +6p of pmf + stutter-aligner LL + the match value base_len positions back in `prev_row`
 * (the matrix row before the block).  The loop around those calls is replayed here line by line (:78-106).
 * out_match[j] = match_matrix of the block's last row; out_art_size / out_art_pos = the best artifact (:96-100).
 */
EXPORT int32_t ltr_ref_stutter_block_row(const char* block_seq, int32_t block_len, int32_t period, int32_t left_align, const double* s6,
                                         const char* seq_0, const char* qual, int32_t seq_len, const double* prev_row,
                                         double* out_match, int32_t* out_art_size, int32_t* out_art_pos) {
  { static bool ready = false; if (!ready) { precompute_integer_logs(); ready = true; } }       /* hipstr_main.cpp:375 */
  StutterModel sm(s6[0], s6[1], s6[2], s6[3], s6[4], s6[5], std::string((size_t)period, 'A'));
  const std::string block(block_seq, block_seq + block_len);
  RepeatStutterInfo rep_info(period, block, &sm);
  StutterAlignerClass aligner(block, period, left_align != 0, &rep_info);
  BaseQuality bq;
  std::vector<double> wrong((size_t)seq_len), correct((size_t)seq_len);
  for (int j = 0; j < seq_len; ++j) { wrong[j] = bq.log_prob_error(qual[j]); correct[j] = bq.log_prob_correct(qual[j]); }   /* :877-880 */
  const double* base_log_wrong = wrong.data(); const double* base_log_correct = correct.data();
  const int num_stutter_artifacts = (rep_info.max_insertion() - rep_info.max_deletion()) / period + 1;
  aligner.load_read(seq_len, seq_0 + seq_len - 1, base_log_wrong + seq_len - 1, base_log_correct + seq_len - 1);           /* :76 */
  std::vector<double> block_probs((size_t)num_stutter_artifacts);
  int offset = seq_len - 1;
  for (int j = 0; j < seq_len; ++j, --offset) {
    int art_idx = 0;
    double best_LL = -1000000000.0;                                                                                          /* IMPOSSIBLE */
    out_art_size[j] = -10000; out_art_pos[j] = -1;
    for (int artifact_size = rep_info.max_deletion(); artifact_size <= rep_info.max_insertion(); artifact_size += period) {
      int art_pos = -1;
      const int base_len = std::min(block_len + artifact_size, j + 1);
      if (base_len >= 0) {
        const double prob = aligner.align_stutter_region_reverse(base_len, seq_0 + j, offset, base_log_wrong + j, base_log_correct + j, artifact_size, art_pos);
        const double pre_prob = (j - base_len < 0 ? 0 : prev_row[j - base_len]);
        block_probs[art_idx] = rep_info.log_prob_pcr_artifact(0, artifact_size) + prob + pre_prob;
      } else
        block_probs[art_idx] = -1000000000.0;
      if (block_probs[art_idx] > best_LL) { out_art_size[j] = artifact_size; out_art_pos[j] = art_pos; best_LL = block_probs[art_idx]; }
      art_idx++;
    }
    out_match[j] = fast_log_sum_exp(block_probs);
  }
  return 0;
}

/* ---- the OUTER functions of the short path that link without Haplotype.cpp ------------
 * HapAligner::compute_aln_logprob (HapAligner.cpp:165-233), calc_best_seed_position (:467-493) and calc_seed_base
 * (:494-542) touch the Haplotype only through its inline accessors (Haplotype.h:64-82: get_seq(block), get_first_char,
 * get_last_char, get_block, get_first_block / get_last_block, num_blocks, cur_size) and int_log / fast_log_sum_exp
 * (mathops.cpp).  The objects are filled by hand as above, for ANY number of blocks; cur_size_ (set by
 * Haplotype::init / next, Haplotype.cpp) is set here to the summed length of the chosen alleles. */
namespace {
struct GenLocus {
  std::vector<HapBlock*> blocks;
  StutterModel* sm = nullptr;
  Haplotype* hap = nullptr;
  HapAligner* aligner = nullptr;
};

GenLocus* make_general_locus(int32_t n_blocks, const int32_t* block_start, const int32_t* block_end, const int32_t* is_repeat,
                             const int32_t* period, const int32_t* n_alleles, const char* allele_bytes, const int64_t* allele_off,
                             const int32_t* counts) {
  GenLocus* L = new GenLocus();
  L->sm = new StutterModel(0.95, 0.05, 0.05, 0.95, 0.01, 0.01, std::string(1, 'A'));
  int64_t k = 0;
  for (int32_t b = 0; b < n_blocks; b++) {
    const std::string ref_seq(allele_bytes + allele_off[k], allele_bytes + allele_off[k + 1]);
    HapBlock* blk = is_repeat[b] ? new RepeatBlock(block_start[b], block_end[b], ref_seq, period[b], L->sm)
                                 : new HapBlock(block_start[b], block_end[b], ref_seq);
    for (int32_t a = 1; a < n_alleles[b]; a++)
      blk->add_alternate(std::make_pair(std::string(allele_bytes + allele_off[k + a], allele_bytes + allele_off[k + a + 1]), false));
    k += n_alleles[b];
    L->blocks.push_back(blk);
  }
  void* hmem = ::operator new(sizeof(Haplotype));
  std::memset(hmem, 0, sizeof(Haplotype));
  Haplotype* H = reinterpret_cast<Haplotype*>(hmem);
  new (&H->blocks_) std::vector<HapBlock*>(L->blocks);
  new (&H->nopts_) std::vector<int>();
  new (&H->dirs_) std::vector<int>();
  new (&H->factors_) std::vector<int>();
  new (&H->counts_) std::vector<int>(counts, counts + n_blocks);
  new (&H->nchanges_) std::vector<int>();
  new (&H->hap_aln_info_) std::vector<std::string>();
  H->ncombs_ = 1; H->counter_ = 0; H->last_changed_ = -1; H->fixed_ = false; H->inc_rev_ = false;
  H->cur_size_ = 0;
  for (int32_t b = 0; b < n_blocks; b++) H->cur_size_ += (int)H->get_seq(b).size();
  L->hap = H;
  void* amem = ::operator new(sizeof(HapAligner));
  std::memset(amem, 0, sizeof(HapAligner));
  HapAligner* A = reinterpret_cast<HapAligner*>(amem);
  A->fw_haplotype_ = H; A->rev_haplotype_ = nullptr;
  new (&A->realign_to_hap_) std::vector<bool>();
  new (&A->rev_blocks_) std::vector<HapBlock*>();
  new (&A->repeat_starts_) std::vector<int32_t>();
  new (&A->repeat_ends_) std::vector<int32_t>();
  for (int i = 0; i < H->num_blocks(); i++) {                  // HapAligner.h:103-109
    HapBlock* block = H->get_block(i);
    if (block->get_repeat_info() != NULL) { A->repeat_starts_.push_back(block->start()); A->repeat_ends_.push_back(block->end()); }
  }
  L->aligner = A;
  return L;
}

void free_general_locus(GenLocus* L) {
  L->aligner->realign_to_hap_.~vector(); L->aligner->rev_blocks_.~vector();
  L->aligner->repeat_starts_.~vector(); L->aligner->repeat_ends_.~vector();
  ::operator delete(L->aligner);
  L->hap->blocks_.~vector(); L->hap->nopts_.~vector(); L->hap->dirs_.~vector(); L->hap->factors_.~vector();
  L->hap->counts_.~vector(); L->hap->nchanges_.~vector(); L->hap->hap_aln_info_.~vector();
  ::operator delete(L->hap);
  for (HapBlock* b : L->blocks) delete b;
  delete L->sm;
  delete L;
}
}  // namespace

/* HapAligner::compute_aln_logprob (HapAligner.cpp:165-233) on caller-supplied match matrices:
 * lM = lflank_len x hapsize doubles, row-major by haplotype position (lflank_len = seed_base), rM = rflank_len x hapsize
 * (rflank_len = base_seq_len - seed_base - 1); only the match matrices are read (:190,:196,:208-223). */
EXPORT double ltr_ref_compute_aln_logprob(int32_t n_blocks, const int32_t* block_start, const int32_t* block_end, const int32_t* is_repeat,
                                          const int32_t* period, const int32_t* n_alleles, const char* allele_bytes, const int64_t* allele_off,
                                          const int32_t* counts, int32_t base_seq_len, int32_t seed_base, int32_t seed_char,
                                          double log_seed_wrong, double log_seed_correct, double* lM, double l_prob,
                                          double* rM, double r_prob, int32_t* max_index) {
  { static bool ready = false; if (!ready) { precompute_integer_logs(); ready = true; } }       /* hipstr_main.cpp:375 */
  GenLocus* L = make_general_locus(n_blocks, block_start, block_end, is_repeat, period, n_alleles, allele_bytes, allele_off, counts);
  int mi = -1;
  const double v = L->aligner->compute_aln_logprob(base_seq_len, seed_base, (char)seed_char, log_seed_wrong, log_seed_correct,
                                                   lM, nullptr, nullptr, l_prob, rM, nullptr, nullptr, r_prob, mi);
  *max_index = mi;
  free_general_locus(L);
  return v;
}

/* HapAligner::calc_best_seed_position (HapAligner.cpp:467-493) on caller-supplied repeat_starts_ / repeat_ends_ */
EXPORT void ltr_ref_calc_best_seed_position(const int32_t* repeat_starts, const int32_t* repeat_ends, int32_t n_repeats,
                                            int32_t region_start, int32_t region_end, int32_t* best_dist, int32_t* best_pos) {
  void* amem = ::operator new(sizeof(HapAligner));
  std::memset(amem, 0, sizeof(HapAligner));
  HapAligner* A = reinterpret_cast<HapAligner*>(amem);
  new (&A->repeat_starts_) std::vector<int32_t>(repeat_starts, repeat_starts + n_repeats);
  new (&A->repeat_ends_) std::vector<int32_t>(repeat_ends, repeat_ends + n_repeats);
  int32_t d = 0, p = 0;
  A->calc_best_seed_position(region_start, region_end, d, p);
  *best_dist = d; *best_pos = p;
  A->repeat_starts_.~vector(); A->repeat_ends_.~vector();
  ::operator delete(A);
}

/* HapAligner::calc_seed_base (HapAligner.cpp:494-542); CIGAR ops outside = X I D make the reference exit(1): not sent here */
EXPORT int32_t ltr_ref_calc_seed_base(int32_t n_blocks, const int32_t* block_start, const int32_t* block_end, const int32_t* is_repeat,
                                      const int32_t* period, const int32_t* n_alleles, const char* allele_bytes, const int64_t* allele_off,
                                      int32_t aln_start, int32_t aln_stop, const char* seq, int32_t seq_len,
                                      const char* ctype, const int32_t* cnum, int32_t n_cigar) {
  for (int32_t k = 0; k < n_cigar; k++)
    if (ctype[k] != '=' && ctype[k] != 'X' && ctype[k] != 'I' && ctype[k] != 'D') return -2;
  std::vector<int32_t> counts((size_t)n_blocks, 0);
  GenLocus* L = make_general_locus(n_blocks, block_start, block_end, is_repeat, period, n_alleles, allele_bytes, allele_off, counts.data());
  Alignment aln = make_alignment(aln_start, aln_stop, seq, seq_len, ctype, cnum, n_cigar);
  const int32_t seed = L->aligner->calc_seed_base(aln);
  free_general_locus(L);
  return seed;
}

EXPORT const char* ltr_ref_describe() {
  return "reference HapAligner.cpp/read_pooler.cpp compiled from source where it lies; harness oracle/ref_driver.cpp";
}

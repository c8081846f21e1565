/*
 * ltr_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-thread CPU restatement of LongTR's read-vs-haplotype
 * alignment path (SURVEY.md section 8a rows a-1 .. a-6).  It exists to check the
 * HIP path and to be timed as the "cpu_baseline" in bench.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing
 * under longtr_amd/ or include/ links, imports or calls it.
 *
 * Parity status (see also DESIGN.md "Oracle"):
 *   a-1 align_long, a-2 trim_alignment, a-3 process_reads (long branch),
 *   a-5 pooling: PINNED against the reference's own HapAligner.cpp /
 *       read_pooler.cpp compiled from /root/reference (oracle/_ref, built by
 *       oracle/Makefile) -- bit-exact on every generated case -- and against
 *       the committed golden vectors in tests/golden/.
 *   a-4 calc_hap_aln_probs scatter, a-6 posteriors: PARITY UNPINNED by a
 *       reference build (seq_stutter_genotyper.cpp / genotyper.cpp need htslib
 *       headers that this image lacks); pinned only by the known-answer point
 *       recorded in SURVEY.md section 8c.
 */
#ifndef LTR_ORACLE_H_
#define LTR_ORACLE_H_

#include <stdint.h>
#include "../include/ltr_gpu.h"   /* ltr_align_params, ltr_alignment, ltr_haplotype_blocks: types only */

#ifdef __cplusplus
extern "C" {
#endif

/* HapAligner::align_seq_to_hap (HapAligner.cpp:236-343).  hap = FULL haplotype
 * string; read = trimmed read.  cells_executed (optional) += rows actually
 * filled * m ("reference-executed" cells: stops at the aborting row). */
double ltr_oracle_align_long(const uint8_t* hap, int64_t hap_len,
                             const uint8_t* read, int64_t read_len,
                             const ltr_align_params* p, double* cells_executed);

/* Same recurrence with two rolling rows instead of three n*m matrices; used
 * only to cross-check the restatement against itself at sizes where the
 * materialised version would need GBs. */
double ltr_oracle_align_long_rolling(const uint8_t* hap, int64_t hap_len,
                                     const uint8_t* read, int64_t read_len,
                                     const ltr_align_params* p);

/* HapAligner::trim_alignment (HapAligner.cpp:346-465). Returns 0 or LTR_ERR_*. */
int ltr_oracle_trim_alignment(const ltr_alignment* aln, int32_t repeat_start, int32_t repeat_end,
                              int32_t padding, int32_t* ltrim, int32_t* rtrim);

/* Haplotype iteration (Haplotype.cpp:123-196). */
int64_t ltr_oracle_haplotype_num_combs(const ltr_haplotype_blocks* hap);
int64_t ltr_oracle_haplotype_seq(const ltr_haplotype_blocks* hap, int64_t index, uint8_t* out, int64_t cap);

/* HapAligner::process_reads, long branch (HapAligner.cpp:545-581, :812-854). */
int ltr_oracle_process_reads(const ltr_align_params* p, const ltr_haplotype_blocks* hap,
                             const uint8_t* realign_to_hap,
                             const ltr_alignment* alns, int32_t n_alns, int32_t init_read_index,
                             const uint8_t* realign_read,
                             double* aln_probs, int32_t* seed_positions);

/* Short (seeded, stutter-aware) path, row a-7: HapAligner::process_reads with short_ == 1
 * (HapAligner.cpp:27-233, :467-542, :855-990; StutterAlignerClass.cpp).  PARITY UNPINNED by a
 * reference build (see ltr_oracle_short.c). */
int ltr_oracle_calc_seed_base(const ltr_alignment* aln, const ltr_haplotype_blocks* hap);
/* compute_aln_logprob (HapAligner.cpp:165-233) / calc_best_seed_position (:467-493) on caller-supplied inputs: the
 * pin against oracle/_ref (ltr_ref_compute_aln_logprob, ltr_ref_calc_best_seed_position); counts = allele per block */
double ltr_oracle_compute_aln_logprob(const ltr_haplotype_blocks* hap, const int32_t* counts, int32_t base_seq_len, int32_t seed_base,
                                      int32_t seed_char, double log_seed_wrong, double log_seed_correct,
                                      const double* lM, double l_prob, const double* rM, double r_prob);
void ltr_oracle_calc_best_seed_position(const int32_t* repeat_starts, const int32_t* repeat_ends, int32_t n_repeats,
                                        int32_t region_start, int32_t region_end, int32_t* best_dist, int32_t* best_pos);
/* ... and its pieces by themselves, pinned to the compiled reference (oracle/_ref): */
double ltr_oracle_log_stutter_pmf(const ltr_stutter_params* sp, int32_t motif_len, int32_t sample_bps, int32_t read_bps);
double ltr_oracle_log_prob_pcr_artifact(const ltr_stutter_params* sp, int32_t period, int32_t allele_size, int32_t artifact_size);
void ltr_oracle_base_quality(int32_t quality_char, double* log_error, double* log_correct);
double ltr_oracle_fast_log_sum_exp_vec(const double* vals, int32_t n);
int ltr_oracle_stutter_block_row(const ltr_stutter_params* sp, const char* block_seq, int32_t block_len, int32_t period, int32_t left_align,
                                 const char* seq_0, const char* qual, int32_t seq_len, const double* prev_row, double* out_match);
int ltr_oracle_process_reads_short(const ltr_align_params* p, const ltr_stutter_params* sp,
                                   const ltr_haplotype_blocks* hap, const uint8_t* realign_to_hap,
                                   const ltr_alignment* alns, int32_t n_alns, int32_t init_read_index,
                                   const uint8_t* realign_read, double* aln_probs, int32_t* seed_positions);

/* Flattened batch scorer: same contract as ltr_align_batch. */
int ltr_oracle_align_batch(const ltr_align_params* p, const ltr_locus_batch* batch,
                           double* out_ll, int32_t* out_seed, double* cells_executed);

/* ReadPooler::add_alignment (read_pooler.cpp:3-20). */
int32_t ltr_oracle_pool_reads(const uint8_t* const* seqs, const int32_t* seq_lens, int32_t n_reads,
                              int32_t* pool_index);

/* SeqStutterGenotyper::calc_hap_aln_probs scatter (seq_stutter_genotyper.cpp:526-559). */
int ltr_oracle_scatter_pool_probs(const double* log_pool_aln_probs, const int32_t* pool_seed_positions,
                                  const int32_t* pool_index, int32_t n_reads, int32_t n_alleles,
                                  const uint8_t* realign_to_hap, const uint8_t* copy_read,
                                  const uint8_t* second_mate,
                                  double* log_aln_probs, int32_t* seed_positions);

/* Genotyper::calc_log_sample_posteriors + get_optimal_haplotypes (genotyper.cpp:21-100). */
int ltr_oracle_posteriors(int32_t n_samples, int32_t n_reads, int32_t n_alleles,
                          double* log_aln_probs, const double* log_p1, const double* log_p2,
                          const int32_t* sample_label, int32_t haploid,
                          double* log_sample_posteriors, double* sample_total_ll,
                          int32_t* gts, double* total_ll);

/* Genotype fields, SURVEY.md 8f next-2 (ltr_oracle_genotype.c): Genotyper::extract_genotypes_and_likelihoods
 * (genotyper.cpp:132-256) and the mathops.cpp helpers it is made of (pinned against oracle/_ref). */
double ltr_oracle_fast_log_sum_exp2(double log_v1, double log_v2);
double ltr_oracle_log_sum_exp2(double log_v1, double log_v2);
double ltr_oracle_streaming_log_sum_exp(const double* vals, int32_t n);
double ltr_oracle_int_log(int32_t v);
int ltr_oracle_extract_genotypes(int32_t num_samples, int32_t num_alleles, int32_t num_variants,
                                 const int32_t* hap_to_allele, int32_t haploid,
                                 const double* log_sample_posteriors, const double* sample_total_LLs,
                                 const int32_t* best_haplotypes, const ltr_genotype_fields* out);

/* Haplotype::aln_haps_to_ref for one (reference haplotype, haplotype) pair, SURVEY.md 8f next-1 (ltr_oracle_nw.c;
 * PARITY UNPINNED): NeedlemanWunsch::Align + adjust_indels + the M / I / D string.  Returns its length. */
int64_t ltr_oracle_nw_aln_info(const uint8_t* refseq, int32_t L1, const uint8_t* readseq, int32_t L2,
                               int32_t ref_pos, int32_t str_pos, char* out);

/* The genotyper's last steps, SURVEY.md 8f next-2 (ltr_oracle_vcf.c; PARITY UNPINNED, see its header). */
int ltr_oracle_haps_to_alleles(const ltr_haplotype_blocks* hap, int32_t block, int32_t* out);
int32_t ltr_oracle_unused_alleles(int32_t num_samples, const int32_t* haps, const uint8_t* aligned_read, const uint8_t* filtered,
                                  const int32_t* hap_to_allele, int32_t num_options, int32_t* out);
int ltr_oracle_remap_haplotypes(const ltr_haplotype_blocks* old_hap, const ltr_haplotype_blocks* new_hap, int32_t* mapping, uint8_t* realign);
int32_t ltr_oracle_get_alleles(const ltr_vcf_locus* v, int32_t* pos, char* out, int64_t cap, int64_t* off);
int64_t ltr_oracle_vcf_header(const char* fasta_path, const char* full_command, const char* contig_lines, const ltr_vcf_options* opt,
                              const char* const* sample_names, int32_t n_samples, char* out, int64_t cap);
int64_t ltr_oracle_vcf_record(const ltr_vcf_locus* v, const ltr_vcf_options* opt, char* out, int64_t cap, int32_t* pos);

#ifdef __cplusplus
}
#endif
#endif

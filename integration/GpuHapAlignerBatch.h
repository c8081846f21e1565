// GpuHapAlignerBatch.h -- the THROUGHPUT form of the drop-in: many loci per GPU call.
//
// The reference scores one locus per SeqStutterGenotyper::calc_hap_aln_probs call (seq_stutter_genotyper.cpp:514-563:
// pool the reads, HapAligner::process_reads over the pools, copy pool rows to read rows, sum mate rows).  With default
// flags every locus makes exactly one such call and loci share no state (bam_processor.cpp:563-627), so a host may
// stage K loci and score them in ONE ltr_calc_hap_aln_probs call -- the call every headline number of this library
// comes from ("Batching across loci").  This header is that binding, written against the REFERENCE's
// own types (Haplotype, Alignment): compiled inside LongTR's tree; in this repository only by the dev-container test
// (oracle/Makefile target `adapter`), which runs it on the golden loci (tests/test_adapter.py).
//
//   GpuHapAlignerBatch batch(INDEL_FLANK_LEN, SWITCH_OLD_ALIGN_LEN, alignment_parameters, device);
//   for every locus:   batch.add_locus(haplotype_, alns_, second_mate_, log_aln_probs_, seed_positions_);
//   batch.run();       // fills every staged locus' log_aln_probs_ [R x H] and seed_positions_ [R]
//
// add_locus keeps flattened COPIES of the haplotype blocks and CIGARs; the read sequences themselves are borrowed
// (Alignment::get_sequence()): the caller's Alignment objects must stay alive and unmoved until run().
#ifndef GPU_HAP_ALIGNER_BATCH_H_
#define GPU_HAP_ALIGNER_BATCH_H_

#include <deque>

#include "GpuHapAligner.h"

class GpuHapAlignerBatch {
 private:
  struct Staged {
    GpuHapAligner::FlatHaplotype hap;
    GpuHapAligner::FlatAlignments alns;
    std::vector<uint8_t> second_mate, realign_to_hap, realign_pool, copy_read;
    double* log_aln_probs;
    int* seed_positions;
  };
  ltr_ctx* ctx_;
  std::deque<Staged> staged_;            // (deque: the flattened arrays never move once staged)

  GpuHapAlignerBatch(const GpuHapAlignerBatch&);
  GpuHapAlignerBatch& operator=(const GpuHapAlignerBatch&);

  static void bits(const std::vector<bool>* v, std::vector<uint8_t>* out) { if (v) out->assign(v->begin(), v->end()); }

 public:
  GpuHapAlignerBatch(int indel_flank_len, int switch_old_align_len, const std::vector<float>& alignment_model_params, int device = 0)
      : ctx_(GpuContext::get(device, GpuContext::params(indel_flank_len, switch_old_align_len, alignment_model_params))) {}

  size_t size() const { return staged_.size(); }

  // One locus as SeqStutterGenotyper holds it when calc_hap_aln_probs runs: haplotype_, alns_ (after left_align_reads),
  // second_mate_ (:491-497), where its log_aln_probs_ [R x H] and seed_positions_ [R] live; the three masks of the
  // form add_and_remove_alleles uses (:390-391; NULL = all set, the call at :634).
  void add_locus(Haplotype* haplotype, const std::vector<Alignment>& alns, const std::vector<bool>* second_mate,
                 double* log_aln_probs, int* seed_positions, const std::vector<bool>* realign_to_haplotype = NULL,
                 const std::vector<bool>* realign_pool = NULL, const std::vector<bool>* copy_read = NULL) {
    staged_.push_back(Staged());
    Staged& S = staged_.back();
    GpuHapAligner::flatten(haplotype, &S.hap);
    GpuHapAligner::flatten(alns, &S.alns);
    bits(second_mate, &S.second_mate); bits(realign_to_haplotype, &S.realign_to_hap);
    bits(realign_pool, &S.realign_pool); bits(copy_read, &S.copy_read);
    S.log_aln_probs = log_aln_probs; S.seed_positions = seed_positions;
  }

  // Everything staged in one GPU pass; the batch is empty afterwards.
  void run() {
    static_assert(sizeof(int) == sizeof(int32_t), "seed_positions_ is int* in the reference");
    const size_t n = staged_.size();
    std::vector<ltr_locus> loci(n);
    std::vector<double*> probs(n);
    std::vector<int32_t*> seeds(n);
    for (size_t i = 0; i < n; i++) {
      Staged& S = staged_[i];
      S.hap.view.block_start = S.hap.start.data(); S.hap.view.block_end = S.hap.end.data(); S.hap.view.is_repeat = S.hap.is_repeat.data();
      S.hap.view.period = S.hap.period.data(); S.hap.view.n_alleles = S.hap.n_alleles.data();
      S.hap.view.allele_bytes = S.hap.bytes.data(); S.hap.view.allele_off = S.hap.off.data();
      ltr_locus& L = loci[i];
      L.hap = &S.hap.view;
      L.alns = S.alns.la.empty() ? NULL : S.alns.la.data();
      L.n_alns = (int32_t)S.alns.la.size();
      L.second_mate = S.second_mate.empty() ? NULL : S.second_mate.data();
      L.realign_to_hap = S.realign_to_hap.empty() ? NULL : S.realign_to_hap.data();
      L.realign_pool = S.realign_pool.empty() ? NULL : S.realign_pool.data();
      L.copy_read = S.copy_read.empty() ? NULL : S.copy_read.data();
      probs[i] = S.log_aln_probs; seeds[i] = (int32_t*)S.seed_positions;
    }
    const int rc = ltr_calc_hap_aln_probs(ctx_, loci.data(), (int64_t)n, probs.data(), seeds.data());
    if (rc != LTR_OK) printErrorAndDie(std::string("GpuHapAlignerBatch::run: ") + ltr_last_error(ctx_));   // error.cpp:6-10
    staged_.clear();
  }
};

#endif

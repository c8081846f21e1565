/*
 * adapter_harness.cpp -- TEST INFRASTRUCTURE.  Compiles integration/GpuHapAligner.h (the adapter a
 * LongTR maintainer adds) against the REFERENCE's own headers where they lie under
 * $(LONGTR_REF)/src/SeqAlignment, links it with the reference objects of oracle/_ref (HapBlock.o ...)
 * and with the product library libltr_gpu.so, and drives it with reference objects:
 *
 *     reference HapBlock / RepeatBlock / Haplotype / Alignment objects
 *        -> GpuHapAligner::process_reads  (the adapter)
 *        -> ltr_process_reads             (C-ABI, GPU)
 *        -> aln_probs / seed_positions
 *
 * Like oracle/ref_driver.cpp it fills the Haplotype object's fields by hand (Haplotype.cpp needs
 * htslib and is not part of the build); every accessor the adapter calls is the reference's own
 * inline code.  Modes:
 *     adapter_check flatten   < locus     flattening only, no GPU: prints every haplotype string
 *                                         (ltr_haplotype_seq over the adapter's flattened blocks)
 *                                         next to Haplotype::get_seq() for the same allele
 *     adapter_check run       < loci      the whole chain on the GPU, a GpuHapAligner per locus like the reference
 *                                         (seq_stutter_genotyper.cpp:517-523): prints R x H hex doubles + seeds per locus
 *     adapter_check batch     < loci      integration/GpuHapAlignerBatch.h: every locus staged, ONE ltr_calc_hap_aln_probs
 *     adapter_check latency N [trace] < loci      N rounds of `run` without printing: mean wall time per locus incl. construction
 * Any number of loci on stdin, one after the other.  Locus format (text, whitespace separated): 7 floats (hex), indel_flank_len, start, lflank, H,
 * H allele strings, rflank, period, R, then per alignment: start stop seq n_cigar (type num)*.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <algorithm>
#include <deque>
#include <map>
#include <new>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

/* the harness fills Haplotype's fields by hand; access control does not change object layout
 * (standard headers are included above, before the redefinition) */
#define private public
#define protected public
#include "SeqAlignment/Haplotype.h"
#include "SeqAlignment/RepeatBlock.h"
#include "stutter_model.h"
#undef private
#undef protected

#include <chrono>

#include "GpuHapAligner.h"
#include "GpuHapAlignerBatch.h"

namespace {

struct TestLocus {
  std::vector<float> p7; int flank = 5; int32_t start = 0; std::string lflank, rflank; int H = 0, period = 1, R = 0;
  std::vector<std::string> alleles;
  std::vector<Alignment> alns;
  StutterModel* sm = NULL;
  std::vector<HapBlock*> blocks;
  Haplotype* hap = NULL;
};

// one locus from stdin; false at end of input
bool read_locus(TestLocus* L) {
  L->p7.resize(7);
  for (int k = 0; k < 7; k++) { std::string t; if (!(std::cin >> t)) return false; L->p7[k] = (float)std::strtod(t.c_str(), NULL); }
  std::cin >> L->flank >> L->start >> L->lflank >> L->H;
  L->alleles.resize(L->H);
  for (int k = 0; k < L->H; k++) std::cin >> L->alleles[k];
  std::cin >> L->rflank >> L->period >> L->R;
  for (int i = 0; i < L->R; i++) {
    int32_t s, e; std::string seq; int nc;
    std::cin >> s >> e >> seq >> nc;
    Alignment aln(s, e, false, false, "r", std::string(seq.size(), 'I'), seq, "");
    for (int c = 0; c < nc; c++) { char t; int num; std::cin >> t >> num; aln.add_cigar_element(CigarElement(t, num)); }
    L->alns.push_back(aln);
  }
  if (!std::cin) { std::fprintf(stderr, "adapter_check: malformed input\n"); std::exit(2); }
  // [left flank][repeat block with alleles][right flank], the layout SeqStutterGenotyper::build_haplotype produces
  L->sm = new StutterModel(0.95, 0.05, 0.05, 0.95, 0.01, 0.01, std::string(L->period, 'A'));
  const int32_t s1 = L->start + (int32_t)L->lflank.size(), e1 = s1 + (int32_t)L->alleles[0].size();
  L->blocks.push_back(new HapBlock(L->start, s1, L->lflank));
  RepeatBlock* rb = new RepeatBlock(s1, e1, L->alleles[0], L->period, L->sm);
  for (int k = 1; k < L->H; k++) rb->add_alternate(std::make_pair(L->alleles[k], false));
  L->blocks.push_back(rb);
  L->blocks.push_back(new HapBlock(e1, e1 + (int32_t)L->rflank.size(), L->rflank));
  void* hmem = ::operator new(sizeof(Haplotype));
  std::memset(hmem, 0, sizeof(Haplotype));
  Haplotype* hap = reinterpret_cast<Haplotype*>(hmem);
  new (&hap->blocks_) std::vector<HapBlock*>(L->blocks);
  new (&hap->nopts_) std::vector<int>();
  new (&hap->dirs_) std::vector<int>();
  new (&hap->factors_) std::vector<int>();
  new (&hap->counts_) std::vector<int>(L->blocks.size(), 0);
  new (&hap->nchanges_) std::vector<int>();
  new (&hap->hap_aln_info_) std::vector<std::string>();
  hap->ncombs_ = L->H; hap->counter_ = 0; hap->last_changed_ = -1; hap->fixed_ = false; hap->inc_rev_ = false;
  L->hap = hap;
  return true;
}

void print_locus_result(const TestLocus& L, const std::vector<double>& probs, const std::vector<int>& seeds) {
  for (size_t k = 0; k < probs.size(); k++) std::printf("%a\n", probs[k]);
  for (int i = 0; i < L.R; i++) std::printf("seed %d\n", seeds[i]);
}

}  // namespace

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "run";
  std::deque<TestLocus> loci;
  for (;;) { loci.push_back(TestLocus()); if (!read_locus(&loci.back())) { loci.pop_back(); break; } }
  if (loci.empty()) { std::fprintf(stderr, "adapter_check: no locus on stdin\n"); return 2; }

  if (mode == "flatten") {
    for (TestLocus& L : loci) {
      GpuHapAligner::FlatHaplotype fh;
      GpuHapAligner::flatten(L.hap, &fh);
      if (ltr_haplotype_num_combs(&fh.view) != L.H) { std::fprintf(stderr, "num_combs mismatch\n"); return 1; }
      std::vector<uint8_t> buf(1 << 20);
      for (int k = 0; k < L.H; k++) {
        L.hap->counts_[1] = k;                                   // with one multi-allele block, haplotype k == allele k (Haplotype.cpp:151-196)
        const int64_t len = ltr_haplotype_seq(&fh.view, k, buf.data(), (int64_t)buf.size());
        const std::string mine(buf.begin(), buf.begin() + (len > 0 ? len : 0));
        std::printf("%s %s\n", mine.c_str(), L.hap->get_seq().c_str());
      }
      std::printf("blocks %d repeat_block_period %d\n", fh.view.n_blocks, fh.view.period[1]);
    }
    return 0;
  }

  if (mode == "run" || mode == "latency") {
    // per locus, the way the reference does it: a GpuHapAligner constructed for the locus (seq_stutter_genotyper.cpp:517),
    // one process_reads call (:523), destroyed.  "latency N": N rounds over the loci, mean wall time per locus
    // INCLUDING construction (the GPU context is the process-wide one of GpuContext).
    const int rounds = (mode == "latency" && argc > 2) ? std::atoi(argv[2]) : 1;
    if (argc > 3 && std::string(argv[3]) == "trace") {          // phase prints of the library on stderr
      ltr_align_params prm; ltr_default_params(&prm);
      ltr_ctx_set_debug(GpuContext::get(0, prm), "trace", 1);
    }
    double total_s = 0.0; long calls = 0;
    for (int r = 0; r < rounds; r++)
      for (TestLocus& L : loci) {
        std::vector<bool> realign_to_hap(L.H, true), realign_read(L.R, true);
        std::vector<double> probs((size_t)L.R * L.H, 0.0);
        std::vector<int> seeds(L.R, -1);
        const auto t0 = std::chrono::steady_clock::now();
        {
          GpuHapAligner aligner(L.hap, realign_to_hap, L.flank, 0, L.p7, 0);
          aligner.process_reads(L.alns, 0, NULL, realign_read, probs.data(), seeds.data());
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (r > 0 || rounds == 1) { total_s += dt; calls++; }      // (round 0 of a latency run warms the context up)
        if (mode == "run") print_locus_result(L, probs, seeds);
      }
    if (mode == "latency") std::printf("latency_ms_per_locus %.4f over %ld calls (%zu loci, construction + process_reads, one context per process)\n",
                                       total_s / (double)std::max(calls, 1L) * 1e3, calls, loci.size());
    return 0;
  }

  if (mode == "batch") {
    // every locus staged, ONE ltr_calc_hap_aln_probs: reads are pooled inside the call, rows come back per read
    // (second_mate: none; the golden alignments are single reads)
    GpuHapAlignerBatch batch(loci[0].flank, 0, loci[0].p7, 0);
    std::vector<std::vector<double> > probs(loci.size());
    std::vector<std::vector<int> > seeds(loci.size());
    for (size_t i = 0; i < loci.size(); i++) {
      probs[i].assign((size_t)loci[i].R * loci[i].H, 0.0); seeds[i].assign(loci[i].R, -1);
      batch.add_locus(loci[i].hap, loci[i].alns, NULL, probs[i].data(), seeds[i].data());
    }
    batch.run();
    for (size_t i = 0; i < loci.size(); i++) print_locus_result(loci[i], probs[i], seeds[i]);
    return 0;
  }
  std::fprintf(stderr, "adapter_check: unknown mode %s\n", mode.c_str());
  return 2;
}

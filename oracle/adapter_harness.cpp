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
 *     adapter_check run       < locus     the whole chain on the GPU: prints R x H hex doubles + seeds
 * Locus format (text, whitespace separated): 7 floats (hex), indel_flank_len, start, lflank, H,
 * H allele strings, rflank, period, R, then per alignment: start stop seq n_cigar (type num)*.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <algorithm>
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

#include "GpuHapAligner.h"

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "run";
  std::vector<float> p7(7);
  for (int k = 0; k < 7; k++) { std::string t; std::cin >> t; p7[k] = (float)std::strtod(t.c_str(), NULL); }
  int flank; int32_t start; std::string lflank, rflank; int H, period, R;
  std::cin >> flank >> start >> lflank >> H;
  std::vector<std::string> alleles(H);
  for (int k = 0; k < H; k++) std::cin >> alleles[k];
  std::cin >> rflank >> period >> R;
  std::vector<Alignment> alns;
  for (int i = 0; i < R; i++) {
    int32_t s, e; std::string seq; int nc;
    std::cin >> s >> e >> seq >> nc;
    Alignment aln(s, e, false, false, "r", std::string(seq.size(), 'I'), seq, "");
    for (int c = 0; c < nc; c++) { char t; int num; std::cin >> t >> num; aln.add_cigar_element(CigarElement(t, num)); }
    alns.push_back(aln);
  }
  if (!std::cin) { std::fprintf(stderr, "adapter_check: malformed input\n"); return 2; }

  // [left flank][repeat block with alleles][right flank], the layout SeqStutterGenotyper::build_haplotype produces
  StutterModel sm(0.95, 0.05, 0.05, 0.95, 0.01, 0.01, std::string(period, 'A'));
  const int32_t s1 = start + (int32_t)lflank.size(), e1 = s1 + (int32_t)alleles[0].size();
  std::vector<HapBlock*> blocks;
  blocks.push_back(new HapBlock(start, s1, lflank));
  RepeatBlock* rb = new RepeatBlock(s1, e1, alleles[0], period, &sm);
  for (int k = 1; k < H; k++) rb->add_alternate(std::make_pair(alleles[k], false));
  blocks.push_back(rb);
  blocks.push_back(new HapBlock(e1, e1 + (int32_t)rflank.size(), rflank));
  void* hmem = ::operator new(sizeof(Haplotype));
  std::memset(hmem, 0, sizeof(Haplotype));
  Haplotype* hap = reinterpret_cast<Haplotype*>(hmem);
  new (&hap->blocks_) std::vector<HapBlock*>(blocks);
  new (&hap->nopts_) std::vector<int>();
  new (&hap->dirs_) std::vector<int>();
  new (&hap->factors_) std::vector<int>();
  new (&hap->counts_) std::vector<int>(blocks.size(), 0);
  new (&hap->nchanges_) std::vector<int>();
  new (&hap->hap_aln_info_) std::vector<std::string>();
  hap->ncombs_ = H; hap->counter_ = 0; hap->last_changed_ = -1; hap->fixed_ = false; hap->inc_rev_ = false;

  if (mode == "flatten") {
    GpuHapAligner::FlatHaplotype fh;
    GpuHapAligner::flatten(hap, &fh);
    if (ltr_haplotype_num_combs(&fh.view) != H) { std::fprintf(stderr, "num_combs mismatch\n"); return 1; }
    std::vector<uint8_t> buf(1 << 20);
    for (int k = 0; k < H; k++) {
      hap->counts_[1] = k;                                     // with one multi-allele block, haplotype k == allele k (Haplotype.cpp:151-196)
      const int64_t len = ltr_haplotype_seq(&fh.view, k, buf.data(), (int64_t)buf.size());
      const std::string mine(buf.begin(), buf.begin() + (len > 0 ? len : 0));
      std::printf("%s %s\n", mine.c_str(), hap->get_seq().c_str());
    }
    std::printf("blocks %d repeat_block_period %d\n", fh.view.n_blocks, fh.view.period[1]);
    return 0;
  }

  std::vector<bool> realign_to_hap(H, true), realign_read(R, true);
  GpuHapAligner aligner(hap, realign_to_hap, flank, 0, p7, 0);
  std::vector<double> probs((size_t)R * H, 0.0);
  std::vector<int> seeds(R, -1);
  aligner.process_reads(alns, 0, NULL, realign_read, probs.data(), seeds.data());
  for (size_t k = 0; k < probs.size(); k++) std::printf("%a\n", probs[k]);
  for (int i = 0; i < R; i++) std::printf("seed %d\n", seeds[i]);
  return 0;
}

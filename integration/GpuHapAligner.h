// GpuHapAligner.h -- the reference-side adapter a LongTR maintainer adds next to
// src/SeqAlignment/HapAligner.h to re-point SeqStutterGenotyper::calc_hap_aln_probs
// (src/seq_stutter_genotyper.cpp:517-523) at libltr_gpu.so.
//
// It includes the REFERENCE's own headers (Haplotype.h, HapBlock.h, AlignmentData.h, error.h):
// it is compiled inside LongTR's source tree, never inside this repository's product build.  The
// dev-container test compiles it against /root/reference/src (oracle/Makefile target `adapter`)
// so that a signature drift on either side breaks the build, and runs it on the golden loci.
//
// process_reads has the argument meaning of HapAligner::process_reads (HapAligner.h:137-138,
// HapAligner.cpp:545-581): aln_probs[(init_read_index+i)*H + k] and
// seed_positions[init_read_index+i] are written for realign_read[i] && realign_to_hap[k] only;
// errors end the process through printErrorAndDie like the reference's (error.cpp:6-10).
#ifndef GPU_HAP_ALIGNER_H_
#define GPU_HAP_ALIGNER_H_

#include <stdint.h>

#include <string>
#include <vector>

#include "ltr_gpu.h"

#include "AlignmentData.h"
#include "HapBlock.h"
#include "Haplotype.h"
// (printErrorAndDie: error.h, included by AlignmentData.h)

// One GPU context per process and device, shared by every GpuHapAligner / GpuHapAlignerBatch: the reference constructs a
// HapAligner per LOCUS (seq_stutter_genotyper.cpp:517), and a context (streams, model tables, device memory pool) is
// not a per-locus object.  Created on first use, kept to the end of the process; the alignment parameters are re-sent
// only when they differ from the ones the context holds.
// Threads: LongTR is single-threaded.  For a multi-threaded host the cache below is guarded (pthread mutex: the reference
// builds with -std=c++0x -pthread), and the library serialises the calls that stage in the context -- ltr_calc_hap_aln_probs,
// ltr_haplotype_align_to_ref -- per context (include/ltr_gpu.h, "Threads"); threads that should score side by side take a
// context each (ltr_ctx_create), they are cheap next to the device.  Changing the PARAMETERS of a shared context while
// another thread is inside a call on it is the caller's race, as it is with any shared object.
#include <pthread.h>
class GpuContext {
 public:
  static ltr_ctx* get(int device, const ltr_align_params& prm) {
    static const int kMaxDevices = 64;
    static ltr_ctx* ctx[kMaxDevices];
    static ltr_align_params held[kMaxDevices];
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    struct Guard { pthread_mutex_t* m; explicit Guard(pthread_mutex_t* m_) : m(m_) { pthread_mutex_lock(m); } ~Guard() { pthread_mutex_unlock(m); } } guard(&mu);
    if (device < 0 || device >= kMaxDevices) printErrorAndDie("GpuHapAligner: bad device ordinal");
    if (ctx[device] == NULL) {
      if (ltr_ctx_create(device, &ctx[device]) != LTR_OK) printErrorAndDie("GpuHapAligner: no usable HIP device (libltr_gpu has no CPU fallback)");
      ltr_default_params(&held[device]);
    }
    if (!same(held[device], prm)) {
      if (ltr_ctx_set_params(ctx[device], &prm) != LTR_OK) printErrorAndDie(std::string("GpuHapAligner: ") + ltr_last_error(ctx[device]));
      held[device] = prm;
    }
    return ctx[device];
  }
  // Host threads.  LongTR scales out as N single-threaded processes per node (README.md:78-82); this library's host loops run on
  // worker threads under a per-process budget (include/ltr_gpu.h, "host threads": affinity mask, cgroup quota, LOCAL_WORLD_SIZE).
  // A driver that starts its own processes and knows how many share the host says so once: 64 cores and 8 processes -> 8 threads each.
  static void set_local_ranks(int device, const ltr_align_params& prm, int n_local_ranks) {
    ltr_ctx* c = get(device, prm);
    if (ltr_ctx_set_host_threads(c, ltr_host_threads_rule(n_local_ranks)) != LTR_OK) printErrorAndDie(std::string("GpuHapAligner: ") + ltr_last_error(c));
  }
  // AlignmentModel(10, ...) of HapAligner.h:111-119 as ltr_align_params
  static ltr_align_params params(int indel_flank_len, int switch_old_align_len, const std::vector<float>& alignment_model_params) {
    ltr_align_params prm;
    ltr_default_params(&prm);                                   // HapAligner.h:118 defaults
    if (!alignment_model_params.empty()) {                      // HapAligner.h:111-116: ins->ins, ins->match, del->del, del->match, match->match, match->ins, match->del
      if (alignment_model_params.size() != 7) printErrorAndDie("GpuHapAligner: --alignment-params needs 7 values");
      prm.log_ins_to_ins = alignment_model_params[0]; prm.log_ins_to_match = alignment_model_params[1];
      prm.log_del_to_del = alignment_model_params[2]; prm.log_del_to_match = alignment_model_params[3];
      prm.log_match_to_match = alignment_model_params[4];
      prm.log_match_to_ins = alignment_model_params[5]; prm.log_match_to_del = alignment_model_params[6];
    }
    prm.indel_flank_len = indel_flank_len;
    prm.use_short_path = switch_old_align_len;
    return prm;
  }
 private:
  static bool same(const ltr_align_params& a, const ltr_align_params& b) {
    return a.log_ins_to_ins == b.log_ins_to_ins && a.log_ins_to_match == b.log_ins_to_match && a.log_del_to_del == b.log_del_to_del &&
           a.log_del_to_match == b.log_del_to_match && a.log_match_to_match == b.log_match_to_match && a.log_match_to_ins == b.log_match_to_ins &&
           a.log_match_to_del == b.log_match_to_del && a.indel_flank_len == b.indel_flank_len && a.use_short_path == b.use_short_path;
  }
};

class GpuHapAligner {
 private:
  ltr_ctx* ctx_;                       // the process-wide context of the device (GpuContext): borrowed, never destroyed here
  Haplotype* fw_haplotype_;
  std::vector<bool> realign_to_hap_;

  GpuHapAligner(const GpuHapAligner&);
  GpuHapAligner& operator=(const GpuHapAligner&);

 public:
  // Mirrors HapAligner(Haplotype*, std::vector<bool>&, int, int, std::vector<float>) (HapAligner.h:94-120)
  // plus the HIP device ordinal (one process per GPU: LOCAL_RANK).
  GpuHapAligner(Haplotype* haplotype, const std::vector<bool>& realign_to_haplotype, int indel_flank_len,
                int switch_old_align_len, const std::vector<float>& alignment_model_params, int device = 0)
      : ctx_(NULL), fw_haplotype_(haplotype), realign_to_hap_(realign_to_haplotype) {
    ctx_ = GpuContext::get(device, GpuContext::params(indel_flank_len, switch_old_align_len, alignment_model_params));
  }
  ~GpuHapAligner() {}

  // Haplotype -> ltr_haplotype_blocks (plain arrays owned by the caller-provided vectors).
  struct FlatHaplotype {
    std::vector<int32_t> start, end, period, n_alleles;
    std::vector<uint8_t> is_repeat, bytes;
    std::vector<int64_t> off;
    ltr_haplotype_blocks view;
  };
  static void flatten(Haplotype* hap, FlatHaplotype* f) {
    f->off.assign(1, 0);
    for (int b = 0; b < hap->num_blocks(); b++) {
      HapBlock* blk = hap->get_block(b);
      f->start.push_back(blk->start()); f->end.push_back(blk->end()); f->n_alleles.push_back(blk->num_options());
      f->is_repeat.push_back(blk->get_repeat_info() != NULL ? 1 : 0);
      f->period.push_back(blk->get_repeat_info() != NULL ? blk->get_repeat_info()->get_period() : 0);
      for (int k = 0; k < blk->num_options(); k++) {
        const std::string& s = blk->get_seq(k);
        f->bytes.insert(f->bytes.end(), s.begin(), s.end());
        f->off.push_back((int64_t)f->bytes.size());
      }
    }
    if (f->bytes.empty()) f->bytes.push_back(0);
    f->view.n_blocks = hap->num_blocks();
    f->view.block_start = f->start.data(); f->view.block_end = f->end.data(); f->view.is_repeat = f->is_repeat.data();
    f->view.period = f->period.data(); f->view.n_alleles = f->n_alleles.data();
    f->view.allele_bytes = f->bytes.data(); f->view.allele_off = f->off.data();
  }

  // std::vector<Alignment> -> ltr_alignment records (the CIGARs as plain arrays owned by `f`)
  struct FlatAlignments {
    std::vector<ltr_alignment> la;
    std::vector<std::string> ctype;
    std::vector<std::vector<int32_t> > cnum;
  };
  static void flatten(const std::vector<Alignment>& alignments, FlatAlignments* f) {
    f->la.resize(alignments.size()); f->ctype.resize(alignments.size()); f->cnum.resize(alignments.size());
    for (size_t i = 0; i < alignments.size(); i++) {
      const std::vector<CigarElement>& cl = alignments[i].get_cigar_list();
      for (size_t c = 0; c < cl.size(); c++) { f->ctype[i] += cl[c].get_type(); f->cnum[i].push_back(cl[c].get_num()); }
      if (f->cnum[i].empty()) f->cnum[i].push_back(0);
      ltr_alignment& a = f->la[i];
      a.start = alignments[i].get_start(); a.stop = alignments[i].get_stop();
      a.seq = (const uint8_t*)alignments[i].get_sequence().data();
      a.seq_len = (int32_t)alignments[i].get_sequence().size();
      a.n_cigar = (int32_t)cl.size();
      a.cigar_type = f->ctype[i].c_str(); a.cigar_num = f->cnum[i].data();
      a.qual = (const uint8_t*)alignments[i].get_base_qualities().data();
    }
  }

  // Same argument list as HapAligner::process_reads (HapAligner.h:137-138); base_quality is read
  // from the alignments themselves (Alignment::get_base_qualities) on the short path.
  void process_reads(const std::vector<Alignment>& alignments, int init_read_index, const BaseQuality* /*base_quality*/,
                     const std::vector<bool>& realign_read, double* aln_probs, int* seed_positions) {
    FlatHaplotype fh;
    flatten(fw_haplotype_, &fh);
    FlatAlignments fa;
    flatten(alignments, &fa);
    std::vector<ltr_alignment>& la = fa.la;
    std::vector<uint8_t> mh(realign_to_hap_.begin(), realign_to_hap_.end()), mr(realign_read.begin(), realign_read.end());
    static_assert(sizeof(int) == sizeof(int32_t), "seed_positions is int* in the reference");
    const int rc = ltr_process_reads(ctx_, &fh.view, mh.empty() ? NULL : mh.data(), la.empty() ? NULL : la.data(), (int32_t)la.size(),
                                     init_read_index, mr.empty() ? NULL : mr.data(), aln_probs, (int32_t*)seed_positions);
    if (rc != LTR_OK) printErrorAndDie(std::string("GpuHapAligner::process_reads: ") + ltr_last_error(ctx_));   // error.cpp:6-10
  }
};

#endif

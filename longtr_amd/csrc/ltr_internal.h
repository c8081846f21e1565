// ltr_internal.h -- shared between the HIP side (ltr_gpu.hip) and the host mirror (ltr_host.cpp).
#ifndef LTR_INTERNAL_H_
#define LTR_INTERNAL_H_

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <cstdio>
#include <cstdlib>

#include "../../include/ltr_gpu.h"

namespace ltr {

constexpr int kRefFlankLen = 35;        // REF_FLANK_LEN, reference HapAligner.cpp:245
constexpr double kImpossible = -1000000000.0;

// Window of the haplotype the DP sees: hap.substr(35-F, size-2*(35-F)) with
// std::string::substr's clamping and size_t wrap (reference HapAligner.cpp:246).
inline int64_t hap_window(int64_t hap_len, int flank, int64_t* pos_out) {
  const int64_t pos = kRefFlankLen - flank;
  int64_t cnt = hap_len - 2 * pos;
  const int64_t rest = hap_len - pos;
  if (cnt < 0 || cnt > rest) cnt = rest;
  *pos_out = pos;
  return cnt;
}

// ---- host-thread budget (round 6) -------------------------------------------------------------------------------------
// The reference is one thread per process and N processes per node (README.md:78-82).  This library's host loops (pooling,
// trimming, planning: ltr_calc_hap_aln_probs, seq_stutter_genotyper.cpp:514-563) run on worker threads, and eight ranks on a
// 64-core host must not start 8 x 32 of them.  The budget of a PROCESS (the worker pools are per process):
//   min(CPUs of the affinity mask, cgroup CPU quota rounded up, hardware threads) / ranks on this host, clamped to 1 .. 16.
// Ranks on this host: the launcher's LOCAL_WORLD_SIZE (torch.distributed.run sets it) unless the caller names the number
// (ltr_host_threads_rule) or the budget itself (ltr_ctx_set_host_threads).
constexpr int kMaxHostThreads = 16;
constexpr int kPrepAheadMinThreads = 12;      // the chunk pipeline's helper thread pays from here on (profiles/r05/prep_ahead_ab.log: level at ~12)

inline int detect_host_cpus() {
  int n = (int)std::thread::hardware_concurrency();
  if (n <= 0) n = 1;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0) n = std::min(n, a); }
  auto quota_cpus = [](const char* path_quota, const char* path_period) -> int {
    // cgroup v2: one file "max 100000" / "400000 100000"; cgroup v1: quota and period in files of their own (-1: no quota)
    long long q = -1, per = 100000;
    if (FILE* f = std::fopen(path_quota, "r")) {
      char word[64] = {0};
      if (std::fscanf(f, "%63s", word) == 1 && word[0] != 'm') q = std::atoll(word);
      if (!path_period) { long long p2 = 0; if (std::fscanf(f, "%lld", &p2) == 1 && p2 > 0) per = p2; }
      std::fclose(f);
    } else return 0;
    if (path_period) if (FILE* f = std::fopen(path_period, "r")) { long long p2 = 0; if (std::fscanf(f, "%lld", &p2) == 1 && p2 > 0) per = p2; std::fclose(f); }
    if (q <= 0) return 0;
    return (int)std::min<long long>((q + per - 1) / per, 1 << 20);
  };
  int q = quota_cpus("/sys/fs/cgroup/cpu.max", nullptr);
  if (q <= 0) q = quota_cpus("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
  if (q > 0) n = std::min(n, q);
  return std::max(n, 1);
}
// local_world_size <= 0: LOCAL_WORLD_SIZE of the environment (1 when unset)
inline int host_threads_rule(int local_world_size) {
  static const int cpus = detect_host_cpus();
  int lw = local_world_size;
  if (lw <= 0) { const char* e = std::getenv("LOCAL_WORLD_SIZE"); lw = e ? std::atoi(e) : 1; }
  if (lw <= 0) lw = 1;
  return std::max(1, std::min(cpus / lw, kMaxHostThreads));
}
inline std::atomic<int>& host_thread_setting() { static std::atomic<int> v(0); return v; }    // 0 = the rule
inline int host_thread_budget() {
  const int v = host_thread_setting().load(std::memory_order_relaxed);
  if (v > 0) return std::min(v, kMaxHostThreads);
  static const int rule = host_threads_rule(0);
  return rule;
}

// Host worker threads, started once per process and parked on a condition variable between jobs
// (every parallel_for of a plan used to start and join its own 15 threads: ~0.5 ms each, ten times per
// plan).  One job at a time; a caller that finds the pool busy (another context on another thread), a
// nested call from a worker, or a process forked while the pool existed falls back to short-lived threads.
// (Two pools: the second one serves the thread of ltr_calc_hap_aln_probs that prepares the next chunk of loci while the calling
// thread plans and launches the current one.)
class WorkerPool {
 public:
  static constexpr int kMaxWorkers = 15;
  static WorkerPool* get(int which = 0) {
    static std::once_flag once;
    std::call_once(once, []() {
      instance(0) = new WorkerPool(); instance(1) = new WorkerPool();
      (void)pthread_atfork(nullptr, nullptr, []() { instance(0) = nullptr; instance(1) = nullptr; });
    });
    return instance(which);                  // (leaked on purpose: parked workers outlive static destructors; null in a forked child)
  }
  static bool& on_worker() { static thread_local bool w = false; return w; }
  // job() on `extra` workers and on the caller; returns when all of them are back.  False: not run at all.
  template <class Job>
  bool run(int extra, Job&& job) {
    if (on_worker()) return false;
    std::unique_lock<std::mutex> busy(busy_mu_, std::try_to_lock);
    if (!busy.owns_lock()) return false;
    {
      std::lock_guard<std::mutex> lk(mu_);
      while ((int)threads_ < std::min(extra, kMaxWorkers)) {
        try { std::thread([this]() { worker(); }).detach(); } catch (...) { break; }
        ++threads_;
      }
      extra = std::min<int>(extra, (int)threads_);
      if (extra <= 0) return false;
      fn_ = [&job]() { job(); };
      tickets_ = extra; pending_ = extra; ++generation_;
    }
    cv_work_.notify_all();
    job();
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [this]() { return pending_ == 0; });
    fn_ = nullptr;
    return true;
  }

 private:
  static WorkerPool*& instance(int which) { static WorkerPool* p[2] = {nullptr, nullptr}; return p[which & 1]; }
  void worker() {
    on_worker() = true;
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_work_.wait(lk, [&]() { return generation_ != seen && tickets_ > 0; });
      seen = generation_;
      --tickets_;
      std::function<void()> fn = fn_;
      lk.unlock();
      fn();                                   // (the job catches its own exceptions)
      lk.lock();
      if (--pending_ == 0) cv_done_.notify_all();
    }
  }
  std::mutex busy_mu_, mu_;
  std::condition_variable cv_work_, cv_done_;
  std::function<void()> fn_;
  int threads_ = 0, tickets_ = 0, pending_ = 0;
  uint64_t generation_ = 0;
};

// f(i) for i in [0, n) on up to host_thread_budget() host threads (chunks of `grain` from a shared counter); serial when
// the range is too short to pay for the hand-over.  (pool, max_threads: see WorkerPool)
template <class F>
inline void parallel_for(int64_t n, int64_t min_per_thread, F&& f, int64_t grain = 64, int which_pool = 0, int max_threads = kMaxHostThreads) {
  const int hw = host_thread_budget();
  const int64_t nt = std::min<int64_t>(std::min<int64_t>(hw, std::max(max_threads, 1)), n / std::max<int64_t>(min_per_thread, 1));
  if (nt <= 1) { for (int64_t i = 0; i < n; ++i) f(i); return; }
  std::atomic<int64_t> next(0);
  // an exception on a worker thread would end the process (std::terminate): the first one is kept
  // and re-thrown on the calling thread after the join, where the entry points turn it into a status
  std::exception_ptr first_error;
  std::atomic<bool> failed(false);
  auto work = [&]() {
    try {
      for (;;) {
        const int64_t i0 = next.fetch_add(grain);
        if (i0 >= n || failed.load(std::memory_order_relaxed)) break;
        for (int64_t i = i0; i < std::min(i0 + grain, n); ++i) f(i);
      }
    } catch (...) {
      if (!failed.exchange(true)) first_error = std::current_exception();
    }
  };
  WorkerPool* pool = WorkerPool::get(which_pool);
  if (!pool || !pool->run((int)nt - 1, work)) {
    std::vector<std::thread> th;
    try {
      for (int64_t k = 1; k < nt; ++k) th.emplace_back(work);
    } catch (...) {                                            // could not start every thread: the ones running finish the range
    }
    work();
    for (std::thread& t : th) t.join();
  }
  if (failed.load()) std::rethrow_exception(first_error);
}

// Entry points never let an exception cross the C-ABI: LTR_GUARD(ctx, body) maps bad_alloc to
// LTR_ERR_NOMEM and anything else to LTR_ERR_INVALID.
#define LTR_GUARD_BEGIN try {
#define LTR_GUARD_END(ctx)                                                                                   \
  } catch (const std::bad_alloc&) { ltr::set_error(ctx, "out of host memory"); return LTR_ERR_NOMEM; }       \
  catch (const std::exception& e_) { ltr::set_error(ctx, std::string("internal error: ") + e_.what()); return LTR_ERR_INVALID; } \
  catch (...) { ltr::set_error(ctx, "internal error"); return LTR_ERR_INVALID; }

enum { kTimerHapBuild = 0, kTimerHapAln = 1, kTimerPosterior = 2, kTimerNwKernel = 3, kTimerShortKernel = 4 };   // (3, 4: kernel_ms only)
void add_time(ltr_ctx* ctx, int which, double seconds, double kernel_ms = 0.0);

// Wall-clock scope of one entry point; nested entry points (ltr_process_reads -> ltr_align_batch) count once.
struct TimedCall {
  ltr_ctx* ctx; int which; bool outer;
  std::chrono::steady_clock::time_point t0;
  static int& depth() { static thread_local int d = 0; return d; }
  TimedCall(ltr_ctx* c, int w) : ctx(c), which(w), outer(depth()++ == 0), t0(std::chrono::steady_clock::now()) {}
  ~TimedCall() {
    --depth();
    if (outer) add_time(ctx, which, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  }
};

// Measurement switches of a context (ltr_ctx_set_debug): A/B runs of tests/manual/*.py and profiles/collect.sh.
// 0 / unset = the library's own rule.  Scheduling only: results never depend on them.
struct DebugKnobs {
  int fan_lanes = 0;            // streams the certificate launches of a plan are dealt over (1 .. 4; rule: 2)
  int64_t fan_pairs = 0;        // plans with at least this many pairs stay on one stream
  int64_t chunks = 0;           // ltr_calc_hap_aln_probs: number of chunks (rule: 2 from 1500 loci up)
  int chunk_streams = 0;        // ... streams the chunks' plans alternate between (rule: 2)
  double chunk_growth = 0.0;    // ... weights 1, g, g^2, .. (g > 0); 1, 2, 3, .. (0); 1, 2, .., k, k, .., 2, 1 (g < 0); rule: 3
  bool chunk_growth_set = false;
  int fold_rounds = 0;          // automatic mode: fold launch classes below this many rounds of resident waves (rule: ltrp::kFoldRounds)
  int short_lane_kernel = 0;    // short path: the lane-per-pair kernel even where the wavefront-per-pair kernel applies (A/B)
  int plan_kernel = 0;          // A/B: 1 = a launch per class / the multi-width launches as before round 5; 0 (and -1, kept for old scripts) = the rule: the plan kernel for every automatic-mode plan
  int wave_clock = 0;           // 1: the plan kernel records every wavefront's first / last wall clock (ltr_plan_debug_wave_clocks)
  int chain = 0;                // 1 = the plan kernel's one-wave classes of strip widths 11 .. 20 by the chained walk (ltr_dp_chain.hpp; measured slower: off)
  int chain_min_w = 0, chain_max_w = 0;   // ... the chained walk for these strip widths only (0: 11 .. 20)
  int plan_share = 0;           // A/B: 1 = every wavefront of the plan kernel starts at the top of its table (default: spread over the entries in proportion to their work)
  int wg_first_pass = 0;        // first pass of the workgroup classes: 0 = what the context has learnt (rule), 1 = always the certificate kernels, 2 = always the threshold kernels (exact in one pass)
  int wgt_keep_waves = 0;       // A/B: 1 = the threshold first pass keeps the four waves of the wide four-wave classes (rule: eight waves, strips half as wide)
  int short_split = 0;          // 1: the seeded stutter path records events between its launches (ltr_ctx_short_kernel_split)
  int compact_plan = 0;         // A/B: -1 = no compact plans (every plan through the separate allocations, copies and fills of the large ones)
  int no_multi = 0;             // A/B: 1 = a launch per class (no multi-width launches), -1 = multi-width launches whatever the plan's size (rule: 512 .. 4096 pairs per CU)
  int pack_rule = 0;            // A/B: 3 = the per-length floor on the lanes per pair of the packed classes (rule until round 4), 2 = no floor at all
  int prep_ahead = 0;           // ltr_calc_hap_aln_probs: -1 = chunk c + 1 is pooled, trimmed and laid out only after chunk c's launches are queued (as before round 5); n > 0: the helper thread on, with n threads of its own; rule: on from a host-thread budget of kPrepAheadMinThreads, with the whole budget
  int trace = 0;                // ltr_calc_hap_aln_probs prints a timestamped phase profile to stderr
};
DebugKnobs ctx_debug(const ltr_ctx* ctx);

void set_error(ltr_ctx* ctx, const std::string& msg);
ltr_align_params ctx_params(const ltr_ctx* ctx);
ltr_stutter_params ctx_stutter_params(const ltr_ctx* ctx);
int ctx_device(const ltr_ctx* ctx);
void* ctx_stream(const ltr_ctx* ctx);      // hipStream_t
int ctx_pool_alloc(ltr_ctx* ctx, void** out, size_t bytes);   // device memory from the context's pool; 0 = ok, else a hipError_t
void ctx_pool_release(ltr_ctx* ctx, void* p);
void* ctx_big_scratch(ltr_ctx* ctx, size_t bytes);   // one grow-only device block kept by the context (NW trace); NULL = out of memory; one user at a time
std::unique_lock<std::mutex> ctx_call_lock(ltr_ctx* ctx);   // held for a whole ltr_calc_hap_aln_probs / haplotype-alignment call: they stage in the two arrays below / in ctx_big_scratch
uint8_t* ctx_host_bytes(ltr_ctx* ctx, int which, size_t bytes);   // one of four grow-only staging arrays kept by the context (uninitialised)
void* ctx_side_stream(const ltr_ctx* ctx, int k);
void ctx_note_short_split(ltr_ctx* ctx, const double ms4[4]);   // ltr_short.hip -> the context: device time of the seeded path's launches, summed over calls   // k % 8 == 0: the context's stream, else one of its seven side streams

// HapAligner::process_reads with short_ == 1 (ltr_short.hip)
int process_reads_short(ltr_ctx* ctx, const ltr_haplotype_blocks* hap, const uint8_t* realign_to_hap,
                        const ltr_alignment* alns, int32_t n_alns, int32_t init_read_index,
                        const uint8_t* realign_read, double* aln_probs, int32_t* seed_positions);

// The same in two halves, for many loci in one launch (ltr_calc_hap_aln_probs): add() prepares a
// locus on the host (and writes its seeds / all-zero rows at once), run() scores everything queued.
// aln_probs passed to add() must stay valid until run().
struct ShortBatch;
ShortBatch* short_batch_new();
void short_batch_free(ShortBatch* b);
int short_batch_add(ltr_ctx* ctx, ShortBatch* b, const ltr_haplotype_blocks* hap, const uint8_t* realign_to_hap,
                    const ltr_alignment* alns, int32_t n_alns, int32_t init_read_index,
                    const uint8_t* realign_read, double* aln_probs, int32_t* seed_positions);
int short_batch_merge(ltr_ctx* ctx, ShortBatch* dst, ShortBatch* src);   // dst += src; src is left empty
int short_batch_run(ltr_ctx* ctx, ShortBatch* b);

// Haplotype::next() order (reference Haplotype.cpp:123-196): allele index per block for
// every combination, combination-major.
int haplotype_counts(const ltr_haplotype_blocks* hap, std::vector<int32_t>* counts, int64_t* ncombs);

}  // namespace ltr

#endif

// ltr_gpu.hip -- MI355X (gfx950) implementation of the LongTR read-vs-haplotype alignment DP.
//
// Replaces HapAligner::align_seq_to_hap (reference src/SeqAlignment/HapAligner.cpp:236-343)
// for whole batches of (pooled read, candidate haplotype) pairs, behind the C-ABI of
// include/ltr_gpu.h.  Design notes live in DESIGN.md; the short version:
//
//  * device side: ltr_dp_kernel.hpp (anti-diagonal wavefront DP, one wave per pair, DPP
//    hand-off between lanes, column blocks with boundary strips, certificate + exact redo);
//  * host side (this file): context / model tables / plan (pack, bin by strip width, upload),
//    launches (one persistent grid per strip width, longest pairs first), the C-ABI.
//
// No CPU fallback exists in this file: every compute entry point fails with
// LTR_ERR_NO_DEVICE when there is no HIP device.

#include <hip/hip_runtime.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <set>
#include <string>
#include <vector>

#include <atomic>
#include "ltr_internal.h"
#include "ltr_kernels.h"
#include "ltr_plan.h"

#define LTR_VERSION_STR "longtr_amd 0.6 (gfx950; ABI 6)"
static double ltr_dbg_ms() {
  static const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
static std::atomic<int> g_trace{0};             // ltr_ctx_set_debug(ctx, "trace", 1) on any context: phase prints to stderr
#define LTR_DBG(...) do { if (g_trace.load(std::memory_order_relaxed)) { std::fprintf(stderr, "[ltr %10.2f ms] ", ltr_dbg_ms()); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); std::fflush(stderr); } } while (0)

// ------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------
namespace {

// The LUT kernels stream each haplotype base as the byte offset of its block of the emission table
// ('A','C','T','G' -> ((byte >> 1) & 3) * 4096; the zero padding maps to 0): formed here from the uploaded bytes,
// one pass at HBM speed instead of a host loop plus a second upload twice the size.
__global__ __launch_bounds__(256) void ltr_hap_codes_kernel(const uint8_t* __restrict__ haps, uint16_t* __restrict__ codes, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (size_t k = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; k < n; k += stride) {
    if (k + 4 <= n) {
      const uint32_t w = *(const uint32_t*)(haps + k);           // (hipMalloc'ed / pool blocks: 256-byte aligned)
      ushort4 c;
      c.x = (uint16_t)(((w >> 1) & 3u) << 12); c.y = (uint16_t)(((w >> 9) & 3u) << 12);
      c.z = (uint16_t)(((w >> 17) & 3u) << 12); c.w = (uint16_t)(((w >> 25) & 3u) << 12);
      *(ushort4*)(codes + k) = c;
    } else {
      for (size_t i = k; i < n; ++i) codes[i] = (uint16_t)((((uint32_t)haps[i] >> 1) & 3u) << 12);
    }
  }
}

// ------------------------------------------------------------------------------------------
// posterior kernel (consumer): Genotyper::calc_log_sample_posteriors, genotyper.cpp:45-83
// one workgroup per sample; thread (a1,a2) loops over the sample's reads in read order.
// ------------------------------------------------------------------------------------------
__global__ void ltr_posterior_kernel(int S, int R, int H, double* __restrict__ ll,
                                     const double* __restrict__ lp1, const double* __restrict__ lp2,
                                     const int* __restrict__ sample_label, double homoz, double hetz,
                                     double* __restrict__ post) {
  const int s = blockIdx.x;
  const double LOG_ONE_HALF = -0.6931471805599453094;          // log(0.5), mathops.cpp:10
  const int nd = H * H;
  for (int idx = threadIdx.x; idx < nd; idx += blockDim.x) {
    const int a1 = idx / H, a2 = idx % H;
    double acc = (a1 == a2) ? homoz : hetz;                    // init_log_sample_priors, :35-43
    for (int r = 0; r < R; ++r) {
      if (sample_label[r] != s) continue;
      double v1 = ll[(size_t)r * H + a1], v2 = ll[(size_t)r * H + a2];
      if (v1 < -600.0) v1 = -600.0;                            // clamp, :57-58 (written back by the clamp kernel)
      if (v2 < -600.0) v2 = -600.0;
      acc += log(exp(v1 + lp1[r] + LOG_ONE_HALF) + exp(v2 + lp2[r] + LOG_ONE_HALF));   // :59
    }
    post[(size_t)s * nd + idx] = acc;
  }
}

__global__ void ltr_clamp_kernel(double* ll, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count && ll[i] < -600.0) ll[i] = -600.0;
}

// per sample: log_sum_exp normalise (genotyper.cpp:67-75, mathops.cpp:47-53) + argmax (:85-100).
// Single thread per sample on purpose: the sum must run in index order to match the
// reference's rounding, and H*H is tiny.
__global__ void ltr_posterior_finish_kernel(int S, int H, double* __restrict__ post,
                                            double* __restrict__ stl, int* __restrict__ gts) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const int nd = H * H;
  double* p = post + (size_t)s * nd;
  double mx = p[0];
  for (int k = 1; k < nd; ++k) if (mx < p[k]) mx = p[k];
  double tot = 0.0;
  for (int k = 0; k < nd; ++k) tot += exp(p[k] - mx);
  const double total = mx + log(tot);
  stl[s] = total;
  double best = -1.7976931348623157e308; int b1 = -1, b2 = -1;
  for (int k = 0; k < nd; ++k) {
    const double v = p[k] - total;
    p[k] = v;
    if (v > best) { best = v; b1 = k / H; b2 = k % H; }
  }
  gts[2 * s] = b1; gts[2 * s + 1] = b2;
}

// ---- batched, plan-resident posteriors: one workgroup per (locus, sample) --------------------
struct PostUnit {            // one (locus, sample)
  int64_t ll_off;            // locus block in the LL buffer ([P x H])
  int64_t post_off;          // this unit's [H x H] block in the posterior buffer
  int32_t r0, r1;            // reads of the locus
  int32_t H, sample;
  double homoz, hetz;        // priors, genotyper.cpp:21-33 (host libm)
};

__global__ void ltr_posterior_batch_kernel(const PostUnit* __restrict__ units, const double* __restrict__ ll,
                                           const int32_t* __restrict__ pool_index, const double* __restrict__ lp1,
                                           const double* __restrict__ lp2, const int32_t* __restrict__ label,
                                           double* __restrict__ post) {
  const PostUnit u = units[blockIdx.x];
  const double LOG_ONE_HALF = -0.6931471805599453094;          // log(0.5), mathops.cpp:10
  const int H = u.H, nd = H * H;
  for (int idx = threadIdx.x; idx < nd; idx += blockDim.x) {
    const int a1 = idx / H, a2 = idx % H;
    double acc = (a1 == a2) ? u.homoz : u.hetz;
    for (int r = u.r0; r < u.r1; ++r) {                         // reads in order, like :52-63
      if (label[r] != u.sample) continue;
      const double* row = ll + u.ll_off + (int64_t)pool_index[r] * H;   // the read's pool row (seq_stutter_genotyper.cpp:531-537)
      double v1 = row[a1], v2 = row[a2];
      if (v1 < -600.0) v1 = -600.0;                              // :57-58
      if (v2 < -600.0) v2 = -600.0;
      acc += log(exp(v1 + lp1[r] + LOG_ONE_HALF) + exp(v2 + lp2[r] + LOG_ONE_HALF));
    }
    post[u.post_off + idx] = acc;
  }
}

// normalise + argmax per unit (index-ordered sum, first maximum: genotyper.cpp:67-75, :85-100)
__global__ void ltr_posterior_batch_finish_kernel(int n_units, const PostUnit* __restrict__ units, double* __restrict__ post,
                                                  double* __restrict__ stl, int* __restrict__ gts) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_units) return;
  const int H = units[k].H, nd = H * H;
  double* p = post + units[k].post_off;
  double mx = p[0];
  for (int i = 1; i < nd; ++i) if (mx < p[i]) mx = p[i];
  double tot = 0.0;
  for (int i = 0; i < nd; ++i) tot += exp(p[i] - mx);
  const double total = mx + log(tot);
  stl[k] = total;
  double best = -1.7976931348623157e308; int b1 = -1, b2 = -1;
  for (int i = 0; i < nd; ++i) {
    const double v = p[i] - total;
    p[i] = v;
    if (v > best) { best = v; b1 = i / H; b2 = i % H; }
  }
  gts[2 * k] = b1; gts[2 * k + 1] = b2;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side: context / plan
// ------------------------------------------------------------------------------------------
// Device allocations of a context are recycled: a plan for one locus needs ten small buffers, and
// hipMalloc / hipFree (a device-wide synchronisation each) would dominate the per-locus call.
// Blocks up to 64 MB are rounded to a power of two and parked here on release (at most 512 MB);
// larger ones go straight back to the runtime.
// Grow-only host array of trivially copyable elements that keeps its storage between uses and never
// initialises it: the per-plan work arrays (tens of MB: descriptors, costs, sort order) would otherwise be
// mapped, zero-filled page by page and unmapped again for every plan.
template <class T>
struct RawBuf {
  T* p = nullptr; size_t n = 0, cap = 0;
  // pinned: page-locked host memory (hipHostMalloc) -- for arrays the library uploads itself: a copy from pinned memory goes
  // over the DMA engines; from pageable memory it is staged by a copy KERNEL that waits for wave slots behind the persistent
  // DP launches of the previous chunk (measured on MI355X, ltr_calc_hap_aln_probs on the catalogue: the uploads of chunks
  // 1 and 2 took 1.1 - 1.6 ms against 0.3 - 0.4 ms for chunk 0, which finds the GPU idle).  Falls back to malloc.
  bool pinned = false, p_is_pinned = false;
  RawBuf() = default;
  RawBuf(const RawBuf&) = delete;
  RawBuf& operator=(const RawBuf&) = delete;
  ~RawBuf() { release(); }
  void release() { if (p) { if (p_is_pinned) (void)hipHostFree(p); else std::free(p); } p = nullptr; cap = 0; n = 0; }
  // NOTE: the contents are UNDEFINED after a growth (resize is not std::vector's: every user rewrites the whole buffer).
  void resize(size_t m) {
    if (m > cap) {
      const size_t c = std::max(m + m / 4, (size_t)1024);
      // (the contents are never kept across a growth: the old block goes first, so that a context never holds both -- hundreds of
      // MB of pinned memory each on a 30 000-locus call; hipHostMallocPortable: usable from whichever device is current)
      release();
      T* q = nullptr;
      bool q_pinned = false;
      if (pinned) {
        void* v = nullptr;
        if (hipHostMalloc(&v, c * sizeof(T), hipHostMallocPortable) == hipSuccess) { q = (T*)v; q_pinned = true; }
        else (void)hipGetLastError();
      }
      if (!q) q = (T*)std::malloc(c * sizeof(T));
      if (!q) throw std::bad_alloc();
      p = q; cap = c; p_is_pinned = q_pinned;
    }
    n = m;
  }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T* data() { return p; }
  T* begin() { return p; }
  T* end() { return p + n; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

struct DevPool {
  // (sized for 288 GB of HBM: the blocks of a 10 k-locus plan -- 130 MB of reads, 180 MB of haplotype codes -- are
  // parked too, so a pipeline of large plans never waits in hipMalloc / hipFree, which synchronise the device)
  static constexpr size_t kMaxBlock = (size_t)2 << 30, kMaxCached = (size_t)8 << 30;
  std::multimap<size_t, void*> idle;
  std::map<void*, size_t> live;
  size_t cached = 0;
  std::mutex mu;
  static size_t size_class(size_t n) { size_t c = 256; while (c < n) c <<= 1; return c; }
  hipError_t alloc(void** out, size_t n) {
    std::lock_guard<std::mutex> lk(mu);
    size_t c = n;
    if (n <= kMaxBlock) {
      c = size_class(n);
      auto it = idle.find(c);
      if (it != idle.end()) { *out = it->second; idle.erase(it); cached -= c; live[*out] = c; return hipSuccess; }
    }
    hipError_t e = hipMalloc(out, c);
    if (e == hipErrorOutOfMemory) {                             // give the parked blocks (up to 8 GB) back and try once more
      for (auto& kv : idle) (void)hipFree(kv.second);
      idle.clear(); cached = 0;
      (void)hipGetLastError();
      e = hipMalloc(out, c);
    }
    if (e == hipSuccess) live[*out] = c;
    return e;
  }
  void release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu);
    auto it = live.find(p);
    if (it == live.end()) { (void)hipFree(p); return; }
    const size_t c = it->second;
    live.erase(it);
    if (c <= kMaxBlock && cached + c <= kMaxCached) { idle.emplace(c, p); cached += c; }
    else (void)hipFree(p);
  }
  void clear() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& kv : idle) (void)hipFree(kv.second);
    idle.clear(); cached = 0;
  }
};

struct ltr_ctx {
  int device = -1;
  DevPool pool;
  std::set<ltr_plan*> plans;            // plans created on this context and not destroyed yet (under mu)
  // host work arrays of ltr_plan_create (used under mu) and the chunk staging bytes of ltr_calc_hap_aln_probs
  struct PlanScratch {
    RawBuf<PairDesc> pairs, sorted; RawBuf<int16_t> key, bin; RawBuf<int32_t> order; RawBuf<uint8_t> read_acgt, hap_acgt;
  } scratch;
  RawBuf<uint8_t> host_bytes[4];         // (two pairs: the chunks of ltr_calc_hap_aln_probs alternate, one is laid out while the other is uploaded)
  void* d_big = nullptr; size_t big_bytes = 0;      // ctx_big_scratch
  hipStream_t stream = nullptr;
  hipStream_t up_stream = nullptr;      // device-side input preparation of new plans (never behind another plan's DP kernels)
  static constexpr int kAux = 10;
  hipStream_t aux[kAux] = {};          // side streams: independent plans (the chunks of ltr_calc_hap_aln_probs) run side by side
  ltr_align_params params;
  ltr_stutter_params stutter;
  ModelConsts mc;
  // device model tables
  int64_t table_len = 0;
  double* d_lpc = nullptr;
  double* d_colXZ = nullptr;
  double* d_thr = nullptr;               // exact row-test thresholds of the LUT exact kernels (ltrp::build_threshold_table)
  double* d_row0XY = nullptr;            // first row: record j = {X(0,j), Y(0,j)} for emit(hap[j], read[0]) = mismatch, match (packed kernels)
  std::string arch;
  int n_cu = 0, clock_mhz = 0;
  int pair_packing = -1;                // two pairs per wavefront: -1 by batch size, 0 never, 1 whenever the read fits
  // resident workgroups per launch class (occupancy x CUs), asked from the runtime once per context
  bool have_grids = false;
  int full_grid[ltrp::kNumFast] = {0};
  int full_multi_grid = 0;              // the multi-width one-wave launch
  int full_pmulti_grid = 0;             // ... packed launch
  int full_plan_grid = 0;               // the plan kernel
  int full_x_wide_grid = 0;             // the W = 20 exact kernel (reads of 1026 .. 1281 bases out of the 4-wave list)
  int full_wgt_grid[2][kWgWMax + 1] = {{0}};   // threshold kernels as the first pass of a workgroup class: [0] four waves, [1] eight, by (even) strip width
  int full_redo_grid = 0;               // ... of the exact kernels
  int full_x_grid[kNumExact] = {0};
  ltr::DebugKnobs dbg;                  // ltr_ctx_set_debug
  // First pass of the workgroup classes (ltr_dp_wg.hpp): certificate kernels (11 operations a cell; a pair whose certificate
  // fails is scored a second time by an exact kernel) or threshold kernels (13 operations, exact in one pass).  Which one pays
  // depends on the READS -- HiFi reads finish, ONT reads under the default model abort (every pair of BASELINE config 5) -- so
  // the context learns it: every execute leaves {pairs the first pass could not finish, pairs it scored} of its workgroup
  // classes in a pinned slot, and the next execute reads the slots that have arrived (wg_stats_poll; never a wait).
  struct WgStatSlot { hipEvent_t ev = nullptr; bool busy = false; int mode = 0; int epoch = 0; };
  static constexpr int kWgStatSlots = 8;
  WgStatSlot wg_stat[kWgStatSlots];
  uint32_t* wg_stat_pin = nullptr;      // kWgStatSlots x 2 words, pinned
  int wg_thr_first = 0;                 // 1: the threshold kernels go first
  int wg_epoch = 0;                     // bumped by ltr_ctx_set_params: slots of the old model are ignored
  uint32_t wg_last_unfinished = 0, wg_last_scored = 0;     // the last slot read (ltr_ctx_wg_first_pass)
  std::string err;
  std::mutex mu;
  std::mutex pin_mu;                    // the pinned download block below (ltr_plan_fetch)
  void* pin = nullptr; size_t pin_bytes = 0;
  hipStream_t copy_stream = nullptr;
  // compact plans: the pinned image the host fills (one at a time: compact_ev = the copy out of it), recycled events, recycled
  // pinned blocks for the scores (hipEventCreate / hipHostMalloc per one-locus call would cost more than the call's kernel)
  RawBuf<uint8_t> compact_stage;
  hipEvent_t compact_ev = nullptr; bool compact_ev_pending = false;
  std::mutex cache_mu;
  std::vector<hipEvent_t> ev_cache[2];  // [0]: hipEventDisableTiming, [1]: timing
  struct PinBlock { void* p; void* dev; size_t cap; bool busy; };     // dev: the address the device reaches it by
  std::vector<PinBlock> pin_blocks;
  std::mutex call_mu;                   // one ltr_calc_hap_aln_probs / NW call at a time per context: they stage in host_bytes / d_big (ctx_call_lock)
  std::mutex err_mu;                    // error text and timers are written from worker threads too
  ltr_timers tm = {};
  double short_split_ms[4] = {0, 0, 0, 0};   // ltr_ctx_set_debug "short_split": [prep + flank rows before the block, block row, flank rows after, seed log-sum]
};

namespace ltr {
void set_error(ltr_ctx* ctx, const std::string& msg) { if (ctx) { std::lock_guard<std::mutex> lk(ctx->err_mu); ctx->err = msg; } }
void add_time(ltr_ctx* ctx, int which, double seconds, double kernel_ms) {
  if (!ctx) return;
  std::lock_guard<std::mutex> lk(ctx->err_mu);
  if (which == kTimerHapBuild) { ctx->tm.hap_build_s += seconds; if (seconds > 0) ctx->tm.hap_build_calls++; }
  else if (which == kTimerHapAln) { ctx->tm.hap_aln_s += seconds; if (seconds > 0) ctx->tm.hap_aln_calls++; }
  else if (which == kTimerPosterior) { ctx->tm.posterior_s += seconds; if (seconds > 0) ctx->tm.posterior_calls++; }
  if (which == kTimerNwKernel) ctx->tm.nw_kernel_ms += kernel_ms;
  else if (which == kTimerShortKernel) ctx->tm.short_kernel_ms += kernel_ms;
  else ctx->tm.dp_kernel_ms += kernel_ms;
}
void ctx_note_short_split(ltr_ctx* ctx, const double ms4[4]) { std::lock_guard<std::mutex> lk(ctx->err_mu); for (int k = 0; k < 4; ++k) ctx->short_split_ms[k] += ms4[k]; }
ltr_align_params ctx_params(const ltr_ctx* ctx) { return ctx->params; }
DebugKnobs ctx_debug(const ltr_ctx* ctx) { return ctx->dbg; }
ltr_stutter_params ctx_stutter_params(const ltr_ctx* ctx) { return ctx->stutter; }
int ctx_device(const ltr_ctx* ctx) { return ctx->device; }
void* ctx_stream(const ltr_ctx* ctx) { return (void*)ctx->stream; }
int ctx_pool_alloc(ltr_ctx* ctx, void** out, size_t bytes) { return (int)ctx->pool.alloc(out, bytes); }
void ctx_pool_release(ltr_ctx* ctx, void* p) { ctx->pool.release(p); }
void* ctx_big_scratch(ltr_ctx* ctx, size_t bytes) {
  if (bytes > ctx->big_bytes) {
    if (ctx->d_big) { (void)hipDeviceSynchronize(); (void)hipFree(ctx->d_big); ctx->d_big = nullptr; ctx->big_bytes = 0; }
    if (hipMalloc(&ctx->d_big, bytes) != hipSuccess) { (void)hipGetLastError(); ctx->d_big = nullptr; return nullptr; }
    ctx->big_bytes = bytes;
  }
  return ctx->d_big;
}
std::unique_lock<std::mutex> ctx_call_lock(ltr_ctx* ctx) { return std::unique_lock<std::mutex>(ctx->call_mu); }
uint8_t* ctx_host_bytes(ltr_ctx* ctx, int which, size_t bytes) {
  (void)hipSetDevice(ctx->device);              // (the caller may be a helper thread of ltr_calc_hap_aln_probs: pinned memory is allocated against the context's device)
  ctx->host_bytes[which & 3].resize(bytes); return ctx->host_bytes[which & 3].data(); }
void* ctx_side_stream(const ltr_ctx* ctx, int k) { k %= (ltr_ctx::kAux + 1); return (void*)(k == 0 ? ctx->stream : ctx->aux[k - 1]); }
}

namespace {

constexpr int kReadPad = 1024;                  // bytes behind the device read buffer: a packed kernel's lane loads its strip (up to 641 + 24 bytes past a read's start) unclamped
constexpr size_t kCompactImageMax = (size_t)1 << 20;   // a plan whose device image (control words, tables, pairs, reads, haplotypes + codes) is at most this is uploaded as ONE block (ltr_plan_create)
constexpr size_t kCompactLlMax = (size_t)256 << 10;    // ... and its scores go straight into pinned host memory
constexpr int kHapPad = 96;                     // zero bytes either side of the device haplotype buffer
using namespace ltrp;                            // class table, Rules, classify_pair, sort_by_class (ltr_plan.h)

#define HIP_TRY(ctx, call)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      ltr::set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_));                \
      return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? LTR_ERR_NO_DEVICE : LTR_ERR_HIP; \
    }                                                                                        \
  } while (0)

void fill_model_consts(const ltr_align_params& p, ModelConsts* mc) {
  mc->a = p.log_ins_to_ins; mc->b = p.log_ins_to_match; mc->c = p.log_del_to_del; mc->d = p.log_del_to_match;
  mc->e = p.log_match_to_match; mc->f = p.log_match_to_ins; mc->g = p.log_match_to_del;
  mc->match = (float)(-0.000100005);     // float MATCH = -0.000100005;  HapAligner.cpp:261
  mc->mismatch = (float)(-9.0);          // float MISMATCH = -9.0;       HapAligner.cpp:260
  volatile float mf = mc->match + mc->f; // float + float, evaluated in float (HapAligner.cpp:277)
  mc->match_plus_f = mf;
}

// Boundary tables of the first row / first column (HapAligner.cpp:267-280): pure functions
// of the model, so they are built once per parameter set instead of once per pair.
int build_tables(ltr_ctx* ctx, int64_t len, bool same_size = false) {
  // (the kernels stream the column table with a pointer that keeps advancing while the last lanes
  // drain: keep 80 records of slack beyond the longest haplotype)
  if (!same_size) {
    if (len + 80 <= ctx->table_len) return LTR_OK;
    len = std::max<int64_t>(len + len / 4 + 80, 4096);
  }
  const ModelConsts& mc = ctx->mc;
  const double IMP = ltr::kImpossible;
  std::vector<double> lpc(len + 2), cx[2], cz[2];
  lpc[0] = 0.0; lpc[1] = 0.0;
  for (int64_t j = 1; j <= len; ++j) lpc[j + 1] = lpc[j] + (double)mc.c;       // left_prob += LOG_DEL_TO_DEL
  for (int e = 0; e < 2; ++e) {
    cx[e].assign(len + 2, IMP); cz[e].assign(len + 2, IMP);
    const double emit = e ? (double)mc.match : (double)mc.mismatch;
    double lpa = 0.0;                       // left_prob of the column loop
    double I_prev = IMP;                    // insertion_matrix[0]
    for (int64_t i = 1; i <= len + 1; ++i) {
      const double Mv = (I_prev + (double)mc.b) + emit;        // match_matrix[i*m], :276
      const double Iv = (double)mc.match_plus_f + lpa;         // insertion_matrix[i*m], :277
      const double Dv = IMP;                                    // :278
      cx[e][i] = std::max(Mv + (double)mc.e, std::max(Dv + (double)mc.d, Iv + (double)mc.b));
      cz[e][i] = std::max(Mv + (double)mc.g, Dv + (double)mc.c);
      I_prev = Iv;
      lpa += (double)mc.a;                                      // :279
    }
  }
  auto up = [&](double** dst, const std::vector<double>& src) -> int {
    if (*dst) (void)hipFree(*dst);
    *dst = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)dst, src.size() * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(double), hipMemcpyHostToDevice));
    return LTR_OK;
  };
  // plans may be executing on caller-supplied streams: nothing may still read the old tables
  HIP_TRY(ctx, hipDeviceSynchronize());
  int rc;
  if ((rc = up(&ctx->d_lpc, lpc))) return rc;
  {
    std::vector<double> xz((size_t)(len + 2) * 4);
    for (int64_t i = 0; i < len + 2; ++i)
      for (int e = 0; e < 2; ++e) { xz[(size_t)(i * 4 + e * 2)] = cx[e][i]; xz[(size_t)(i * 4 + e * 2 + 1)] = cz[e][i]; }
    if ((rc = up(&ctx->d_colXZ, xz))) return rc;
  }
  {
    // first row (HapAligner.cpp:267-272) as the packed kernels consume it: X(0,j), Y(0,j) -- the two max-terms row 1
    // reads -- for both outcomes of the row's emission test; the operations and their order are the kernels' own
    // set-up code (ltr_dp_kernel.hpp, column_block: row0), so the bits are
    std::vector<double> xy((size_t)(len + 2) * 4, IMP);
    const double cg = (double)mc.g, cd = (double)mc.d, ce = (double)mc.e, cb = (double)mc.b, cf = (double)mc.f, ca = (double)mc.a;
    for (int64_t j = 1; j <= len + 1; ++j)
      for (int e = 0; e < 2; ++e) {
        const double lp1 = lpc[(size_t)std::max<int64_t>(j - 1, 0)], lp = lpc[(size_t)j];
        const double D0jm1 = (j == 1) ? IMP : (cg + lp1);                 // deletion_matrix[j-1]
        const double D0j = cg + lp;                                       // deletion_matrix[j] = g + left_prob
        const double M0 = (D0jm1 + cd) + (e ? (double)mc.match : (double)mc.mismatch);
        xy[(size_t)(j * 4 + e * 2)] = std::max(M0 + ce, std::max(D0j + cd, IMP + cb));
        xy[(size_t)(j * 4 + e * 2 + 1)] = std::max(M0 + cf, IMP + ca);
      }
    if ((rc = up(&ctx->d_row0XY, xy))) return rc;
  }
  {
    std::vector<double> thr((size_t)kPenTabDoubles);
    ltrp::build_threshold_table(mc.c, thr.data());
    if ((rc = up(&ctx->d_thr, thr))) return rc;
  }
  ctx->table_len = len;
  return LTR_OK;
}

int validate_params(const ltr_align_params* p) {
  if (!p) return LTR_ERR_INVALID;
  if (p->indel_flank_len < 0 || p->indel_flank_len > ltr::kRefFlankLen) return LTR_ERR_INVALID;
  const float v[7] = {p->log_ins_to_ins, p->log_ins_to_match, p->log_del_to_del, p->log_del_to_match,
                      p->log_match_to_match, p->log_match_to_ins, p->log_match_to_del};
  for (float x : v) if (!(x < 0.0f) || !std::isfinite(x)) return LTR_ERR_INVALID;   // hipstr_main.cpp:429-430
  return LTR_OK;
}

}  // namespace

#ifdef LTR_KDEBUG
static uint32_t* g_dbg_host = nullptr;
extern "C" uint32_t* ltr_debug_buffer(void) { return g_dbg_host; }
#endif

struct ltr_plan {
  ltr_ctx* ctx = nullptr;
  int64_t n_pairs = 0, ll_size = 0, n_reads = 0;
  double cells = 0.0, input_bytes = 0.0;
  int32_t max_len = 0;
  // device buffers
  uint8_t* d_reads = nullptr; uint8_t* d_haps = nullptr; uint16_t* d_hap_codes = nullptr;
  PairDesc* d_pairs = nullptr;
  double* d_ll = nullptr;
  uint32_t* d_queue = nullptr;          // one counter per bin
  double* d_scratch = nullptr;
  int32_t scratch_stride = 0;
  int bin_first[kNumKernels + 1] = {0};    // classes kNumFast + c: pairs that start out in exact list c (non-ACGT pairs; mode 4: all)
  int bin_grid[kNumFast] = {0};
  bool bin_small[kNumFast] = {false};     // the class cannot fill the GPU's wave slots once
  int max_grid = 0;
  int max_grid_wide = 1;                // grid of the W = 20 exact launch (candidates of the 4-wave list)
  std::vector<int32_t> seed;            // host: read length - 1 (or -1 when the read is masked out)
  double* last_out = nullptr;
  hipStream_t last_stream = nullptr;
  std::vector<hipStream_t> streams;     // every stream an execute of this plan was queued on (synchronised before its buffers are released)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_up = nullptr;            // device-side input preparation (hap codes) done
  int fan_lanes = 1;                     // certificate launches dealt over this many streams (own scratch region each)
  size_t scratch_lane_stride = 0;        // doubles per stream region of d_scratch
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fast = nullptr, ev_x[kNumExact + 1] = {nullptr};   // exact launches side by side: after the certificate launches / joined back (+ 1: the W = 20 launch)
  hipEvent_t ev_close[kNumExact][4] = {{nullptr}};   // "every certificate launch that can feed exact list c has been queued", one per launch stream
  std::vector<int> order;               // certificate classes with pairs, longest reads first: the launch order
  int order_pos[kNumKernels] = {0};     // position of every class in it (-1: empty class); exact class c: order.size() + c
  int32_t cls_cmax[kNumFast] = {0};     // longest read (columns, m - 1) of every certificate class
  // the one-wave classes of strip widths kMultiMinW .. kWMax as ONE persistent launch (ltr_dp_multi_kernel), listed under
  // the widest of them: multi_rep (-1: no such launch), its classes widest first in multi_classes
  int multi_rep = -1, multi_grid = 0;
  bool multi_small = false;
  std::vector<int> multi_classes;
  // ... and the packed launches of strip widths kPackMultiMinW .. kPackWMax (ltr_dp_pack_multi_kernel), listed under the
  // representative class of the widest of them: pmulti_rep (-1: none), the representatives of its widths widest first
  int pmulti_rep = -1, pmulti_grid = 0;
  bool pmulti_small = false;
  std::vector<int> pmulti_reps;
  PackTable* d_pk_tabs = nullptr;
  // the PLAN KERNEL (ltr_dp_plan.hpp): every one-wave class and every packed strip width of the plan in ONE persistent launch,
  // listed under plan_rep (-1: a launch per class / the multi-width launches); its entries longest pairs first
  bool use_plan = false;
  int plan_rep = -1, plan_grid = 0;
  bool plan_small = false;
  std::vector<PlanEntry> plan_entries;
  PlanEntry* d_pl_entries = nullptr;
  unsigned long long* d_wave_clock = nullptr;   // (debug) two wall-clock words per wavefront of the plan kernel
  std::vector<int> order2;              // the launch order with those launches split into their classes again (ltr_plan_set_timing level 2)
  int order_pos2[kNumKernels] = {0};
  int pack_rep[kNumPack] = {0};         // packed class j: the class its launch is listed under (one launch per strip width), -1 = no pairs
  hipEvent_t bin_ev[kNumKernels + 1] = {nullptr};   // bracket every DP launch on the launch stream
  double bin_cells[kNumFast] = {0};
  double x_cells[kNumExact] = {0};      // nominal cells of the pairs pre-seeded into every exact list
  std::vector<int32_t> locus_P, locus_H;   // per locus: pools, haplotypes
  std::vector<int64_t> locus_ll_off;       // per locus: offset of its [P x H] block
  int x_seed[kNumExact] = {0};          // pairs pre-seeded into every exact list (sorted array ranges bin_first[kNumFast + c] ..)
  uint32_t* d_ctrl_init = nullptr;      // image of the control words (queues = 0, redo count = n_generic)
  int32_t* d_redo_init = nullptr;       // indices of the generic pairs: copied over the head of the redo list every execute
  bool sym_at_create = true;            // indel model was symmetric when the pairs were binned
  bool uses_wg = false;                 // some pairs sit in workgroup-kernel classes (symmetric models only)
  bool last_wg_thr = false;             // ... and the last execute scored them with the threshold kernels first
  // compact plans (ltr_plan_create): every device array below is a piece of ONE block; the scores live in pinned host memory
  void* d_block = nullptr;
  double* h_ll = nullptr; size_t h_ll_cap = 0;
  bool ctrl_fresh = false;              // the control words arrived with the upload: the first execute skips their reset
  int timed = 0;                        // the last execute recorded per-launch events (level)
  int timing = 0;                       // record a HIP event around every launch (ltr_plan_set_timing): 1 = as launched, 2 = the multi-width launch class by class
  int32_t* d_redo_list = nullptr;       // kNumExact lists (capacity n_pairs each): pairs the certificate kernels handed to the exact kernels
  uint32_t* d_redo_count = nullptr;     // their lengths (control words)
  int64_t redo_cap = 1;
  int redo_grid = 0;
  int x_grid[kNumExact] = {0};          // launch grid of every exact kernel; 0 = no pair of this plan can land in its list
  uint32_t seed_total = 0;
  bool xlut = false;                    // LUT / penalty-table exact kernels usable (symmetric model, k600 <= kPenKMax)
  int last_launches = 0;
  bool executed = false;
  bool kernel_ms_counted = true;
};

extern "C" {

const char* ltr_version(void) { return LTR_VERSION_STR; }
int ltr_abi_version(void) { return LTR_ABI_VERSION; }
int ltr_num_kernels(void) { return kNumKernels; }
int ltr_kernel_lanes_per_pair(int k) {
  if (k < 0 || k >= kNumKernels) return 64;
  if (k >= kNumFast) return (k - kNumFast == kXWg4) ? 256 : ((k - kNumFast == kXWg8) ? 512 : 64);
  const ClassInfo ci = class_info(k);
  return ci.family == kFamPack ? (1 << ci.lp_shift) : 64 * ci.waves;
}
int ltr_kernel_family(int k) {
  if (k < 0 || k >= kNumKernels) return -1;
  return k >= kNumFast ? 3 : class_info(k).family;
}

void ltr_default_params(ltr_align_params* p) {
  // AlignmentModel(10, -1.0, -0.458675, -1.0, -0.458675, -0.00005800168, -10.448214728, -10.448214728)
  // (reference HapAligner.h:118): double literals narrowed to the float members.
  p->log_ins_to_ins = (float)(-1.0);
  p->log_ins_to_match = (float)(-0.458675);
  p->log_del_to_del = (float)(-1.0);
  p->log_del_to_match = (float)(-0.458675);
  p->log_match_to_match = (float)(-0.00005800168);
  p->log_match_to_ins = (float)(-10.448214728);
  p->log_match_to_del = (float)(-10.448214728);
  p->indel_flank_len = 5;
  p->use_short_path = 0;
}

void ltr_default_stutter_params(ltr_stutter_params* p) {
  // the CLI always installs this fixed model (reference hipstr_main.cpp:140,362-363)
  p->in_geom = 0.95; p->in_up = 0.05; p->in_down = 0.05; p->out_geom = 0.95; p->out_up = 0.01; p->out_down = 0.01;
}

int ltr_ctx_set_pair_packing(ltr_ctx* ctx, int mode) {
  if (!ctx || mode < -1 || mode > 8) return LTR_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->pair_packing = mode;
  return LTR_OK;
}

int ltr_ctx_set_debug(ltr_ctx* ctx, const char* key, double value) {
  if (!ctx || !key) return LTR_ERR_INVALID;
  std::unique_lock<std::mutex> lk(ctx->mu);
  const std::string k(key);
  if (k == "fan_lanes") ctx->dbg.fan_lanes = (int)value;
  else if (k == "fan_pairs") ctx->dbg.fan_pairs = (int64_t)value;
  else if (k == "chunks") ctx->dbg.chunks = (int64_t)value;
  else if (k == "prep_ahead") ctx->dbg.prep_ahead = (int)value;
  else if (k == "chunk_streams") ctx->dbg.chunk_streams = (int)value;
  else if (k == "chunk_growth") { ctx->dbg.chunk_growth = value; ctx->dbg.chunk_growth_set = true; }
  else if (k == "trace") { ctx->dbg.trace = (int)value; g_trace.store((int)value); }
  else if (k == "fold_rounds") ctx->dbg.fold_rounds = (int)value;
  else if (k == "pack_rule") ctx->dbg.pack_rule = (int)value;
  else if (k == "no_multi") ctx->dbg.no_multi = (int)value;
  else if (k == "wg_first_pass") ctx->dbg.wg_first_pass = (int)value;
  else if (k == "compact_plan") ctx->dbg.compact_plan = (int)value;
  else if (k == "short_split") ctx->dbg.short_split = (int)value;
  else if (k == "wgt_keep_waves") ctx->dbg.wgt_keep_waves = (int)value;
  else if (k == "plan_kernel") ctx->dbg.plan_kernel = (int)value;
  else if (k == "plan_share") ctx->dbg.plan_share = (int)value;
  else if (k == "chain") ctx->dbg.chain = (int)value;
  else if (k == "chain_min_w") ctx->dbg.chain_min_w = (int)value;
  else if (k == "chain_max_w") ctx->dbg.chain_max_w = (int)value;
  else if (k == "wave_clock") ctx->dbg.wave_clock = (int)value;
  else if (k == "pageable_staging" || k == "reset") {            // A/B: 1 = the library's own staging arrays in pageable memory again ("reset": pinned, the default)
    const bool pin = (k == "reset") || value == 0.0;
    if (pin != ctx->host_bytes[0].pinned) {
      // (the staging arrays belong to the call that is running: wait for it -- lock order call_mu before mu, as ltr_calc_hap_aln_probs takes them)
      lk.unlock();
      std::lock_guard<std::mutex> call_lk(ctx->call_mu);
      lk.lock();
      for (RawBuf<uint8_t>& b : ctx->host_bytes) { b.release(); b.n = 0; b.pinned = pin; }
      ctx->scratch.sorted.release(); ctx->scratch.sorted.n = 0; ctx->scratch.sorted.pinned = pin;
    }
    if (k == "reset") ctx->dbg = ltr::DebugKnobs();
  }
  else if (k == "short_lane_kernel") ctx->dbg.short_lane_kernel = (int)value;
  else { ltr::set_error(ctx, "ltr_ctx_set_debug: unknown key " + k); return LTR_ERR_INVALID; }
  return LTR_OK;
}

int ltr_ctx_set_stutter_params(ltr_ctx* ctx, const ltr_stutter_params* p) {
  if (!ctx || !p) return LTR_ERR_INVALID;
  // StutterModel constructor asserts (stutter_model.h:37-42)
  if (!(p->in_geom > 0 && p->in_geom < 1 && p->out_geom > 0 && p->out_geom < 1 && p->in_up > 0 && p->in_down > 0 &&
        p->out_up > 0 && p->out_down > 0 && p->in_up + p->in_down + p->out_up + p->out_down < 1)) {
    ltr::set_error(ctx, "invalid stutter model"); return LTR_ERR_INVALID;
  }
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->stutter = *p;
  return LTR_OK;
}

int ltr_ctx_create(int device_ordinal, ltr_ctx** out) {
  if (!out) return LTR_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LTR_ERR_NO_DEVICE;
  if (device_ordinal < 0 || device_ordinal >= ndev) return LTR_ERR_NO_DEVICE;
  ltr_ctx* ctx = new ltr_ctx();
  ctx->device = device_ordinal;
  if (hipSetDevice(device_ordinal) != hipSuccess) { delete ctx; return LTR_ERR_NO_DEVICE; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) { delete ctx; return LTR_ERR_HIP; }
  ctx->arch = prop.gcnArchName;
  ctx->n_cu = prop.multiProcessorCount;
  ctx->clock_mhz = prop.clockRate / 1000;
  for (RawBuf<uint8_t>& hb : ctx->host_bytes) hb.pinned = true;   // staging of ltr_calc_hap_aln_probs' chunks: uploaded by ltr_plan_create
  ctx->scratch.sorted.pinned = true;                              // the sorted pair descriptors: uploaded by ltr_plan_create
  ctx->compact_stage.pinned = true;                               // the one-block image of a compact plan
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return LTR_ERR_HIP; }
  if (hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking) != hipSuccess) { ltr_ctx_destroy(ctx); return LTR_ERR_HIP; }
  for (int k = 0; k < ltr_ctx::kAux; ++k)
    if (hipStreamCreateWithFlags(&ctx->aux[k], hipStreamNonBlocking) != hipSuccess) { ltr_ctx_destroy(ctx); return LTR_ERR_HIP; }
  ltr_default_params(&ctx->params);
  ltr_default_stutter_params(&ctx->stutter);
  fill_model_consts(ctx->params, &ctx->mc);
  *out = ctx;
  return LTR_OK;
}

static void release_plan_buffers(ltr_plan* plan, ltr_ctx* ctx);

// recycled events / pinned blocks of a context (see ltr_ctx: compact plans)
static hipEvent_t ctx_take_event(ltr_ctx* ctx, bool timing) {
  {
    std::lock_guard<std::mutex> lk(ctx->cache_mu);
    std::vector<hipEvent_t>& c = ctx->ev_cache[timing ? 1 : 0];
    if (!c.empty()) { hipEvent_t e = c.back(); c.pop_back(); return e; }
  }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return e;
}
static void ctx_give_event(ltr_ctx* ctx, hipEvent_t e, bool timing) {
  if (!e) return;
  if (!ctx) { (void)hipEventDestroy(e); return; }
  std::lock_guard<std::mutex> lk(ctx->cache_mu);
  std::vector<hipEvent_t>& c = ctx->ev_cache[timing ? 1 : 0];
  if (c.size() < 64) c.push_back(e); else (void)hipEventDestroy(e);
}
static double* ctx_take_pinned(ltr_ctx* ctx, size_t bytes, size_t* cap_out, double** dev_out) {
  std::lock_guard<std::mutex> lk(ctx->cache_mu);
  for (ltr_ctx::PinBlock& b : ctx->pin_blocks) if (!b.busy && b.cap >= bytes) { b.busy = true; *cap_out = b.cap; *dev_out = (double*)b.dev; return (double*)b.p; }
  size_t cap = 4096;
  while (cap < bytes) cap <<= 1;
  void* p = nullptr;
  void* d = nullptr;
  if (hipHostMalloc(&p, cap, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostFree(p); return nullptr; }
  ctx->pin_blocks.push_back({p, d, cap, true});
  *cap_out = cap; *dev_out = (double*)d;
  return (double*)p;
}
static void ctx_give_pinned(ltr_ctx* ctx, void* p) {
  if (!p) return;
  if (!ctx) return;                                              // (the context freed its blocks when it went)
  std::lock_guard<std::mutex> lk(ctx->cache_mu);
  for (ltr_ctx::PinBlock& b : ctx->pin_blocks) if (b.p == p) { b.busy = false; return; }
}

void ltr_ctx_destroy(ltr_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  // plans that outlive their context keep working as handles (destroy is still legal) but lose
  // their device memory here
  for (ltr_plan* plan : ctx->plans) { release_plan_buffers(plan, ctx); plan->ctx = nullptr; plan->last_stream = nullptr; }
  ctx->plans.clear();
  if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
  if (ctx->up_stream) { (void)hipStreamSynchronize(ctx->up_stream); (void)hipStreamDestroy(ctx->up_stream); }
  for (int k = 0; k < ltr_ctx::kAux; ++k) if (ctx->aux[k]) { (void)hipStreamSynchronize(ctx->aux[k]); (void)hipStreamDestroy(ctx->aux[k]); }
  ctx->pool.clear();
  if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  for (ltr_ctx::WgStatSlot& sl : ctx->wg_stat) if (sl.ev) (void)hipEventDestroy(sl.ev);
  for (std::vector<hipEvent_t>& c : ctx->ev_cache) for (hipEvent_t e : c) (void)hipEventDestroy(e);
  for (ltr_ctx::PinBlock& b : ctx->pin_blocks) (void)hipHostFree(b.p);
  if (ctx->compact_ev) (void)hipEventDestroy(ctx->compact_ev);
  if (ctx->wg_stat_pin) (void)hipHostFree(ctx->wg_stat_pin);
  if (ctx->d_big) (void)hipFree(ctx->d_big);
  if (ctx->d_lpc) (void)hipFree(ctx->d_lpc);
  if (ctx->d_colXZ) (void)hipFree(ctx->d_colXZ);
  if (ctx->d_row0XY) (void)hipFree(ctx->d_row0XY);
  if (ctx->d_thr) (void)hipFree(ctx->d_thr);
  delete ctx;
}

int ltr_ctx_set_params(ltr_ctx* ctx, const ltr_align_params* p) {
  if (!ctx) return LTR_ERR_INVALID;
  if (validate_params(p) != LTR_OK) { ltr::set_error(ctx, "invalid alignment parameters (transitions must be < 0, 0 <= indel_flank_len <= 35)"); return LTR_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(ctx->mu);
  ctx->params = *p;
  fill_model_consts(ctx->params, &ctx->mc);
  ctx->wg_thr_first = 0; ++ctx->wg_epoch;                        // (what was learnt about the workgroup classes' first pass belonged to the old model)
  const int64_t want = ctx->table_len;
  ctx->table_len = 0;                       // force rebuild with the new transitions
  (void)hipSetDevice(ctx->device);
  return want > 0 ? build_tables(ctx, want, true) : LTR_OK;      // same length: plans made earlier stay covered
}

const char* ltr_last_error(const ltr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int ltr_ctx_device_info(const ltr_ctx* ctx, char* arch, int arch_len, int* n_cu, int* clock_mhz) {
  if (!ctx) return LTR_ERR_INVALID;
  if (arch && arch_len > 0) { std::snprintf(arch, (size_t)arch_len, "%s", ctx->arch.c_str()); }
  if (n_cu) *n_cu = ctx->n_cu;
  if (clock_mhz) *clock_mhz = ctx->clock_mhz;
  return LTR_OK;
}

// Give a plan's device buffers back (to the context's pool, or to the runtime when the context is gone).
static void release_plan_buffers(ltr_plan* plan, ltr_ctx* ctx) {
  for (hipStream_t st : plan->streams) (void)hipStreamSynchronize(st);       // nothing in flight may still use the buffers
  if (plan->ev_up) (void)hipEventSynchronize(plan->ev_up);                   // (... nor the input preparation of a plan that never ran)
  plan->streams.clear();
  void** bufs[] = {(void**)&plan->d_reads, (void**)&plan->d_haps, (void**)&plan->d_hap_codes, (void**)&plan->d_pairs,
                   (void**)&plan->d_ll, (void**)&plan->d_queue, (void**)&plan->d_scratch, (void**)&plan->d_redo_list,
                   (void**)&plan->d_ctrl_init, (void**)&plan->d_redo_init, (void**)&plan->d_pk_tabs, (void**)&plan->d_pl_entries, (void**)&plan->d_wave_clock};
  if (plan->d_block || plan->h_ll) {
    // a compact plan: one block holds every array but the scratch strips; the scores are a pinned block of the context
    void* const scr = plan->d_scratch;
    for (void** p : bufs) *p = nullptr;
    plan->d_scratch = (double*)scr;
    if (ctx) ctx->pool.release(plan->d_block); else if (plan->d_block) (void)hipFree(plan->d_block);
    plan->d_block = nullptr;
    ctx_give_pinned(ctx, plan->h_ll);
    plan->h_ll = nullptr;
  }
  for (void** p : bufs) { if (ctx) ctx->pool.release(*p); else if (*p) (void)hipFree(*p); *p = nullptr; }
  plan->d_redo_count = nullptr;
}

static void destroy_plan(ltr_plan* plan, const bool ctx_locked) {
  ltr_ctx* ctx = plan->ctx;                                    // nullptr: the context was destroyed first
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    if (ctx_locked) ctx->plans.erase(plan);
    else { std::lock_guard<std::mutex> lk(ctx->mu); ctx->plans.erase(plan); }
  }
  release_plan_buffers(plan, ctx);
  ctx_give_event(ctx, plan->ev_up, false);
  if (plan->ev_fast) (void)hipEventDestroy(plan->ev_fast);
  if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
  for (int k = 0; k < 3; ++k) if (plan->ev_join[k]) (void)hipEventDestroy(plan->ev_join[k]);
  for (int c = 0; c <= kNumExact; ++c) if (plan->ev_x[c]) (void)hipEventDestroy(plan->ev_x[c]);
  for (int c = 0; c < kNumExact; ++c) for (int k = 0; k < 4; ++k) if (plan->ev_close[c][k]) (void)hipEventDestroy(plan->ev_close[c][k]);
  ctx_give_event(ctx, plan->ev0, true);
  ctx_give_event(ctx, plan->ev1, true);
  for (int k = 0; k <= kNumKernels; ++k) if (plan->bin_ev[k]) (void)hipEventDestroy(plan->bin_ev[k]);
  delete plan;
}

void ltr_plan_destroy(ltr_plan* plan) {
  if (plan) destroy_plan(plan, false);
}

// ---- units of ltr_plan_create (host side; the class rule and the sort live in ltr_plan.cpp) ----------------------------

// Nominal cells and longest read of every class: blocks of the sorted pairs on the host cores (a block spans a few
// classes; a class's pairs are one contiguous range), partial sums merged in block order.
static void plan_class_stats(ltr_plan* plan, const RawBuf<PairDesc>& sorted, const RawBuf<int32_t>& order, const RawBuf<int16_t>& key) {
  {
    const size_t np = sorted.size();
    const int64_t n_blk = (int64_t)((np + kPlanBlock - 1) / kPlanBlock);
    struct Part { int k0 = 0, k1 = -1; std::vector<double> cl; std::vector<int32_t> cm; };
    std::vector<Part> parts((size_t)n_blk);
    auto class_of = [&](size_t i) { int k = 0; while (plan->bin_first[k + 1] <= (int)i) ++k; return k; };
    ltr::parallel_for(n_blk, np < 20000 ? n_blk + 1 : 1, [&](int64_t c) {                         // (a one-locus plan: not worth waking the worker pool)
      const size_t i0 = (size_t)c * kPlanBlock, i1 = std::min(np, i0 + kPlanBlock);
      Part& P = parts[(size_t)c];
      int k = class_of(i0);
      P.k0 = k; P.k1 = k;
      double cl = 0.0; int32_t cm = 0;
      for (size_t i = i0; i < i1; ++i) {
        while (plan->bin_first[k + 1] <= (int)i) { P.cl.push_back(cl); P.cm.push_back(cm); cl = 0.0; cm = 0; ++k; P.k1 = k; }
        if (key[(size_t)order[i]] > 0) cl += (double)sorted[i].n * (double)sorted[i].m;
        cm = std::max(cm, sorted[i].m - 1);
      }
      P.cl.push_back(cl); P.cm.push_back(cm);
    }, 1);
    for (int k = 0; k < kNumFast; ++k) { plan->bin_cells[k] = 0.0; plan->cls_cmax[k] = 0; }
    for (int c = 0; c < kNumExact; ++c) plan->x_cells[c] = 0.0;
    for (const Part& P : parts)
      for (int k = P.k0; k <= P.k1 && np > 0; ++k) {
        const double cl = P.cl[(size_t)(k - P.k0)];
        if (k < kNumFast) { plan->bin_cells[k] += cl; plan->cls_cmax[k] = std::max(plan->cls_cmax[k], P.cm[(size_t)(k - P.k0)]); }
        else if (k < kNumKernels) plan->x_cells[k - kNumFast] += cl;
      }
  }
}

// Launch order of the certificate classes and which of them share a launch (packed: one launch per strip width; automatic
// mode, large plans: the multi-width launches).  Pure function of plan->bin_first / cls_cmax.
static void plan_launch_order(ltr_plan* plan, const bool use_multi) {
  // launch order of the certificate classes: longest reads first (the classes that can feed the exact lists of long reads
  // are through early, and those lists' launches -- a handful of pairs, each as long as its longest pair -- run beside
  // the remaining certificate launches instead of behind the last one)
  // (the packed classes of one strip width -- 32, 16, 8, 4, 2 lanes per pair -- are ONE launch, ltr_dp_pack.hpp: it is listed
  // under the first of them that has pairs, its representative, and is as long as the longest read of any of them)
  for (int k = kPackFirst; k < kWg4First; ++k) plan->pack_rep[k - kPackFirst] = -1;
  for (int w = 1; w <= kPackWMax; ++w) {
    int rep = -1, cm = 0;
    for (int sft = kPackMaxShift; sft >= kPackMinShift; --sft) {
      const int k = ltrp::pack_class(sft, w);
      if (plan->bin_first[k + 1] <= plan->bin_first[k]) continue;
      if (rep < 0) rep = k;
      cm = std::max(cm, plan->cls_cmax[k]);
    }
    if (rep < 0) continue;
    for (int sft = kPackMaxShift; sft >= kPackMinShift; --sft) {
      const int k = ltrp::pack_class(sft, w);
      if (plan->bin_first[k + 1] > plan->bin_first[k]) plan->pack_rep[k - kPackFirst] = rep;
    }
    plan->cls_cmax[rep] = cm;
  }
  // (... and in automatic mode the one-wave classes of strip widths kMultiMinW .. kWMax are ONE launch too,
  // ltr_dp_multi_kernel: listed under the widest of them that has pairs)
  // (... or, under the plan kernel, EVERY one-wave class and every packed width: one launch, listed under plan_rep)
  const bool use_plan = plan->use_plan;
  if (use_multi || use_plan) {
    for (int k = kNumBins - 1; k >= (use_plan ? 0 : kMultiMinW - 1); --k) if (plan->bin_first[k + 1] > plan->bin_first[k]) plan->multi_classes.push_back(k);
    if (plan->multi_classes.size() >= (use_plan ? 1u : 2u)) plan->multi_rep = plan->multi_classes[0]; else plan->multi_classes.clear();
  }
  int32_t multi_cmax = 0, pmulti_cmax = 0;
  for (int k : plan->multi_classes) multi_cmax = std::max(multi_cmax, plan->cls_cmax[k]);
  if (use_multi || use_plan) {
    for (int w = kPackWMax; w >= (use_plan ? 1 : kPackMultiMinW); --w) {
      int r2 = -1;
      for (int sft = kPackMaxShift; sft >= kPackMinShift && r2 < 0; --sft) r2 = plan->pack_rep[ltrp::pack_class(sft, w) - kPackFirst];
      if (r2 >= 0) plan->pmulti_reps.push_back(r2);
    }
    if (plan->pmulti_reps.size() >= (use_plan ? 1u : 2u)) plan->pmulti_rep = plan->pmulti_reps[0]; else plan->pmulti_reps.clear();
  }
  for (int k : plan->pmulti_reps) pmulti_cmax = std::max(pmulti_cmax, plan->cls_cmax[k]);
  if (use_plan) plan->plan_rep = plan->multi_rep >= 0 ? plan->multi_rep : plan->pmulti_rep;
  auto in_multi = [&](int k) {
    if (use_plan) return k < kWg4First;
    if (plan->multi_rep >= 0 && k < kNumBins && k >= kMultiMinW - 1) return true;
    return plan->pmulti_rep >= 0 && k >= kPackFirst && k < kWg4First && class_info(k).W >= kPackMultiMinW;
  };
  for (int k = kNumFast - 1; k >= 0; --k) {
    if (plan->bin_first[k + 1] <= plan->bin_first[k]) continue;
    if (k >= kPackFirst && k < kWg4First && plan->pack_rep[k - kPackFirst] != k) continue;
    plan->order2.push_back(k);
    if (use_plan ? (in_multi(k) && k != plan->plan_rep) : (in_multi(k) && k != plan->multi_rep && k != plan->pmulti_rep)) continue;
    plan->order.push_back(k);
  }
  std::stable_sort(plan->order2.begin(), plan->order2.end(), [&](int x, int y) { return plan->cls_cmax[x] > plan->cls_cmax[y]; });
  for (int k = 0; k < kNumKernels; ++k) plan->order_pos2[k] = -1;
  for (size_t i = 0; i < plan->order2.size(); ++i) plan->order_pos2[plan->order2[i]] = (int)i;
  for (int c = 0; c < kNumExact; ++c) plan->order_pos2[kNumFast + c] = (int)plan->order2.size() + c;
  if (plan->multi_rep >= 0) plan->cls_cmax[plan->multi_rep] = multi_cmax;       // (>= its own: the exact lists close no earlier for it)
  if (plan->pmulti_rep >= 0) plan->cls_cmax[plan->pmulti_rep] = pmulti_cmax;
  if (plan->plan_rep >= 0) plan->cls_cmax[plan->plan_rep] = std::max(multi_cmax, pmulti_cmax);
  std::stable_sort(plan->order.begin(), plan->order.end(), [&](int x, int y) { return plan->cls_cmax[x] > plan->cls_cmax[y]; });
  for (int k = 0; k < kNumKernels; ++k) plan->order_pos[k] = -1;
  for (size_t i = 0; i < plan->order.size(); ++i) plan->order_pos[plan->order[i]] = (int)i;
  for (int c = 0; c < kNumExact; ++c) plan->order_pos[kNumFast + c] = (int)plan->order.size() + c;

}

#define GRID_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)
// Resident workgroups of every launch class (occupancy x CUs), asked from the runtime once per context.
static hipError_t ctx_query_grids(ltr_ctx* ctx) {
  for (int k = 0; k < kNumFast; ++k) {
    const ClassInfo ci = class_info(k);
    int per_cu = 0;
    GRID_TRY(ci.family == kFamOne ? ltrk::occ_onewave(ci.W, &per_cu) : (ci.family == kFamPack ? ltrk::occ_pack(ci.W, &per_cu) : ltrk::occ_wg(ci.waves, ci.W, &per_cu)));
    ctx->full_grid[k] = std::max(per_cu, 1) * ctx->n_cu;
  }
  {
    int per_cu = 0;
    GRID_TRY(ltrk::occ_multi(&per_cu));
    ctx->full_multi_grid = std::max(per_cu, 1) * ctx->n_cu;
    per_cu = 0;
    GRID_TRY(ltrk::occ_pack_multi(&per_cu));
    ctx->full_pmulti_grid = std::max(per_cu, 1) * ctx->n_cu;
    per_cu = 0;
    GRID_TRY(ltrk::occ_plan(true, &per_cu));
    int per_cu_general = 0;
    GRID_TRY(ltrk::occ_plan(false, &per_cu_general));           // (same launch bounds and LDS: the smaller of the two sizes the grid for both)
    ctx->full_plan_grid = std::max(std::min(per_cu, per_cu_general), 1) * ctx->n_cu;
  }
  for (int c = 0; c <= kNumExact; ++c) {                     // (kNumExact: the W = 20 launch that shares the four-wave list)
    int per_cu = 0;
    GRID_TRY(ltrk::occ_exact(c, &per_cu));
    if (c < kNumExact) ctx->full_x_grid[c] = std::max(per_cu, 1) * ctx->n_cu;
    else ctx->full_x_wide_grid = std::max(per_cu, 1) * ctx->n_cu;
  }
  for (int nw = 0; nw < 2; ++nw)
    for (int w = (nw ? kWg8MinW : 6); w <= kWgWMax; w += 2) {
      int per_cu = 0;
      GRID_TRY(ltrk::occ_wgt(nw ? 8 : 4, w, &per_cu));
      ctx->full_wgt_grid[nw][w] = std::max(per_cu, 1) * ctx->n_cu;
    }
  ctx->full_redo_grid = ctx->full_x_grid[kXGeneric];
  ctx->have_grids = true;
  return hipSuccess;
}
#undef GRID_TRY

// Persistent grid of every launch of the plan, "small" flags (a launch that cannot fill the GPU's wave slots once), the range
// tables of the multi-width packed launch (uploaded by the caller).  counts: pairs per class after folding; xcand: pairs that
// could end up in each exact list.
static void plan_size_grids(ltr_ctx* ctx, ltr_plan* plan, const int* counts, const int64_t* xcand_in, std::vector<PackTable>* pack_tabs) {
  int64_t xcand[kNumExact];
  for (int c = 0; c < kNumExact; ++c) xcand[c] = xcand_in[c];
  const int* g = ctx->full_grid;
  plan->redo_grid = ctx->full_redo_grid;
  for (int k = 0; k < kNumFast; ++k) {
    const ClassInfo ci = class_info(k);
    if (ci.family == kFamWg) {                                                // one pair per workgroup, no scratch strips
      plan->bin_grid[k] = std::min(g[k], std::max(counts[k], 1));
      plan->bin_small[k] = counts[k] < g[k];
      continue;
    }
    int waves = counts[k];
    if (k == plan->multi_rep) {
      // the multi-width launch takes every class of its group; its grid is kept next to the class's own (level-2 timing launches the classes one by one)
      int all = 0;
      for (int k2 : plan->multi_classes) all += counts[k2];
      plan->multi_grid = std::min(ctx->full_multi_grid, std::max((all + kBlockWaves - 1) / kBlockWaves, 1));
      plan->multi_small = (all + kBlockWaves - 1) / kBlockWaves < ctx->full_multi_grid;
      plan->max_grid = std::max(plan->max_grid, plan->multi_grid);
    }
    if (ci.family == kFamPack) {
      // a packed wave takes 64 / LP pairs; the launch (listed under its representative) takes every lanes-per-pair block of the width
      waves = 0;
      if (plan->pack_rep[k - kPackFirst] == k)
        for (int sft = kPackMinShift; sft <= kPackMaxShift; ++sft) { const int per = 64 >> sft; waves += (counts[ltrp::pack_class(sft, ci.W)] + per - 1) / per; }
    }
    plan->bin_grid[k] = std::min(g[k], std::max((waves + kBlockWaves - 1) / kBlockWaves, 1));
    plan->bin_small[k] = (waves + kBlockWaves - 1) / kBlockWaves < g[k];
    if (ci.family == kFamOne) plan->max_grid = std::max(plan->max_grid, plan->bin_grid[k]);
  }
  if (plan->pmulti_rep >= 0) {
    // the multi-width packed launch: one table per strip width (its ranges as the single-width launch would get them), widest first
    std::vector<PackTable>& tabs = *pack_tabs;
    tabs.clear();
    int groups_all = 0;
    for (int rep : plan->pmulti_reps) {
      PackTable T;
      std::memset(&T, 0, sizeof(T));
      T.W = class_info(rep).W; T.queue_class = rep;
      int nr = 0, groups = 0;
      for (int sft = kPackMaxShift; sft >= kPackMinShift; --sft) {
        const int k2 = ltrp::pack_class(sft, T.W);
        const int c2 = plan->bin_first[k2 + 1] - plan->bin_first[k2];
        if (c2 <= 0) continue;
        const int per = 64 >> sft;
        groups += (c2 + per - 1) / per;
        T.shift[nr] = sft; T.first[nr] = plan->bin_first[k2]; T.end[nr] = plan->bin_first[k2 + 1]; T.grp_end[nr] = groups;
        ++nr;
      }
      for (; nr < 5; ++nr) { T.shift[nr] = kPackMaxShift; T.first[nr] = 0; T.end[nr] = 0; T.grp_end[nr] = groups; }
      groups_all += groups;
      tabs.push_back(T);
    }
    plan->pmulti_grid = std::min(ctx->full_pmulti_grid, std::max((groups_all + kBlockWaves - 1) / kBlockWaves, 1));
    plan->pmulti_small = (groups_all + kBlockWaves - 1) / kBlockWaves < ctx->full_pmulti_grid;
  }
  if (plan->use_plan && plan->plan_rep >= 0) {
    // The plan kernel's entries: every one-wave class, every packed width (table t of pack_tabs), the entry with the longest
    // pairs first (modelled steps x strip cost of its longest read); the launch's wavefronts start spread over the entries in
    // proportion to their modelled work (cells x (1 + per-step overhead / W)) and walk the table from the top afterwards.
    struct Ent { PlanEntry e; double longest, work; };
    std::vector<Ent> ents;
    int waves_all = 0;
    for (int k : plan->multi_classes) {
      const int w = class_info(k).W, np = plan->bin_first[k + 1] - plan->bin_first[k];
      PlanEntry e; std::memset(&e, 0, sizeof(e));
      e.kind = 0; e.W = w; e.first = plan->bin_first[k]; e.n_pairs = np; e.queue_class = k; e.tab = 0; e.limit = np;
      // the chained walk (ltr_dp_chain.hpp: no fill and drain of the skew between the pairs of a class; on request) for the strip
      // widths whose reads fill the wave's scratch strip, where the next pair's first row is parked
      {
        const int64_t need = 2 * (int64_t)w * 64 + ((w + 3) / 4) * 32 + 2;
        const int64_t have = 6 * (int64_t)(((plan->max_len + 2 + 15) / 16) * 16);
        const int lo = ctx->dbg.chain_min_w > 0 ? ctx->dbg.chain_min_w : kMultiMinW, hi = ctx->dbg.chain_max_w > 0 ? ctx->dbg.chain_max_w : kWMax;
        if (ctx->dbg.chain > 0 && plan->sym_at_create && w >= std::max(lo, (int)kMultiMinW) && w <= hi && need <= have) e.kind = 3;   // (off by default: measured slower, ltr_dp_chain.hpp)
      }
      const int ncb = (plan->cls_cmax[k] + 64 * w - 1) / (64 * w);
      ents.push_back({e, (double)std::max(ncb, 1) * (plan->cls_cmax[k] + 64.0) * (w + 1.5), plan->bin_cells[k] * (1.0 + 1.5 / w)});
      waves_all += np;
    }
    for (size_t t = 0; t < pack_tabs->size(); ++t) {
      const PackTable& T = (*pack_tabs)[t];
      PlanEntry e; std::memset(&e, 0, sizeof(e));
      e.kind = 1; e.W = T.W; e.queue_class = T.queue_class; e.tab = (int32_t)t; e.limit = T.grp_end[4];
      double cells = 0.0; int cmax = 0, lp = 2;
      for (int sft = kPackMinShift; sft <= kPackMaxShift; ++sft) {
        const int k2 = ltrp::pack_class(sft, T.W);
        if (plan->bin_first[k2 + 1] > plan->bin_first[k2]) { cells += plan->bin_cells[k2]; lp = 1 << sft; }
      }
      cmax = plan->cls_cmax[T.queue_class];
      ents.push_back({e, (cmax + (double)lp) * (T.W + 1.5), cells * (1.0 + 1.5 / T.W)});
      waves_all += e.limit;
    }
    {
      // The pairs that START OUT in an exact list -- bytes outside ACGT, length differences no certificate can hold
      // (Rules::risky_dd_pos / _neg) -- are scored by the plan kernel itself, FIRST: they are the longest jobs of the plan (an exact body of
      // 1 - 3 ms per pair on one wavefront).  (Measured on MI355X, 1250 loci of config 3: as launches of their own beside the plan
      // kernel they found no free wave slot before its workgroups left and the pass ended 3 ms after the plan kernel, 34.0 ms; as
      // its last work they were its tail, 33.8 ms; first, 30.4 ms.)  The exact launches of such a plan only take what the
      // workgroup classes queue on the device.
      for (int c = 0; c < kNumExact; ++c) {
        const int np = plan->bin_first[kNumFast + c + 1] - plan->bin_first[kNumFast + c];
        plan->x_seed[c] = 0;
        if (np <= 0) continue;
        PlanEntry e; std::memset(&e, 0, sizeof(e));
        e.kind = 2; e.W = (c == kXGeneric) ? 0 : 1; e.first = plan->bin_first[kNumFast + c]; e.n_pairs = np; e.queue_class = ltrp::kStartQueueSlot + c; e.limit = np;   // (a counter of its own: list c's exact launch may run as well, fed by the workgroup classes)
        ents.push_back({e, 1e30 - c, plan->x_cells[c] * 1.4});
        waves_all += np;
      }
    }
    std::stable_sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) { return x.longest > y.longest; });
    plan->plan_grid = std::min(ctx->full_plan_grid, std::max((waves_all + kBlockWaves - 1) / kBlockWaves, 1));
    plan->plan_small = (waves_all + kBlockWaves - 1) / kBlockWaves < ctx->full_plan_grid;
    plan->max_grid = std::max(plan->max_grid, plan->plan_grid);
    double total = 0.0, run = 0.0;
    for (const Ent& x : ents) total += x.work;
    const double n_waves = (double)plan->plan_grid * kBlockWaves;
    plan->plan_entries.clear();
    for (Ent& x : ents) {
      x.e.first_wave = total > 0.0 ? (int32_t)std::min(n_waves, std::floor(n_waves * run / total)) : 0;
      run += x.work;
      plan->plan_entries.push_back(x.e);
    }
    if (!plan->plan_entries.empty()) plan->plan_entries[0].first_wave = 0;
    // (Measured on MI355X, plan kernel with the shares against every wavefront starting at the top of the table
    // (ltr_ctx_set_debug "plan_share" = 1): shards of config 3 of 625 / 1250 / 2500 / 5000 loci 15.5 - 15.6 against 15.8 - 16.0 ms,
    // 30.2 against 30.4, 59.4 both, 125.8 against 125.1 - 125.5; shards of the catalogue of 6250 / 12 500 loci 4.57 against 4.99,
    // 8.09 against 8.54 -- 3072 wavefronts racing down a table of 30 - 40 short entries pop every counter 3072 times.  While
    // failed certificates still ended the launch (first version) the shares looked worse: 32.28 against 31.68 ms at 1250 loci.)
    if (ctx->dbg.plan_share == 1) for (size_t i = 1; i < plan->plan_entries.size(); ++i) plan->plan_entries[i].first_wave = 0x7fffffff;
  }
  // exact kernels: launched only when some pair of the plan can land in their list
  for (int c = 0; c < kNumExact; ++c) {
    if (xcand[c] <= 0) { plan->x_grid[c] = 0; continue; }
    const bool wgx = (c == kXWg4 || c == kXWg8);
    const int64_t wgs = wgx ? xcand[c] : (xcand[c] + kBlockWaves - 1) / kBlockWaves;
    plan->x_grid[c] = (int)std::min<int64_t>(ctx->full_x_grid[c], std::max<int64_t>(wgs, 1));
    if (!wgx) plan->max_grid = std::max(plan->max_grid, plan->x_grid[c]);      // (the one-wave kernels park column blocks in scratch strips)
  }
  plan->redo_grid = plan->x_grid[kXGeneric];
  plan->max_grid = std::max(plan->max_grid, 1);
  plan->max_grid_wide = (int)std::max<int64_t>(1, std::min<int64_t>((xcand[kXWg4] + kBlockWaves - 1) / kBlockWaves, 1 << 20));
}

// The body of ltr_plan_create, under ctx->mu.  `plan` is the caller's pointer: the plan being built is owned through it until it is
// handed out (*out), so that an exception on the way (bad_alloc in a host array) can be answered by the caller's handler.
static int plan_create_locked(ltr_ctx* ctx, const ltr_locus_batch* b, ltr_plan** out, ltr_plan*& plan) {
  if (b->n_loci < 0 || b->n_reads < 0 || b->n_haps < 0) { ltr::set_error(ctx, "negative counts"); return LTR_ERR_INVALID; }
  if (b->n_loci > 0 && (!b->locus_read_off || !b->locus_hap_off || !b->read_off || !b->hap_off)) { ltr::set_error(ctx, "null offset array"); return LTR_ERR_INVALID; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  plan = new ltr_plan();
  plan->ctx = ctx;
  plan->n_reads = b->n_reads;
  const int F = ctx->params.indel_flank_len;
  // ---- what the batch as a whole decides: packing, workgroup kernels, exact kernel flavour (ltr_plan.cpp) ----
  int64_t pairs_upper = 0, n_long_pairs = 0;
  int64_t by_bucket[ltrp::kLengthBuckets] = {0};                          // pairs by read length (quarter octaves)
  {
    // (blocks of loci on the host cores, partial sums merged under a lock: serial, this loop and the two below were 1.1 ms of the
    // 3.2 ms a 10 000-locus chunk of ltr_calc_hap_aln_probs spends in here)
    std::mutex acc_mu;
    std::atomic<int> bad(0);
    const int64_t n_blk0 = (b->n_loci + 255) / 256;
    ltr::parallel_for(n_blk0, b->n_loci < 2048 ? n_blk0 + 1 : 1, [&](int64_t c) {
      int64_t pu = 0, nlp = 0, bb[ltrp::kLengthBuckets] = {0};
      for (int64_t l = c * 256; l < std::min<int64_t>(b->n_loci, (c + 1) * 256); ++l) {
        const int64_t r0 = b->locus_read_off[l], r1 = b->locus_read_off[l + 1], h0 = b->locus_hap_off[l], h1 = b->locus_hap_off[l + 1];
        if (r0 < 0 || r1 < r0 || r1 > b->n_reads || h0 < 0 || h1 < h0 || h1 > b->n_haps) { bad.store(1, std::memory_order_relaxed); return; }
        pu += (r1 - r0) * (h1 - h0);
        int64_t nl = 0;
        for (int64_t r = r0; r < r1; ++r) {
          const int64_t C = b->read_off[r + 1] - b->read_off[r] - 1;
          nl += (C > 64 * kWMax);
          bb[ltrp::length_bucket((int)std::max<int64_t>(std::min<int64_t>(C, 1 << 24), 0))] += h1 - h0;
        }
        nlp += nl * (h1 - h0);
      }
      std::lock_guard<std::mutex> lk2(acc_mu);
      pairs_upper += pu; n_long_pairs += nlp;
      for (int q = 0; q < ltrp::kLengthBuckets; ++q) by_bucket[q] += bb[q];
    }, 1);
    if (bad.load()) { ltr::set_error(ctx, "locus offsets out of range"); delete plan; plan = nullptr; return LTR_ERR_INVALID; }
  }
  LTR_DBG("plan: lengths counted");
  const ltrp::Rules rules = ltrp::make_rules(ctx->mc, F, ctx->pair_packing, ctx->n_cu, pairs_upper, n_long_pairs, by_bucket, ctx->dbg.pack_rule, ctx->dbg.plan_kernel);
  plan->sym_at_create = rules.sym_model;
  plan->xlut = rules.xlut;
  // The plan kernel (ltr_dp_plan.hpp): every plan of the automatic mode, whatever its size and whatever the indel model (round 6:
  // the general-model instance; ltrp::make_rules; ltr_ctx_set_debug "plan_kernel": 1 = never).  Measured on MI355X, cost shards of
  // config 3 (tests/manual/gpu_plan_ab.py), plan kernel against round 4's launches: profiles/r05/plan_kernel/.
  plan->use_plan = rules.plan_kernel;
  int64_t xcand[kNumExact] = {0};               // pairs that could end up in each exact kernel's list
  int64_t xstart[kNumExact] = {0};              // (plan kernel: the pairs that start out in a list, counted apart -- the plan kernel scores them itself)

  // ---- validate + enumerate pairs --------------------------------------------------------
  RawBuf<PairDesc>& pairs = ctx->scratch.pairs;
  RawBuf<int16_t>& key = ctx->scratch.key;      // launch-order key of every pair
  RawBuf<int16_t>& bin = ctx->scratch.bin;      // launch class of every pair
  int64_t ll_off = 0;
  int32_t max_len = 1;
  plan->seed.assign((size_t)b->n_reads, -1);
  double in_bytes = 0.0, cells = 0.0;
  // which sequences are pure upper-case ACGT (the LUT emission of the fast kernels needs that)
  // (16 bytes per step with SSE2 -- part of x86-64 --: the byte loop, which hipcc's host pass does not vectorise, was 1.4 - 1.5 ms
  // of the 5.3 ms of plan creation per 10 000 catalogue loci; 4.4 x faster per thread here)
  auto acgt_only = [](const uint8_t* p, int64_t len) {
    int64_t k = 0;
#if defined(__SSE2__)
    const __m128i cA = _mm_set1_epi8('A'), cC = _mm_set1_epi8('C'), cG = _mm_set1_epi8('G'), cT = _mm_set1_epi8('T');
    __m128i all = _mm_set1_epi8((char)0xff);
    for (; k + 16 <= len; k += 16) {
      const __m128i v = _mm_loadu_si128((const __m128i*)(p + k));
      all = _mm_and_si128(all, _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(v, cA), _mm_cmpeq_epi8(v, cC)), _mm_or_si128(_mm_cmpeq_epi8(v, cG), _mm_cmpeq_epi8(v, cT))));
    }
    if (_mm_movemask_epi8(all) != 0xffff) return false;
#endif
    unsigned ok = 1;
    for (; k < len; ++k) { const uint8_t c = p[k]; ok &= (unsigned)((c == 'A') | (c == 'C') | (c == 'G') | (c == 'T')); }
    return ok != 0;
  };
  RawBuf<uint8_t>& read_acgt = ctx->scratch.read_acgt; RawBuf<uint8_t>& hap_acgt = ctx->scratch.hap_acgt;
  read_acgt.resize((size_t)b->n_reads); hap_acgt.resize((size_t)b->n_haps);
  {
    std::atomic<int> bad(0);                                              // 1: reads, 2: haplotypes
    ltr::parallel_for((b->n_reads + 4095) / 4096, 1, [&](int64_t c) {
      for (int64_t r = c * 4096; r < std::min<int64_t>(b->n_reads, (c + 1) * 4096); ++r) if (b->read_off[r + 1] < b->read_off[r]) { bad.store(1, std::memory_order_relaxed); return; }
    }, 1);
    ltr::parallel_for((b->n_haps + 4095) / 4096, 1, [&](int64_t c) {
      for (int64_t h = c * 4096; h < std::min<int64_t>(b->n_haps, (c + 1) * 4096); ++h) if (b->hap_off[h + 1] < b->hap_off[h]) { int e = 0; bad.compare_exchange_strong(e, 2); return; }
    }, 1);
    if (bad.load() == 1) { ltr::set_error(ctx, "read offsets not ascending"); delete plan; plan = nullptr; return LTR_ERR_INVALID; }
    if (bad.load() == 2) { ltr::set_error(ctx, "haplotype offsets not ascending"); delete plan; plan = nullptr; return LTR_ERR_INVALID; }
  }
  LTR_DBG("plan: offsets checked");
  // (the byte scans run on the host cores: ~180 MB per 10 k loci)
  ltr::parallel_for(b->n_reads, 512, [&](int64_t r) {
    read_acgt[(size_t)r] = acgt_only(b->read_bytes + b->read_off[r], b->read_off[r + 1] - b->read_off[r]); });
  ltr::parallel_for(b->n_haps, 512, [&](int64_t h) {
    hap_acgt[(size_t)h] = acgt_only(b->hap_bytes + b->hap_off[h], b->hap_off[h + 1] - b->hap_off[h]); });
  LTR_DBG("plan: bytes scanned");
  // ---- pass 1 (serial, cheap): per-locus output offsets and pair counts -> where every locus' pairs go ----
  std::vector<int64_t> pair_base((size_t)b->n_loci + 1, 0);
  plan->locus_P.reserve((size_t)b->n_loci); plan->locus_H.reserve((size_t)b->n_loci); plan->locus_ll_off.reserve((size_t)b->n_loci);
  for (int64_t l = 0; l < b->n_loci; ++l) {
    const int64_t r0 = b->locus_read_off[l], r1 = b->locus_read_off[l + 1];
    const int64_t h0 = b->locus_hap_off[l], h1 = b->locus_hap_off[l + 1];
    const int64_t H = h1 - h0;
    plan->locus_P.push_back((int32_t)(r1 - r0)); plan->locus_H.push_back((int32_t)H); plan->locus_ll_off.push_back(ll_off);
    in_bytes += (double)(b->read_off[r1] - b->read_off[r0]) + (double)(b->hap_off[h1] - b->hap_off[h0]) + 8.0 * (double)(r1 - r0) * (double)H;
    int64_t nr = r1 - r0, nh = H;
    if (b->realign_read) { nr = 0; for (int64_t r = r0; r < r1; ++r) nr += b->realign_read[r] ? 1 : 0; }
    if (b->realign_hap) { nh = 0; for (int64_t h = h0; h < h1; ++h) nh += b->realign_hap[h] ? 1 : 0; }
    pair_base[(size_t)l + 1] = pair_base[(size_t)l] + nr * nh;
    ll_off += (r1 - r0) * H;
  }
  const int64_t n_pairs_total = pair_base[(size_t)b->n_loci];
  LTR_DBG("plan: %ld pairs counted", (long)n_pairs_total);
  if (n_pairs_total > 0x7fffffff) { ltr::set_error(ctx, "too many pairs in one batch"); delete plan; plan = nullptr; return LTR_ERR_INVALID; }
  // (the plan kernel's per-wave notes carry flags in bits 30 and 31 of a pair index -- kNotePlain, "generic body": a plan of 2^30
  // pairs and more, > 40 GB of descriptors, keeps a launch per class)
  if (n_pairs_total >= ((int64_t)1 << 30)) plan->use_plan = false;
  pairs.resize((size_t)n_pairs_total); key.resize((size_t)n_pairs_total); bin.resize((size_t)n_pairs_total);
  LTR_DBG("plan: arrays sized");
  // ---- pass 2 (all host cores): one descriptor, launch class and launch-order key per pair (ltrp::classify_pair) ----
  struct ClassMemo { uint64_t tag = ~0ull, plan_id = 0; ltrp::PairClass pc; };
  constexpr int kClassMemoBits = 12;
  static std::atomic<uint64_t> plan_counter{0};
  const uint64_t plan_id = plan_counter.fetch_add(1) + 1;            // (the rules differ from plan to plan)
  struct LocusAcc { double cells = 0.0; int32_t max_len = 1; int64_t xcand[kNumExact] = {0}, xstart[kNumExact] = {0}; uint8_t uses_wg = 0; int8_t err = 0; };
  std::vector<LocusAcc> acc((size_t)b->n_loci);
  ltr::parallel_for(b->n_loci, 256, [&](int64_t l) {
    const int64_t r0 = b->locus_read_off[l], r1 = b->locus_read_off[l + 1];
    const int64_t h0 = b->locus_hap_off[l], h1 = b->locus_hap_off[l + 1];
    const int64_t H = h1 - h0, ll_base = plan->locus_ll_off[(size_t)l];
    LocusAcc& A2 = acc[(size_t)l];
    int64_t at = pair_base[(size_t)l];
    static thread_local std::vector<ClassMemo> memo_store;
    if (memo_store.empty()) memo_store.resize((size_t)1 << kClassMemoBits);
    ClassMemo* memo = memo_store.data();
    for (int64_t r = r0; r < r1; ++r) {
      if (b->realign_read && !b->realign_read[r]) continue;
      const int64_t m = b->read_off[r + 1] - b->read_off[r];
      if (m <= 0 || m > (1 << 20)) { A2.err = 1; return; }
      plan->seed[(size_t)r] = (int32_t)m - 1;
      for (int64_t h = h0; h < h1; ++h) {
        if (b->realign_hap && !b->realign_hap[h]) continue;
        const int64_t hl = b->hap_off[h + 1] - b->hap_off[h];
        if (hl < 0 || hl > (1 << 20)) { A2.err = 2; return; }
        PairDesc pd;
        pd.read_off = b->read_off[r]; pd.out_idx = ll_base + (r - r0) * H + (h - h0);
        pd.m = (int32_t)m; pd.hap_full_len = (int32_t)hl;
        pd.generic = (read_acgt[(size_t)r] && hap_acgt[(size_t)h]) ? 0 : 1;
        int64_t pos = 0, n = 0;
        if (hl > 60) {
          n = ltr::hap_window(hl, F, &pos);
          if (n <= 0) { A2.err = 3; return; }
        }
        pd.hap_off = b->hap_off[h] + pos; pd.n = (int32_t)n;
        // (the rule's answer for (n, m, hl, generic) is kept per host thread in a direct-mapped table, tagged with the plan: a
        // catalogue of short repeats asks for the same few thousand combinations over and over, and the rule -- five packed
        // segment widths costed per pair, a logarithm -- was 120 ns per pair, 1.8 ms per 235 000-pair chunk on 16 threads)
        ltrp::PairClass pc;
        {
          const uint64_t tag = (uint64_t)n | ((uint64_t)m << 21) | ((uint64_t)hl << 42) | ((uint64_t)pd.generic << 63);   // (n, m, hl <= 2^20)
          ClassMemo& E = memo[(size_t)((tag * 0x9E3779B97F4A7C15ull) >> (64 - kClassMemoBits))];
          if (E.tag != tag || E.plan_id != plan_id) { E.pc = ltrp::classify_pair(rules, n, m, hl, pd.generic != 0); E.tag = tag; E.plan_id = plan_id; }
          pc = E.pc;
        }
        if (!pc.shortcut) {
          A2.cells += (double)n * (double)m;
          A2.max_len = std::max<int32_t>(A2.max_len, (int32_t)std::max(n, m));
        }
        // (under the plan kernel the one-wave and packed classes score their failed certificates themselves, and so it does the
        // pairs that start out in a list: only the workgroup classes feed the exact launches)
        if (pc.x_candidate && (!plan->use_plan || pc.uses_wg)) A2.xcand[pc.xc]++;
        else if (pc.x_candidate && pc.cls >= kNumFast) A2.xstart[pc.xc]++;
        if (pc.uses_wg) A2.uses_wg = 1;
        pairs[(size_t)at] = pd; bin[(size_t)at] = pc.cls; key[(size_t)at] = pc.key;
        ++at;
      }
    }
    // (Launch order inside a class is longest pair first, nothing else.  Keeping the pairs of a locus together --
    // one launch-order key per locus, in steps of 1/16 octave, so that the 30 reads of a haplotype are popped by
    // neighbouring waves -- was measured on MI355X: the L2 fetch volume of the large launches did not move
    // (125 MB each: consecutive pops land on different XCDs, each with its own L2) and the pass went from 245.6
    // to 257.6 ms on the coarser longest-first order.)
  });
  for (int64_t l = 0; l < b->n_loci; ++l) {
    const LocusAcc& A2 = acc[(size_t)l];
    if (A2.err) {
      ltr::set_error(ctx, A2.err == 1 ? "empty or oversized read (the reference is undefined for an empty read)"
                                      : (A2.err == 2 ? "bad haplotype length"
                                                     : "haplotype window is empty (only possible with indel_flank_len < 5; undefined in the reference)"));
      delete plan; plan = nullptr; return LTR_ERR_INVALID;
    }
    cells += A2.cells; max_len = std::max(max_len, A2.max_len);
    for (int c = 0; c < kNumExact; ++c) { xcand[c] += A2.xcand[c]; xstart[c] += A2.xstart[c]; }
    if (A2.uses_wg) plan->uses_wg = true;
  }
  plan->ll_size = ll_off; plan->n_pairs = n_pairs_total;
  plan->cells = cells; plan->input_bytes = in_bytes; plan->max_len = max_len;

  // Whole rounds of four-wave workgroups for the reads of 3.6 - 5.1 kb, the rest of them on eight waves (make_rules): the
  // pairs beyond the quota -- the last ones in batch order -- move to the eight-wave class of their length.
  if (rules.wg_wide4 && rules.wide4_quota != INT64_MAX) {
    int64_t seen = 0;
    for (int64_t i = 0; i < n_pairs_total; ++i) {
      const int k = bin[(size_t)i];
      if (k < kWg4First + (ltrp::kWg4WideMinW - kWg4MinW) || k >= kWg8First) continue;
      if (++seen <= rules.wide4_quota) continue;
      const int C = pairs[(size_t)i].m - 1;
      bin[(size_t)i] = (int16_t)(kWg8First + std::max((C + 511) / 512, kWg8MinW) - kWg8MinW);
    }
  }
  LTR_DBG("plan: pairs described");
  // ---- bin by launch class, longest first inside a class (ltrp::sort_by_class; all host cores) ----
  // pairs with bytes outside ACGT ("generic") sit behind every certificate class: they skip the LUT kernels
  // and are pre-seeded into the exact kernel's list
  int counts[kNumKernels] = {0};
  // Multi-width launches (ltr_dp_multi_kernel, ltr_dp_pack_multi_kernel: the one-wave classes of strip widths 11 .. 20 /
  // the packed widths 13 .. 20 as one persistent launch each) in automatic mode for plans of 512 .. 4096 pairs per CU.
  // Measured on MI355X against a launch per class (tests/manual/gpu_multi_ab.py): a 1250-locus shard of config 3 (717 pairs
  // per CU) 32.7 against 33.5 ms per pass, 8 certificate launches against 17; a 12 500-locus shard of the catalogue 10.6
  // against 11.6.  Below: a 625-locus shard -- 360 pairs per CU -- 18.3 against 17.6: with a handful of launches left the
  // two launch streams have little to fill each other's ends with.  Above: config 3 whole (5730 per CU) 240.0 against 240.4,
  // the catalogue whole 62.3 against 62.3 -- nothing to gain, and every call into a class's body saves its callee-saved
  // registers: ~0.9 GB of scratch write-backs per config-3 pass (rocprofv3 WRITE_SIZE; no time, but 5 x the pass's
  // algorithmic bytes) that a launch per class does not write.
  const bool use_multi = !plan->use_plan && ctx->pair_packing < 0 && ctx->dbg.no_multi <= 0 &&
                         (ctx->dbg.no_multi < 0 || (n_pairs_total >= (int64_t)512 * ctx->n_cu && n_pairs_total < (int64_t)4096 * ctx->n_cu));
  RawBuf<int32_t>& order = ctx->scratch.order;
  order.resize(pairs.size());
  ltrp::sort_by_class(bin.data(), key.data(), (int64_t)pairs.size(), ctx->pair_packing < 0 ? (ctx->dbg.fold_rounds > 0 ? ctx->dbg.fold_rounds : ltrp::kFoldRounds) : 0, ctx->n_cu,
                      order.data(), plan->bin_first, counts, plan->use_plan ? 2 : (use_multi ? 1 : 0));
  for (int c = 0; c < kNumExact; ++c) plan->x_seed[c] = counts[kNumFast + c];
  LTR_DBG("plan: sorted");
  RawBuf<PairDesc>& sorted = ctx->scratch.sorted;
  sorted.resize(pairs.size());
  ltr::parallel_for((int64_t)((pairs.size() + kPlanBlock - 1) / kPlanBlock), 1, [&](int64_t c) {
    for (size_t i = (size_t)c * kPlanBlock; i < std::min(pairs.size(), ((size_t)c + 1) * kPlanBlock); ++i) sorted[i] = pairs[(size_t)order[i]];
  }, 1);
  LTR_DBG("plan: gathered");
  plan_class_stats(plan, sorted, order, key);
  plan_launch_order(plan, use_multi);
  if (plan->use_plan && plan->plan_rep < 0) {
    // nothing for the plan kernel to score (workgroup classes and list starters only): the exact launches take the starters
    plan->use_plan = false;
    for (int c = 0; c < kNumExact; ++c) xcand[c] += xstart[c];
  }

  LTR_DBG("plan: %zu pairs, max_len %d", pairs.size(), max_len);
  int rc = build_tables(ctx, (int64_t)max_len + 2);
  if (rc != LTR_OK) { delete plan; plan = nullptr; return rc; }
  LTR_DBG("tables built");

  // ---- upload ---------------------------------------------------------------------------
  auto fail = [&](int code) { (void)hipStreamSynchronize(ctx->up_stream); destroy_plan(plan, true); plan = nullptr; return code; };       // (ctx->mu is held here; nothing of this plan in flight when its buffers go back to the pool)
#define PLAN_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ltr::set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_)); return fail(LTR_ERR_HIP); } } while (0)
  const int64_t rbytes = b->n_reads > 0 ? b->read_off[b->n_reads] : 0;
  const int64_t hbytes = b->n_haps > 0 ? b->hap_off[b->n_haps] : 0;
  // 96 bytes of zero padding either side: the kernel streams haplotype rows without clamping
  // ... and the two-pairs-per-wave kernels keep streaming rows of the SHORTER haplotype of a wave
  // until the longer one ends: the tail pad also covers the longest window of the batch
  const size_t hap_tail = (size_t)kHapPad + (size_t)max_len + 384;   // (+ the workgroup kernels' 64-row chunks, two ahead)
  const size_t hap_buf = (size_t)std::max<int64_t>(hbytes, 1) + kHapPad + hap_tail;
  const size_t read_buf = (size_t)std::max<int64_t>(rbytes, 1) + kReadPad;       // (the packed kernels load a lane's strip of bytes unclamped)
  // persistent grid per launch (occupancy x CUs, asked from the runtime once per context), the tables of the multi-width packed launch,
  // the images of the control words and of the pre-seeded list heads: everything the device is given besides the batch itself
  if (!ctx->have_grids) PLAN_TRY(ctx_query_grids(ctx));
  std::vector<PackTable> tabs;
  plan_size_grids(ctx, plan, counts, xcand, &tabs);
  std::vector<uint32_t> ctrl(kCtrlWords, 0);
  for (int c = 0; c < kNumExact; ++c) ctrl[kRedoCountSlot + c] = (uint32_t)plan->x_seed[c];
  // image of the pre-seeded list heads: the sorted-array indices bin_first[kNumFast] .. n_pairs, in order
  const int n_seed = plan->bin_first[kNumKernels] - plan->bin_first[kNumFast];
  std::vector<int32_t> init((size_t)std::max(n_seed, 1), 0);
  for (int g2 = 0; g2 < n_seed; ++g2) init[(size_t)g2] = plan->bin_first[kNumFast] + g2;
  plan->redo_cap = (int64_t)std::max<size_t>(sorted.size(), 1);
  const bool want_clock = !plan->plan_entries.empty() && ctx->dbg.wave_clock > 0;

  // COMPACT plans (round 6): a one-locus call -- HapAligner::process_reads as the reference calls it, once per locus
  // (seq_stutter_genotyper.cpp:517-523) -- is a few hundred pairs and a few hundred KB; what it costs is not the DP (0.1 ms) but
  // the dozen allocations, the eight copies (four of them synchronous), the three fills, the hap-code launch, the events and
  // the two downloads around it.  Such a plan is ONE device block laid out below, filled from ONE pinned image the host writes
  // (the haplotype codes too: a host loop over a few KB) by ONE copy over the DMA engines; the control words arrive with it (the
  // first execute skips its reset); the scores are written by the kernel straight into pinned host memory the device can
  // reach, so ltr_plan_fetch is an event wait and a memcpy.  Same kernels, same launch, same bits.
  auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_queue = 0, o_ctrl = up256(o_queue + kCtrlWords * sizeof(uint32_t)), o_ent = up256(o_ctrl + kCtrlWords * sizeof(uint32_t)),
               o_tabs = up256(o_ent + std::max<size_t>(plan->plan_entries.size(), 1) * sizeof(PlanEntry)),
               o_init = up256(o_tabs + std::max<size_t>(tabs.size(), 1) * sizeof(PackTable)), o_pairs = up256(o_init + init.size() * sizeof(int32_t)),
               o_reads = up256(o_pairs + std::max<size_t>(sorted.size(), 1) * sizeof(PairDesc)), o_haps = up256(o_reads + read_buf),
               o_codes = up256(o_haps + hap_buf), image_bytes = up256(o_codes + hap_buf * sizeof(uint16_t)),
               o_list = image_bytes, block_bytes = up256(o_list + (size_t)plan->redo_cap * kNumExact * sizeof(int32_t));
  const size_t ll_bytes = (size_t)std::max<int64_t>(plan->ll_size, 1) * sizeof(double);
  const bool compact = ctx->dbg.compact_plan >= 0 && !want_clock && image_bytes <= kCompactImageMax && ll_bytes <= kCompactLlMax;
  hipEvent_t ev_copied = nullptr;
  auto copies_done = [&]() { if (!ev_copied) return hipSuccess; const hipError_t e_ = hipEventSynchronize(ev_copied); (void)hipEventDestroy(ev_copied); ev_copied = nullptr; return e_; };
#define PLAN_TRY2(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (void)copies_done(); ltr::set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_)); return fail(LTR_ERR_HIP); } } while (0)
  if (compact) {
    PLAN_TRY(ctx->pool.alloc((void**)&plan->d_block, block_bytes));
    uint8_t* const base = (uint8_t*)plan->d_block;
    plan->d_queue = (uint32_t*)(base + o_queue); plan->d_ctrl_init = (uint32_t*)(base + o_ctrl); plan->d_pl_entries = (PlanEntry*)(base + o_ent);
    plan->d_pk_tabs = (PackTable*)(base + o_tabs); plan->d_redo_init = (int32_t*)(base + o_init); plan->d_pairs = (PairDesc*)(base + o_pairs);
    plan->d_reads = base + o_reads; plan->d_haps = base + o_haps; plan->d_hap_codes = (uint16_t*)(base + o_codes);
    plan->d_redo_list = (int32_t*)(base + o_list);
    plan->d_redo_count = plan->d_queue + kRedoCountSlot;
    if (plan->plan_entries.empty()) plan->d_pl_entries = nullptr;
    if (tabs.empty()) plan->d_pk_tabs = nullptr;
    // the image: the context's pinned staging block, free again once the previous compact plan's copy is through
    if (ctx->compact_ev_pending) { PLAN_TRY(hipEventSynchronize(ctx->compact_ev)); ctx->compact_ev_pending = false; }
    ctx->compact_stage.resize(image_bytes);
    uint8_t* const img = ctx->compact_stage.data();
    std::memset(img, 0, o_pairs);                                                      // control words, table padding
    std::memcpy(img + o_ctrl, ctrl.data(), ctrl.size() * sizeof(uint32_t));
    std::memcpy(img + o_queue, ctrl.data(), ctrl.size() * sizeof(uint32_t));       // (the control words themselves: the first execute needs no reset)
    if (!plan->plan_entries.empty()) std::memcpy(img + o_ent, plan->plan_entries.data(), plan->plan_entries.size() * sizeof(PlanEntry));
    if (!tabs.empty()) std::memcpy(img + o_tabs, tabs.data(), tabs.size() * sizeof(PackTable));
    std::memcpy(img + o_init, init.data(), init.size() * sizeof(int32_t));
    if (!sorted.empty()) std::memcpy(img + o_pairs, sorted.data(), sorted.size() * sizeof(PairDesc));
    std::memset(img + o_reads, 0, o_haps - o_reads);
    if (rbytes) std::memcpy(img + o_reads, b->read_bytes, (size_t)rbytes);
    std::memset(img + o_haps, 0, o_codes - o_haps);
    if (hbytes) std::memcpy(img + o_haps + kHapPad, b->hap_bytes, (size_t)hbytes);
    {
      // the haplotypes as emission-table block offsets (what ltr_hap_codes_kernel forms on the device for the large plans)
      const uint8_t* hsrc = img + o_haps;
      uint16_t* hc = (uint16_t*)(img + o_codes);
      for (size_t i = 0; i < hap_buf; ++i) hc[i] = (uint16_t)((((uint32_t)hsrc[i] >> 1) & 3u) << 12);
      std::memset(img + o_codes + hap_buf * sizeof(uint16_t), 0, image_bytes - (o_codes + hap_buf * sizeof(uint16_t)));
    }
    PLAN_TRY(hipMemcpyAsync(plan->d_block, img, image_bytes, hipMemcpyHostToDevice, ctx->up_stream));
    if (!ctx->compact_ev) PLAN_TRY(hipEventCreateWithFlags(&ctx->compact_ev, hipEventDisableTiming));
    PLAN_TRY(hipEventRecord(ctx->compact_ev, ctx->up_stream));
    ctx->compact_ev_pending = true;
    plan->ev_up = ctx_take_event(ctx, false);
    if (!plan->ev_up) { ltr::set_error(ctx, "hipEventCreate failed"); return fail(LTR_ERR_HIP); }
    PLAN_TRY(hipEventRecord(plan->ev_up, ctx->up_stream));
    // the scores: pinned host memory the kernel writes into (zero = what a masked cell keeps)
    plan->h_ll = ctx_take_pinned(ctx, ll_bytes, &plan->h_ll_cap, &plan->d_ll);
    if (!plan->h_ll) { plan->d_ll = nullptr; ltr::set_error(ctx, "out of pinned host memory"); return fail(LTR_ERR_NOMEM); }
    std::memset(plan->h_ll, 0, ll_bytes);
    plan->ctrl_fresh = true;
  } else {
  PLAN_TRY(ctx->pool.alloc((void**)&plan->d_reads, read_buf));
  PLAN_TRY(ctx->pool.alloc((void**)&plan->d_haps, hap_buf));
  PLAN_TRY(ctx->pool.alloc((void**)&plan->d_hap_codes, hap_buf * sizeof(uint16_t)));
  PLAN_TRY(ctx->pool.alloc((void**)&plan->d_pairs, std::max<size_t>(sorted.size(), 1) * sizeof(PairDesc)));
  PLAN_TRY(ctx->pool.alloc((void**)&plan->d_ll, ll_bytes));
  // Everything on the context's upload stream, the COPIES FIRST: they go over the DMA engines, while a fill (hipMemset) or the
  // hap-code kernel needs wave slots -- and behind the persistent launch of the previous chunk of ltr_calc_hap_aln_probs there are
  // none until that launch drains.  The host waits for the copies only (ev_copied, at the end of this call: the caller's arrays and
  // the context's staging are free again on return); the plan's executes wait for all of it (ev_up).  (Measured on MI355X, 30 000
  // catalogue loci in three chunks: with hipMemset + hipMemcpy on the null stream the uploads of chunks 1 and 2 took 2.0 - 2.1 ms
  // against 0.24 for chunk 0 -- the copy sat behind the fill, the fill behind the running plan kernel; profiles/r05/e2e_prep_ahead.log.)
  if (rbytes) PLAN_TRY(hipMemcpyAsync(plan->d_reads, b->read_bytes, (size_t)rbytes, hipMemcpyHostToDevice, ctx->up_stream));
  if (hbytes) PLAN_TRY(hipMemcpyAsync(plan->d_haps + kHapPad, b->hap_bytes, (size_t)hbytes, hipMemcpyHostToDevice, ctx->up_stream));
  if (!sorted.empty()) PLAN_TRY(hipMemcpyAsync(plan->d_pairs, sorted.data(), sorted.size() * sizeof(PairDesc), hipMemcpyHostToDevice, ctx->up_stream));
  PLAN_TRY(hipEventCreateWithFlags(&ev_copied, hipEventDisableTiming));
  {
    const hipError_t e_ = hipEventRecord(ev_copied, ctx->up_stream);
    if (e_ != hipSuccess) { (void)hipEventDestroy(ev_copied); ev_copied = nullptr; ltr::set_error(ctx, std::string("hipEventRecord: ") + hipGetErrorString(e_)); return fail(LTR_ERR_HIP); }
  }
  LTR_DBG("upload: copies queued");
  {
    // the zero padding either side of the haplotype bytes, then the hap codes (see ltr_hap_codes_kernel), then the output rows
    PLAN_TRY2(hipMemsetAsync(plan->d_haps, 0, kHapPad, ctx->up_stream));
    PLAN_TRY2(hipMemsetAsync(plan->d_haps + kHapPad + hbytes, 0, hap_buf - kHapPad - (size_t)hbytes, ctx->up_stream));
    const int blocks = (int)std::min<size_t>((hap_buf / 4 + 255) / 256 + 1, (size_t)ctx->n_cu * 8);
    hipLaunchKernelGGL(ltr_hap_codes_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->up_stream, plan->d_haps, plan->d_hap_codes, hap_buf);
    PLAN_TRY2(hipGetLastError());
    PLAN_TRY2(hipMemsetAsync(plan->d_ll, 0, ll_bytes, ctx->up_stream));
    plan->ev_up = ctx_take_event(ctx, false);
    if (!plan->ev_up) { (void)copies_done(); ltr::set_error(ctx, "hipEventCreate failed"); return fail(LTR_ERR_HIP); }
    PLAN_TRY2(hipEventRecord(plan->ev_up, ctx->up_stream));
  }
  PLAN_TRY2(ctx->pool.alloc((void**)&plan->d_queue, kCtrlWords * sizeof(uint32_t)));      // work queues + exact list lengths (ltr_plan.h)
  plan->d_redo_count = plan->d_queue + kRedoCountSlot;
  LTR_DBG("uploaded");
  if (!tabs.empty()) {
    PLAN_TRY2(ctx->pool.alloc((void**)&plan->d_pk_tabs, tabs.size() * sizeof(PackTable)));
    PLAN_TRY2(hipMemcpy(plan->d_pk_tabs, tabs.data(), tabs.size() * sizeof(PackTable), hipMemcpyHostToDevice));
  }
  if (want_clock) {
    const size_t nb = ((size_t)ctx->full_plan_grid * kBlockWaves * 4 + 4096 + 256) * sizeof(unsigned long long);
    PLAN_TRY2(ctx->pool.alloc((void**)&plan->d_wave_clock, nb));
    PLAN_TRY2(hipMemset(plan->d_wave_clock, 0, nb));
  }
  if (!plan->plan_entries.empty()) {
    PLAN_TRY2(ctx->pool.alloc((void**)&plan->d_pl_entries, plan->plan_entries.size() * sizeof(PlanEntry)));
    PLAN_TRY2(hipMemcpy(plan->d_pl_entries, plan->plan_entries.data(), plan->plan_entries.size() * sizeof(PlanEntry), hipMemcpyHostToDevice));
  }
  PLAN_TRY2(ctx->pool.alloc((void**)&plan->d_redo_list, (size_t)plan->redo_cap * kNumExact * sizeof(int32_t)));
  PLAN_TRY2(ctx->pool.alloc((void**)&plan->d_ctrl_init, ctrl.size() * sizeof(uint32_t)));
  PLAN_TRY2(hipMemcpy(plan->d_ctrl_init, ctrl.data(), ctrl.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  PLAN_TRY2(ctx->pool.alloc((void**)&plan->d_redo_init, init.size() * sizeof(int32_t)));
  PLAN_TRY2(hipMemcpy(plan->d_redo_init, init.data(), init.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  plan->scratch_stride = ((max_len + 2 + 15) / 16) * 16;
  {
    // boundary strips: 6 arrays x stride doubles per resident wave; for very long reads shrink
    // the persistent grids instead of allocating more than ~8 GB
    const size_t per_wave = (size_t)6 * plan->scratch_stride * sizeof(double);
    // The certificate launches of a plan are dealt over TWO streams (regions of their own in the scratch strips): every
    // launch ends in a last, partly filled round of pairs per wave slot, and the next class fills it.  Measured on
    // MI355X, config 3, same box: 10 000 loci 245.5 ms per pass on one stream, 241.6 on two, 243.0 on three, 244.7 on
    // four; 1250 loci (one GPU's share of the catalogue sharded over eight) 38.6 / 32.2 / 32.4 / 33.8 ms -- 2.16e12 ->
    // 2.58e12 cells/s.  (ltr_ctx_set_debug "fan_lanes" / "fan_pairs": the A/B switches of those runs.)
    const int64_t fan_below = ctx->dbg.fan_pairs > 0 ? ctx->dbg.fan_pairs : INT64_MAX;
    const int fan_n = ctx->dbg.fan_lanes > 0 ? std::min(4, ctx->dbg.fan_lanes) : 4;       // two lanes for the big classes + two for the small ones
    // (... or, with fewer pairs, when pairs go to workgroup kernels: a launch of a few hundred 5-kb pairs is a handful of
    // rounds of one pair per workgroup, each milliseconds long -- config5hifi: the 180 pairs of the W = 11 class, 4.2 ms,
    // used to start behind the 23 ms of the W = 10 class)
    plan->fan_lanes = (ctx->pair_packing < 0 && (plan->n_pairs >= (int64_t)16 * ctx->n_cu || plan->uses_wg) && plan->n_pairs < fan_below) ? fan_n : 1;
    const int cap = (int)std::max<size_t>(16, ((size_t)8 << 30) / (per_wave * kBlockWaves * ((size_t)plan->fan_lanes + 1)));
    for (int k = 0; k < kNumBins; ++k) plan->bin_grid[k] = std::min(plan->bin_grid[k], cap);
    plan->multi_grid = std::min(plan->multi_grid, cap);
    plan->plan_grid = std::min(plan->plan_grid, cap);
    for (int c = 0; c <= kXLong; ++c) plan->x_grid[c] = std::min(plan->x_grid[c], cap);
    plan->redo_grid = plan->x_grid[kXGeneric];
    plan->max_grid = std::min(plan->max_grid, cap);
    plan->scratch_lane_stride = (size_t)plan->max_grid * kBlockWaves * 6 * (size_t)plan->scratch_stride;
    PLAN_TRY2(ctx->pool.alloc((void**)&plan->d_scratch, plan->scratch_lane_stride * sizeof(double) * ((size_t)plan->fan_lanes + 1)));   // (+ 1: the kXLong exact launch, see ltr_plan_execute)
    if (plan->fan_lanes > 1) {
      PLAN_TRY2(hipEventCreateWithFlags(&plan->ev_fork, hipEventDisableTiming));
      for (int k = 0; k < 3; ++k) PLAN_TRY2(hipEventCreateWithFlags(&plan->ev_join[k], hipEventDisableTiming));
    }
  }
  plan->ev0 = ctx_take_event(ctx, true);
  plan->ev1 = ctx_take_event(ctx, true);
  if (!plan->ev0 || !plan->ev1) { (void)copies_done(); ltr::set_error(ctx, "hipEventCreate failed"); return fail(LTR_ERR_HIP); }
  // (the per-launch events are created by ltr_plan_set_timing, only for plans that ask for them)
  PLAN_TRY(copies_done());                                     // the caller's arrays / the context's staging are free again
  LTR_DBG("upload: copies done");
#undef PLAN_TRY2
#undef PLAN_TRY
  ctx->plans.insert(plan);                                     // (ctx->mu is held)
  *out = plan;
  plan = nullptr;                                              // (handed out)
  return LTR_OK;
}

int ltr_plan_create(ltr_ctx* ctx, const ltr_locus_batch* b, ltr_plan** out) {
  if (!ctx || !b || !out) return LTR_ERR_INVALID;
  *out = nullptr;
  LTR_DBG("plan: create");
  std::lock_guard<std::mutex> lk(ctx->mu);
  // (no exception crosses the C-ABI: a host array that cannot grow ends the call with LTR_ERR_NOMEM and the half-built plan is taken
  // down -- after the upload stream has drained: copies out of the caller's arrays may be in flight)
  ltr_plan* plan = nullptr;
  auto take_down = [&]() { if (plan) { (void)hipStreamSynchronize(ctx->up_stream); destroy_plan(plan, true); plan = nullptr; } };
  try { return plan_create_locked(ctx, b, out, plan); }
  catch (const std::bad_alloc&) { take_down(); ltr::set_error(ctx, "out of host memory"); return LTR_ERR_NOMEM; }
  catch (const std::exception& e_) { take_down(); ltr::set_error(ctx, std::string("internal error: ") + e_.what()); return LTR_ERR_INVALID; }
  catch (...) { take_down(); ltr::set_error(ctx, "internal error"); return LTR_ERR_INVALID; }
}

int64_t ltr_plan_num_pairs(const ltr_plan* p) { return p ? p->n_pairs : 0; }
int64_t ltr_plan_ll_size(const ltr_plan* p) { return p ? p->ll_size : 0; }
double ltr_plan_cells(const ltr_plan* p) { return p ? p->cells : 0.0; }
double ltr_plan_input_bytes(const ltr_plan* p) { return p ? p->input_bytes : 0.0; }

// The statistics slots that have arrived (under ctx->mu).  Certificate pass: more than half of the pairs failed -> the threshold
// kernels go first from now on; threshold pass: fewer than a quarter aborted -> back to the certificates (a pair that aborts
// would have failed its certificate too; the converse does not hold, so the way back is the cautious one).
static void wg_stats_poll(ltr_ctx* ctx) {
  if (!ctx->wg_stat_pin) return;
  for (int i = 0; i < ltr_ctx::kWgStatSlots; ++i) {
    ltr_ctx::WgStatSlot& sl = ctx->wg_stat[i];
    if (!sl.busy || hipEventQuery(sl.ev) != hipSuccess) continue;
    sl.busy = false;
    const uint32_t unfinished = ctx->wg_stat_pin[2 * i], scored = ctx->wg_stat_pin[2 * i + 1];
    if (sl.epoch != ctx->wg_epoch || scored == 0) continue;
    ctx->wg_last_unfinished = unfinished; ctx->wg_last_scored = scored;
    if (sl.mode == 0) ctx->wg_thr_first = ((uint64_t)unfinished * 2 > scored) ? 1 : 0;
    else ctx->wg_thr_first = ((uint64_t)unfinished * 4 >= scored) ? 1 : 0;
  }
  (void)hipGetLastError();                                       // (hipEventQuery's hipErrorNotReady is no error)
}

int ltr_ctx_wg_first_pass(ltr_ctx* ctx, int64_t* last_unfinished, int64_t* last_scored) {
  if (!ctx) return LTR_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  (void)hipSetDevice(ctx->device);
  // (the query waits for the statistics of the executes queued so far -- ltr_plan_execute itself never does)
  for (ltr_ctx::WgStatSlot& sl : ctx->wg_stat) if (sl.busy && sl.ev) (void)hipEventSynchronize(sl.ev);
  wg_stats_poll(ctx);
  if (last_unfinished) *last_unfinished = ctx->wg_last_unfinished;
  if (last_scored) *last_scored = ctx->wg_last_scored;
  return ctx->dbg.wg_first_pass == 1 ? 0 : (ctx->dbg.wg_first_pass == 2 ? 1 : ctx->wg_thr_first);
}

static int plan_execute_impl(ltr_plan* plan, ltr_ctx* ctx, double* d_out_ll, void* stream_v);

int ltr_plan_execute(ltr_plan* plan, double* d_out_ll, void* stream_v) {
  if (!plan) return LTR_ERR_INVALID;
  ltr_ctx* ctx = plan->ctx;
  if (!ctx) return LTR_ERR_INVALID;                          // the context was destroyed before this plan
  // (no exception crosses the C-ABI: the launch-order lists are host vectors)
  try { return plan_execute_impl(plan, ctx, d_out_ll, stream_v); }
  catch (const std::bad_alloc&) { ltr::set_error(ctx, "out of host memory"); return LTR_ERR_NOMEM; }
  catch (const std::exception& e_) { ltr::set_error(ctx, std::string("internal error: ") + e_.what()); return LTR_ERR_INVALID; }
  catch (...) { ltr::set_error(ctx, "internal error"); return LTR_ERR_INVALID; }
}

static int plan_execute_impl(ltr_plan* plan, ltr_ctx* ctx, double* d_out_ll, void* stream_v) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream_v ? (hipStream_t)stream_v : ctx->stream;
  double* out = d_out_ll ? d_out_ll : plan->d_ll;
  KernelArgs A;
  int wg_learnt = 0;
  A.pairs = plan->d_pairs; A.index = nullptr; A.n_pairs_dev = nullptr; A.queue = nullptr;
  for (int c = 0; c < kNumExact; ++c) A.xlist[c] = plan->d_redo_list + (int64_t)c * plan->redo_cap;
  A.xcount = plan->d_redo_count;
  A.read_bytes = plan->d_reads; A.hap_bytes = plan->d_haps + kHapPad; A.hap_codes = plan->d_hap_codes + kHapPad;
  A.out_ll = out;
  {
    // (a plan being created on another thread may be rebuilding the model tables: snapshot them under the lock)
    std::lock_guard<std::mutex> lk(ctx->mu);
    A.lpc = ctx->d_lpc;
    A.colXZ = ctx->d_colXZ; A.row0XY = ctx->d_row0XY; A.thr_tab = ctx->d_thr; A.table_len = (int32_t)std::min<int64_t>(ctx->table_len + 1, 0x7fffffff);
    A.mc = ctx->mc;
    if (plan->uses_wg) { wg_stats_poll(ctx); wg_learnt = ctx->wg_thr_first; }
  }
  A.scratch = plan->d_scratch; A.scratch_stride = plan->scratch_stride;
  A.c_lo = 0; A.c_hi = 0x7fffffff; A.lp_shift = 6;
  A.mk_n = 0; A.queue_base = plan->d_queue; A.pk_tabs = nullptr; A.pk_ntabs = 0;
  A.pl_entries = nullptr; A.pl_n = 0; A.wave_clock = plan->d_wave_clock;
  for (int r = 0; r < kMultiMax; ++r) { A.mk_w[r] = kWMax; A.mk_first[r] = 0; A.mk_np[r] = 0; A.mk_class[r] = 0; }
  for (int r = 0; r < 5; ++r) { A.pk_shift[r] = kPackMaxShift; A.pk_first[r] = 0; A.pk_end[r] = 0; A.pk_grp_end[r] = 0; }
  // symmetric indel model (ins->match == del->match, match->ins == match->del): 11-op cell body
  const bool sym = (A.mc.b == A.mc.d) && (A.mc.f == A.mc.g);
  if (plan->uses_wg && !sym) {
    ltr::set_error(ctx, "the alignment parameters changed from a symmetric to an asymmetric indel model after this plan was created: create it again");
    return LTR_ERR_INVALID;
  }
  {
    const float cabs = std::fabs(A.mc.c);
    const bool pen_ok = (cabs * 1.0e9f > 600.0f) && ((int64_t)(600.0f / cabs) + 2 <= kPenKMax);
    A.xlut = (plan->xlut && sym && pen_ok) ? 1 : 0;           // (parameters may have changed since the plan was binned: then everything goes to the generic exact kernel)
    A.thr_ok = pen_ok ? 1 : 0;                                 // (the threshold table is rebuilt with every parameter set: valid whenever it fits)
    if (!A.xlut) for (int c = 1; c < kNumExact; ++c) A.xlist[c] = A.xlist[kXGeneric];
  }
  // first pass of the workgroup classes: threshold kernels when the context has learnt that certificates fail here (or on request);
  // they need the threshold table (A.xlut)
  const bool wg_thr = plan->uses_wg && A.xlut && (ctx->dbg.wg_first_pass == 2 || (ctx->dbg.wg_first_pass == 0 && wg_learnt != 0));
  // the generic list starts as the non-ACGT pairs; the certificate kernels append to the lists
  // one D2D copy resets the work queues (zeros) and the redo count (= number of generic pairs)
  // (the plan's upload first: a compact plan's control-word image and list heads arrive with it -- the copies below read them)
  if (plan->ev_up) HIP_TRY(ctx, hipStreamWaitEvent(st, plan->ev_up, 0));
  if (plan->ctrl_fresh) plan->ctrl_fresh = false;               // (a compact plan's first execute: the control words came with the upload)
  else HIP_TRY(ctx, hipMemcpyAsync(plan->d_queue, plan->d_ctrl_init, kCtrlWords * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  for (int c = 0; c < kNumExact; ++c)
    if (plan->x_seed[c] > 0) {
      // (when the LUT exact kernels are off for this execute every list is the generic one: seeds pile up behind each other)
      int64_t at = 0;
      if (!A.xlut) for (int c2 = 0; c2 < c; ++c2) at += plan->x_seed[c2];
      HIP_TRY(ctx, hipMemcpyAsync(A.xlist[c] + at, plan->d_redo_init + (plan->bin_first[kNumFast + c] - plan->bin_first[kNumFast]),
                                  (size_t)plan->x_seed[c] * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    }
  if (!A.xlut) {
    // ... and the generic list's length is the sum of the seeds (one word, rewritten after the control block reset)
    uint32_t tot = 0;
    for (int c = 0; c < kNumExact; ++c) tot += (uint32_t)plan->x_seed[c];
    if (tot != (uint32_t)plan->x_seed[kXGeneric]) {
      plan->seed_total = tot;
      HIP_TRY(ctx, hipMemcpyAsync(plan->d_redo_count + kXGeneric, &plan->seed_total, sizeof(uint32_t), hipMemcpyHostToDevice, st));
    }
  }
  HIP_TRY(ctx, hipEventRecord(plan->ev0, st));
  int launches = 0;
  // Launch order: the certificate classes longest reads first (plan->order), then the exact kernels; with per-launch
  // timing, launch number o runs between bin_ev[o] and bin_ev[o+1] (plan->order_pos).
  // (The classes are independent and every launch ends in a tail in which only its longest pairs still run: the
  // launches alternate between two streams, see ltr_plan_create.)
  int o = 0;
  if (plan->timing) HIP_TRY(ctx, hipEventRecord(plan->bin_ev[o], st));
  // The launches go round-robin over the plan's stream and side streams of the context: the classes that fill the GPU's
  // wave slots ("big") over lanes 0 .. nb-1, the classes that do not ("small": a chain of launches each as long as its
  // longest pair, whatever the GPU could do meanwhile) over lanes nb .. nl-1, queued FIRST: they trickle into the tails
  // of the big launches all along the plan instead of following the last of them one after the other (measured on
  // MI355X, a 1250-locus plan: ten small launches of 0.3 - 3 ms each behind the last big one, 4.5 ms of 33).
  hipStream_t lanes[4] = {st, ctx->aux[2], ctx->aux[3], ctx->aux[1]};
  const bool fan = plan->fan_lanes > 1 && !plan->timing && st != lanes[1] && st != lanes[2] && st != lanes[3];
  const int nl = fan ? plan->fan_lanes : 1;
  const int nb = nl >= 4 ? 2 : nl;                              // big lanes; with four lanes the last two take the small classes
  if (fan) {
    HIP_TRY(ctx, hipEventRecord(plan->ev_fork, st));
    for (int k = 1; k < nl; ++k) HIP_TRY(ctx, hipStreamWaitEvent(lanes[k], plan->ev_fork, 0));
  }
  // Exact kernels over whatever the certificate kernels queued (the list lengths live on the device); a kernel no pair
  // of the plan can reach is not launched.  The exact launches are independent of each other (own list, own queue word;
  // the generic kernel and kXLong park column blocks in strip regions of their own) and mostly latency: a handful of
  // pairs each, as long as their longest pair.  Without per-launch timing they run on side streams of the context,
  // each as soon as no certificate launch still to come can feed its list -- list c takes reads of at least
  // kListMinC[c] columns and the certificate classes run longest reads first -- and the plan's stream waits for them at
  // the end: a plan of 1250 loci used to end in ~4 ms of exact launches behind its last certificate launch.
  // (Not for plans of a few hundred pairs -- config 2: the cross-stream waits cost more than they hide, 0.14 ms per
  // pass against 0.10 -- and not when the lists are the bulk of the work, mode 4: 1.35e12 against 1.45e12 cells/s.)
  static const int kListMinC[kNumExact] = {0, 0, 64 * kXShortW + 1, 64 * kXMidW + 1, 64 * kXLongW + 1, kXWg4MaxC + 1};
  int64_t seeded = 0;
  for (int c = 0; c < kNumExact; ++c) seeded += plan->x_seed[c];
  const bool x_fan = !plan->timing && plan->n_pairs >= (int64_t)32 * ctx->n_cu && seeded * 16 < plan->n_pairs;
  if (x_fan && !plan->ev_fast) {
    // created into locals and published together: a failure half way must not leave the plan with ev_fast set and null events behind it
    hipEvent_t made[1 + (kNumExact + 1) + kNumExact * 4] = {nullptr};
    int n_made = 0;
    hipError_t e2 = hipSuccess;
    for (; n_made < (int)(sizeof(made) / sizeof(made[0])) && e2 == hipSuccess; ++n_made) e2 = hipEventCreateWithFlags(&made[n_made], hipEventDisableTiming);
    if (e2 != hipSuccess) {
      for (int i = 0; i < n_made; ++i) if (made[i]) (void)hipEventDestroy(made[i]);
      ltr::set_error(ctx, std::string("hipEventCreateWithFlags: ") + hipGetErrorString(e2));
      return LTR_ERR_HIP;
    }
    int at = 1;
    for (int c = 0; c <= kNumExact; ++c) plan->ev_x[c] = made[at++];
    for (int c = 0; c < kNumExact; ++c) for (int k = 0; k < 4; ++k) plan->ev_close[c][k] = made[at++];
    plan->ev_fast = made[0];
  }
  // side stream of every exact launch (x_fan): short / mid / long / the W = 20 launch / four-wave / eight-wave; generic stays on st
  auto exact_stream = [&](int which) -> hipStream_t {
    if (!x_fan || which == kXGeneric) return st;
    static const int kIdx[kNumExact + 1] = {-1, 4, 5, 6, 8, 9, 7};
    hipStream_t xs = ctx->aux[kIdx[which]];
    return xs == st ? st : xs;
  };
  bool x_done[kNumExact] = {false}, x_launched[kNumExact] = {false};
  // one exact list: launched on its side stream behind the certificate launches queued so far on every lane (x_fan), or
  // on the plan's stream
  auto launch_exact_list = [&](int c, bool small_events) -> int {
    x_done[c] = true;
    const bool usable = (c == kXGeneric) || A.xlut;
    const int grid = (c == kXGeneric && !A.xlut) ? std::max(plan->x_grid[c], (plan->n_pairs > 0) ? 1 : 0) : plan->x_grid[c];
    if (!(usable && grid > 0 && plan->n_pairs > 0)) return LTR_OK;
    KernelArgs X = A;
    X.first_pair = 0; X.n_pairs = 0; X.index = A.xlist[c]; X.n_pairs_dev = plan->d_redo_count + c;
    X.queue = plan->d_queue + kNumFast + c;
    X.scratch = plan->d_scratch; X.c_lo = 0; X.c_hi = 0x7fffffff; X.lp_shift = 6;
    const dim3 g((unsigned)grid);
    hipStream_t xs = exact_stream(c);
    auto behind_the_lanes = [&](hipStream_t s2) -> int {          // s2 waits for what every lane has queued so far
      if (s2 == st) return LTR_OK;
      for (int k = 0; k < nl; ++k) {
        // (big lanes: whatever is queued now; small lanes: the event recorded when their last feeder of this list was
        // queued -- or now, when the list stayed open to the end)
        if (k < nb || !small_events) HIP_TRY(ctx, hipEventRecord(plan->ev_close[c][k], lanes[k]));
        HIP_TRY(ctx, hipStreamWaitEvent(s2, plan->ev_close[c][k], 0));
      }
      return LTR_OK;
    };
    int rc2;
    if ((rc2 = behind_the_lanes(xs)) != LTR_OK) return rc2;
    if (c == kXWg4) {
      // the list of 1026 .. 3585-base reads is worked off by two launches: reads that fit one wavefront's widest
      // strips (<= 1281 bases) by the one-wave exact kernel with W = 20 -- 0.8e12 cells/s on four-wave workgroups
      // (W = 5) in round 2a -- the rest by the workgroup kernel; each skips the other's pairs (c_lo / c_hi)
      KernelArgs B = X;
      B.queue = plan->d_queue + kNumKernels;                   // (a queue word of its own: zeroed with the others)
      B.c_hi = 64 * kXWideW;
      const int gw = std::max(1, std::min(ctx->full_x_wide_grid, plan->max_grid_wide));
      hipStream_t ws = exact_stream(kNumExact);
      if (ws != xs && (rc2 = behind_the_lanes(ws)) != LTR_OK) return rc2;
      ltrk::launch_exact(ltrk::kXWideLaunch, sym, dim3((unsigned)gw), ws, B);
      if (ws != st) {
        HIP_TRY(ctx, hipEventRecord(plan->ev_x[kNumExact], ws));
        HIP_TRY(ctx, hipStreamWaitEvent(st == xs ? st : xs, plan->ev_x[kNumExact], 0));     // (joined through the list's own stream / event below)
      }
      X.c_lo = 64 * kXWideW + 1;
      ltrk::launch_exact(c, sym, g, xs, X);
    } else if (c == kXWg8) {
      // the list of 3586 .. 10241-base reads, two launches that skip each other's pairs: reads of up to 5121 bases on strips of
      // 10 columns at four waves per SIMD (the threshold bodies fit 128 registers up to there), the longer ones on 12 / 16 / 20
      // columns at three.  ONE AFTER THE OTHER on the list's stream, the long pairs first: side by side they do not share a CU
      // (an eight-wave workgroup of 168 registers leaves room for four waves of 128, not for eight), each kernel keeps half-empty
      // CUs from the other, and the pass takes longer than the two alone (rocprofv3 per dispatch, config5hifi through the lists:
      // 5.7 ms + 21 ms alone, 40 ms side by side: profiles/r06/pmc_dispatch_config5hifi_exact.txt)
      KernelArgs B = X;
      B.queue = plan->d_queue + kNumKernels + 1;               // (a queue word of its own: zeroed with the others)
      B.c_hi = ltrk::kXWg8NarrowMaxC;
      X.c_lo = ltrk::kXWg8NarrowMaxC + 1;
      // (the narrow launch IS the first-pass threshold kernel of 10-column strips, given the list: one strip width in the function --
      // a kernel of its own with an 8- and a 10-column body spilled inside its step loops, 5.7 GB of scratch writes and 30 ms per
      // config5hifi pass against 21 ms for the same pairs: profiles/r06/pmc_dispatch_config5hifi_*.txt)
      const int gn = std::max(1, std::min(ctx->full_wgt_grid[1][10], grid * 2));
      ltrk::launch_exact(c, sym, g, xs, X);
      ltrk::launch_wgt(8, 10, dim3((unsigned)gn), xs, B);
    } else {
      // kXLong walks the column blocks of reads beyond the eight-wave workgroups' 10241 bases through scratch strips and
      // may run beside the generic exact kernel (which does the same for non-ACGT pairs): a strip region of its own
      if (c == kXLong) X.scratch = plan->d_scratch + (size_t)plan->fan_lanes * plan->scratch_lane_stride;
      ltrk::launch_exact(c, sym, g, xs, X);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (xs != st) HIP_TRY(ctx, hipEventRecord(plan->ev_x[c], xs));
    x_launched[c] = true;
    LTR_DBG("launched exact kernel %d grid %d", c, grid);
    ++launches;
    return LTR_OK;
  };
  // level-2 timing: the multi-width launch class by class (the single-class kernels: same bodies).  Not under the plan kernel:
  // its classes score their failed certificates in line, and no exact launch is sized for what a single-class kernel would queue
  const bool use_plan = plan->use_plan && plan->plan_rep >= 0;      // (either model: the launch picks the instance of the parameters in force now)
  const bool split_multi = plan->timing >= 2 && !use_plan;
  const std::vector<int>& launch_order = split_multi ? plan->order2 : plan->order;
  auto is_plan = [&](int k) { return use_plan && k == plan->plan_rep; };
  auto is_multi = [&](int k) { return !use_plan && !split_multi && k == plan->multi_rep; };
  auto is_pmulti = [&](int k) { return !use_plan && !split_multi && k == plan->pmulti_rep; };
  std::vector<int> big, small;                                  // both longest reads first
  for (int k : launch_order) ((nl > nb && (is_plan(k) ? plan->plan_small : (is_multi(k) ? plan->multi_small : (is_pmulti(k) ? plan->pmulti_small : plan->bin_small[k])))) ? small : big).push_back(k);
  // Threshold first pass: which kernel scores a workgroup class, and which classes share a launch.  The threshold kernels carry
  // two quads of thresholds on top of the certificate body's registers: the wide four-wave strips (W = 15 .. 20, reads of 3586 ..
  // 5121 bases) would run at two waves per SIMD -- such a class goes to EIGHT waves with strips half as wide (8 / 10 columns: 128
  // registers, two workgroups = four waves per SIMD); the geometry follows the kernel.  Classes that are neighbours in the sorted
  // pair list and end up on eight waves with strips of up to LTR_WGT_LB4_MAXW columns (the wide four-wave classes and the eight-wave
  // classes W = 8 .. 10: a batch of 5-kb pairs that the certificate rules split into whole rounds of four-wave workgroups plus a
  // rest) are ONE launch of the widest of those kernels: one ramp, one tail, no two persistent launches fighting for the same
  // wave slots.  (Not under per-launch timing: every class keeps its own launch and its own time then.)
  int thr_nw[kNumFast] = {0}, thr_w[kNumFast] = {0}, thr_np[kNumFast] = {0};      // per class: its kernel; pairs of the launch it leads (0: led by a class before it)
  if (wg_thr) {
    int lead = -1;
    for (int k = kWg4First; k < kWg1First; ++k) {
      const int np = plan->bin_first[k + 1] - plan->bin_first[k];
      if (np <= 0) continue;
      const ClassInfo ci = class_info(k);
      int nw = ci.waves, w = ci.W;
      if (nw == 4 && w >= ltrp::kWg4WideMinW && ctx->dbg.wgt_keep_waves <= 0) { nw = 8; w = std::max((int)kWg8MinW, (w + 1) / 2); }
      w = ltrk::wgt_width(w);
      thr_nw[k] = nw; thr_w[k] = w; thr_np[k] = np;
      const bool narrow8 = (nw == 8 && w <= 10);                     // (lead: the narrow eight-wave class the current run of such classes began with)
      if (narrow8 && lead >= 0 && !plan->timing && ctx->dbg.wgt_keep_waves <= 0) {
        thr_np[lead] += np; thr_np[k] = 0;                           // (pairs of consecutive non-empty classes are consecutive in the sorted list)
        thr_w[lead] = std::max(thr_w[lead], w);
      } else lead = narrow8 ? k : -1;
    }
  }
  auto launch_class = [&](int k, int li) -> int {
    int np = plan->bin_first[k + 1] - plan->bin_first[k];
    if (wg_thr && thr_nw[k] != 0) { if (thr_np[k] == 0) return LTR_OK; np = thr_np[k]; }      // (scored by the launch of the class that leads its group)
    A.first_pair = plan->bin_first[k]; A.n_pairs = np; A.queue = plan->d_queue + k;
    const dim3 grid((unsigned)(is_plan(k) ? plan->plan_grid : (is_multi(k) ? plan->multi_grid : (is_pmulti(k) ? plan->pmulti_grid : plan->bin_grid[k]))));
    const ClassInfo ci = class_info(k);
    hipStream_t ls = lanes[li];
    A.scratch = plan->d_scratch + (size_t)li * plan->scratch_lane_stride;
    A.lp_shift = ci.lp_shift;
    if (ci.family == kFamPack) {
      // the ranges of the launch, widest segments first (their groups last longest)
      int nr = 0, groups = 0;
      for (int sft = kPackMaxShift; sft >= kPackMinShift; --sft) {
        const int k2 = ltrp::pack_class(sft, ci.W);
        const int c2 = plan->bin_first[k2 + 1] - plan->bin_first[k2];
        if (c2 <= 0) continue;
        const int per = 64 >> sft;
        groups += (c2 + per - 1) / per;
        A.pk_shift[nr] = sft; A.pk_first[nr] = plan->bin_first[k2]; A.pk_end[nr] = plan->bin_first[k2 + 1]; A.pk_grp_end[nr] = groups;
        ++nr;
      }
      for (; nr < 5; ++nr) { A.pk_shift[nr] = kPackMaxShift; A.pk_first[nr] = 0; A.pk_end[nr] = 0; A.pk_grp_end[nr] = groups; }
    }
    if (is_plan(k)) {
      A.pk_tabs = plan->d_pk_tabs; A.pk_ntabs = (int32_t)plan->pmulti_reps.size(); A.queue_base = plan->d_queue;
      A.pl_entries = plan->d_pl_entries; A.pl_n = (int32_t)plan->plan_entries.size();
      ltrk::launch_plan(sym, grid, ls, A);
    } else if (is_pmulti(k)) {
      A.pk_tabs = plan->d_pk_tabs; A.pk_ntabs = (int32_t)plan->pmulti_reps.size(); A.queue_base = plan->d_queue;
      ltrk::launch_pack_multi(sym, grid, ls, A);
    } else if (is_multi(k)) {
      A.mk_n = 0; A.queue_base = plan->d_queue;
      for (int k2 : plan->multi_classes) {                       // widest strips first
        A.mk_w[A.mk_n] = class_info(k2).W; A.mk_first[A.mk_n] = plan->bin_first[k2]; A.mk_np[A.mk_n] = plan->bin_first[k2 + 1] - plan->bin_first[k2];
        A.mk_class[A.mk_n] = k2;
        ++A.mk_n;
      }
      ltrk::launch_multi(sym, grid, ls, A);
    } else if (ci.family == kFamOne) ltrk::launch_onewave(ci.W, sym, grid, ls, A);
    else if (ci.family == kFamPack) ltrk::launch_pack(ci.W, sym, grid, ls, A);
    else if (wg_thr && thr_nw[k] != 0) {
      // (all of them on the plan's own stream, one after the other: two persistent eight-wave launches of different register
      // budgets side by side keep half-empty CUs from each other -- config5hifi, thresholds first: 21 + 12 ms alone, 41 ms side by side)
      const int gt = std::max(1, std::min(np, ctx->full_wgt_grid[thr_nw[k] == 8 ? 1 : 0][thr_w[k]]));
      ltrk::launch_wgt(thr_nw[k], thr_w[k], dim3((unsigned)gt), lanes[0], A);
    }
    else ltrk::launch_wg(ci.waves, ci.W, grid, ls, A);
    HIP_TRY(ctx, hipGetLastError());
    LTR_DBG("launched class %d grid %d pairs %d on lane %d", k, plan->bin_grid[k], np, li);
    ++launches;
    return LTR_OK;
  };
  // the small classes first, on their own lanes; per exact list an event on each of those lanes once nothing small still
  // to come can feed it
  bool small_closed[kNumExact] = {false};
  for (size_t p = 0; p < small.size(); ++p) {
    const int rc2 = launch_class(small[p], nb + (int)(p % (size_t)(nl - nb)));
    if (rc2 != LTR_OK) return rc2;
    if (x_fan && A.xlut) {
      const int next_cmax = (p + 1 < small.size()) ? plan->cls_cmax[small[p + 1]] : -1;
      for (int c = kNumExact - 1; c > kXShort; --c)
        if (!small_closed[c] && next_cmax < kListMinC[c]) {
          small_closed[c] = true;
          for (int k = nb; k < nl; ++k) HIP_TRY(ctx, hipEventRecord(plan->ev_close[c][k], lanes[k]));
        }
    }
  }
  for (size_t p = 0; p < big.size(); ++p) {
    // (round-robin; giving every launch to the stream with less work queued so far measured 0.4 % slower)
    const int rc2 = launch_class(big[p], (int)(p % (size_t)nb));
    if (rc2 != LTR_OK) return rc2;
    if (plan->timing) HIP_TRY(ctx, hipEventRecord(plan->bin_ev[++o], st));
    // exact lists nothing still to come can feed: the next class holds only shorter reads than the list takes
    if (x_fan && A.xlut) {
      const int next_cmax = (p + 1 < big.size()) ? plan->cls_cmax[big[p + 1]] : -1;
      for (int c = kNumExact - 1; c > kXShort; --c)
        if (!x_done[c] && next_cmax < kListMinC[c] && next_cmax >= 0) { const int rc3 = launch_exact_list(c, small_closed[c]); if (rc3 != LTR_OK) return rc3; }
    }
  }
  if (fan)
    for (int k = 1; k < nl; ++k) {
      HIP_TRY(ctx, hipEventRecord(plan->ev_join[k - 1], lanes[k]));
      HIP_TRY(ctx, hipStreamWaitEvent(st, plan->ev_join[k - 1], 0));
    }
  A.scratch = plan->d_scratch;
  // the lists still open (x_fan: the short reads' and the generic one; else all of them), in list order
  for (int c = 0; c < kNumExact; ++c) {
    if (!x_done[c]) { const int rc2 = launch_exact_list(c, false); if (rc2 != LTR_OK) return rc2; }
    if (plan->timing) HIP_TRY(ctx, hipEventRecord(plan->bin_ev[++o], st));
  }
  // ... and the plan's stream joins the side streams
  if (x_fan)
    for (int c = 0; c < kNumExact; ++c) if (exact_stream(c) != st && x_launched[c]) HIP_TRY(ctx, hipStreamWaitEvent(st, plan->ev_x[c], 0));
  HIP_TRY(ctx, hipEventRecord(plan->ev1, st));
  if (plan->uses_wg && ctx->dbg.wg_first_pass == 0) {
    // what this execute's workgroup classes met -> a pinned slot the next execute reads (no wait here, none there)
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ctx->wg_stat_pin && hipHostMalloc((void**)&ctx->wg_stat_pin, ltr_ctx::kWgStatSlots * 2 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { ctx->wg_stat_pin = nullptr; (void)hipGetLastError(); }
    if (ctx->wg_stat_pin)
      for (int i = 0; i < ltr_ctx::kWgStatSlots; ++i) {
        ltr_ctx::WgStatSlot& sl = ctx->wg_stat[i];
        if (sl.busy) continue;
        if (!sl.ev && hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) { sl.ev = nullptr; (void)hipGetLastError(); break; }
        if (hipMemcpyAsync(ctx->wg_stat_pin + 2 * i, plan->d_redo_count + kWgStatOff, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st) == hipSuccess &&
            hipEventRecord(sl.ev, st) == hipSuccess) { sl.busy = true; sl.mode = wg_thr ? 1 : 0; sl.epoch = ctx->wg_epoch; }
        else (void)hipGetLastError();
        break;
      }
  }
  plan->last_wg_thr = wg_thr;
  plan->last_out = out; plan->last_stream = st; plan->last_launches = launches; plan->executed = true;
  if (std::find(plan->streams.begin(), plan->streams.end(), st) == plan->streams.end()) plan->streams.push_back(st);
  plan->timed = (use_plan && plan->timing >= 2) ? 1 : plan->timing;     // (the plan kernel is never split: its launches were timed as launched)
  plan->kernel_ms_counted = false;
  return LTR_OK;
}

int ltr_plan_fetch(ltr_plan* plan, double* out_ll, int32_t* out_seed) {
  if (!plan || !plan->executed) return LTR_ERR_INVALID;
  ltr_ctx* ctx = plan->ctx;
  if (!ctx) return LTR_ERR_INVALID;                          // the context was destroyed before this plan
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  LTR_DBG("fetch: waiting");
  HIP_TRY(ctx, hipEventSynchronize(plan->ev1));               // this plan's last execute (work queued behind it on the stream keeps running)
  LTR_DBG("fetch: plan done");
  if (!plan->kernel_ms_counted) {                              // device time of this execute's DP kernels -> the context's timers
    float t = 0.f;
    if (hipEventElapsedTime(&t, plan->ev0, plan->ev1) == hipSuccess) ltr::add_time(ctx, -1, 0.0, (double)t);
    { std::lock_guard<std::mutex> lk(ctx->err_mu); ctx->tm.dp_cells += ltr_plan_cells(plan); ctx->tm.dp_pairs += ltr_plan_num_pairs(plan); }
    plan->kernel_ms_counted = true;
  }
  if (out_ll && plan->ll_size > 0 && plan->h_ll && plan->last_out == plan->d_ll) {
    // a compact plan: the kernel wrote into pinned host memory, visible now that ev1 has passed
    std::memcpy(out_ll, plan->h_ll, (size_t)plan->ll_size * sizeof(double));
  } else if (out_ll && plan->ll_size > 0) {
    // Through a pinned staging block on a copy stream of its own: hipMemcpy into pageable memory is done by a copy KERNEL,
    // and behind the persistent DP launches of later plans it waited for wave slots -- measured on MI355X, the three chunks of
    // a 30 000-locus ltr_calc_hap_aln_probs call: the 2 MB of chunk 0 arrived 8 ms after its plan had finished, when chunks
    // 1 and 2 were through as well.  A pinned destination goes over the DMA engines.
    std::lock_guard<std::mutex> lk(ctx->pin_mu);
    const size_t total = (size_t)plan->ll_size * sizeof(double);
    const size_t block = std::min<size_t>(total, (size_t)8 << 20);      // (8 MB pieces: a 10 000-locus plan's 12 MB already takes two)
    if (ctx->pin_bytes < block) {
      if (ctx->pin) (void)hipHostFree(ctx->pin);
      ctx->pin = nullptr; ctx->pin_bytes = 0;
      HIP_TRY(ctx, hipHostMalloc(&ctx->pin, block, hipHostMallocDefault));
      ctx->pin_bytes = block;
    }
    if (!ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (size_t at = 0; at < total; at += block) {
      const size_t nb = std::min(block, total - at);
      HIP_TRY(ctx, hipMemcpyAsync(ctx->pin, (const char*)plan->last_out + at, nb, hipMemcpyDeviceToHost, ctx->copy_stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
      std::memcpy((char*)out_ll + at, ctx->pin, nb);
    }
  }
  if (out_seed)
    for (int64_t r = 0; r < plan->n_reads; ++r) if (plan->seed[(size_t)r] >= 0) out_seed[r] = plan->seed[(size_t)r];
  return LTR_OK;
}

int ltr_plan_last_kernel_ms(ltr_plan* plan, float* ms, int* n_launches) {
  if (!plan || !plan->executed) return LTR_ERR_INVALID;
  ltr_ctx* ctx = plan->ctx;
  if (!ctx) return LTR_ERR_INVALID;                          // the context was destroyed before this plan
  HIP_TRY(ctx, hipEventSynchronize(plan->ev1));
  float t = 0.f;
  HIP_TRY(ctx, hipEventElapsedTime(&t, plan->ev0, plan->ev1));
  if (ms) *ms = t;
  if (n_launches) *n_launches = plan->last_launches;
  return LTR_OK;
}

int ltr_plan_set_timing(ltr_plan* plan, int on) {
  if (!plan) return LTR_ERR_INVALID;
  if (!plan->ctx) return LTR_ERR_INVALID;                     // the context was destroyed before this plan
  if (on && !plan->bin_ev[0]) {
    ltr_ctx* ctx = plan->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k <= kNumKernels; ++k) HIP_TRY(ctx, hipEventCreate(&plan->bin_ev[k]));
  }
  plan->timing = on <= 0 ? 0 : (on >= 2 ? 2 : 1);
  return LTR_OK;
}

int ltr_plan_kernel_stats(ltr_plan* plan, int k, int* strip_width, int64_t* n_pairs, double* cells, float* ms) {
  if (!plan || k < 0 || k >= kNumKernels) return LTR_ERR_INVALID;
  ltr_ctx* ctx = plan->ctx;
  if (!ctx) return LTR_ERR_INVALID;                          // the context was destroyed before this plan
  const bool redo = (k >= kNumFast);
  const int xc = k - kNumFast;
  static const int kXW[kNumExact] = {kExactW, kXShortW, kXMidW, kXLongW, 0, 0};     // (workgroup exact kernels pick the width per pair)
  if (strip_width) *strip_width = redo ? kXW[xc] : class_info(k).W;
  // a packed launch scores every lanes-per-pair block of its strip width: its pairs, cells and time are reported under
  // its representative class (ltr_plan_kernel_ranges names the blocks), the other classes of the width report nothing
  const bool pack = !redo && k >= kPackFirst && k < kWg4First;
  double cl = redo ? plan->x_cells[xc] : plan->bin_cells[k];
  int64_t np = redo ? 0 : plan->bin_first[k + 1] - plan->bin_first[k];
  const bool plan_member = !redo && plan->use_plan && plan->plan_rep >= 0 && k < kWg4First;
  if (plan_member) {
    // the plan kernel: every one-wave class and packed width in one launch, reported under plan_rep
    cl = 0.0; np = 0;
    if (k == plan->plan_rep) for (int k2 = 0; k2 < kWg4First; ++k2) { cl += plan->bin_cells[k2]; np += plan->bin_first[k2 + 1] - plan->bin_first[k2]; }
  }
  const bool multi_member = !plan_member && !redo && plan->multi_rep >= 0 && plan->timed < 2 && k < kNumBins && k >= kMultiMinW - 1;
  if (multi_member) {
    // ... and so does the multi-width one-wave launch (unless the last execute ran it class by class: timing level 2)
    cl = 0.0; np = 0;
    if (k == plan->multi_rep) for (int k2 : plan->multi_classes) { cl += plan->bin_cells[k2]; np += plan->bin_first[k2 + 1] - plan->bin_first[k2]; }
  }
  if (pack && !plan_member) {
    cl = 0.0; np = 0;
    auto add_width = [&](int w) {
      for (int sft = kPackMinShift; sft <= kPackMaxShift; ++sft) {
        const int k2 = ltrp::pack_class(sft, w);
        cl += plan->bin_cells[k2]; np += plan->bin_first[k2 + 1] - plan->bin_first[k2];
      }
    };
    const bool in_pm = plan->pmulti_rep >= 0 && plan->timed < 2 && class_info(k).W >= kPackMultiMinW;
    if (in_pm) { if (k == plan->pmulti_rep) for (int rep : plan->pmulti_reps) add_width(class_info(rep).W); }
    else if (plan->pack_rep[k - kPackFirst] == k) add_width(class_info(k).W);
  }
  if (cells) *cells = cl;
  if (n_pairs) {
    *n_pairs = np;
    if (redo && plan->executed) {                      // pairs the certificates could not clear (+ the non-ACGT ones, generic list)
      uint32_t c[kInlineCountOff + kNumExact] = {0};
      HIP_TRY(ctx, hipStreamSynchronize(plan->last_stream));
      HIP_TRY(ctx, hipMemcpy(c, plan->d_redo_count, sizeof(c), hipMemcpyDeviceToHost));
      *n_pairs = (int64_t)c[xc] + c[kInlineCountOff + xc];      // its list + what the plan kernel scored in line for it
    }
  }
  if (ms) {
    *ms = 0.f;
    if (plan->executed && plan->timed) {
      // launch number o ran between bin_ev[o] and bin_ev[o+1] (see ltr_plan_execute)
      const int o = plan->timed >= 2 ? plan->order_pos2[k] : plan->order_pos[k];
      if (o >= 0) {
        HIP_TRY(ctx, hipEventSynchronize(plan->bin_ev[o + 1]));
        HIP_TRY(ctx, hipEventElapsedTime(ms, plan->bin_ev[o], plan->bin_ev[o + 1]));
      }
    }
  }
  return LTR_OK;
}

int ltr_plan_kernel_ranges(ltr_plan* plan, int k, int32_t* lanes_per_pair, int32_t* strip_width, int64_t* n_pairs) {
  if (!plan || k < 0 || k >= kNumKernels) return LTR_ERR_INVALID;
  int nr = 0;
  if (plan->use_plan && plan->plan_rep >= 0) {
    if (k >= kWg4First || k != plan->plan_rep) return 0;
    // (at most kNumBins + kNumPack ranges: the caller's arrays hold ltr_num_kernels() entries)
    for (int k2 : plan->multi_classes) {
      if (lanes_per_pair) lanes_per_pair[nr] = 64;
      if (strip_width) strip_width[nr] = class_info(k2).W;
      if (n_pairs) n_pairs[nr] = plan->bin_first[k2 + 1] - plan->bin_first[k2];
      ++nr;
    }
    for (int rep : plan->pmulti_reps)
      for (int sft = kPackMaxShift; sft >= kPackMinShift; --sft) {
        const int k2 = ltrp::pack_class(sft, class_info(rep).W);
        const int c2 = plan->bin_first[k2 + 1] - plan->bin_first[k2];
        if (c2 <= 0) continue;
        if (lanes_per_pair) lanes_per_pair[nr] = 1 << sft;
        if (strip_width) strip_width[nr] = class_info(rep).W;
        if (n_pairs) n_pairs[nr] = c2;
        ++nr;
      }
    return nr;
  }
  if (k == plan->multi_rep && plan->timed < 2) {
    for (int k2 : plan->multi_classes) {
      if (lanes_per_pair) lanes_per_pair[nr] = 64;
      if (strip_width) strip_width[nr] = class_info(k2).W;
      if (n_pairs) n_pairs[nr] = plan->bin_first[k2 + 1] - plan->bin_first[k2];
      ++nr;
    }
    return nr;
  }
  if (!(k >= kPackFirst && k < kWg4First) || plan->pack_rep[k - kPackFirst] != k) return 0;
  auto width = [&](int w) {
    for (int sft = kPackMaxShift; sft >= kPackMinShift; --sft) {
      const int k2 = ltrp::pack_class(sft, w);
      const int c2 = plan->bin_first[k2 + 1] - plan->bin_first[k2];
      if (c2 <= 0) continue;
      if (lanes_per_pair) lanes_per_pair[nr] = 1 << sft;
      if (strip_width) strip_width[nr] = w;
      if (n_pairs) n_pairs[nr] = c2;
      ++nr;
    }
  };
  const bool in_pm = plan->pmulti_rep >= 0 && plan->timed < 2 && class_info(k).W >= kPackMultiMinW;
  if (in_pm) { if (k == plan->pmulti_rep) for (int rep : plan->pmulti_reps) width(class_info(rep).W); }
  else width(class_info(k).W);
  return nr;
}

int ltr_plan_kernel_class(const ltr_plan* plan) { return (plan && plan->use_plan) ? plan->plan_rep : -1; }

int ltr_plan_debug_entries(const ltr_plan* plan, int32_t* kind, int32_t* strip_width, int64_t* n_pairs, double* cells, int cap) {
  if (!plan || cap < 0) return LTR_ERR_INVALID;
  const int n = (int)plan->plan_entries.size();
  for (int i = 0; i < std::min(n, cap); ++i) {
    const PlanEntry& e = plan->plan_entries[(size_t)i];
    int64_t np = e.n_pairs; double cl = 0.0;
    if (e.kind == 0 || e.kind == 3) cl = plan->bin_cells[e.queue_class];
    else if (e.kind == 1) {
      np = 0;
      for (int sft = kPackMinShift; sft <= kPackMaxShift; ++sft) { const int k2 = ltrp::pack_class(sft, e.W); cl += plan->bin_cells[k2]; np += plan->bin_first[k2 + 1] - plan->bin_first[k2]; }
    } else cl = plan->x_cells[e.queue_class - ltrp::kStartQueueSlot];
    if (kind) kind[i] = e.kind;
    if (strip_width) strip_width[i] = e.W;
    if (n_pairs) n_pairs[i] = np;
    if (cells) cells[i] = cl;
  }
  return n;
}

int ltr_plan_debug_wave_clocks(ltr_plan* plan, uint64_t* out, int64_t cap) {
  if (!plan || !plan->ctx || !out || cap < 0) return LTR_ERR_INVALID;
  if (!plan->d_wave_clock || !plan->executed) return 0;
  ltr_ctx* ctx = plan->ctx;
  const int64_t n = (int64_t)plan->plan_grid * kBlockWaves;
  if (cap < 4 * n + 4096 + 256) return LTR_ERR_INVALID;
  HIP_TRY(ctx, hipStreamSynchronize(plan->last_stream));
  HIP_TRY(ctx, hipMemcpy(out, plan->d_wave_clock, ((size_t)n * 4 + 4096 + 256) * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemset(plan->d_wave_clock + 4 * n, 0, sizeof(uint64_t)));      // (the log's counter, for the next execute)
  HIP_TRY(ctx, hipMemset(plan->d_wave_clock + 4 * n + 4096, 0, 256 * sizeof(uint64_t)));   // (... and the per-entry sums)
  return (int)n;
}

int ltr_align_batch(ltr_ctx* ctx, const ltr_locus_batch* batch, double* out_ll, int32_t* out_seed) {
  if (!ctx || !batch || !out_ll) return LTR_ERR_INVALID;
  ltr::TimedCall timed(ctx, ltr::kTimerHapAln);
  LTR_GUARD_BEGIN
  ltr_plan* plan = nullptr;
  int rc = ltr_plan_create(ctx, batch, &plan);
  if (rc != LTR_OK) return rc;
  // Masked cells must stay untouched (reference HapAligner.cpp:557-560, :841-845): results go
  // through a staging copy and only the computed entries are scattered into the caller's buffer.
  std::vector<double> tmp((size_t)std::max<int64_t>(plan->ll_size, 1));
  rc = ltr_plan_execute(plan, nullptr, nullptr);
  if (rc == LTR_OK) rc = ltr_plan_fetch(plan, tmp.data(), out_seed);
  if (rc == LTR_OK) {
    if (!batch->realign_read && !batch->realign_hap) {
      std::memcpy(out_ll, tmp.data(), (size_t)plan->ll_size * sizeof(double));
    } else {
      int64_t off = 0;
      for (int64_t l = 0; l < batch->n_loci; ++l) {
        const int64_t r0 = batch->locus_read_off[l], r1 = batch->locus_read_off[l + 1];
        const int64_t h0 = batch->locus_hap_off[l], h1 = batch->locus_hap_off[l + 1];
        const int64_t H = h1 - h0;
        for (int64_t r = r0; r < r1; ++r) {
          if (batch->realign_read && !batch->realign_read[r]) continue;
          for (int64_t h = h0; h < h1; ++h) {
            if (batch->realign_hap && !batch->realign_hap[h]) continue;
            const int64_t k = off + (r - r0) * H + (h - h0);
            out_ll[k] = tmp[(size_t)k];
          }
        }
        off += (r1 - r0) * H;
      }
    }
  }
  ltr_plan_destroy(plan);
  return rc;
  LTR_GUARD_END(ctx)
}

int ltr_ctx_timers(ltr_ctx* ctx, ltr_timers* out, int reset) {
  if (!ctx || !out) return LTR_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->err_mu);
  *out = ctx->tm;
  if (reset) ctx->tm = ltr_timers{};
  return LTR_OK;
}

int ltr_ctx_short_kernel_split(ltr_ctx* ctx, double out_ms[4], int reset) {
  if (!ctx || !out_ms) return LTR_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->err_mu);
  for (int k = 0; k < 4; ++k) { out_ms[k] = ctx->short_split_ms[k]; if (reset) ctx->short_split_ms[k] = 0.0; }
  return LTR_OK;
}

int ltr_ctx_timers_n(ltr_ctx* ctx, void* out, size_t out_bytes, int reset) {
  if (!ctx || !out) return LTR_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->err_mu);
  std::memcpy(out, &ctx->tm, std::min(out_bytes, sizeof(ltr_timers)));
  if (reset) ctx->tm = ltr_timers{};
  return LTR_OK;
}

// Genotyper::calc_log_sample_posteriors + get_optimal_haplotypes (genotyper.cpp:21-100)
int ltr_posteriors(ltr_ctx* ctx, int32_t S, int32_t R, int32_t H,
                   double* ll, const double* lp1, const double* lp2, const int32_t* sample_label,
                   int32_t haploid, double* post, double* stl, int32_t* gts, double* total_ll) {
  if (!ctx || S <= 0 || R < 0 || H <= 0 || !ll || !lp1 || !lp2 || !sample_label || !post || !stl) return LTR_ERR_INVALID;
  ltr::TimedCall timed(ctx, ltr::kTimerPosterior);             // total_posterior_time_, genotyper.cpp:46,:80-81
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (int32_t r = 0; r < R; ++r) if (sample_label[r] < 0 || sample_label[r] >= S) { ltr::set_error(ctx, "sample label out of range"); return LTR_ERR_INVALID; }
  // int_log(v) == log(v) (mathops.cpp:14-22); priors of genotyper.cpp:21-33
  const double lH = std::log((double)H), lH1 = std::log((double)(H + 1));
  const double homoz = haploid ? -lH : std::log(2.0) - lH - lH1;
  const double hetz = haploid ? -1.7976931348623157e308 / 2 : -lH - lH1;
  const size_t nll = (size_t)R * H, npost = (size_t)S * H * H;
  double *d_ll = nullptr, *d_p1 = nullptr, *d_p2 = nullptr, *d_post = nullptr, *d_stl = nullptr;
  int *d_lab = nullptr, *d_gts = nullptr;
  int rc = LTR_OK;
  hipStream_t st = ctx->stream;
#define P_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ltr::set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_)); rc = LTR_ERR_HIP; goto done; } } while (0)
  P_TRY(ctx->pool.alloc((void**)&d_ll, std::max<size_t>(nll, 1) * 8));
  P_TRY(ctx->pool.alloc((void**)&d_p1, std::max<size_t>(R, 1) * 8));
  P_TRY(ctx->pool.alloc((void**)&d_p2, std::max<size_t>(R, 1) * 8));
  P_TRY(ctx->pool.alloc((void**)&d_lab, std::max<size_t>(R, 1) * 4));
  P_TRY(ctx->pool.alloc((void**)&d_post, npost * 8));
  P_TRY(ctx->pool.alloc((void**)&d_stl, (size_t)S * 8));
  P_TRY(ctx->pool.alloc((void**)&d_gts, (size_t)S * 8));
  if (R > 0) {
    P_TRY(hipMemcpyAsync(d_ll, ll, nll * 8, hipMemcpyHostToDevice, st));
    P_TRY(hipMemcpyAsync(d_p1, lp1, (size_t)R * 8, hipMemcpyHostToDevice, st));
    P_TRY(hipMemcpyAsync(d_p2, lp2, (size_t)R * 8, hipMemcpyHostToDevice, st));
    P_TRY(hipMemcpyAsync(d_lab, sample_label, (size_t)R * 4, hipMemcpyHostToDevice, st));
  }
  hipLaunchKernelGGL(ltr_posterior_kernel, dim3((unsigned)S), dim3(256), 0, st, S, R, H, d_ll, d_p1, d_p2, d_lab, homoz, hetz, d_post);
  if (nll) hipLaunchKernelGGL(ltr_clamp_kernel, dim3((unsigned)((nll + 255) / 256)), dim3(256), 0, st, d_ll, (int64_t)nll);
  hipLaunchKernelGGL(ltr_posterior_finish_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, S, H, d_post, d_stl, d_gts);
  P_TRY(hipGetLastError());
  if (nll) P_TRY(hipMemcpyAsync(ll, d_ll, nll * 8, hipMemcpyDeviceToHost, st));
  P_TRY(hipMemcpyAsync(post, d_post, npost * 8, hipMemcpyDeviceToHost, st));
  P_TRY(hipMemcpyAsync(stl, d_stl, (size_t)S * 8, hipMemcpyDeviceToHost, st));
  {
    std::vector<int32_t> g((size_t)2 * S);
    P_TRY(hipMemcpyAsync(g.data(), d_gts, (size_t)S * 8, hipMemcpyDeviceToHost, st));
    P_TRY(hipStreamSynchronize(st));
    if (gts) std::memcpy(gts, g.data(), (size_t)S * 8);
  }
  if (total_ll) { double t = 0.0; for (int32_t s = 0; s < S; ++s) t += stl[s]; *total_ll = t; }   // sum(), genotyper.cpp:78
done:
#undef P_TRY
  if (rc != LTR_OK) (void)hipStreamSynchronize(st);            // (buffers go back to the context's pool: nothing may still use them)
  ctx->pool.release(d_ll);
  ctx->pool.release(d_p1);
  ctx->pool.release(d_p2);
  ctx->pool.release(d_lab);
  ctx->pool.release(d_post);
  ctx->pool.release(d_stl);
  ctx->pool.release(d_gts);
  return rc;
}

// Genotyper::calc_log_sample_posteriors + get_optimal_haplotypes for EVERY locus of a resident
// plan, straight from the LL buffer of the last execute (no host round trip of the LL matrix).
int ltr_plan_posteriors(ltr_plan* plan, const ltr_posterior_batch* pb, double* post, double* sample_total_ll, int32_t* gts) {
  if (!plan || !pb || !post || !sample_total_ll) return LTR_ERR_INVALID;
  ltr_ctx* ctx = plan->ctx;
  if (!ctx) return LTR_ERR_INVALID;                          // the context was destroyed before this plan
  if (!plan->executed) { ltr::set_error(ctx, "ltr_plan_posteriors: execute the plan first"); return LTR_ERR_INVALID; }
  if (pb->n_loci != (int64_t)plan->locus_P.size()) { ltr::set_error(ctx, "posterior batch and plan disagree on the number of loci"); return LTR_ERR_INVALID; }
  ltr::TimedCall timed(ctx, ltr::kTimerPosterior);             // total_posterior_time_, genotyper.cpp:46,:80-81
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<PostUnit> units;
  int64_t post_off = 0;
  for (int64_t l = 0; l < pb->n_loci; ++l) {
    const int64_t r0 = pb->locus_read_off[l], r1 = pb->locus_read_off[l + 1];
    const int32_t H = plan->locus_H[(size_t)l], P = plan->locus_P[(size_t)l], S = pb->n_samples[l];
    if (r0 < 0 || r1 < r0 || r1 > pb->n_reads || S < 0) { ltr::set_error(ctx, "bad posterior batch offsets"); return LTR_ERR_INVALID; }
    for (int64_t r = r0; r < r1; ++r)
      if (pb->pool_index[r] < 0 || pb->pool_index[r] >= P || pb->sample_label[r] < 0 || pb->sample_label[r] >= S) {
        ltr::set_error(ctx, "pool index / sample label out of range"); return LTR_ERR_INVALID;
      }
    // int_log(v) == log(v) (mathops.cpp:14-22); priors of genotyper.cpp:21-33
    const double lH = std::log((double)H), lH1 = std::log((double)(H + 1));
    for (int32_t sm = 0; sm < S; ++sm) {
      PostUnit u;
      u.ll_off = plan->locus_ll_off[(size_t)l]; u.post_off = post_off; u.r0 = (int32_t)r0; u.r1 = (int32_t)r1; u.H = H; u.sample = sm;
      u.homoz = pb->haploid ? -lH : std::log(2.0) - lH - lH1;
      u.hetz = pb->haploid ? -1.7976931348623157e308 / 2 : -lH - lH1;
      units.push_back(u);
      post_off += (int64_t)H * H;
    }
  }
  const size_t nu = units.size(), nr = (size_t)pb->n_reads;
  if (nu == 0) return LTR_OK;
  PostUnit* d_units = nullptr; int32_t *d_pool = nullptr, *d_lab = nullptr; int* d_gts = nullptr;
  double *d_p1 = nullptr, *d_p2 = nullptr, *d_post = nullptr, *d_stl = nullptr;
  int rc = LTR_OK;
  hipStream_t st = plan->last_stream;
#define P_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { ltr::set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_)); rc = LTR_ERR_HIP; goto done; } } while (0)
  P_TRY(ctx->pool.alloc((void**)&d_units, nu * sizeof(PostUnit)));
  P_TRY(ctx->pool.alloc((void**)&d_pool, std::max<size_t>(nr, 1) * 4));
  P_TRY(ctx->pool.alloc((void**)&d_lab, std::max<size_t>(nr, 1) * 4));
  P_TRY(ctx->pool.alloc((void**)&d_p1, std::max<size_t>(nr, 1) * 8));
  P_TRY(ctx->pool.alloc((void**)&d_p2, std::max<size_t>(nr, 1) * 8));
  P_TRY(ctx->pool.alloc((void**)&d_post, (size_t)post_off * 8));
  P_TRY(ctx->pool.alloc((void**)&d_stl, nu * 8));
  P_TRY(ctx->pool.alloc((void**)&d_gts, nu * 8));
  P_TRY(hipMemcpyAsync(d_units, units.data(), nu * sizeof(PostUnit), hipMemcpyHostToDevice, st));
  if (nr) {
    P_TRY(hipMemcpyAsync(d_pool, pb->pool_index, nr * 4, hipMemcpyHostToDevice, st));
    P_TRY(hipMemcpyAsync(d_lab, pb->sample_label, nr * 4, hipMemcpyHostToDevice, st));
    P_TRY(hipMemcpyAsync(d_p1, pb->log_p1, nr * 8, hipMemcpyHostToDevice, st));
    P_TRY(hipMemcpyAsync(d_p2, pb->log_p2, nr * 8, hipMemcpyHostToDevice, st));
  }
  hipLaunchKernelGGL(ltr_posterior_batch_kernel, dim3((unsigned)nu), dim3(128), 0, st, d_units, plan->last_out, d_pool, d_p1, d_p2, d_lab, d_post);
  hipLaunchKernelGGL(ltr_posterior_batch_finish_kernel, dim3((unsigned)((nu + 63) / 64)), dim3(64), 0, st, (int)nu, d_units, d_post, d_stl, d_gts);
  P_TRY(hipGetLastError());
  P_TRY(hipMemcpyAsync(post, d_post, (size_t)post_off * 8, hipMemcpyDeviceToHost, st));
  P_TRY(hipMemcpyAsync(sample_total_ll, d_stl, nu * 8, hipMemcpyDeviceToHost, st));
  {
    std::vector<int32_t> g(2 * nu);
    P_TRY(hipMemcpyAsync(g.data(), d_gts, nu * 8, hipMemcpyDeviceToHost, st));
    P_TRY(hipStreamSynchronize(st));
    if (gts) std::memcpy(gts, g.data(), nu * 8);
  }
done:
#undef P_TRY
  if (rc != LTR_OK) (void)hipStreamSynchronize(st);
  ctx->pool.release(d_units);
  ctx->pool.release(d_pool);
  ctx->pool.release(d_lab);
  ctx->pool.release(d_p1);
  ctx->pool.release(d_p2);
  ctx->pool.release(d_post);
  ctx->pool.release(d_stl);
  ctx->pool.release(d_gts);
  return rc;
}

}  // extern "C"

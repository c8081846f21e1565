// tests/host_sanitize/harness.cpp -- TEST INFRASTRUCTURE.
// Fuzzes the host-only entry points of the C-ABI library (ltr_host.cpp, ltr_genotype.cpp) under
// AddressSanitizer + UBSan on the CPU: trimming, haplotype enumeration, pooling, scatter, genotype
// fields, and ltr_process_reads' host half (trim + haplotype strings) up to the point where it
// would hand the batch to the GPU.  The four library-internal symbols those files need from the HIP
// translation units are defined here (a context that holds parameters, a batch scorer that
// reports "no device").  Built and run by tests/test_host_sanitizers.py.
#include <cmath>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include "../../longtr_amd/csrc/ltr_internal.h"

struct ltr_ctx { ltr_align_params p; std::string err; std::vector<uint8_t> host_bytes[4]; std::mutex call_mu, err_mu; ltr::DebugKnobs knobs; bool fake_device = false; };
struct ltr_plan { int64_t ll_size = 0, pairs = 0; };               // (the fake device's plan: sizes only)
namespace ltr {
void set_error(ltr_ctx* ctx, const std::string& msg) { if (ctx) { std::lock_guard<std::mutex> lk(ctx->err_mu); ctx->err = msg; } }
ltr_align_params ctx_params(const ltr_ctx* ctx) { return ctx->p; }
DebugKnobs ctx_debug(const ltr_ctx* ctx) { return ctx->knobs; }
void add_time(ltr_ctx*, int, double, double) {}
void* ctx_side_stream(const ltr_ctx*, int) { return nullptr; }
std::unique_lock<std::mutex> ctx_call_lock(ltr_ctx* ctx) { return std::unique_lock<std::mutex>(ctx->call_mu); }
uint8_t* ctx_host_bytes(ltr_ctx* ctx, int which, size_t bytes) { ctx->host_bytes[which & 3].resize(bytes); return ctx->host_bytes[which & 3].data(); }
int process_reads_short(ltr_ctx*, const ltr_haplotype_blocks*, const uint8_t*, const ltr_alignment*, int32_t, int32_t,
                        const uint8_t*, double*, int32_t*) { return LTR_ERR_NO_DEVICE; }
struct ShortBatch { int n = 0; };
ShortBatch* short_batch_new() { return new ShortBatch(); }
void short_batch_free(ShortBatch* b) { delete b; }
int short_batch_add(ltr_ctx*, ShortBatch* b, const ltr_haplotype_blocks*, const uint8_t*, const ltr_alignment*, int32_t, int32_t,
                    const uint8_t*, double*, int32_t*) { b->n++; return LTR_OK; }
int short_batch_merge(ltr_ctx*, ShortBatch* dst, ShortBatch* src) { dst->n += src->n; src->n = 0; return LTR_OK; }
int short_batch_run(ltr_ctx*, ShortBatch*) { return LTR_ERR_NO_DEVICE; }
}
static std::atomic<long> g_batches(0), g_pairs(0);                 // (the stub scorer is called from two threads at once below)
// A context with fake_device set gets plans that "run": every row the library asks for comes back as its own index, so that the
// whole chunk pipeline of ltr_calc_hap_aln_probs (staging ahead on a thread of its own, fetch, fan-out) runs under the sanitizers.
extern "C" int ltr_plan_execute(ltr_plan* p, double*, void*) { return p ? LTR_OK : LTR_ERR_NO_DEVICE; }
extern "C" int ltr_plan_fetch(ltr_plan* p, double* ll, int32_t*) {
  if (!p) return LTR_ERR_NO_DEVICE;
  for (int64_t i = 0; i < p->ll_size; ++i) ll[i] = -(double)(i % 1000) - 1.0;
  return LTR_OK;
}
extern "C" int64_t ltr_plan_ll_size(const ltr_plan* p) { return p ? p->ll_size : 0; }
extern "C" int64_t ltr_plan_num_pairs(const ltr_plan* p) { return p ? p->pairs : 0; }
extern "C" void ltr_plan_destroy(ltr_plan* p) { delete p; }
static int touch_batch(const ltr_locus_batch* b);
extern "C" int ltr_plan_create(ltr_ctx* ctx, const ltr_locus_batch* b, ltr_plan** out) {
  *out = nullptr;
  const int rc = touch_batch(b);
  if (!ctx->fake_device) return rc;
  ltr_plan* p = new ltr_plan();
  for (int64_t l = 0; l < b->n_loci; ++l) p->ll_size += (b->locus_read_off[l + 1] - b->locus_read_off[l]) * (b->locus_hap_off[l + 1] - b->locus_hap_off[l]);
  p->pairs = p->ll_size;
  *out = p;
  return LTR_OK;
}
extern "C" int ltr_align_batch(ltr_ctx*, const ltr_locus_batch* b, double*, int32_t*) { return touch_batch(b); }
static int touch_batch(const ltr_locus_batch* b) {
  // touch every byte the library handed over: ASan checks the extents
  long sum = 0;
  for (int64_t r = 0; r < b->n_reads; ++r)
    for (int64_t k = b->read_off[r]; k < b->read_off[r + 1]; ++k) sum += b->read_bytes[k];
  for (int64_t h = 0; h < b->n_haps; ++h)
    for (int64_t k = b->hap_off[h]; k < b->hap_off[h + 1]; ++k) sum += b->hap_bytes[k];
  g_batches++; g_pairs += (long)(b->n_reads * b->n_haps) + (sum & 1);
  return LTR_ERR_NO_DEVICE;
}

static std::mt19937_64 rng(20250225);
static int ri(int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); }
static std::string rseq(int n) { std::string s((size_t)n, 'A'); for (auto& c : s) c = "ACGT"[rng() & 3]; return s; }

int main(int argc, char** argv) {
  long checks = 0;
  // ---- trim_alignment: random CIGARs (valid and corrupt) ---------------------------------------
  for (int it = 0; it < 20000; ++it) {
    const int nops = ri(0, 8);
    std::string types; std::vector<int32_t> nums; int qlen = 0, rlen = 0;
    for (int k = 0; k < nops; ++k) {
      const char t = (it % 17 == 0 && k == 1) ? 'Q' : "M=XIDSH"[ri(0, 6)];
      const int n = ri(it % 13 == 0 ? 0 : 1, 40);
      types += t; nums.push_back(n);
      if (t == 'M' || t == '=' || t == 'X') { qlen += n; rlen += n; } else if (t == 'I' || t == 'S') qlen += n; else if (t == 'D') rlen += n;
    }
    if (it % 11 == 0) qlen = std::max(0, qlen + ri(-3, 3));           // sequence / CIGAR length mismatch
    const std::string seq = rseq(qlen);
    const int start = ri(0, 2000);
    ltr_alignment a = {start, start + rlen - 1, (const uint8_t*)seq.data(), (int32_t)seq.size(), (int32_t)nums.size(),
                       types.data(), nums.data(), nullptr};
    int32_t lt = -1, rt = -1;
    const int rs = start + ri(-20, rlen + 20), re = rs + ri(0, 60);
    const int rc = ltr_trim_alignment(&a, rs, re, ri(0, 35), &lt, &rt);
    if (rc == LTR_OK && (lt < 0 || rt < 0 || lt + rt > (int)seq.size() + 1)) { std::printf("trim out of range\n"); return 1; }
    checks++;
  }
  // ---- haplotype enumeration + process_reads host half ----------------------------------------
  for (int it = 0; it < 600; ++it) {
    const int nb = ri(1, 4);
    std::vector<int32_t> bs, be, per, na; std::vector<uint8_t> rep, bytes; std::vector<int64_t> off(1, 0);
    int pos = ri(100, 1000);
    std::vector<std::string> keep;
    for (int b = 0; b < nb; ++b) {
      const bool is_rep = (b % 2 == 1);
      const int len = ri(is_rep ? 0 : 5, 60), nall = is_rep ? ri(1, 5) : 1;
      bs.push_back(pos); be.push_back(pos + len); pos += len;
      rep.push_back(is_rep); per.push_back(is_rep ? ri(1, 6) : 0); na.push_back(nall);
      for (int k = 0; k < nall; ++k) { const std::string s = rseq(k == 0 ? len : ri(0, 80)); bytes.insert(bytes.end(), s.begin(), s.end()); off.push_back((int64_t)bytes.size()); }
    }
    ltr_haplotype_blocks hb = {nb, bs.data(), be.data(), rep.data(), per.data(), na.data(), bytes.data(), off.data()};
    const int64_t H = ltr_haplotype_num_combs(&hb);
    if (H < 1) { std::printf("num_combs %ld\n", (long)H); return 1; }
    std::vector<uint8_t> buf(400);
    for (int64_t k = -1; k <= H; ++k) {
      const int64_t n = ltr_haplotype_seq(&hb, k, buf.data(), (int64_t)buf.size());
      if (k >= 0 && k < H && n < 0 && n != LTR_ERR_INVALID) { std::printf("haplotype_seq rc %ld\n", (long)n); return 1; }
    }
    // reads over the locus, through ltr_process_reads (host half; the stub scorer says "no device")
    ltr_ctx ctx; std::memset(&ctx.p, 0, sizeof(ctx.p)); ctx.p.indel_flank_len = ri(0, 35); ctx.p.use_short_path = (it % 5 == 0);
    const int R = ri(0, 6);
    std::vector<std::string> seqs, types; std::vector<std::vector<int32_t>> nums; std::vector<ltr_alignment> alns;
    for (int r = 0; r < R; ++r) {
      const int len = ri(1, 300);
      seqs.push_back(rseq(len)); types.push_back(std::string(1, '=')); nums.push_back({len});
    }
    for (int r = 0; r < R; ++r) {
      const int st = bs[0] + ri(-150, 50);
      alns.push_back({st, st + nums[(size_t)r][0] - 1, (const uint8_t*)seqs[(size_t)r].data(), (int32_t)seqs[(size_t)r].size(), 1,
                      types[(size_t)r].data(), nums[(size_t)r].data(), nullptr});
    }
    std::vector<double> probs((size_t)std::max<int64_t>(1, R * H), 0.0); std::vector<int32_t> seeds((size_t)std::max(1, R), 0);
    (void)ltr_process_reads(&ctx, &hb, nullptr, alns.data(), R, 0, nullptr, probs.data(), seeds.data());
    checks++;
  }
  // ---- ltr_calc_hap_aln_probs: threaded per-locus preparation + concatenation --------------------
  for (int it = 0; it < 6; ++it) {
    const int NL = 300 + 40 * it;
    struct Loc { std::vector<int32_t> bs, be, per, na; std::vector<uint8_t> rep, bytes; std::vector<int64_t> off;
                 std::vector<std::string> seqs, types; std::vector<std::vector<int32_t>> nums; std::vector<ltr_alignment> alns;
                 ltr_haplotype_blocks hb; std::vector<double> probs; std::vector<int32_t> seeds; };
    std::vector<Loc> L((size_t)NL);
    std::vector<ltr_locus> loci((size_t)NL);
    std::vector<double*> pp((size_t)NL); std::vector<int32_t*> sp((size_t)NL);
    for (int l = 0; l < NL; ++l) {
      Loc& X = L[(size_t)l];
      int pos = ri(100, 1000);
      X.off.push_back(0);
      for (int b = 0; b < 3; ++b) {
        const bool is_rep = (b == 1);
        const int len = ri(5, 40), nall = is_rep ? ri(1, 4) : 1;
        X.bs.push_back(pos); X.be.push_back(pos + len); pos += len;
        X.rep.push_back(is_rep); X.per.push_back(is_rep ? ri(2, 6) : 0); X.na.push_back(nall);
        for (int k = 0; k < nall; ++k) { const std::string s = rseq(k == 0 ? len : ri(1, 60)); X.bytes.insert(X.bytes.end(), s.begin(), s.end()); X.off.push_back((int64_t)X.bytes.size()); }
      }
      X.hb = {3, X.bs.data(), X.be.data(), X.rep.data(), X.per.data(), X.na.data(), X.bytes.data(), X.off.data()};
      const int R = ri(0, 8);
      for (int r = 0; r < R; ++r) {
        const int len = ri(1, 200);
        X.seqs.push_back(r > 0 && (rng() & 1) ? X.seqs[0] : rseq(len));            // duplicates: pools
        X.types.push_back(std::string(1, (it == 5 && l == 123 && r == 1) ? 'Q' : '=')); X.nums.push_back({(int32_t)X.seqs.back().size()});
      }
      for (int r = 0; r < R; ++r) {
        const int st = X.bs[0] + ri(-100, 40);
        X.alns.push_back({st, st + X.nums[(size_t)r][0] - 1, (const uint8_t*)X.seqs[(size_t)r].data(), (int32_t)X.seqs[(size_t)r].size(), 1,
                          X.types[(size_t)r].data(), X.nums[(size_t)r].data(), nullptr});
      }
      const int64_t H = ltr_haplotype_num_combs(&X.hb);
      X.probs.assign((size_t)std::max<int64_t>(1, R * H), 0.0); X.seeds.assign((size_t)std::max(1, R), 0);
      loci[(size_t)l] = {&X.hb, X.alns.data(), R, nullptr};
      pp[(size_t)l] = X.probs.data(); sp[(size_t)l] = X.seeds.data();
    }
    ltr_ctx ctx; std::memset(&ctx.p, 0, sizeof(ctx.p)); ctx.p.indel_flank_len = 5;
    const int rc = ltr_calc_hap_aln_probs(&ctx, loci.data(), NL, pp.data(), sp.data());
    if (rc != LTR_ERR_NO_DEVICE && rc != LTR_ERR_CIGAR && rc != LTR_ERR_INVALID) { std::printf("calc_hap_aln_probs rc %d\n", rc); return 1; }
    checks++;
  }
  // ---- the whole chunk pipeline on a fake device: five growing chunks, chunk c + 1 staged by the helper thread while the
  // calling thread "plans" chunk c; then the same one chunk after the other: the rows must be the same ----
  {
    struct Loc { std::vector<int32_t> bs, be, per, na; std::vector<uint8_t> rep, bytes; std::vector<int64_t> off;
                 std::vector<std::string> seqs, types; std::vector<std::vector<int32_t>> nums; std::vector<ltr_alignment> alns;
                 ltr_haplotype_blocks hb; std::vector<double> probs; std::vector<int32_t> seeds; };
    const int NL = 2600;
    std::vector<Loc> L((size_t)NL); std::vector<ltr_locus> loci((size_t)NL); std::vector<double*> pp((size_t)NL); std::vector<int32_t*> sp((size_t)NL);
    for (int l = 0; l < NL; ++l) {
      Loc& X = L[(size_t)l];
      int pos = ri(100, 1000);
      X.off.push_back(0);
      for (int b = 0; b < 3; ++b) {
        const bool is_rep = (b == 1);
        const int len = ri(36, 60), nall = is_rep ? ri(1, 3) : 1;
        X.bs.push_back(pos); X.be.push_back(pos + len); pos += len;
        X.rep.push_back(is_rep); X.per.push_back(is_rep ? ri(2, 6) : 0); X.na.push_back(nall);
        for (int k = 0; k < nall; ++k) { const std::string s2 = rseq(k == 0 ? len : ri(1, 60)); X.bytes.insert(X.bytes.end(), s2.begin(), s2.end()); X.off.push_back((int64_t)X.bytes.size()); }
      }
      X.hb = {3, X.bs.data(), X.be.data(), X.rep.data(), X.per.data(), X.na.data(), X.bytes.data(), X.off.data()};
      const int R = ri(1, 6);
      for (int r = 0; r < R; ++r) { X.seqs.push_back(r > 0 && (rng() & 1) ? X.seqs[0] : rseq(ri(1, 150))); X.types.push_back("="); X.nums.push_back({(int32_t)X.seqs.back().size()}); }
      for (int r = 0; r < R; ++r) {
        const int st = X.bs[0] + ri(-80, 30);
        X.alns.push_back({st, st + X.nums[(size_t)r][0] - 1, (const uint8_t*)X.seqs[(size_t)r].data(), (int32_t)X.seqs[(size_t)r].size(), 1,
                          X.types[(size_t)r].data(), X.nums[(size_t)r].data(), nullptr});
      }
      const int64_t H = ltr_haplotype_num_combs(&X.hb);
      X.probs.assign((size_t)std::max<int64_t>(1, R * H), 0.0); X.seeds.assign((size_t)R, 0);
      loci[(size_t)l] = {&X.hb, X.alns.data(), R, nullptr};
      pp[(size_t)l] = X.probs.data(); sp[(size_t)l] = X.seeds.data();
    }
    std::vector<std::vector<double>> first;
    for (int mode = 0; mode < 3; ++mode) {
      ltr_ctx ctx; std::memset(&ctx.p, 0, sizeof(ctx.p)); ctx.p.indel_flank_len = 5; ctx.fake_device = true;
      ctx.knobs.chunks = 5; ctx.knobs.chunk_growth = 1.3; ctx.knobs.chunk_growth_set = true;
      ctx.knobs.prep_ahead = mode == 0 ? 0 : (mode == 1 ? 3 : -1);          // helper thread on 16 threads / on 3 / off
      for (Loc& X : L) std::fill(X.probs.begin(), X.probs.end(), 0.0);
      const int rc = ltr_calc_hap_aln_probs(&ctx, loci.data(), NL, pp.data(), sp.data());
      if (rc != LTR_OK) { std::printf("chunked calc_hap_aln_probs on the fake device: rc %d (%s)\n", rc, ctx.err.c_str()); return 1; }
      if (mode == 0) for (const Loc& X : L) first.push_back(X.probs);
      else for (int l = 0; l < NL; ++l) if (first[(size_t)l] != L[(size_t)l].probs) { std::printf("chunk pipeline: mode %d differs at locus %d\n", mode, l); return 1; }
      checks++;
    }
    // a bad record in the fourth chunk: the call's error, whatever thread found it
    {
      Loc& X = L[2000];
      X.types[0] = "Q";
      X.alns[0].cigar_type = X.types[0].data();
      ltr_ctx ctx; std::memset(&ctx.p, 0, sizeof(ctx.p)); ctx.p.indel_flank_len = 5; ctx.fake_device = true;
      ctx.knobs.chunks = 5; ctx.knobs.chunk_growth = 1.3; ctx.knobs.chunk_growth_set = true;
      const int rc = ltr_calc_hap_aln_probs(&ctx, loci.data(), NL, pp.data(), sp.data());
      if (rc != LTR_ERR_CIGAR) { std::printf("bad record in a late chunk: rc %d\n", rc); return 1; }
      checks++;
    }
  }
  // ---- two callers at once, own contexts: one gets the worker pool, the other finds it busy and falls back to
  // short-lived threads (ltr_internal.h) ----
  {
    struct Loc { std::vector<int32_t> bs, be, per, na; std::vector<uint8_t> rep, bytes; std::vector<int64_t> off;
                 std::vector<std::string> seqs, types; std::vector<std::vector<int32_t>> nums; std::vector<ltr_alignment> alns;
                 ltr_haplotype_blocks hb; std::vector<double> probs; std::vector<int32_t> seeds; };
    auto make = [&](int NL, std::vector<Loc>& L, std::vector<ltr_locus>& loci, std::vector<double*>& pp, std::vector<int32_t*>& sp) {
      L.resize((size_t)NL); loci.resize((size_t)NL); pp.resize((size_t)NL); sp.resize((size_t)NL);
      for (int l = 0; l < NL; ++l) {
        Loc& X = L[(size_t)l];
        int pos = ri(100, 1000);
        X.off.push_back(0);
        for (int b = 0; b < 3; ++b) {
          const bool is_rep = (b == 1);
          const int len = ri(5, 40), nall = is_rep ? ri(1, 3) : 1;
          X.bs.push_back(pos); X.be.push_back(pos + len); pos += len;
          X.rep.push_back(is_rep); X.per.push_back(is_rep ? ri(2, 6) : 0); X.na.push_back(nall);
          for (int k = 0; k < nall; ++k) { const std::string s2 = rseq(k == 0 ? len : ri(1, 60)); X.bytes.insert(X.bytes.end(), s2.begin(), s2.end()); X.off.push_back((int64_t)X.bytes.size()); }
        }
        X.hb = {3, X.bs.data(), X.be.data(), X.rep.data(), X.per.data(), X.na.data(), X.bytes.data(), X.off.data()};
        const int R = ri(1, 6);
        for (int r = 0; r < R; ++r) { X.seqs.push_back(rseq(ri(1, 150))); X.types.push_back("="); X.nums.push_back({(int32_t)X.seqs.back().size()}); }
        for (int r = 0; r < R; ++r) {
          const int st = X.bs[0] + ri(-80, 30);
          X.alns.push_back({st, st + X.nums[(size_t)r][0] - 1, (const uint8_t*)X.seqs[(size_t)r].data(), (int32_t)X.seqs[(size_t)r].size(), 1,
                            X.types[(size_t)r].data(), X.nums[(size_t)r].data(), nullptr});
        }
        const int64_t H = ltr_haplotype_num_combs(&X.hb);
        X.probs.assign((size_t)std::max<int64_t>(1, R * H), 0.0); X.seeds.assign((size_t)R, 0);
        loci[(size_t)l] = {&X.hb, X.alns.data(), R, nullptr};
        pp[(size_t)l] = X.probs.data(); sp[(size_t)l] = X.seeds.data();
      }
    };
    std::vector<Loc> LA, LB; std::vector<ltr_locus> la, lb; std::vector<double*> pa, pb; std::vector<int32_t*> sa, sb;
    make(700, LA, la, pa, sa); make(800, LB, lb, pb, sb);
    int rca = 0, rcb = 0;
    std::thread ta([&]() { ltr_ctx c; std::memset(&c.p, 0, sizeof(c.p)); c.p.indel_flank_len = 5; for (int k = 0; k < 3; ++k) rca = ltr_calc_hap_aln_probs(&c, la.data(), 700, pa.data(), sa.data()); });
    std::thread tb([&]() { ltr_ctx c; std::memset(&c.p, 0, sizeof(c.p)); c.p.indel_flank_len = 5; for (int k = 0; k < 3; ++k) rcb = ltr_calc_hap_aln_probs(&c, lb.data(), 800, pb.data(), sb.data()); });
    ta.join(); tb.join();
    if (rca != LTR_ERR_NO_DEVICE || rcb != LTR_ERR_NO_DEVICE) { std::printf("concurrent calc_hap_aln_probs rc %d %d\n", rca, rcb); return 1; }
    checks++;
  }
  // ---- on-disk formats: region reader (valid, malformed, truncated lines), VCF writer (plain + BGZF) ----
  {
    const std::string dir = "/tmp/ltr_harness_" + std::to_string((long)getpid());
    (void)mkdir(dir.c_str(), 0700);
    for (int it = 0; it < 300; ++it) {
      const std::string bed = dir + "/r.bed";
      FILE* f = std::fopen(bed.c_str(), "w");
      const int nl = ri(0, 12);
      for (int l = 0; l < nl; ++l) {
        const int st = ri(it % 7 == 0 ? 0 : 1, 5000);
        std::string line = "chr" + std::to_string(ri(1, 3)) + "\t" + std::to_string(st) + "\t" + std::to_string(st + ri(it % 5 == 0 ? -2 : 1, 80)) + "\t" + rseq(ri(1, 6));
        if (it % 3 == 0) line += ",AC";
        if (it % 4 == 0) line += "\tname" + std::to_string(l);
        if (it % 11 == 0 && l == 1) line = line.substr(0, (size_t)ri(0, (int)line.size()));     // truncated line
        std::fprintf(f, "%s\n", line.c_str());
      }
      std::fclose(f);
      ltr_region_set* rs = nullptr; char err[64];                  // (a short error buffer: messages are cut, not overrun)
      const int rc = ltr_read_regions(bed.c_str(), (uint32_t)ri(1, 20), it % 2 ? "chr1" : nullptr, &rs, err, (int)sizeof(err));
      if (rc == LTR_OK) {
        ltr_region_set_order(rs);
        for (int64_t i = -1; i <= ltr_region_set_size(rs); ++i) { (void)ltr_region_chrom(rs, i); (void)ltr_region_period(rs, i); (void)ltr_region_period_str(rs, i); }
        ltr_region_set_free(rs);
      } else if (rc != LTR_ERR_INVALID || std::strlen(err) >= sizeof(err)) { std::printf("read_regions rc %d\n", rc); return 1; }
      checks++;
    }
    for (int it = 0; it < 6; ++it) {
      ltr_vcf_writer* w = nullptr;
      const std::string path = dir + (it % 2 ? "/o.vcf.gz" : "/o.vcf");
      if (ltr_vcf_writer_open(path.c_str(), &w) != LTR_OK) { std::printf("vcf_writer_open\n"); return 1; }
      (void)ltr_vcf_writer_header(w, "##fileformat=VCFv4.1\n");
      int pos = 100;
      for (int k = 0; k < 4000; ++k) {
        pos += ri(0, 90);
        const std::string rec = "chr" + std::to_string(1 + k / 1500) + "\t" + std::to_string(pos + ri(-25, 25)) + "\t" + rseq(ri(0, 300));
        if (ltr_vcf_writer_add_record(w, ("chr" + std::to_string(1 + k / 1500)).c_str(), pos + ri(-25, 25), rec.c_str()) != LTR_OK) { std::printf("add_record\n"); return 1; }
      }
      if (ltr_vcf_writer_close(w) != LTR_OK) { std::printf("vcf_writer_close\n"); return 1; }
      checks++;
    }
    (void)std::remove((dir + "/r.bed").c_str()); (void)std::remove((dir + "/o.vcf").c_str()); (void)std::remove((dir + "/o.vcf.gz").c_str());
    (void)rmdir(dir.c_str());
  }
  // ---- BAM reader on a real file (argv[1]): every chromosome, random regions, a truncated and a corrupted copy ----
  if (argc > 1) {
    const char* paths[1] = {argv[1]};
    ltr_bam* b = nullptr; char err[256];
    if (ltr_bam_open(paths, 1, 1, &b, err, (int)sizeof(err)) != LTR_OK) { std::printf("bam_open: %s\n", err); return 1; }
    long n_rec = 0, n_bases = 0;
    ltr_bam_record rec;
    for (int32_t t = 0; t < ltr_bam_num_refs(b); ++t) {
      if (ltr_bam_set_region(b, ltr_bam_ref_name(b, t), 0, (int32_t)std::min<int64_t>(ltr_bam_ref_len(b, t), 0x7fffffff)) != LTR_OK) { std::printf("set_region\n"); return 1; }
      int rc;
      while ((rc = ltr_bam_next(b, &rec)) == 1) {
        n_rec++; n_bases += (long)std::strlen(rec.bases) + (long)std::strlen(rec.quals) + rec.n_cigar;
        int64_t iv; double dv; char cv;
        (void)ltr_bam_aux_int(&rec, "NM", &iv); (void)ltr_bam_aux_float(&rec, "rq", &dv); (void)ltr_bam_aux_char(&rec, "XX", &cv); (void)ltr_bam_aux_string(&rec, "RG");
        if (n_rec % 40 == 0 && ltr_bam_set_region(b, ltr_bam_ref_name(b, t), rec.pos + ri(-500, 500), rec.pos + ri(0, 30000)) != LTR_OK) { std::printf("set_region (inner)\n"); return 1; }
      }
      if (rc < 0) { std::printf("bam_next rc %d\n", rc); return 1; }
    }
    ltr_bam_close(b);
    if (n_rec == 0 || n_bases == 0) { std::printf("no BAM records read\n"); return 1; }
    // damaged copies: any status is fine, any memory error is not
    std::vector<uint8_t> raw; { FILE* f = std::fopen(argv[1], "rb"); uint8_t tmp[65536]; size_t n; while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) raw.insert(raw.end(), tmp, tmp + n); std::fclose(f); }
    std::vector<uint8_t> bai; { FILE* f = std::fopen((std::string(argv[1]) + ".bai").c_str(), "rb"); uint8_t tmp[65536]; size_t n; while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) bai.insert(bai.end(), tmp, tmp + n); std::fclose(f); }
    const std::string dir = "/tmp/ltr_harness_bam_" + std::to_string((long)getpid());
    (void)mkdir(dir.c_str(), 0700);
    for (int it = 0; it < 12; ++it) {
      std::vector<uint8_t> d = raw, x = bai;
      if (it % 3 == 0) d.resize((size_t)ri(100, (int)d.size() - 1));
      else if (it % 3 == 1) for (int k = 0; k < 30; ++k) d[(size_t)ri(2000, (int)d.size() - 1)] ^= (uint8_t)ri(1, 255);
      else for (int k = 0; k < 6; ++k) x[(size_t)ri(8, (int)x.size() - 1)] ^= (uint8_t)ri(1, 255);
      const std::string p = dir + "/d.bam";
      { FILE* f = std::fopen(p.c_str(), "wb"); std::fwrite(d.data(), 1, d.size(), f); std::fclose(f); }
      { FILE* f = std::fopen((p + ".bai").c_str(), "wb"); std::fwrite(x.data(), 1, x.size(), f); std::fclose(f); }
      const char* pp[1] = {p.c_str()};
      ltr_bam* q = nullptr;
      if (ltr_bam_open(pp, 1, 1, &q, err, (int)sizeof(err)) == LTR_OK) {
        for (int32_t t = 0; t < std::min(ltr_bam_num_refs(q), 3); ++t)
          if (ltr_bam_set_region(q, ltr_bam_ref_name(q, t), 0, 0x7fffffff) == LTR_OK) { int guard = 0; while (ltr_bam_next(q, &rec) == 1 && ++guard < 100000) {} }
        ltr_bam_close(q);
      }
      checks++;
    }
    (void)std::remove((dir + "/d.bam").c_str()); (void)std::remove((dir + "/d.bam.bai").c_str()); (void)rmdir(dir.c_str());
    checks++;
  }
  // ---- pooling + scatter ------------------------------------------------------------------------
  for (int it = 0; it < 2000; ++it) {
    const int R = ri(0, 40), H = ri(1, 9);
    std::vector<std::string> pool; for (int k = 0; k < 6; ++k) pool.push_back(rseq(ri(0, 30)));
    std::vector<const uint8_t*> ptr; std::vector<int32_t> len, idx((size_t)std::max(1, R));
    for (int r = 0; r < R; ++r) { const std::string& s = pool[(size_t)ri(0, 5)]; ptr.push_back((const uint8_t*)s.data()); len.push_back((int32_t)s.size()); }
    const int P = ltr_pool_reads(ptr.data(), len.data(), R, idx.data());
    if (P < 0 || P > R) { std::printf("pool_reads %d\n", P); return 1; }
    std::vector<double> pp((size_t)std::max(1, P * H), -1.0), out((size_t)std::max(1, R * H), -5.0);
    std::vector<int32_t> ps((size_t)std::max(1, P), 7), seeds((size_t)std::max(1, R), -1);
    std::vector<uint8_t> mh((size_t)H), cr((size_t)std::max(1, R)), sm((size_t)std::max(1, R));
    for (auto& x : mh) x = rng() & 1; for (auto& x : cr) x = rng() & 1; for (auto& x : sm) x = (rng() % 5 == 0);
    if (R > 0) sm[0] = 0;
    (void)ltr_scatter_pool_probs(pp.data(), ps.data(), idx.data(), R, H, it % 3 ? mh.data() : nullptr, it % 2 ? cr.data() : nullptr,
                                 it % 4 ? nullptr : sm.data(), out.data(), seeds.data());
    checks++;
  }
  // ---- genotype fields --------------------------------------------------------------------------
  for (int it = 0; it < 3000; ++it) {
    const int S = ri(0, 4), H = ri(1, 8), V = ri(1, H), hap = it & 1;
    std::vector<double> post((size_t)std::max(1, S * H * H)), stl((size_t)std::max(1, S), -30.0);
    for (auto& x : post) x = -(double)(rng() % 100000) / 1000.0 - ((rng() % 50 == 0) ? 8.9e307 : 0.0);
    std::vector<int32_t> h2a((size_t)H), best((size_t)std::max(1, 2 * S));
    for (auto& x : h2a) x = ri(0, V - 1);
    for (auto& x : best) x = ri(0, H - 1);
    if (it % 37 == 0) best[0] = H;                                     // invalid on purpose
    const int ngl = hap ? V : V * (V + 1) / 2, npgl = hap ? V : V * V;
    std::vector<int32_t> gts((size_t)std::max(1, 2 * S)), pls((size_t)std::max(1, S * ngl));
    std::vector<double> a((size_t)std::max(1, S)), b2(a), c(a), d(a), gl((size_t)std::max(1, S * ngl)), gd(a), pg((size_t)std::max(1, S * npgl));
    ltr_genotype_fields f = {gts.data(), a.data(), b2.data(), c.data(), d.data(), it % 3 ? gl.data() : nullptr, it % 2 ? gd.data() : nullptr,
                             it % 5 ? pls.data() : nullptr, it % 7 ? pg.data() : nullptr};
    (void)ltr_extract_genotypes(S, H, V, h2a.data(), hap, post.data(), stl.data(), best.data(), &f);
    checks++;
  }
  std::printf("host sanitizer harness: %ld cases, %ld batches reached the scorer (%ld pairs)\n", checks, g_batches.load(), g_pairs.load());
  return 0;
}

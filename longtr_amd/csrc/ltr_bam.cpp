// ltr_bam.cpp -- indexed BAM input without htslib: BGZF blocks through zlib, the BAM
// record layout and the BAI binning index as the SAM specification publishes them (sections 4.1, 4.2, 5.1.1-5.2).
//
// Replaces, for position-sorted *.bam files with a *.bam.bai (or *.bai) next to them:
//   BamHeader (reference names / lengths, @RG ID / SM / LB)            reference src/bam_io.h:360-420, src/bam_io.cpp:43-70
//   BamCramReader::SetRegion / GetNextAlignment                         src/bam_io.cpp:142-197
//     (region "chrom:start+1-end" = records overlapping [start, end); the stream ends at the first record
//      whose position is > end + 1)
//   BamAlignment::ExtractSequenceFields (bases, qualities + 33, CIGAR)  src/bam_io.cpp:14-40
//   BamAlignment's tag getters (bam_aux_get: A, c C s S i I, f, Z)      src/bam_io.h:182-212
//   BamCramMultiReader::SetRegion / GetNextAlignment                    src/bam_io.cpp:201-244
//     (several files as one stream: a heap of (-position, file) or (-file, file))
// Not here: CRAM (needs the reference FASTA codec of htslib), remote paths, BamWriter.  A record with more than
// 65535 CIGAR operations (the CG:B,I convention) is returned with the placeholder CIGAR the file holds.
// Parity: htslib is the implementation behind the reference and is not in this tree -- unpinned; checked on
// the reference's bundled BAM files against an independent Python decoder (tests/test_bam_reader.py).

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include <unistd.h>
#include <sys/stat.h>
#include <zlib.h>

#include "ltr_internal.h"

namespace {

void put_error(char* err, int cap, const std::string& msg) {
  if (!err || cap <= 0) return;
  const size_t n = std::min(msg.size(), (size_t)cap - 1);
  std::memcpy(err, msg.data(), n);
  err[n] = 0;
}

uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// ---- BGZF reader: one inflated block at a time, addressed by virtual offsets (coffset << 16 | uoffset) ----
struct Bgzf {
  FILE* f = nullptr;
  int64_t block_addr = -1, next_addr = 0;         // file offset of the block in `data` / of the one after it
  std::vector<uint8_t> data;                      // inflated bytes of the current block
  size_t at = 0;
  bool bad = false;
  ~Bgzf() { if (f) std::fclose(f); }
  bool load(int64_t addr) {                       // false: end of file or a damaged block (bad)
    uint8_t head[18];
    if (fseeko(f, (off_t)addr, SEEK_SET) != 0) { bad = true; return false; }
    const size_t got = std::fread(head, 1, 18, f);
    if (got == 0) return false;
    if (got != 18 || head[0] != 0x1f || head[1] != 0x8b || head[2] != 8 || !(head[3] & 4)) { bad = true; return false; }
    // the extra field holds the BC subfield (possibly after others)
    const int xlen = le16(head + 10);
    std::vector<uint8_t> extra((size_t)xlen);
    std::memcpy(extra.data(), head + 12, std::min<size_t>(6, (size_t)xlen));
    if (xlen > 6 && std::fread(extra.data() + 6, 1, (size_t)xlen - 6, f) != (size_t)xlen - 6) { bad = true; return false; }
    int bsize = -1;
    for (int k = 0; k + 4 <= xlen;) {
      const int slen = le16(extra.data() + k + 2);
      if (extra[(size_t)k] == 'B' && extra[(size_t)k + 1] == 'C' && slen == 2 && k + 6 <= xlen) bsize = le16(extra.data() + k + 4) + 1;
      k += 4 + slen;
    }
    if (bsize < 12 + xlen + 8) { bad = true; return false; }
    const size_t clen = (size_t)bsize - 12 - (size_t)xlen - 8;
    std::vector<uint8_t> comp(clen + 8);
    if (std::fread(comp.data(), 1, clen + 8, f) != clen + 8) { bad = true; return false; }
    const uint32_t isize = le32(comp.data() + clen + 4);
    if (isize > 65536) { bad = true; return false; }
    // (inflated into a buffer of its own and committed only once the block is whole: a damaged block never replaces --
    // or half-overwrites -- the bytes block_addr stands for)
    std::vector<uint8_t> fresh(isize);
    if (isize) {
      z_stream zs; std::memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, -15) != Z_OK) { bad = true; return false; }
      zs.next_in = comp.data(); zs.avail_in = (uInt)clen; zs.next_out = fresh.data(); zs.avail_out = isize;
      const int rc = inflate(&zs, Z_FINISH);
      inflateEnd(&zs);
      if (rc != Z_STREAM_END || zs.total_out != isize || (uint32_t)crc32(crc32(0L, Z_NULL, 0), fresh.data(), isize) != le32(comp.data() + clen)) { bad = true; return false; }
    }
    data.swap(fresh);
    block_addr = addr; next_addr = addr + bsize; at = 0;
    return true;
  }
  bool seek(uint64_t voff) {
    const int64_t addr = (int64_t)(voff >> 16);
    if (addr != block_addr && !load(addr)) return false;
    at = (size_t)(voff & 0xffff);
    return at <= data.size();
  }
  uint64_t tell() const { return at < data.size() || block_addr < 0 ? (((uint64_t)block_addr) << 16) | at : ((uint64_t)next_addr) << 16; }
  bool read(void* dst, size_t n) {                // false: fewer than n bytes left (end of file when !bad and nothing was read)
    uint8_t* out = (uint8_t*)dst;
    while (n) {
      if (block_addr < 0 || at >= data.size()) {
        do { if (!load(block_addr < 0 ? 0 : next_addr)) return false; } while (data.empty());    // (empty blocks: the end-of-file marker)
      }
      const size_t k = std::min(n, data.size() - at);
      std::memcpy(out, data.data() + at, k);
      out += k; at += k; n -= k;
    }
    return true;
  }
};

struct Chunk { uint64_t beg, end; };
struct RefIndex { std::map<uint32_t, std::vector<Chunk>> bins; std::vector<uint64_t> linear; };

struct ReadGroup { std::string id, sample, library; };

struct Record {
  std::vector<uint8_t> raw;                       // the record after block_size
  int32_t ref_id = -1, pos = -1, end_pos = -1, mate_ref_id = -1, mate_pos = -1, tlen = 0, l_seq = 0, n_cigar = 0;
  uint16_t flag = 0; uint8_t mapq = 0;
  std::string name, bases, quals, cigar_type;
  std::vector<int32_t> cigar_num;
  size_t aux_off = 0;
};

struct Reader {
  std::string path;
  Bgzf z;
  std::string text;
  std::vector<std::string> ref_names; std::vector<int64_t> ref_lens;
  std::map<std::string, int32_t> ref_ids;
  std::vector<ReadGroup> read_groups;
  std::vector<RefIndex> index;
  // region state
  int32_t tid = -1, beg = 0, end = 0;
  std::vector<Chunk> todo; size_t chunk_i = 0; bool in_chunk = false, done = true;
  Record cached; bool have = false;
};

bool parse_record(Record& r) {
  const uint8_t* p = r.raw.data();
  if (r.raw.size() < 32) return false;
  r.ref_id = (int32_t)le32(p); r.pos = (int32_t)le32(p + 4);
  const int l_name = p[8]; r.mapq = p[9];
  r.n_cigar = le16(p + 12); r.flag = le16(p + 14); r.l_seq = (int32_t)le32(p + 16);
  r.mate_ref_id = (int32_t)le32(p + 20); r.mate_pos = (int32_t)le32(p + 24); r.tlen = (int32_t)le32(p + 28);
  size_t at = 32;
  if (r.l_seq < 0 || at + (size_t)l_name + 4u * (size_t)r.n_cigar + ((size_t)r.l_seq + 1) / 2 + (size_t)r.l_seq > r.raw.size()) return false;
  r.name.assign((const char*)p + at, l_name > 0 ? (size_t)l_name - 1 : 0); at += (size_t)l_name;
  r.cigar_type.resize((size_t)r.n_cigar); r.cigar_num.resize((size_t)r.n_cigar);
  int64_t rlen = 0;
  for (int k = 0; k < r.n_cigar; ++k, at += 4) {
    const uint32_t c = le32(p + at);
    const int op = (int)(c & 15); const int32_t len = (int32_t)(c >> 4);
    r.cigar_type[(size_t)k] = op < 9 ? "MIDNSHP=X"[op] : '?';                 // bam_cigar_opchr
    r.cigar_num[(size_t)k] = len;
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += len;      // bam_cigar2rlen: M D N = X
  }
  // ExtractSequenceFields (bam_io.cpp:14-40): 4-bit bases through "=ACMGRSVTWYHKDBN", qualities + 33
  r.bases.resize((size_t)r.l_seq); r.quals.resize((size_t)r.l_seq);
  for (int32_t i = 0; i < r.l_seq; ++i) r.bases[(size_t)i] = "=ACMGRSVTWYHKDBN"[(p[at + (size_t)(i >> 1)] >> ((~i & 1) << 2)) & 15];
  at += ((size_t)r.l_seq + 1) / 2;
  for (int32_t i = 0; i < r.l_seq; ++i) r.quals[(size_t)i] = (char)(p[at + (size_t)i] + 33);
  at += (size_t)r.l_seq;
  r.aux_off = at;
  // bam_endpos: unmapped or CIGAR-less records cover one base
  if ((r.flag & 4) || r.n_cigar == 0 || rlen == 0) rlen = 1;
  r.end_pos = (int32_t)(r.pos + rlen);
  return true;
}

// 1 = record read, 0 = clean end of file, < 0 = damaged file
int read_record(Bgzf& z, Record& r) {
  uint8_t szb[4];
  if (!z.read(szb, 4)) return z.bad ? LTR_ERR_INVALID : 0;
  const uint32_t sz = le32(szb);
  if (sz < 32 || sz > (1u << 28)) return LTR_ERR_INVALID;
  r.raw.resize(sz);
  if (!z.read(r.raw.data(), sz)) return LTR_ERR_INVALID;
  return parse_record(r) ? 1 : LTR_ERR_INVALID;
}

// reg2bins of the SAM specification (5.3): the bins a region [beg, end) can overlap
void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t>& bins) {
  --end;
  bins.push_back(0);
  for (int64_t k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) bins.push_back((uint32_t)k);
  for (int64_t k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) bins.push_back((uint32_t)k);
  for (int64_t k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) bins.push_back((uint32_t)k);
  for (int64_t k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) bins.push_back((uint32_t)k);
  for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k) bins.push_back((uint32_t)k);
}

bool load_index(Reader& R, std::string* err) {
  FILE* f = std::fopen((R.path + ".bai").c_str(), "rb");
  if (!f && R.path.size() > 4) f = std::fopen((R.path.substr(0, R.path.size() - 4) + ".bai").c_str(), "rb");
  if (!f) { *err = "Failed to load the index of " + R.path; return false; }
  std::vector<uint8_t> b;
  uint8_t tmp[65536]; size_t n;
  while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) b.insert(b.end(), tmp, tmp + n);
  std::fclose(f);
  size_t at = 8;
  if (b.size() < 8 || std::memcmp(b.data(), "BAI\1", 4) != 0) { *err = "Failed to load the index of " + R.path; return false; }
  const uint32_t n_ref = le32(b.data() + 4);
  if ((size_t)n_ref * 8 > b.size()) { *err = "Failed to load the index of " + R.path; return false; }      // (>= 8 bytes per reference)
  R.index.resize(n_ref);
  auto need = [&](size_t k) { return at + k <= b.size(); };
  for (uint32_t r = 0; r < n_ref; ++r) {
    if (!need(4)) return false;
    const uint32_t n_bin = le32(b.data() + at); at += 4;
    for (uint32_t k = 0; k < n_bin; ++k) {
      if (!need(8)) return false;
      const uint32_t bin = le32(b.data() + at), n_chunk = le32(b.data() + at + 4); at += 8;
      if (!need((size_t)n_chunk * 16)) return false;
      std::vector<Chunk> ch((size_t)n_chunk);
      for (uint32_t c = 0; c < n_chunk; ++c, at += 16) ch[c] = {le64(b.data() + at), le64(b.data() + at + 8)};
      if (bin != 37450) R.index[r].bins[bin] = std::move(ch);                 // (37450: the metadata pseudo-bin)
    }
    if (!need(4)) return false;
    const uint32_t n_intv = le32(b.data() + at); at += 4;
    if (!need((size_t)n_intv * 8)) return false;
    R.index[r].linear.resize(n_intv);
    for (uint32_t k = 0; k < n_intv; ++k, at += 8) R.index[r].linear[k] = le64(b.data() + at);
  }
  return true;
}

bool open_reader(Reader& R, std::string* err) {
  R.z.f = std::fopen(R.path.c_str(), "rb");
  if (!R.z.f) { *err = "Failed to open " + R.path; return false; }
  uint8_t h[8];
  if (!R.z.read(h, 8) || std::memcmp(h, "BAM\1", 4) != 0) { *err = "Not a BAM file: " + R.path; return false; }
  const uint32_t l_text = le32(h + 4);
  // (sizes from the file are checked before anything is allocated for them: a header cannot inflate to more than
  // ~1000 x the file's own size, and a reference name is at most a few kilobytes)
  int64_t file_size = 0;
  { struct stat st; if (stat(R.path.c_str(), &st) == 0) file_size = (int64_t)st.st_size; }
  if ((int64_t)l_text > std::max<int64_t>(file_size, 1) * 1100 || l_text > (1u << 30)) { *err = "Truncated BAM header in " + R.path; return false; }
  R.text.resize(l_text);
  if (l_text && !R.z.read(&R.text[0], l_text)) { *err = "Truncated BAM header in " + R.path; return false; }
  while (!R.text.empty() && R.text.back() == 0) R.text.pop_back();
  uint8_t nb[4];
  if (!R.z.read(nb, 4)) { *err = "Truncated BAM header in " + R.path; return false; }
  const uint32_t n_ref = le32(nb);
  if ((int64_t)n_ref * 9 > std::max<int64_t>(file_size, 1) * 1100) { *err = "Truncated BAM header in " + R.path; return false; }   // (>= 9 bytes per reference)
  for (uint32_t r = 0; r < n_ref; ++r) {
    uint8_t lb[4];
    if (!R.z.read(lb, 4)) { *err = "Truncated BAM header in " + R.path; return false; }
    const uint32_t l_name = le32(lb);
    if (l_name > 4096) { *err = "Truncated BAM header in " + R.path; return false; }
    std::string name(l_name, 0);
    if ((l_name && !R.z.read(&name[0], l_name)) || !R.z.read(lb, 4)) { *err = "Truncated BAM header in " + R.path; return false; }
    if (!name.empty() && name.back() == 0) name.pop_back();
    R.ref_ids[name] = (int32_t)R.ref_names.size();
    R.ref_names.push_back(name); R.ref_lens.push_back((int64_t)le32(lb));
  }
  // BamHeader::parse_read_groups (bam_io.cpp:43-70): @RG lines, tab-separated ID: / SM: / LB:
  std::stringstream ss(R.text);
  std::string line;
  while (std::getline(ss, line)) {
    if (line.compare(0, 3, "@RG") != 0) continue;
    ReadGroup rg;
    std::stringstream ls(line); std::string tok; bool first = true;
    while (std::getline(ls, tok, '\t')) {
      if (first) { first = false; continue; }
      if (tok.compare(0, 3, "ID:") == 0) rg.id = tok.substr(3);
      else if (tok.compare(0, 3, "SM:") == 0) rg.sample = tok.substr(3);
      else if (tok.compare(0, 3, "LB:") == 0) rg.library = tok.substr(3);
    }
    R.read_groups.push_back(rg);
  }
  if (!load_index(R, err)) { if (err->empty()) *err = "Damaged index of " + R.path; return false; }
  return true;
}

// BamCramReader::SetRegion (bam_io.cpp:142-169): "chrom:start+1-end" = [start, end)
bool set_region(Reader& R, const std::string& chrom, int32_t start, int32_t end) {
  R.done = true; R.have = false; R.todo.clear(); R.chunk_i = 0; R.in_chunk = false;
  auto it = R.ref_ids.find(chrom);
  if (it == R.ref_ids.end()) return false;                                   // sam_itr_querys: unknown name -> NULL
  R.tid = it->second; R.beg = std::max(start, 0); R.end = end;
  R.done = false;
  if ((size_t)R.tid >= R.index.size() || R.end <= R.beg) { R.done = true; return true; }
  const RefIndex& ix = R.index[(size_t)R.tid];
  std::vector<uint32_t> bins;
  reg2bins(R.beg, R.end, bins);
  uint64_t min_off = 0;
  if (!ix.linear.empty()) min_off = ix.linear[std::min<size_t>((size_t)(R.beg >> 14), ix.linear.size() - 1)];
  for (uint32_t bn : bins) {
    auto b = ix.bins.find(bn);
    if (b == ix.bins.end()) continue;
    for (const Chunk& c : b->second) if (c.end > min_off) R.todo.push_back(c);
  }
  std::sort(R.todo.begin(), R.todo.end(), [](const Chunk& a, const Chunk& b) { return a.beg < b.beg; });
  std::vector<Chunk> merged;
  for (const Chunk& c : R.todo) {
    if (!merged.empty() && c.beg <= merged.back().end) merged.back().end = std::max(merged.back().end, c.end);
    else merged.push_back(c);
  }
  R.todo.swap(merged);
  if (R.todo.empty()) R.done = true;
  return true;
}

// the next record of the region; 1 / 0 / < 0
int next_in_region(Reader& R, Record& out) {
  while (!R.done) {
    if (!R.in_chunk) {
      if (R.chunk_i >= R.todo.size()) { R.done = true; break; }
      if (!R.z.seek(R.todo[R.chunk_i].beg)) return LTR_ERR_INVALID;
      R.in_chunk = true;
    }
    if (R.z.tell() >= R.todo[R.chunk_i].end) { R.in_chunk = false; ++R.chunk_i; continue; }
    const int rc = read_record(R.z, out);
    if (rc < 0) return rc;
    if (rc == 0) { R.done = true; break; }
    if (out.ref_id != R.tid || out.pos >= R.end) { R.done = true; break; }     // position-sorted: nothing further can overlap
    if (out.end_pos > R.beg) return 1;
  }
  return 0;
}

}  // namespace

struct ltr_bam {
  std::vector<std::unique_ptr<Reader>> readers;
  bool by_position = true;
  std::vector<std::pair<int32_t, int32_t>> heap;                    // BamCramMultiReader::aln_heap_
  Record current;
  int32_t current_file = -1, region_end = 0;
  std::vector<ReadGroup> read_groups; std::vector<int32_t> read_group_file;
};

extern "C" {

// BamCramMultiReader(paths, merge_type) (bam_io.h:537-560); all files must name the same reference sequences
// in the same order (compare_bam_headers, bam_io.cpp:247-270)
int ltr_bam_open(const char* const* paths, int32_t n_files, int32_t merge_by_position, ltr_bam** out, char* err, int err_cap) {
  if (!paths || n_files < 1 || !out) return LTR_ERR_INVALID;
  *out = nullptr;
  try {
  std::unique_ptr<ltr_bam> b(new ltr_bam());
  b->by_position = merge_by_position != 0;
  for (int32_t k = 0; k < n_files; ++k) {
    if (!paths[k]) return LTR_ERR_INVALID;
    std::unique_ptr<Reader> R(new Reader());
    R->path = paths[k];
    std::string msg;
    if (!open_reader(*R, &msg)) { put_error(err, err_cap, msg); return LTR_ERR_INVALID; }
    if (k > 0 && (R->ref_names != b->readers[0]->ref_names || R->ref_lens != b->readers[0]->ref_lens)) {
      put_error(err, err_cap, "BAM header mismatch issue. BAM headers for files " + b->readers[0]->path + " and " + R->path + " must have the same reference sequences");
      return LTR_ERR_INVALID;
    }
    for (const ReadGroup& rg : R->read_groups) { b->read_groups.push_back(rg); b->read_group_file.push_back(k); }
    b->readers.push_back(std::move(R));
  }
  *out = b.release();
  return LTR_OK;
  } catch (const std::bad_alloc&) { put_error(err, err_cap, "out of host memory"); return LTR_ERR_NOMEM; }
  catch (const std::exception& e) { put_error(err, err_cap, std::string("internal error: ") + e.what()); return LTR_ERR_INVALID; }
}
void ltr_bam_close(ltr_bam* b) { delete b; }
int32_t ltr_bam_num_refs(const ltr_bam* b) { return b ? (int32_t)b->readers[0]->ref_names.size() : 0; }
const char* ltr_bam_ref_name(const ltr_bam* b, int32_t i) { return (b && i >= 0 && i < ltr_bam_num_refs(b)) ? b->readers[0]->ref_names[(size_t)i].c_str() : nullptr; }
int64_t ltr_bam_ref_len(const ltr_bam* b, int32_t i) { return (b && i >= 0 && i < ltr_bam_num_refs(b)) ? b->readers[0]->ref_lens[(size_t)i] : -1; }
int32_t ltr_bam_num_read_groups(const ltr_bam* b) { return b ? (int32_t)b->read_groups.size() : 0; }
#define LTR_RG(field) (b && i >= 0 && i < (int32_t)b->read_groups.size()) ? b->read_groups[(size_t)i].field.c_str() : nullptr
const char* ltr_bam_read_group_id(const ltr_bam* b, int32_t i) { return LTR_RG(id); }
const char* ltr_bam_read_group_sample(const ltr_bam* b, int32_t i) { return LTR_RG(sample); }
const char* ltr_bam_read_group_library(const ltr_bam* b, int32_t i) { return LTR_RG(library); }
#undef LTR_RG
int32_t ltr_bam_read_group_file(const ltr_bam* b, int32_t i) { return (b && i >= 0 && i < (int32_t)b->read_group_file.size()) ? b->read_group_file[(size_t)i] : -1; }

// BamCramMultiReader::SetRegion (bam_io.cpp:201-220).  LTR_OK, or LTR_ERR_INVALID when a file does not know the chromosome
// (the reference's SetRegion returns false).
int ltr_bam_set_region(ltr_bam* b, const char* chrom, int32_t start, int32_t end) {
  if (!b || !chrom) return LTR_ERR_INVALID;
  try {
  b->heap.clear(); b->region_end = end; b->current_file = -1;
  for (size_t k = 0; k < b->readers.size(); ++k) {
    Reader& R = *b->readers[k];
    R.z.bad = false;                                                  // (a damaged block met in an earlier region does not poison this one)
    if (!set_region(R, chrom, start, end)) return LTR_ERR_INVALID;
    const int rc = next_in_region(R, R.cached);
    if (rc < 0) return rc;
    R.have = rc == 1;
    if (R.have && (int64_t)R.cached.pos > (int64_t)end + 1) R.have = false;      // (64-bit: the reference's end_ + 1 overflows for SetChromosome's INT32_MAX)                      // GetNextAlignment's stop test, bam_io.cpp:174
    if (R.have) b->heap.push_back(b->by_position ? std::make_pair(-R.cached.pos, (int32_t)k) : std::make_pair(-(int32_t)k, (int32_t)k));
  }
  std::make_heap(b->heap.begin(), b->heap.end());
  return LTR_OK;
  } catch (const std::bad_alloc&) { return LTR_ERR_NOMEM; } catch (const std::exception&) { return LTR_ERR_INVALID; }
}

// BamCramMultiReader::GetNextAlignment (bam_io.cpp:222-244): 1 = a record (fields valid until the next call), 0 = the region is exhausted
int ltr_bam_next(ltr_bam* b, ltr_bam_record* rec) {
  if (!b || !rec) return LTR_ERR_INVALID;
  if (b->heap.empty()) return 0;
  try {
  std::pop_heap(b->heap.begin(), b->heap.end());
  const int32_t k = b->heap.back().second;
  b->heap.pop_back();
  Reader& R = *b->readers[(size_t)k];
  std::swap(b->current, R.cached);
  b->current_file = k;
  const int rc = next_in_region(R, R.cached);
  if (rc < 0) return rc;
  R.have = rc == 1 && !((int64_t)R.cached.pos > (int64_t)b->region_end + 1);
  if (R.have) {
    b->heap.push_back(b->by_position ? std::make_pair(-R.cached.pos, k) : std::make_pair(-k, k));
    std::push_heap(b->heap.begin(), b->heap.end());
  }
  const Record& c = b->current;
  rec->name = c.name.c_str(); rec->file_index = k;
  rec->ref_id = c.ref_id; rec->pos = c.pos; rec->end_pos = c.end_pos; rec->mapq = c.mapq; rec->flag = c.flag;
  rec->mate_ref_id = c.mate_ref_id; rec->mate_pos = c.mate_pos; rec->tlen = c.tlen; rec->length = c.l_seq;
  rec->bases = c.bases.c_str(); rec->quals = c.quals.c_str();
  rec->n_cigar = c.n_cigar; rec->cigar_type = c.cigar_type.c_str(); rec->cigar_num = c.cigar_num.data();
  rec->aux = c.raw.data() + c.aux_off; rec->aux_len = (int32_t)(c.raw.size() - c.aux_off);
  return 1;
  } catch (const std::bad_alloc&) { return LTR_ERR_NOMEM; } catch (const std::exception&) { return LTR_ERR_INVALID; }
}

// bam_aux_get + the typed readers behind GetIntTag / GetFloatTag / GetStringTag / GetCharTag (bam_io.h:182-212):
// returns 1 when the tag exists with a matching type, 0 otherwise
static const uint8_t* aux_find(const ltr_bam_record* rec, const char tag[2]) {
  if (!rec || !rec->aux || !tag) return nullptr;
  const uint8_t* p = rec->aux; const uint8_t* e = p + rec->aux_len;
  while (p + 3 <= e) {
    const uint8_t* val = p + 2;
    const char t = (char)val[0];
    size_t len = 0;
    switch (t) {
      case 'A': case 'c': case 'C': len = 1; break;
      case 's': case 'S': len = 2; break;
      case 'i': case 'I': case 'f': len = 4; break;
      case 'Z': case 'H': { const uint8_t* q = val + 1; while (q < e && *q) ++q; len = (size_t)(q - (val + 1)) + 1; break; }
      case 'B': {
        if (val + 6 > e) return nullptr;
        const char st = (char)val[1]; const uint32_t n = le32(val + 2);
        const size_t w = (st == 'c' || st == 'C') ? 1 : ((st == 's' || st == 'S') ? 2 : 4);
        len = 5 + (size_t)n * w; break;
      }
      default: return nullptr;
    }
    if (val + 1 + len > e) return nullptr;
    if (p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1]) return val;
    p = val + 1 + len;
  }
  return nullptr;
}
int ltr_bam_aux_int(const ltr_bam_record* rec, const char tag[2], int64_t* value) {
  const uint8_t* v = aux_find(rec, tag);
  if (!v || !value) return 0;
  switch ((char)v[0]) {
    case 'c': *value = (int8_t)v[1]; return 1;
    case 'C': *value = v[1]; return 1;
    case 's': *value = (int16_t)le16(v + 1); return 1;
    case 'S': *value = le16(v + 1); return 1;
    case 'i': *value = (int32_t)le32(v + 1); return 1;
    case 'I': *value = le32(v + 1); return 1;
    default: return 0;
  }
}
int ltr_bam_aux_float(const ltr_bam_record* rec, const char tag[2], double* value) {
  const uint8_t* v = aux_find(rec, tag);
  if (!v || !value || (char)v[0] != 'f') return 0;
  float f; const uint32_t bits = le32(v + 1); std::memcpy(&f, &bits, 4);
  *value = f;
  return 1;
}
int ltr_bam_aux_char(const ltr_bam_record* rec, const char tag[2], char* value) {
  const uint8_t* v = aux_find(rec, tag);
  if (!v || !value || (char)v[0] != 'A') return 0;
  *value = (char)v[1];
  return 1;
}
const char* ltr_bam_aux_string(const ltr_bam_record* rec, const char tag[2]) {
  const uint8_t* v = aux_find(rec, tag);
  return (v && ((char)v[0] == 'Z' || (char)v[0] == 'H')) ? (const char*)v + 1 : nullptr;
}

}  // extern "C"

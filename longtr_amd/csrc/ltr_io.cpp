// ltr_io.cpp -- the on-disk formats of the path that need no htslib: the region (BED)
// reader, the indexed-FASTA reader and the position-ordered, BGZF-compressed VCF writer.  Host code behind
// the C-ABI of include/ltr_gpu.h.  BAM / CRAM input stays with the host program (htslib); the alignments it
// yields enter through ltr_left_align_reads / ltr_calc_hap_aln_probs.
//
// Replaces:
//   readRegions, orderRegions, Region::computePeriod / period_str     reference src/region.cpp:17-69, src/region.h:16-94
//   FastaReader (one indexed FASTA file or a directory of *.fa)        src/fasta_reader.h:25-118, src/fasta_reader.cpp:10-95
//     on top of htslib's faidx (fai_load / faidx_fetch_seq / faidx_seq_len: the published .fai format --
//     NAME LENGTH OFFSET LINEBASES LINEWIDTH -- and its clamping of a fetch to the sequence)
//   VCFWriter (heap of records by position, MAX_RECORD_PAD = 50)       src/vcf_writer.h:25-84, src/vcf_writer.cpp:3-36
//     on top of bgzfostream (src/bgzf_streams.h): BGZF = gzip members of <= 64 KB with a 'BC' extra field and
//     the 28-byte end-of-file block (the SAM specification, section 4.1)
// The reference ends the process on a malformed input (printErrorAndDie, src/error.cpp:6-10); these entry
// points return LTR_ERR_INVALID and hand the same message back.

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "ltr_internal.h"

namespace {

void put_error(char* err, int cap, const std::string& msg) {
  if (!err || cap <= 0) return;
  const size_t n = std::min(msg.size(), (size_t)cap - 1);
  std::memcpy(err, msg.data(), n);
  err[n] = 0;
}

// ---- regions -------------------------------------------------------------------------------------------
struct Region {
  std::string chrom, name, motifs;
  int32_t start = 0, stop = 0;
  int period = -1;
  std::string period_str;
  bool operator<(const Region& r) const {                       // region.h:86-90
    if (chrom != r.chrom) return chrom < r.chrom;
    if (start != r.start) return start < r.start;
    return stop < r.stop;
  }
};

// Region::splitMotifs (region.h:22-30): std::getline on ',' -- a trailing comma adds no empty item, a leading one does
std::vector<std::string> split_motifs(const std::string& motifs) {
  std::vector<std::string> out;
  std::stringstream ss(motifs);
  std::string item;
  while (std::getline(ss, item, ',')) out.push_back(item);
  return out;
}

void finish_region(Region& r) {
  const std::vector<std::string> list = split_motifs(r.motifs);
  std::set<int> periods;                                        // computePeriod, region.h:32-39
  for (const std::string& m : list) periods.insert((int)m.size());
  r.period = (periods.size() == 1) ? *periods.begin() : -1;
  std::ostringstream oss;                                       // period_str, region.h:63-71
  for (size_t i = 0; i < list.size(); ++i) { if (i > 0) oss << ","; oss << list[i].size(); }
  r.period_str = oss.str();
}

bool valid_motif(const std::string& motif) {                    // isValidMotif, region.cpp:17-24
  for (char ch : motif) if (!std::isalpha((unsigned char)ch) && ch != ',') return false;
  return true;
}

}  // namespace

struct ltr_region_set { std::vector<Region> regions; int32_t lines = 0; };

// ---- FASTA ---------------------------------------------------------------------------------------------
namespace {
struct FaiEntry { int64_t len = 0, offset = 0, line_bases = 0, line_width = 0; int file = 0; };
}
struct ltr_fasta {
  std::vector<std::string> paths;
  std::vector<FILE*> files;
  std::map<std::string, FaiEntry> index;
  std::vector<std::string> order;                               // sequence names in index order, file by file
  ~ltr_fasta() { for (FILE* f : files) if (f) std::fclose(f); }
};

namespace {

bool file_exists(const std::string& p) { return access(p.c_str(), F_OK) != -1; }
bool is_file(const std::string& p) { struct stat st; if (stat(p.c_str(), &st) != 0) return false; return S_ISREG(st.st_mode); }

// FastaReader::add_index (fasta_reader.cpp:10-40)
int add_fasta(ltr_fasta* fa, const std::string& path, std::string* err) {
  if (!file_exists(path)) { *err = "FASTA file " + path + " does not exist"; return LTR_ERR_INVALID; }
  if (!file_exists(path + ".fai")) {
    *err = "No FASTA index file exists for " + path + "\nPlease rerun the analysis after generating the index using the command:\n\tsamtools faidx " + path + "\n";
    return LTR_ERR_INVALID;
  }
  std::ifstream idx((path + ".fai").c_str());
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!idx.is_open() || !f) { if (f) std::fclose(f); *err = "Failed to load FASTA index file for " + path; return LTR_ERR_INVALID; }
  const int file_no = (int)fa->files.size();
  fa->files.push_back(f); fa->paths.push_back(path);
  std::string line;
  while (std::getline(idx, line)) {
    if (line.empty()) continue;
    std::istringstream iss(line);
    std::string name; FaiEntry e;
    std::getline(iss, name, '\t');
    if (!(iss >> e.len >> e.offset >> e.line_bases >> e.line_width) || e.line_bases <= 0 || e.line_width < e.line_bases || e.len < 0) {
      *err = "Failed to load FASTA index file for " + path; return LTR_ERR_INVALID;
    }
    e.file = file_no;
    if (fa->index.count(name)) { *err = "Multiple entries for chromosome " + name + " exist in FASTA files"; return LTR_ERR_INVALID; }
    fa->index[name] = e; fa->order.push_back(name);
  }
  return LTR_OK;
}

// ---- BGZF ------------------------------------------------------------------------------------------------
constexpr size_t kBgzfBlock = 0xff00;                           // uncompressed bytes per block (htslib's BGZF_BLOCK_SIZE)
const uint8_t kBgzfEof[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

bool bgzf_write_block(FILE* f, const uint8_t* data, size_t n) {
  uint8_t out[0x10000];
  z_stream zs; std::memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
  zs.next_in = const_cast<uint8_t*>(data); zs.avail_in = (uInt)n;
  zs.next_out = out + 18; zs.avail_out = sizeof(out) - 18 - 8;
  const int rc = deflate(&zs, Z_FINISH);
  const size_t clen = zs.total_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END) return false;                         // (0xff00 bytes always fit: deflate's worst case adds 5 bytes per 16 KB)
  const size_t total = 18 + clen + 8;
  const uint8_t head[18] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0,
                            (uint8_t)((total - 1) & 0xff), (uint8_t)((total - 1) >> 8)};
  std::memcpy(out, head, 18);
  const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n);
  uint8_t* tail = out + 18 + clen;
  for (int k = 0; k < 4; ++k) { tail[k] = (uint8_t)(crc >> (8 * k)); tail[4 + k] = (uint8_t)((uint32_t)n >> (8 * k)); }
  return std::fwrite(out, 1, total, f) == total;
}

struct Record { int32_t pos; std::string text; };

}  // namespace

struct ltr_vcf_writer {
  FILE* f = nullptr;
  bool bgzf = false, open = false, failed = false;
  std::vector<uint8_t> buf;                                     // pending uncompressed bytes (BGZF)
  std::string chrom;
  std::vector<Record> heap;                                     // std::push_heap / pop_heap with r1.pos > r2.pos: smallest position on top
  static bool after(const Record& a, const Record& b) { return a.pos > b.pos; }   // tuple_comparator, vcf_writer.cpp:3-5
  void put(const std::string& s) {
    if (!bgzf) { if (std::fwrite(s.data(), 1, s.size(), f) != s.size()) failed = true; return; }
    buf.insert(buf.end(), s.begin(), s.end());
    size_t done = 0;
    while (buf.size() - done >= kBgzfBlock) { if (!bgzf_write_block(f, buf.data() + done, kBgzfBlock)) failed = true; done += kBgzfBlock; }
    if (done) buf.erase(buf.begin(), buf.begin() + (long)done);
  }
  void write_all() {                                            // write_all_records, vcf_writer.h:38-45
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end(), after);
      put(heap.back().text); put("\n");
      heap.pop_back();
    }
  }
};

extern "C" {

// ---- regions -------------------------------------------------------------------------------------------
int ltr_read_regions(const char* path, uint32_t max_regions, const char* chrom_limit, ltr_region_set** out, char* err, int err_cap) {
  if (!path || !out) return LTR_ERR_INVALID;
  *out = nullptr;
  std::ifstream input(path);
  if (!input.is_open()) { put_error(err, err_cap, "Failed to open region file"); return LTR_ERR_INVALID; }
  const std::string limit = chrom_limit ? chrom_limit : "";
  std::unique_ptr<ltr_region_set> rs(new ltr_region_set());
  const std::string lead = "Improperly formatted region file. \n";
  std::string line;
  while (std::getline(input, line) && rs->regions.size() < max_regions) {       // region.cpp:35
    rs->lines++;
    std::istringstream iss(line);
    Region r; std::string name;
    int32_t start, stop;
    if (!(iss >> r.chrom >> start >> stop >> r.motifs)) {
      put_error(err, err_cap, "Improperly formatted region file. \nRequired format is tab-delimited columns CHROM START STOP MOTIF\n Bad line: " + line);
      return LTR_ERR_INVALID;
    }
    if (start < 1) { put_error(err, err_cap, lead + " Region has a START < 1, but START must be >= 1\n Bad line: " + line); return LTR_ERR_INVALID; }
    if (stop <= start) { put_error(err, err_cap, lead + " Region has a STOP <= START. Bad line: " + line); return LTR_ERR_INVALID; }
    if (r.motifs.size() < 1) { put_error(err, err_cap, lead + " Region has a MOTIF with size < 1. Bad line: " + line); return LTR_ERR_INVALID; }
    if (!valid_motif(r.motifs)) { put_error(err, err_cap, lead + " Region has a MOTIF with invalid character. Bad line: " + line); return LTR_ERR_INVALID; }
    if (!limit.empty() && r.chrom.compare(limit) != 0) continue;
    if (iss >> name) r.name = name;
    r.start = start - 1; r.stop = stop;                          // Region(chrom, start-1, stop, ...), region.cpp:53-56
    finish_region(r);
    rs->regions.push_back(std::move(r));
  }
  if (!limit.empty() && rs->regions.empty()) {
    put_error(err, err_cap, std::string("Region file ") + path + " did not contain any regions on the requested chromosome: " + limit);
    return LTR_ERR_INVALID;
  }
  *out = rs.release();
  return LTR_OK;
}

int64_t ltr_region_set_size(const ltr_region_set* rs) { return rs ? (int64_t)rs->regions.size() : 0; }
int32_t ltr_region_set_lines_read(const ltr_region_set* rs) { return rs ? rs->lines : 0; }
void ltr_region_set_order(ltr_region_set* rs) { if (rs) std::sort(rs->regions.begin(), rs->regions.end()); }   // orderRegions, region.cpp:67-69
void ltr_region_set_free(ltr_region_set* rs) { delete rs; }
#define LTR_REGION_FIELD(expr, dflt) (rs && i >= 0 && i < (int64_t)rs->regions.size()) ? (expr) : (dflt)
const char* ltr_region_chrom(const ltr_region_set* rs, int64_t i) { return LTR_REGION_FIELD(rs->regions[(size_t)i].chrom.c_str(), nullptr); }
const char* ltr_region_name(const ltr_region_set* rs, int64_t i) { return LTR_REGION_FIELD(rs->regions[(size_t)i].name.c_str(), nullptr); }
const char* ltr_region_motif(const ltr_region_set* rs, int64_t i) { return LTR_REGION_FIELD(rs->regions[(size_t)i].motifs.c_str(), nullptr); }
const char* ltr_region_period_str(const ltr_region_set* rs, int64_t i) { return LTR_REGION_FIELD(rs->regions[(size_t)i].period_str.c_str(), nullptr); }
int32_t ltr_region_start(const ltr_region_set* rs, int64_t i) { return LTR_REGION_FIELD(rs->regions[(size_t)i].start, -1); }
int32_t ltr_region_stop(const ltr_region_set* rs, int64_t i) { return LTR_REGION_FIELD(rs->regions[(size_t)i].stop, -1); }
int32_t ltr_region_period(const ltr_region_set* rs, int64_t i) { return LTR_REGION_FIELD(rs->regions[(size_t)i].period, -1); }
#undef LTR_REGION_FIELD

// ---- FASTA ---------------------------------------------------------------------------------------------
// FastaReader::init (fasta_reader.cpp:42-68): one indexed file, or every *.fa of a directory
int ltr_fasta_open(const char* path, ltr_fasta** out, char* err, int err_cap) {
  if (!path || !out) return LTR_ERR_INVALID;
  *out = nullptr;
  std::unique_ptr<ltr_fasta> fa(new ltr_fasta());
  std::string msg;
  const std::string p = path;
  if (is_file(p)) {
    const int rc = add_fasta(fa.get(), p, &msg);
    if (rc != LTR_OK) { put_error(err, err_cap, msg); return rc; }
  } else {
    DIR* dir = opendir(path);
    if (!dir) { put_error(err, err_cap, "Failed to access directory " + p); return LTR_ERR_INVALID; }
    std::vector<std::string> names;
    while (struct dirent* de = readdir(dir)) {
      const std::string fn = de->d_name;
      if (fn.size() >= 3 && fn.compare(fn.size() - 3, 3, ".fa") == 0) names.push_back(fn);       // string_ends_with(filename, ".fa")
    }
    closedir(dir);
    for (const std::string& fn : names) {
      const int rc = add_fasta(fa.get(), p + "/" + fn, &msg);
      if (rc != LTR_OK) { put_error(err, err_cap, msg); return rc; }
    }
    if (fa->files.empty()) { put_error(err, err_cap, "Failed to locate any FASTA files in the provided directory: \n\t" + p); return LTR_ERR_INVALID; }
  }
  *out = fa.release();
  return LTR_OK;
}
void ltr_fasta_close(ltr_fasta* fa) { delete fa; }
int64_t ltr_fasta_num_seqs(const ltr_fasta* fa) { return fa ? (int64_t)fa->order.size() : 0; }
const char* ltr_fasta_seq_name(const ltr_fasta* fa, int64_t i) { return (fa && i >= 0 && i < (int64_t)fa->order.size()) ? fa->order[(size_t)i].c_str() : nullptr; }
// get_sequence_length (fasta_reader.h:104-110): -1 for an unknown name
int64_t ltr_fasta_seq_len(const ltr_fasta* fa, const char* chrom) {
  if (!fa || !chrom) return -1;
  auto it = fa->index.find(chrom);
  return it == fa->index.end() ? -1 : it->second.len;
}
// get_sequence(chrom, start, end) (fasta_reader.h:85-101): 0-based, END INCLUSIVE; like faidx_fetch_seq the range is
// clamped to the sequence (end >= length -> length - 1, start < 0 -> 0, start > end -> empty).  Bytes are returned as
// stored (no case change), line breaks skipped.  Returns the number of bases written, or a negative status.
int64_t ltr_fasta_fetch(ltr_fasta* fa, const char* chrom, int64_t start, int64_t end, char* out, int64_t cap, char* err, int err_cap) {
  if (!fa || !chrom || (!out && cap > 0)) return LTR_ERR_INVALID;
  auto it = fa->index.find(chrom);
  if (it == fa->index.end()) { put_error(err, err_cap, std::string("No entry for chromosome ") + chrom + " found in FASTA files"); return LTR_ERR_INVALID; }
  const FaiEntry& e = it->second;
  if (start < 0) start = 0;
  if (end >= e.len) end = e.len - 1;
  if (start > end) return 0;
  const int64_t n = end - start + 1;
  if (n > cap) return LTR_ERR_INVALID;
  FILE* f = fa->files[(size_t)e.file];
  const int64_t first = e.offset + (start / e.line_bases) * e.line_width + start % e.line_bases;
  if (fseeko(f, (off_t)first, SEEK_SET) != 0) return LTR_ERR_INVALID;
  int64_t got = 0;
  char chunk[65536];
  while (got < n) {
    const size_t want = (size_t)std::min<int64_t>((int64_t)sizeof(chunk), (n - got) + (n - got) / e.line_bases * (e.line_width - e.line_bases) + e.line_width);
    const size_t rd = std::fread(chunk, 1, want, f);
    if (rd == 0) break;
    for (size_t k = 0; k < rd && got < n; ++k) if (std::isgraph((unsigned char)chunk[k])) out[got++] = chunk[k];    // (faidx keeps isgraph bytes)
  }
  return got;
}
// write_all_contigs_to_vcf (fasta_reader.cpp:70-79): "##contig=<ID=name,length=len>\n" per sequence, in index order
int64_t ltr_fasta_contig_lines(const ltr_fasta* fa, char* out, int64_t cap) {
  if (!fa) return LTR_ERR_INVALID;
  std::ostringstream oss;
  for (const std::string& name : fa->order) oss << "##contig=<ID=" << name << ",length=" << fa->index.at(name).len << ">" << "\n";
  const std::string s = oss.str();
  if ((int64_t)s.size() + 1 > cap || !out) return LTR_ERR_INVALID;
  std::memcpy(out, s.c_str(), s.size() + 1);
  return (int64_t)s.size();
}

// ---- VCF writer ----------------------------------------------------------------------------------------
// VCFWriter::open (vcf_writer.h:63-68).  The reference always writes BGZF (bgzfostream); here a path ending in ".gz" or
// ".bgz" does, any other path gets the same text uncompressed.
int ltr_vcf_writer_open(const char* path, ltr_vcf_writer** out) {
  if (!path || !out) return LTR_ERR_INVALID;
  *out = nullptr;
  FILE* f = std::fopen(path, "wb");
  if (!f) return LTR_ERR_INVALID;
  ltr_vcf_writer* w = new ltr_vcf_writer();
  const std::string p = path;
  w->f = f; w->open = true;
  w->bgzf = (p.size() >= 3 && p.compare(p.size() - 3, 3, ".gz") == 0) || (p.size() >= 4 && p.compare(p.size() - 4, 4, ".bgz") == 0);
  *out = w;
  return LTR_OK;
}
int ltr_vcf_writer_header(ltr_vcf_writer* w, const char* text) {          // write_header, vcf_writer.h:70-74
  if (!w || !w->open || !text) return LTR_ERR_INVALID;
  w->put(text);
  return w->failed ? LTR_ERR_INVALID : LTR_OK;
}
// add_vcf_record (vcf_writer.cpp:7-36): records may arrive up to MAX_RECORD_PAD = 50 bp out of order inside a chromosome
int ltr_vcf_writer_add_record(ltr_vcf_writer* w, const char* chrom, int32_t record_pos, const char* record_text) {
  if (!w || !w->open || !chrom || !record_text) return LTR_ERR_INVALID;
  constexpr int32_t kMaxRecordPad = 50;
  if (w->chrom.compare(chrom) != 0) {
    w->write_all();
    w->chrom = chrom;
  } else {
    while (!w->heap.empty()) {
      std::pop_heap(w->heap.begin(), w->heap.end(), ltr_vcf_writer::after);
      if (w->heap.back().pos < record_pos - kMaxRecordPad) {
        w->put(w->heap.back().text); w->put("\n");
        w->heap.pop_back();
      } else {
        std::push_heap(w->heap.begin(), w->heap.end(), ltr_vcf_writer::after);
        break;
      }
    }
  }
  w->heap.push_back({record_pos, record_text});
  std::push_heap(w->heap.begin(), w->heap.end(), ltr_vcf_writer::after);
  return w->failed ? LTR_ERR_INVALID : LTR_OK;
}
int ltr_vcf_writer_close(ltr_vcf_writer* w) {                              // close, vcf_writer.h:78-82 (+ the destructor)
  if (!w) return LTR_ERR_INVALID;
  int rc = LTR_OK;
  if (w->open) {
    w->write_all();
    if (w->bgzf) {
      size_t done = 0;
      while (done < w->buf.size()) { const size_t n = std::min(kBgzfBlock, w->buf.size() - done); if (!bgzf_write_block(w->f, w->buf.data() + done, n)) w->failed = true; done += n; }
      if (std::fwrite(kBgzfEof, 1, sizeof(kBgzfEof), w->f) != sizeof(kBgzfEof)) w->failed = true;
    }
    if (std::fclose(w->f) != 0) w->failed = true;
    if (w->failed) rc = LTR_ERR_INVALID;
  }
  delete w;
  return rc;
}

}  // extern "C"

// mm_reads.cpp — host side: BAM / SAM -> alignment tuples (mm_aln + NH) in file order.
// Replaces Reader / SamReader / BamReader and Read::parseCigar of the reference
// (mmannot.cpp:846-903, 1339-1650).  The tuple semantics of SURVEY.md section 8a row 2 are reproduced:
//   start = pos0+1; end = start + sum(M,D,=,X) - 1 (N and unknown operations add nothing and are reported);
//   a record without CIGAR operations gets end = start-1, a SAM '*' gets end = start;
//   NH is honoured only when stored in an unsigned BAM type (C/S/I), signed types give 0; default 1;
//   an XA tag replays its equal-NM alternatives as further reads and sets nHits = alternatives+1;
//   the unmapped flag is ignored; tid -1 is the chromosome "*".
// Read strand: the reference's XamRecord::setFlags ignores its argument, which makes its -s F/R
// undefined.  Default here = the intended meaning (flag 0x10 clear -> '+'); MM_STRAND_COMPAT_* reproduce what a
// given verbatim run did (every record '+', or every record '-').
#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mmannot_amd.h"
#include "mm_hostutil.h"

using namespace mmh;

namespace {

struct CigarOp { char op; int len; };

struct AltHit {
  std::string chromosome;
  bool strand;
  uint64_t start;
  std::vector<CigarOp> cigar;
};

struct ByteSource {  // sequential reader over gzread (plain gzip, BGZF and uncompressed input alike)
  gzFile f = nullptr;
  std::vector<uint8_t> buf;
  size_t pos = 0, len = 0;
  bool eof = false;
  bool open(const char *path) {
    f = gzopen(path, "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    buf.resize(4 << 20);
    return true;
  }
  void close() { if (f) { gzclose(f); f = nullptr; } }
  bool fill() {
    if (eof) return false;
    const int n = gzread(f, buf.data(), (unsigned) buf.size());
    if (n <= 0) { eof = true; len = pos = 0; return false; }
    len = (size_t) n; pos = 0;
    return true;
  }
  // false when fewer than n bytes remain
  bool read(void *dst, size_t n) {
    uint8_t *d = (uint8_t *) dst;
    while (n) {
      if (pos == len && !fill()) return false;
      const size_t k = std::min(n, len - pos);
      if (d) { memcpy(d, buf.data() + pos, k); d += k; }
      pos += k; n -= k;
    }
    return true;
  }
  bool skip(size_t n) { return read(nullptr, n); }
  int getc() {
    if (pos == len && !fill()) return -1;
    return buf[pos++];
  }
};

}  // namespace

struct mm_reader {
  std::string path;
  bool is_bam = false;
  ByteSource src;
  std::ifstream sam;
  std::vector<std::string> bam_chr;
  const mm_annotation *an = nullptr;
  int strand_compat = MM_STRAND_COMPAT_NONE;
  bool unique_only = false;
  bool keep_names = false;
  // current record
  std::string name, chromosome;
  uint64_t rec_start = 0;
  uint32_t nhits = 1, nmismatches = 0;  // nMismatches deliberately survives from record to record (never reset by the reference)
  bool rec_strand = true;
  std::vector<CigarOp> cigar;
  std::vector<AltHit> alts;
  size_t alt_next = 0;
  bool have_record = false;
  bool over = false;
  // dictionaries
  std::unordered_map<std::string, uint32_t> qids;
  std::vector<std::string> names;
  std::unordered_map<std::string, int32_t> chr_tid;  // read chromosome name -> annotation chromosome id (-1 unknown)
  std::vector<std::string> unknown_warned;
  uint64_t n_reads = 0;
  uint64_t cur_end = 0;   // Read::end of the read delivered last (a '*' CIGAR of an alternative keeps it)
  std::string log, err;
  std::string log_out;
};

static void parse_cigar_text(const std::string &s, std::vector<CigarOp> &out) {
  int value = 0;
  for (char c : s) {
    if (c >= '0' && c <= '9') value = value * 10 + (c - '0');
    else { out.push_back({c, value}); value = 0; }
  }
}

// Read::parseCigar: returns the end coordinate; complaints go to the log
static uint64_t cigar_end(const std::vector<CigarOp> &cg, uint64_t start, uint64_t prev_end, std::string &log) {
  if (cg.size() == 1 && cg[0].op == '*') return prev_end;
  uint64_t end = start;
  for (const CigarOp &o : cg) {
    switch (o.op) {
      case 'M': case 'D': case '=': case 'X': end += (uint64_t) (int64_t) o.len; break;
      case 'I': case 'S': case 'H': case 'P': break;
      default: log += std::string("Problem in the cigar: do not understand char ") + o.op + "\n";
    }
  }
  return end - 1;
}

static void parse_xa(mm_reader *r, const std::string &xa) {
  if (xa == "0") return;
  std::stringstream all(xa);
  std::string one;
  while (std::getline(all, one, ';')) {
    if (one.empty()) continue;
    AltHit ah;
    ah.strand = true; ah.start = 0;
    try {
      std::vector<std::string> f;
      split_all(one, ',', f);
      if (f.empty()) f.push_back("");
      auto field = [&](size_t k) -> const std::string & { return f[std::min(k, f.size() - 1)]; };  // a missing field repeats the last one read
      ah.chromosome = field(0);
      ah.strand = (!field(1).empty() && field(1)[0] == '+');
      ah.start = std::stoul(field(1).substr(1));
      const std::string cg = field(2);
      const unsigned long nm = std::stoul(field(3));
      if ((uint32_t) nm == r->nmismatches) {
        parse_cigar_text(cg, ah.cigar);
        r->alts.push_back(ah);
      }
    } catch (...) {
      r->log += "Warning!  Problem while parsing an \"XA\" tag, which is probably too long:\n" + xa + "\n";
    }
  }
  r->nhits = (uint32_t) r->alts.size() + 1;
}

static bool next_sam_record(mm_reader *r) {
  std::string line;
  do {
    if (!std::getline(r->sam, line)) return false;
  } while (line.empty() || line[0] == '@' || line[0] == '#');
  std::vector<std::string> col;
  split_all(line, '\t', col);
  if (col.size() < 12) { r->err = "SAM line with fewer than 12 columns (the reference asserts here, mmannot.cpp:1451): " + line; return false; }
  r->cigar.clear(); r->alts.clear();
  try {
    const int flags = std::stoi(col[1]);
    r->rec_strand = (flags & 0x10) == 0;
    r->chromosome = col[2];
    r->rec_start = std::stoul(col[3]);
    r->name = col[0];
    r->nhits = 1;
    parse_cigar_text(col[5], r->cigar);
    for (size_t i = 11; i < col.size(); ++i) {
      const std::string &part = col[i];
      const size_t p = part.find(':');
      const std::string key = part.substr(0, p);
      if (key == "NH") {
        if (r->alts.empty()) r->nhits = (uint32_t) std::stoul(part.substr(part.find(':', p + 1) + 1));
      } else if (key == "NM") {
        r->nmismatches = (uint32_t) std::stoul(part.substr(part.find(':', p + 1) + 1));
      } else if (key == "XA") {
        parse_xa(r, part.substr(part.find(':', p + 1) + 1));
      }
    }
  } catch (...) {
    r->err = "cannot parse SAM line (the reference terminates on the uncaught exception): " + line;
    return false;
  }
  return true;
}

static bool next_bam_record(mm_reader *r) {
  ByteSource &s = r->src;
  int32_t size;
  if (!s.read(&size, 4)) return false;
  int32_t chr_id, pos, l_seq;
  uint32_t bin_mq_nl, flag_nc;
  r->cigar.clear(); r->alts.clear();
  if (!s.read(&chr_id, 4) || !s.read(&pos, 4) || !s.read(&bin_mq_nl, 4) || !s.read(&flag_nc, 4) || !s.read(&l_seq, 4) || !s.skip(12)) return false;
  if (chr_id == -1) r->chromosome = "*";
  else if (chr_id >= 0 && (size_t) chr_id < r->bam_chr.size()) r->chromosome = r->bam_chr[chr_id];
  else { r->err = "BAM record refers to reference sequence " + std::to_string(chr_id) + " which the header does not declare"; return false; }
  r->rec_start = (uint64_t) (int64_t) (pos + 1);
  const int32_t l_name = (int32_t) (bin_mq_nl & 0xff), n_cigar = (int32_t) (flag_nc & 0xffff);
  const uint32_t flags = flag_nc >> 16;
  r->rec_strand = (flags & 0x10) == 0;
  std::string nm((size_t) l_name, '\0');
  if (l_name && !s.read(&nm[0], (size_t) l_name)) return false;
  r->name = std::string(nm.c_str());
  static const char OPS[] = "MIDNSHP=X";
  for (int32_t i = 0; i < n_cigar; ++i) {
    uint32_t v;
    if (!s.read(&v, 4)) return false;
    const uint32_t k = v & 15u;
    r->cigar.push_back({k <= 9 ? OPS[k] : '?', (int) (v >> 4)});
  }
  if (!s.skip((size_t) ((l_seq + 1) / 2)) || !s.skip((size_t) l_seq)) return false;
  r->nhits = 1;
  std::string v_s;
  for (int32_t i = 33 + l_name + 4 * n_cigar + (l_seq + 1) / 2 + l_seq; i < size;) {
    char key[3] = {0, 0, 0}, c;
    if (!s.read(key, 2) || !s.read(&c, 1)) return false;
    i += 3;
    int8_t n = 1;
    uint32_t v_u32 = 0;
    if (c == 'H') {                       // one count byte, then that many 'C' values
      if (!s.read(&n, 1)) return false;
      c = 'C'; ++i;
    } else if (c == 'B') {                // sub-type, then a count assembled in 8-bit arithmetic from 4 bytes
      int8_t sb = 0, m = 1;
      if (!s.read(&c, 1)) return false;
      n = 0;
      for (int j = 0; j < 4; ++j) {
        if (!s.read(&sb, 1)) return false;
        n = (int8_t) (n + sb * m);
        m = (int8_t) (m * 16);
      }
      i += 5;
    }
    for (int j = 0; j < n; ++j) {
      switch (c) {
        case 'A': case 'c': if (!s.skip(1)) return false; i += 1; break;
        case 'C': { uint8_t b; if (!s.read(&b, 1)) return false; v_u32 = (v_u32 & ~0xffu) | b; i += 1; break; }
        case 's': if (!s.skip(2)) return false; i += 2; break;
        case 'S': { uint16_t h; if (!s.read(&h, 2)) return false; v_u32 = (v_u32 & ~0xffffu) | h; i += 2; break; }
        case 'i': case 'f': if (!s.skip(4)) return false; i += 4; break;
        case 'I': if (!s.read(&v_u32, 4)) return false; i += 4; break;
        case 'Z': {
          v_s.clear();
          int ch;
          while ((ch = s.getc()) > 0) { v_s.push_back((char) ch); ++i; }
          if (ch < 0) return false;
          ++i;
          break;
        }
        default:
          r->log += std::string("Problem with tag type '") + c + "'\n";
          return true;                    // the reference gives up on the record here and keeps reading from this byte
      }
    }
    if (key[0] == 'N' && key[1] == 'H') { if (r->alts.empty()) r->nhits = v_u32; }
    else if (key[0] == 'N' && key[1] == 'M') r->nmismatches = v_u32;
    else if (key[0] == 'X' && key[1] == 'A') parse_xa(r, v_s);
  }
  return true;
}

static int32_t tid_of(mm_reader *r, const std::string &chr, bool reaches_scan) {
  auto it = r->chr_tid.find(chr);
  int32_t tid;
  if (it == r->chr_tid.end()) {
    tid = -1;
    if (r->an) {
      const int32_t nc = mm_annotation_n_chromosomes(r->an);
      for (int32_t c = 0; c < nc; ++c)
        if (chr == mm_annotation_chromosome(r->an, c)) { tid = mm_annotation_chromosome_has_intervals(r->an, c) ? c : -1; break; }
    }
    r->chr_tid.emplace(chr, tid);
  } else tid = it->second;
  if (tid < 0 && reaches_scan) {          // IntervalList::scan warns once per unknown name, never for "*" (mmannot.cpp:1295-1300)
    bool seen = false;
    for (const std::string &u : r->unknown_warned) if (u == chr) { seen = true; break; }
    if (!seen) {
      if (chr != "*") r->log += "\t\tWarning!  Chromosome '" + chr + "' (found in your reads) is not present in your annotation file.\n";
      r->unknown_warned.push_back(chr);
    }
  }
  return tid;
}

extern "C" int mm_reader_open(mm_reader **out, const char *path, int format, const mm_annotation *an, int strand_compat,
                              int unique_only, int keep_names) {
  if (!out || !path) return MM_ERR_ARG;
  *out = nullptr;
  mm_reader *r = new mm_reader();
  r->path = path; r->an = an; r->strand_compat = strand_compat; r->unique_only = unique_only != 0; r->keep_names = keep_names != 0;
  *out = r;
  bool bam;
  if (format == MM_FMT_BAM) bam = true;
  else if (format == MM_FMT_SAM) bam = false;
  else {
    std::string suf = r->path.size() >= 4 ? r->path.substr(r->path.size() - 4) : "";
    for (char &ch : suf) ch = (char) std::tolower((unsigned char) ch);
    if (suf == ".bam") bam = true;
    else if (suf == ".sam") bam = false;
    else { r->err = "Cannot deduce type from file name '" + r->path + "'.  Should be a .sam or .bam file.  Please specify it using the '-f' option."; return MM_ERR_ARG; }
  }
  r->is_bam = bam;
  { std::ifstream t(path); if (!t.good()) { r->err = "Error, file '" + r->path + "' does not exists!"; return MM_ERR_IO; } }
  if (bam) {
    r->log += "Reading BAM file " + r->path + "\n";
    if (!r->src.open(path)) { r->err = "Cannot open file '" + r->path + "'."; return MM_ERR_IO; }
    char magic[5] = {0, 0, 0, 0, 0};
    r->src.read(magic, 4);
    if (strcmp(magic, "BAM\1") != 0) {
      r->err = "Problem with file '" + r->path + "': file does not look like a BAM file (missing magic string, got '" + magic + "' instead of 'BAM\\1').";
      return MM_ERR_PARSE;
    }
    int32_t l_text = 0, n_ref = 0;
    if (!r->src.read(&l_text, 4) || !r->src.skip((size_t) l_text) || !r->src.read(&n_ref, 4)) { r->err = "truncated BAM header"; return MM_ERR_PARSE; }
    for (int32_t i = 0; i < n_ref; ++i) {
      int32_t l = 0;
      if (!r->src.read(&l, 4)) { r->err = "truncated BAM header"; return MM_ERR_PARSE; }
      std::string nm((size_t) l, '\0');
      if (l && !r->src.read(&nm[0], (size_t) l)) { r->err = "truncated BAM header"; return MM_ERR_PARSE; }
      r->bam_chr.push_back(std::string(nm.c_str()));
      r->src.skip(4);
    }
  } else {
    r->log += "Reading SAM file " + r->path + "\n";
    r->sam.open(path);
  }
  return MM_OK;
}

extern "C" const char *mm_reader_error(const mm_reader *r) { return r ? r->err.c_str() : "null reader"; }

// Reader::gotoNextRead: the record itself, then each of its XA alternatives, as separate reads
extern "C" int mm_reader_next(mm_reader *r, mm_aln *aln, uint32_t *nh, size_t cap, size_t *n_out) {
  if (!r || !aln || !nh || !n_out) return MM_ERR_ARG;
  size_t n = 0;
  uint64_t &cur_end = r->cur_end;
  while (n < cap && !r->over) {
    std::string chr;
    bool strand;
    uint64_t start, end;
    if (r->have_record && r->alt_next < r->alts.size()) {
      const AltHit &ah = r->alts[r->alt_next++];
      chr = ah.chromosome; strand = ah.strand; start = ah.start;
      end = cigar_end(ah.cigar, start, cur_end, r->log);
    } else {
      const bool ok = r->is_bam ? next_bam_record(r) : next_sam_record(r);
      if (!ok) {
        r->over = true;
        if (!r->err.empty()) { r->have_record = false; *n_out = n; return MM_ERR_PARSE; }
        // At end of file the reference calls read.reset() once more on the stale last record (mmannot.cpp:1414-1415,
        // 1444-1447, 1527-1532): its CIGAR is parsed again and any complaint about it is printed a second time.
        if (r->have_record) (void) cigar_end(r->cigar, r->rec_start, r->rec_start, r->log);
        r->have_record = false;
        break;
      }
      r->have_record = true; r->alt_next = 0;
      chr = r->chromosome; start = r->rec_start;
      strand = (r->strand_compat == MM_STRAND_COMPAT_PLUS) ? true : (r->strand_compat == MM_STRAND_COMPAT_MINUS ? false : r->rec_strand);
      end = cigar_end(r->cigar, start, start, r->log);
    }
    cur_end = end;
    if (start > 0x7FFFFFF0ull || (end > 0x7FFFFFF0ull && end != start - 1)) {
      r->err = "alignment coordinate outside the supported range [0, 2^31) for read " + r->name; *n_out = n; return MM_ERR_RANGE;
    }
    const bool reaches_scan = !r->unique_only || r->nhits == 1;   // mmannot.cpp:1773
    const int32_t tid = tid_of(r, chr, reaches_scan);
    auto it = r->qids.find(r->name);
    uint32_t q;
    if (it == r->qids.end()) {
      q = (uint32_t) r->qids.size();
      r->qids.emplace(r->name, q);
      if (r->keep_names) r->names.push_back(r->name);
    } else q = it->second;
    aln[n].start = (int32_t) start;
    aln[n].end = (int32_t) (int64_t) end;
    aln[n].tid_strand = MM_ALN_TID_STRAND(tid, strand);
    aln[n].qid = q;
    nh[n] = r->nhits;
    ++n; ++r->n_reads;
  }
  *n_out = n;
  return MM_OK;
}

extern "C" uint64_t mm_reader_qid_bound(const mm_reader *r) { return r ? (uint64_t) r->qids.size() : 0; }
extern "C" uint64_t mm_reader_n_reads(const mm_reader *r) { return r ? r->n_reads : 0; }
extern "C" const char *mm_reader_name(const mm_reader *r, uint32_t qid) {
  return (r && qid < r->names.size()) ? r->names[qid].c_str() : "";
}
// messages produced since the previous call, in the order the reference would have written them to stderr
extern "C" const char *mm_reader_take_log(mm_reader *r) {
  if (!r) return "";
  r->log_out.swap(r->log);
  r->log.clear();
  return r->log_out.c_str();
}
extern "C" void mm_reader_close(mm_reader *r) {
  if (!r) return;
  r->src.close();
  delete r;
}

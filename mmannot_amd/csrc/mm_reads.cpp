// mm_reads.cpp — host side: BAM / SAM -> alignment tuples (mm_aln + NH) in file order.
// Replaces Reader / SamReader / BamReader and Read::parseCigar of the reference
// (mmannot.cpp:846-903, 1339-1650).  The tuple semantics of SURVEY.md section 8a row 2 are reproduced:
//   start = pos0+1; end = start + sum(M,D,=,X) - 1 (N and unknown operations add nothing and are reported);
//   a record without CIGAR operations gets end = start-1, a SAM '*' gets end = start;
//   NH is honoured only when stored in an unsigned BAM type (C/S/I), signed types give 0; default 1;
//   an XA tag replays its equal-NM alternatives as further reads and sets nHits = alternatives+1;
//   the unmapped flag is ignored; tid -1 is the chromosome "*".
// Output: 16-byte tuples with NH packed beside chromosome and strand (include/mmannot_amd.h: mm_aln), NH values beyond the
// field in a side list.
// Read strand: the reference's XamRecord::setFlags ignores its argument, which makes its -s F/R
// undefined.  Default here = the intended meaning (flag 0x10 clear -> '+'); MM_STRAND_COMPAT_* reproduce what a
// given verbatim run did (every record '+', or every record '-').
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <thread>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mmannot_amd.h"
#include "mm_hostutil.h"
#include "mm_inflate.h"

using namespace mmh;

namespace {

struct CigarOp { char op; int len; };

// A handful of threads that stay alive between parallel steps: worker w of one step is worker w of the next, so what it left in
// its core's cache (the blocks it inflated, the sub-dictionary it owns) is still there when it comes back.
struct Pool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, done_cv;
  std::function<void(int)> job;
  int W = 0, remaining = 0;
  uint64_t gen = 0;
  bool stop = false;
  void start(int n) {
    if (W) return;
    W = n < 1 ? 1 : n;
    for (int w = 1; w < W; ++w) th.emplace_back([this, w] { loop(w); });
  }
  void loop(int w) {
    uint64_t seen = 0;
    for (;;) {
      std::function<void(int)> f;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || gen != seen; });
        if (stop) return;
        seen = gen; f = job;
      }
      f(w);
      std::lock_guard<std::mutex> lk(mu);
      if (--remaining == 0) done_cv.notify_one();
    }
  }
  // f(w) for w = 0..W-1, f(0) on the calling thread; returns when all are done
  void run(const std::function<void(int)> &f) {
    if (W <= 1) { f(0); return; }
    {
      std::lock_guard<std::mutex> lk(mu);
      job = f; remaining = W - 1; ++gen;
    }
    cv.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(mu);
    done_cv.wait(lk, [&] { return remaining == 0; });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto &t : th) t.join();
  }
};

// hash of a read name (the dictionary's, computed by whoever parses the record)
static inline uint64_t name_hash(const char *p, size_t n) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xff51afd7ed558ccdull);
  while (n >= 8) { uint64_t w; memcpy(&w, p, 8); h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 32; p += 8; n -= 8; }
  uint64_t w = 0; memcpy(&w, p, n);
  h = (h ^ w) * 0xc4ceb9fe1a85ec53ull; h ^= h >> 29;
  return h;
}

// An ordinary BAM record read in place: fields, end coordinate straight from the packed CIGAR, NH / NM from the tags, the name
// where it lies.  Anything the general parser treats specially - a CIGAR operation it complains about, tag types H / B /
// unknown, an XA tag, a malformed record - gives false, and the general parser (which carries the reference's exact
// behaviour for those cases, mmannot.cpp:1533-1650) takes the record.  p points at the record's block_size field, avail bytes
// are readable there; n_chr = number of reference names incl. the "*" appended behind the header's.
struct FastRec { uint64_t start, end; const char *name; uint32_t name_len, hash, nhits, nm, size; int32_t ref; bool strand, has_nm; };
static bool parse_fast(const uint8_t *p, size_t avail, size_t n_chr, FastRec &out) {
  if (avail < 36) return false;
  const uint8_t *b = p;
  int32_t size, chr_id, pos, l_seq;
  uint32_t bin_mq_nl, flag_nc;
  memcpy(&size, b, 4);
  if (size < 32 || avail - 4 < (size_t) size) return false;
  b += 4;
  memcpy(&chr_id, b, 4); memcpy(&pos, b + 4, 4); memcpy(&bin_mq_nl, b + 8, 4); memcpy(&flag_nc, b + 12, 4); memcpy(&l_seq, b + 16, 4);
  const int32_t l_name = (int32_t) (bin_mq_nl & 0xff), n_cigar = (int32_t) (flag_nc & 0xffff);
  if (l_seq < 0 || l_name == 0) return false;
  const int64_t off = 32 + (int64_t) l_name + 4 * (int64_t) n_cigar + ((int64_t) l_seq + 1) / 2 + (int64_t) l_seq;
  if (off > size) return false;
  int32_t ref;
  if (chr_id == -1) ref = (int32_t) n_chr - 1;
  else if (chr_id >= 0 && (size_t) chr_id + 1 < n_chr) ref = chr_id;
  else return false;
  const char *name = (const char *) (b + 32);
  const size_t name_len = strnlen(name, (size_t) l_name);
  if (name_len == (size_t) l_name) return false;            // no terminating NUL inside the name field
  const uint64_t start = (uint64_t) (int64_t) (pos + 1);
  uint64_t end = start;
  const uint8_t *cg = b + 32 + l_name;
  for (int32_t i = 0; i < n_cigar; ++i) {
    uint32_t v;
    memcpy(&v, cg + 4 * i, 4);
    switch (v & 15u) {
      case 0: case 2: case 7: case 8: end += (uint64_t) (v >> 4); break;   // M D = X   (Read::parseCigar, mmannot.cpp:852-875)
      case 1: case 4: case 5: case 6: break;                             // I S H P
      default: return false;                                             // N and the undefined codes: the general parser complains
    }
  }
  end -= 1;
  uint32_t nh = 1, nm = 0;
  bool has_nm = false;
  int64_t i = off;
  while (i + 1 < size) {                                                  // (the reference's loop bound, mmannot.cpp:1563)
    if (i + 3 > size) return false;
    const uint8_t k0 = b[i], k1 = b[i + 1], ty = b[i + 2];
    i += 3;
    uint32_t v = 0;
    switch (ty) {
      case 'A': case 'c': i += 1; break;
      case 'C': if (i + 1 > size) return false; v = b[i]; i += 1; break;
      case 's': i += 2; break;
      case 'S': { if (i + 2 > size) return false; uint16_t h; memcpy(&h, b + i, 2); v = h; i += 2; break; }
      case 'i': case 'f': i += 4; break;
      case 'I': if (i + 4 > size) return false; memcpy(&v, b + i, 4); i += 4; break;
      case 'Z': {
        if (k0 == 'X' && k1 == 'A') return false;                         // alternative hits: general parser
        const void *z = i < size ? memchr(b + i, 0, (size_t) (size - i)) : nullptr;
        if (!z) return false;
        i = (const uint8_t *) z - b + 1;
        break;
      }
      default: return false;
    }
    if (i > size) return false;
    if (k0 == 'N' && k1 == 'H') nh = v;
    else if (k0 == 'N' && k1 == 'M') { nm = v; has_nm = true; }
  }
  if (i != size) return false;                                           // a stray trailing byte: the general parser reproduces what follows
  out.start = start; out.end = end; out.name = name; out.name_len = (uint32_t) name_len; out.nhits = nh; out.nm = nm; out.has_nm = has_nm;
  out.ref = ref; out.size = (uint32_t) size;
  out.strand = ((flag_nc >> 16) & 0x10) == 0;
  out.hash = (uint32_t) name_hash(name, name_len);
  return true;
}

// One framed record as the parser threads of the BGZF pipe leave it (32 bytes)
struct ParsedRec {
  uint32_t off;              // of the record's block_size field in the batch buffer (the name follows at off + 36)
  int32_t start, end;
  uint32_t nh, nm, hash;
  int32_t ref;
  uint8_t name_len, flags;   // PR_*
  uint16_t pad;
};
enum { PR_FAST = 1, PR_STRAND = 2, PR_HAS_NM = 4 };

struct AltHit {
  std::string chromosome;
  bool strand;
  uint64_t start;
  std::vector<CigarOp> cigar;
};

// BGZF (the blocked gzip of real BAM files): every block is an independent gzip member of <= 64 KiB with its
// compressed size in a 'BC' extra field, so blocks can be inflated in parallel.  A producer thread reads a batch of
// blocks, inflates them with a few helper threads straight into one contiguous buffer, and hands the buffer over
// while the consumer is still working on the previous one (two buffers).
// With `frame` the same helper threads also PARSE: the producer walks the BAM header once, frames the records of each
// batch (block_size chain; a record cut by the end of the batch moves to the front of the next one: `prefix`) and the
// threads parse record ranges in place (parse_fast) into ParsedRec's.  What is left to the consumer's single thread is
// what has to stay in file order: chromosome warnings, the read-name dictionary (first-appearance ids) and the records
// the general parser takes.  Framing stops for good at the first impossible block_size; the bytes from there on are
// delivered raw and the consumer reads them the sequential way.
struct BgzfPipe {
  // the compressed file is mapped, not read: the inflating threads take the bytes straight from the page cache, in parallel
  const uint8_t *map = nullptr;
  size_t fsize = 0, fpos = 0;
  int fd = -1;
  int n_threads = 4;
  bool frame = false;
  bool own_inflate = !(getenv("MM_READER_ZLIB") && *getenv("MM_READER_ZLIB") && *getenv("MM_READER_ZLIB") != '0');   // (A/B and tests)
  bool check_crc = !(getenv("MM_READER_NOCRC") && *getenv("MM_READER_NOCRC") && *getenv("MM_READER_NOCRC") != '0');    // (A/B: what the CRC-32 of every block costs)
  struct Batch {
    std::vector<uint8_t> data;
    size_t len = 0;           // bytes in data (prefix included)
    size_t prefix = 0;        // leading bytes that repeat the unframed tail of the previous batch
    size_t rec_begin = 0;     // framed records cover [rec_begin, framed_end)
    size_t framed_end = 0;
    bool parsed = false;      // recs[] describe the records of [rec_begin, framed_end)
    bool tail_carried = false; // [framed_end, len) reappears as the next batch's prefix
    std::vector<ParsedRec> recs;
    size_t n_recs = 0;
    size_t n_chr = 0;         // reference names of the header + 1
    bool last = false, bad = false;
  };
  Batch slot[2];
  int ready[2] = {0, 0};       // 0 = free for the producer, 1 = filled for the consumer
  std::mutex mu;
  std::condition_variable cv;
  std::thread producer;
  Pool pool;                 // inflates, then parses (worker w: the same blocks' bytes both times)
  bool stop = false;
  int take = 0;
  double t_read = 0, t_inflate = 0, t_frame = 0, t_parse = 0, t_wait_prod = 0, t_wait_cons = 0;   // MM_READER_TIMING=1: printed at close
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  // does the file start with a BGZF block: a gzip member whose extra field holds the 'BC' subfield (anywhere in it)
  static bool is_bgzf(const char *path) {
    FILE *g = fopen(path, "rb");
    if (!g) return false;
    unsigned char h[12 + 256];
    const size_t got = fread(h, 1, sizeof h, g);
    fclose(g);
    if (got < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) return false;
    const size_t xlen = (size_t) (h[10] | (h[11] << 8));
    if (12 + xlen > got) return false;
    for (size_t x = 12; x + 4 <= 12 + xlen;) {
      const size_t sl = (size_t) (h[x + 2] | (h[x + 3] << 8));
      if (h[x] == 'B' && h[x + 1] == 'C' && sl == 2 && x + 6 <= 12 + xlen) return true;
      x += 4 + sl;
    }
    return false;
  }
  bool open(const char *path, bool frame_records) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { ::close(fd); fd = -1; return false; }
    fsize = (size_t) st.st_size;
    void *m = mmap(nullptr, fsize, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { ::close(fd); fd = -1; return false; }
    map = (const uint8_t *) m;
    (void) madvise(m, fsize, MADV_SEQUENTIAL);
    frame = frame_records;
    n_threads = (int) std::max(1u, std::min(16u, std::thread::hardware_concurrency()));   // a GPU's share of the host cores
    if (const char *e = getenv("MM_READER_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n_threads = v; }
    producer = std::thread([this] { run(); });
    return true;
  }
  // BAM header at the start of the stream: -> offset of the first record and the number of reference names, false when it
  // does not lie completely inside buf (or is not a BAM header)
  static bool walk_header(const uint8_t *buf, size_t len, size_t &rec_begin, size_t &n_ref) {
    if (len < 12 || memcmp(buf, "BAM\1", 4) != 0) return false;
    int32_t l_text, nr;
    memcpy(&l_text, buf + 4, 4);
    if (l_text < 0 || (size_t) l_text > len - 12) return false;
    size_t p = 8 + (size_t) l_text;
    memcpy(&nr, buf + p, 4); p += 4;
    if (nr < 0) return false;
    for (int32_t i = 0; i < nr; ++i) {
      int32_t l;
      if (p + 4 > len) return false;
      memcpy(&l, buf + p, 4); p += 4;
      if (l < 0 || p + (size_t) l + 4 > len) return false;
      p += (size_t) l + 4;
    }
    rec_begin = p; n_ref = (size_t) nr;
    return true;
  }
  void run() {
    const size_t BATCH_BLOCKS = 512;
    std::vector<uint8_t> carry;
    std::vector<size_t> coff, clen, uoff, ulen;
    std::vector<uint32_t> offs;
    int put = 0;
    bool eof = false, first = true;
    size_t n_chr = 0;
    while (!eof) {
      {
        const double w0 = now();
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || ready[put] == 0; });
        t_wait_prod += now() - w0;
        if (stop) return;
      }
      const double s0 = now();
      Batch &B = slot[put];
      coff.clear(); clen.clear(); uoff.clear(); ulen.clear();
      size_t total = carry.size();
      B.bad = false; B.prefix = carry.size(); B.parsed = false; B.tail_carried = false; B.n_recs = 0; B.rec_begin = B.framed_end = 0;
      const uint8_t *raw = map;
      while (coff.size() < BATCH_BLOCKS) {
        if (fpos == fsize) { eof = true; break; }
        const uint8_t *h = map + fpos;
        // gzip member header with FEXTRA (RFC 1952); the BGZF 'BC' subfield (SAM specification 4.1) is looked for in every block's
        // extra field - other subfields may stand beside it - and gives the size of the whole block
        if (fsize - fpos < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { B.bad = true; eof = true; break; }
        const size_t xlen = (size_t) (h[10] | (h[11] << 8));
        if (fsize - fpos < 12 + xlen + 8) { B.bad = true; eof = true; break; }
        size_t bsize = 0;
        for (size_t x = 12; x + 4 <= 12 + xlen;) {
          const size_t sl = (size_t) (h[x + 2] | (h[x + 3] << 8));
          if (h[x] == 'B' && h[x + 1] == 'C' && sl == 2 && x + 6 <= 12 + xlen) { bsize = (size_t) (h[x + 4] | (h[x + 5] << 8)) + 1; break; }
          x += 4 + sl;
        }
        const size_t hdr = 12 + xlen;                                   // compressed data starts here (no FNAME / FCOMMENT / FHCRC in BGZF)
        if ((h[3] & ~4) || bsize < hdr + 8 || fsize - fpos < bsize) { B.bad = true; eof = true; break; }
        uint32_t isize;
        memcpy(&isize, h + bsize - 4, 4);
        if (isize > 65536) { B.bad = true; eof = true; break; }
        coff.push_back(fpos + hdr); clen.push_back(bsize - hdr - 8); uoff.push_back(total); ulen.push_back(isize);
        total += isize;
        fpos += bsize;
      }
      if (B.data.size() < total) B.data.resize(total);
      B.len = total;
      if (!carry.empty()) memcpy(B.data.data(), carry.data(), carry.size());
      carry.clear();
      const size_t nb = coff.size();
      const double s1 = now();
      t_read += s1 - s0;
      std::vector<int> fail((size_t) n_threads, 0);
      // every thread inflates a contiguous run of blocks and later parses the records that start in those bytes: the data
      // stays in the cache it was written to (parsing records another core has just inflated cost 4x as much)
      auto work = [&](int tid) {
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) { fail[tid] = 1; return; }
        std::unique_ptr<mminf::Tables> tables(own_inflate ? new mminf::Tables : nullptr);
        const size_t b_lo = nb * (size_t) tid / (size_t) n_threads, b_hi = nb * ((size_t) tid + 1) / (size_t) n_threads;
        for (size_t i = b_lo; i < b_hi; ++i) {
          if (ulen[i] == 0) continue;
          // the block decoder of mm_inflate.h first; whatever it does not take (any non-zero return) goes to zlib
          if (!(tables && mminf::inflate_block(raw + coff[i], clen[i], &B.data[uoff[i]], ulen[i], *tables) == 0)) {
            if (inflateReset(&z) != Z_OK) { fail[tid] = 1; break; }
            z.next_in = const_cast<Bytef *>(raw + coff[i]); z.avail_in = (uInt) clen[i];
            z.next_out = &B.data[uoff[i]]; z.avail_out = (uInt) ulen[i];
            const int rc = inflate(&z, Z_FINISH);
            if (rc != Z_STREAM_END || z.avail_out != 0) { fail[tid] = 1; break; }
          }
          // the member's CRC-32 (its trailer: CRC32, ISIZE): a damaged block that still inflates to ISIZE bytes must not turn into tuples
          if (check_crc) {
            uint32_t want;
            memcpy(&want, raw + coff[i] + clen[i], 4);
            if ((uint32_t) crc32(0L, &B.data[uoff[i]], (uInt) ulen[i]) != want) { fail[tid] = 1; break; }
          }
        }
        inflateEnd(&z);
      };
      pool.start(n_threads);
      pool.run(work);
      for (int v : fail) if (v) B.bad = true;
      const double s2 = now();
      t_inflate += s2 - s1;
      if (frame && !B.bad) {
        // frame: the chain of block_size fields; a record cut by the end of the batch moves to the next one
        size_t pos = 0;
        bool ok = true;
        if (first) { size_t nr = 0; ok = walk_header(B.data.data(), total, pos, nr); n_chr = nr + 1; }
        if (ok) {
          B.rec_begin = pos; B.n_chr = n_chr;
          offs.clear();
          bool broken = false;
          while (pos + 4 <= total) {
            int32_t size;
            memcpy(&size, &B.data[pos], 4);
            if (size < 32 || size > (1 << 28)) { broken = true; break; }     // not a record the fast parser could take: sequential from here
            if (pos + 4 + (size_t) size > total) break;
            __builtin_prefetch(&B.data[std::min(pos + 1536, total - 1)]);   // the bytes were written by other cores a moment ago
            offs.push_back((uint32_t) pos);
            pos += 4 + (size_t) size;
          }
          B.framed_end = pos;
          const size_t nr = offs.size();
          if (B.recs.size() < nr) B.recs.resize(nr);
          B.n_recs = nr;
          auto parse = [&](int tid, int nt) {
            // the records that start in the bytes this thread inflated
            size_t lo = nr * (size_t) tid / (size_t) nt, hi = nr * ((size_t) tid + 1) / (size_t) nt;
            if (nt == n_threads && nb > 0) {
              const size_t b_lo = nb * (size_t) tid / (size_t) nt, b_hi = nb * ((size_t) tid + 1) / (size_t) nt;
              const size_t byte_lo = tid == 0 ? 0 : uoff[b_lo], byte_hi = b_hi < nb ? uoff[b_hi] : total;
              lo = (size_t) (std::lower_bound(offs.begin(), offs.end(), (uint32_t) byte_lo) - offs.begin());
              hi = tid == nt - 1 ? nr : (size_t) (std::lower_bound(offs.begin(), offs.end(), (uint32_t) byte_hi) - offs.begin());
            }
            for (size_t k = lo; k < hi; ++k) {
              ParsedRec &R = B.recs[k];
              FastRec fr;
              R.off = offs[k]; R.flags = 0; R.pad = 0;
              if (parse_fast(&B.data[offs[k]], total - offs[k], n_chr, fr) && fr.start <= 0x7FFFFFF0ull &&
                  (fr.end <= 0x7FFFFFF0ull || fr.end == fr.start - 1)) {      // (out-of-range coordinates: the sequential path reports them)
                R.start = (int32_t) fr.start; R.end = (int32_t) (int64_t) fr.end; R.nh = fr.nhits; R.nm = fr.nm; R.hash = fr.hash; R.ref = fr.ref;
                R.name_len = (uint8_t) fr.name_len;
                R.flags = (uint8_t) (PR_FAST | (fr.strand ? PR_STRAND : 0) | (fr.has_nm ? PR_HAS_NM : 0));
              }
            }
          };
          const double s3 = now();
          t_frame += s3 - s2;
          if (nr >= 4096) pool.run([&](int t) { parse(t, n_threads); });
          else parse(0, 1);
          t_parse += now() - s3;
          B.parsed = true;
          if (broken) frame = false;                       // [framed_end, len) and every later batch: raw, for the sequential reader
          else if (!eof && pos < total) { carry.assign(B.data.begin() + (long) pos, B.data.begin() + (long) total); B.tail_carried = true; }
          // (at the end of the file a cut record stays where it is: the sequential reader fails on it as the reference does)
        } else frame = false;                              // header not inside the first batch: everything stays raw
      }
      first = false;
      B.last = eof;
      {
        std::lock_guard<std::mutex> lk(mu);
        ready[put] = 1;
      }
      cv.notify_all();
      put ^= 1;
    }
  }
  // next filled batch (nullptr at end); the previous one is released
  Batch *next(bool release_prev) {
    std::unique_lock<std::mutex> lk(mu);
    if (release_prev) { ready[take ^ 1] = 0; cv.notify_all(); }
    const double w0 = now();
    cv.wait(lk, [&] { return ready[take] == 1; });
    t_wait_cons += now() - w0;
    Batch *b = &slot[take];
    take ^= 1;
    return b;
  }
  void close() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    if (producer.joinable()) producer.join();
    if (map) { munmap(const_cast<uint8_t *>(map), fsize); map = nullptr; }
    if (fd >= 0) { ::close(fd); fd = -1; }
    if (const char *e = getenv("MM_READER_TIMING")) if (*e && *e != '0')
      fprintf(stderr, "[reader] %d threads: read %.3f s, inflate %.3f, frame %.3f, parse %.3f, producer waited %.3f for a free buffer, consumer waited %.3f for data\n",
              n_threads, t_read, t_inflate, t_frame, t_parse, t_wait_prod, t_wait_cons);
  }
};

struct ByteSource {  // sequential reader: BGZF through BgzfPipe, anything else (plain gzip, uncompressed) through gzread
  gzFile f = nullptr;
  BgzfPipe *pipe = nullptr;
  BgzfPipe::Batch *cur = nullptr;
  bool pipe_done = false;
  std::vector<uint8_t> buf;
  const uint8_t *p = nullptr;
  size_t pos = 0, len = 0;
  bool eof = false;
  std::string err;
  bool skip_prefix = false;   // sequential reading of framed batches: a batch's prefix repeats the previous batch's tail
  bool open(const char *path, bool frame_records) {
    if (BgzfPipe::is_bgzf(path)) {
      pipe = new BgzfPipe();
      if (!pipe->open(path, frame_records)) { delete pipe; pipe = nullptr; return false; }
      return true;
    }
    f = gzopen(path, "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    buf.resize(4 << 20);
    p = buf.data();
    return true;
  }
  void close() {
    if (f) { gzclose(f); f = nullptr; }
    if (pipe) { pipe->close(); delete pipe; pipe = nullptr; }
  }
  bool fill() {
    if (eof) return false;
    if (pipe) {
      for (;;) {
        if (!next_batch()) return false;
        const size_t skip = (skip_prefix && cur->prefix <= cur->len) ? cur->prefix : 0;
        if (cur->len > skip) { pos = skip; return true; }
      }
    }
    const int n = gzread(f, buf.data(), (unsigned) buf.size());
    if (n <= 0) { eof = true; len = pos = 0; return false; }
    p = buf.data();
    len = (size_t) n; pos = 0;
    return true;
  }
  // pipe only: the next inflated batch becomes the current buffer (pos 0); false at the end of the file or on a corrupt block
  bool next_batch() {
    if (eof) return false;
    if (pipe_done) { eof = true; len = pos = 0; return false; }
    cur = pipe->next(cur != nullptr);
    if (cur->bad) { err = "corrupt BGZF block (bad header, DEFLATE stream or CRC-32)"; eof = true; len = pos = 0; return false; }
    if (cur->last) pipe_done = true;
    p = cur->data.data(); len = cur->len; pos = 0;
    return true;
  }
  // false when fewer than n bytes remain
  inline bool read(void *dst, size_t n) {
    if (len - pos >= n) {                       // fast path: inside the current buffer
      if (dst) memcpy(dst, p + pos, n);
      pos += n;
      return true;
    }
    uint8_t *d = (uint8_t *) dst;
    while (n) {
      if (pos == len && !fill()) return false;
      const size_t k = std::min(n, len - pos);
      if (d) { memcpy(d, p + pos, k); d += k; }
      pos += k; n -= k;
    }
    return true;
  }
  bool skip(size_t n) { return read(nullptr, n); }
  int getc() {
    if (pos == len && !fill()) return -1;
    return p[pos++];
  }
};

}  // namespace

namespace {

// Read-name dictionary: name -> dense id in first-appearance order (the reference keys its maps on the name string,
// mmannot.cpp:1656-1662).  Open addressing over an arena of the names; no allocation per record.
struct NameDict {
  std::vector<char> arena;            // names back to back, each NUL-terminated
  std::vector<uint64_t> name_off;     // id -> offset in arena
  // slot: 32-bit hash, id + 1 (0 = empty) and the name's place in the arena - a hit costs two cache lines (slot, name), and a
  // caller that knows the hash ahead of time can have both in flight before it asks (prefetch, prefetch_name)
  struct Slot { uint32_t hash, id1; uint64_t off; };
  std::vector<Slot> table;
  std::vector<uint32_t> hashes;       // id -> 32-bit hash (reused when growing)
  size_t mask = 0;
  uint32_t last = 0xFFFFFFFFu;        // id of the previous lookup: an aligner writes the hits of a read one after the other
  static uint64_t hash(const char *p, size_t n) { return name_hash(p, n); }
  void grow() {
    const size_t cap = table.empty() ? (1u << 16) : table.size() * 2;
    table.assign(cap, Slot{0, 0, 0}); mask = cap - 1;
    for (uint32_t id = 0; id < name_off.size(); ++id) {
      size_t s = hashes[id] & mask;
      while (table[s].id1) s = (s + 1) & mask;
      table[s] = Slot{hashes[id], id + 1, name_off[id]};
    }
  }
  uint32_t size() const { return (uint32_t) name_off.size(); }
  const char *name(uint32_t id) const { return id < name_off.size() ? arena.data() + name_off[id] : ""; }
  bool same_at(uint64_t off, const char *p, size_t n) const {
    const char *q = arena.data() + off;
    return strncmp(q, p, n) == 0 && q[n] == 0;    // (stops at q's NUL: never reads past the stored name)
  }
  bool same(uint32_t id, const char *p, size_t n) const { return same_at(name_off[id], p, n); }
  void prefetch(uint32_t h) const { if (mask) { __builtin_prefetch(&table[h & mask]); __builtin_prefetch(&recent_[(h >> 7) & (RECENT - 1)]); } }
  // second stage, a few lookups after prefetch(h): the slot is here by now; ask for the stored name it points at
  void prefetch_name(uint32_t h) const {
    if (!mask) return;
    const Slot &e = table[h & mask];
    if (e.id1 && e.hash == h) __builtin_prefetch(arena.data() + e.off);
  }
  uint32_t get(const char *p, size_t n) { return get(p, n, (uint32_t) hash(p, n)); }
  // Names seen a moment ago (the other hits of a multi-mapping read are usually close by in the file) are answered from a
  // direct-mapped cache that holds the name itself: one cache line instead of slot + stored name.  1 MB: it stays in the L2 of
  // the core that works on this dictionary.
  struct Recent { uint32_t hash, id; uint8_t len; char name[55]; };
  static constexpr size_t RECENT = 16384;
  std::vector<Recent> recent;
  const Recent *recent_ = nullptr;
  NameDict() { recent.resize(RECENT); for (Recent &r : recent) { r.hash = 0; r.id = 0xFFFFFFFFu; r.len = 0; } recent_ = recent.data(); }
  NameDict(const NameDict &) = delete;
  NameDict &operator=(const NameDict &) = delete;
  uint32_t get(const char *p, size_t n, uint32_t h) {
    if (last != 0xFFFFFFFFu && same(last, p, n)) return last;
    Recent &rc = recent[(h >> 7) & (RECENT - 1)];
    if (rc.id != 0xFFFFFFFFu && rc.hash == h && rc.len == n && memcmp(rc.name, p, n) == 0) return last = rc.id;
    const uint32_t id = lookup(p, n, h);
    if (n <= sizeof(rc.name)) { rc.hash = h; rc.id = id; rc.len = (uint8_t) n; memcpy(rc.name, p, n); }
    return last = id;
  }
  uint32_t lookup(const char *p, size_t n, uint32_t h) {
    if ((name_off.size() + 1) * 2 > table.size()) grow();
    size_t s = h & mask;
    for (;;) {
      const Slot &e = table[s];
      if (!e.id1) break;
      if (e.hash == h && same_at(e.off, p, n)) return last = e.id1 - 1;
      s = (s + 1) & mask;
    }
    const uint32_t id = (uint32_t) name_off.size();
    name_off.push_back(arena.size());
    arena.insert(arena.end(), p, p + n);
    arena.push_back('\0');
    hashes.push_back(h);
    table[s] = Slot{h, id + 1, name_off[id]};
    return last = id;
  }
};

// The read-name dictionary of a file: dense ids in order of first appearance (what the device keys its per-name state on).
// Looking a name up is a chain of cache misses (the record's name, the table slot, the stored name) - at one lookup per
// record the one thread that has to keep the file order would cap the decoder.  So the names are spread over NSUB
// sub-dictionaries by their hash, and a long run of parsed records is resolved by all threads at once (resolve_run):
//   A. the run is cut into contiguous ranges, one per thread; each thread sorts the record indices of its range by owning
//      sub-dictionary;
//   B. each sub-dictionary's thread looks its records up, range after range - i.e. in file order, so its local ids come out
//      in first-appearance order - and notes which of them were new;
//   C. the new names of range w get the global ids behind those of the ranges before it (a prefix sum over the ranges), in
//      file order inside the range; then every range translates its records.
// Every array a thread writes is its own or a contiguous block (no cache line ping-pong); the ids are exactly those a single
// sequential pass gives.
struct QidSpace {
  static constexpr int NSUB = 16;
  NameDict sub[NSUB];
  std::vector<uint32_t> l2g[NSUB];      // local id -> global id
  std::vector<uint8_t> g_owner;         // global id -> sub-dictionary
  std::vector<uint32_t> g_local;        // global id -> local id
  static inline int owner(uint32_t h) { return (int) (h >> 28); }   // (the low bits index the tables)
  uint32_t size() const { return (uint32_t) g_owner.size(); }
  const char *name(uint32_t g) const { return g < g_owner.size() ? sub[g_owner[g]].name(g_local[g]) : ""; }
  void prefetch(uint32_t h) const { sub[owner(h)].prefetch(h); }
  uint32_t get(const char *p, size_t n) { return get(p, n, (uint32_t) name_hash(p, n)); }
  uint32_t get(const char *p, size_t n, uint32_t h) {
    const int o = owner(h);
    const uint32_t before = sub[o].size(), l = sub[o].get(p, n, h);
    if (l == before) { l2g[o].push_back(size()); g_owner.push_back((uint8_t) o); g_local.push_back(l); }
    return l2g[o][l];
  }
  std::vector<std::vector<uint32_t>> bucket, lids;   // [range * NSUB + owner]: record indices / local id | new << 31
  std::vector<uint32_t> newcnt;
  double t_a = 0, t_b = 0, t_c = 0;     // MM_READER_TIMING
  Pool pool;                            // worker w always serves the sub-dictionaries o = w (mod W): their tables stay in its cache
  void parallel(int W, const std::function<void(int)> &f) { pool.start(W); pool.run(f); }
  // ids of the parsed records recs[0..n) (all PR_FAST, names at base + off + 36), in file order -> out[0..n)
  void resolve_run(const ParsedRec *recs, size_t n, const uint8_t *base, uint32_t *out, int n_threads) {
    const int W = std::max(1, std::min(n_threads, 16));
    if (bucket.size() < (size_t) W * NSUB) { bucket.resize((size_t) W * NSUB); lids.resize((size_t) W * NSUB); }
    newcnt.assign((size_t) W * NSUB, 0);
    auto range = [&](int w, size_t &lo, size_t &hi) { lo = n * (size_t) w / (size_t) W; hi = n * ((size_t) w + 1) / (size_t) W; };
    const double c0 = BgzfPipe::now();
    parallel(W, [&](int w) {                                             // A
      size_t lo, hi;
      range(w, lo, hi);
      for (int o = 0; o < NSUB; ++o) bucket[(size_t) w * NSUB + o].clear();
      for (size_t k = lo; k < hi; ++k) bucket[(size_t) w * NSUB + owner(recs[k].hash)].push_back((uint32_t) k);
    });
    const double c1 = BgzfPipe::now();
    t_a += c1 - c0;
    parallel(W, [&](int w) {                                             // B
      for (int o = w; o < NSUB; o += W) {
        NameDict &D = sub[o];
        for (int r = 0; r < W; ++r) {
          const std::vector<uint32_t> &L = bucket[(size_t) r * NSUB + o];
          std::vector<uint32_t> &V = lids[(size_t) r * NSUB + o];
          V.resize(L.size());
          uint32_t fresh = 0;
          for (size_t j = 0; j < L.size(); ++j) {
            if (j + 16 < L.size()) { const ParsedRec &P = recs[L[j + 16]]; D.prefetch(P.hash); __builtin_prefetch(base + P.off + 36); }
            if (j + 6 < L.size()) D.prefetch_name(recs[L[j + 6]].hash);
            const ParsedRec &R = recs[L[j]];
            const uint32_t before = D.size();
            const uint32_t l = D.get((const char *) (base + R.off + 36), R.name_len, R.hash);
            const uint32_t isnew = l == before ? 1u : 0u;
            V[j] = l | (isnew << 31);
            fresh += isnew;
          }
          newcnt[(size_t) r * NSUB + o] = fresh;
        }
        l2g[o].resize(D.size());
      }
    });
    const double c2 = BgzfPipe::now();
    t_b += c2 - c1;
    std::vector<uint32_t> first((size_t) W + 1, size());                 // first global id of every range's new names
    for (int w = 0; w < W; ++w) { uint32_t c = 0; for (int o = 0; o < NSUB; ++o) c += newcnt[(size_t) w * NSUB + o]; first[(size_t) w + 1] = first[(size_t) w] + c; }
    g_owner.resize(first[(size_t) W]); g_local.resize(first[(size_t) W]);
    parallel(W, [&](int w) {                                             // C: global ids of the new names, in file order
      size_t lo, hi, at[NSUB] = {0};
      range(w, lo, hi);
      uint32_t g = first[(size_t) w];
      for (size_t k = lo; k < hi; ++k) {
        const int o = owner(recs[k].hash);
        const uint32_t v = lids[(size_t) w * NSUB + o][at[o]++];
        if (v >> 31) { const uint32_t l = v & 0x7FFFFFFFu; l2g[o][l] = g; g_owner[g] = (uint8_t) o; g_local[g] = l; ++g; }
      }
    });
    parallel(W, [&](int w) {                                             // ... and every record's id
      size_t lo, hi, at[NSUB] = {0};
      range(w, lo, hi);
      for (size_t k = lo; k < hi; ++k) {
        const int o = owner(recs[k].hash);
        out[k] = l2g[o][lids[(size_t) w * NSUB + o][at[o]++] & 0x7FFFFFFFu];
      }
    });
    t_c += BgzfPipe::now() - c2;
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
  QidSpace qids;
  std::vector<uint32_t> run_qid;   // ids of the run of parsed records being delivered
  double t_emit = 0;
  std::unordered_map<std::string, int32_t> chr_tid;  // read chromosome name -> annotation chromosome id (-1 unknown)
  std::vector<int32_t> bam_tid;                      // BAM reference id -> annotation chromosome id (-2 = not resolved yet)
  int32_t cur_ref = -2;                              // reference id of the current BAM record (-2: use the name)
  char name_buf[512];
  size_t name_len = 0;
  bool allow_fast = true;            // MM_READER_GENERAL=1 sends every record through the general parser (differential tests)
  std::vector<std::string> unknown_warned;
  uint64_t n_reads = 0;
  uint64_t cur_end = 0;   // Read::end of the read delivered last (a '*' CIGAR of an alternative keeps it)
  std::string log, err;
  std::string log_out;
  uint64_t batch_qlo = 0;   // smallest qid with NH > 1 of the batch delivered last
  // BGZF input whose records the pipe's threads have framed and parsed (BgzfPipe::frame): position in the current batch
  bool parsed_mode = false, check_slow = false;
  size_t rec_cur = 0, slow_end = 0;
  const void *slow_batch = nullptr;
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
    r->name_len = std::min(col[0].size(), sizeof(r->name_buf) - 1); memcpy(r->name_buf, col[0].data(), r->name_len); r->name_buf[r->name_len] = 0;
    r->cur_ref = -2;
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

// Fast path for a BAM record that lies completely inside the current buffer (parse_fast); false without consuming a byte when
// the general parser has to take it.
static bool fast_bam_record(mm_reader *r, FastRec &out) {
  ByteSource &s = r->src;
  if (!parse_fast(s.p + s.pos, s.len - s.pos, r->bam_chr.size(), out)) return false;
  s.pos += 4 + (size_t) out.size;
  if (out.has_nm) r->nmismatches = out.nm;                               // (survives from record to record, like the reference's)
  r->qids.prefetch(out.hash);
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
  if (chr_id == -1) r->cur_ref = (int32_t) r->bam_chr.size() - 1;      // the "*" entry appended behind the header's names
  else if (chr_id >= 0 && (size_t) chr_id + 1 < r->bam_chr.size()) r->cur_ref = chr_id;
  else { r->err = "BAM record refers to reference sequence " + std::to_string(chr_id) + " which the header does not declare"; return false; }
  r->rec_start = (uint64_t) (int64_t) (pos + 1);
  const int32_t l_name = (int32_t) (bin_mq_nl & 0xff), n_cigar = (int32_t) (flag_nc & 0xffff);
  const uint32_t flags = flag_nc >> 16;
  r->rec_strand = (flags & 0x10) == 0;
  if (l_name && !s.read(r->name_buf, (size_t) l_name)) return false;   // l_read_name is one byte: at most 255
  r->name_buf[l_name] = 0;
  r->name_len = strlen(r->name_buf);
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
  { const char *e = getenv("MM_READER_GENERAL"); r->allow_fast = !(e && *e && *e != '0'); }
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
    if (!r->src.open(path, r->allow_fast)) { r->err = "Cannot open file '" + r->path + "'."; return MM_ERR_IO; }
    char magic[5] = {0, 0, 0, 0, 0};
    r->src.read(magic, 4);
    if (strcmp(magic, "BAM\1") != 0) {
      if (!r->src.err.empty()) { r->err = r->src.err + " in '" + r->path + "'"; return MM_ERR_PARSE; }   // (nothing could be read at all)
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
    r->bam_chr.push_back("*");
    r->bam_tid.assign(r->bam_chr.size(), -2);
    if (r->src.pipe) {
      const BgzfPipe::Batch *b0 = r->src.cur;
      // the pipe's threads parse the records; this thread keeps what has to happen in file order
      r->parsed_mode = b0 && b0->parsed && r->src.pos == b0->rec_begin && b0->n_chr == r->bam_chr.size();
      r->rec_cur = 0;
      r->src.skip_prefix = true;   // whenever the bytes are read sequentially again, a batch's prefix (the previous batch's tail) was seen already
    }
  } else {
    r->log += "Reading SAM file " + r->path + "\n";
    r->sam.open(path);
  }
  return MM_OK;
}

extern "C" const char *mm_reader_error(const mm_reader *r) { return r ? r->err.c_str() : "null reader"; }

// Reader::gotoNextRead: the record itself, then each of its XA alternatives, as separate reads
extern "C" int mm_reader_next(mm_reader *r, mm_aln *aln, size_t cap, size_t *n_out, mm_nh_big *big, size_t big_cap, size_t *n_big, uint32_t *max_nh) {
  if (!r || !aln || !n_out || !big || !n_big || big_cap < 64) return MM_ERR_ARG;
  size_t n = 0, nb = 0;
  uint32_t mx = 0;
  *n_big = 0;
  if (max_nh) *max_nh = 0;
  uint64_t &cur_end = r->cur_end;
  // NH travels inside the tuple; the rare value beyond the field goes to the side list (the batch ends while that list
  // still has room for a whole run of fast-path records)
  r->batch_qlo = ~0ull;
  auto put_nh = [&](size_t i, uint32_t v) { if (v > mx) mx = v; if (v > 1 && aln[i].qid < r->batch_qlo) r->batch_qlo = aln[i].qid; if (v >= MM_ALN_NH_ESCAPE && nb < big_cap) { big[nb].index = (uint32_t) i; big[nb].nh = v; ++nb; } };
  while (n < cap && !r->over && nb + 33 <= big_cap) {
    // BGZF input, records framed and parsed by the pipe's threads: finish them in file order
    if (r->parsed_mode && !(r->have_record && r->alt_next < r->alts.size())) {
      BgzfPipe::Batch *pb = r->src.cur;
      if (r->rec_cur < pb->n_recs) {
        const ParsedRec *recs = pb->recs.data();
        if (recs[r->rec_cur].flags & PR_FAST) {
          size_t k = r->rec_cur;
          const size_t stop = std::min(pb->n_recs, k + (cap - n));
          const uint8_t *base = pb->data.data();
          // the read-name ids of a long run come from all threads at once (QidSpace::resolve_run), those of a short one from here
          // (the run also ends where the NH side list would be full: every record with an NH beyond the tuple's field takes an entry,
          // and the caller's list has big_cap of them - the batch then ends early, the rest of the run opens the next one)
          size_t run_end = k, esc = 0;
          while (run_end < stop && (recs[run_end].flags & PR_FAST)) {
            if (recs[run_end].nh >= MM_ALN_NH_ESCAPE) { if (nb + esc == big_cap) break; ++esc; }
            ++run_end;
          }
          const size_t k0 = k;
          const bool par = run_end - k0 >= 8192 && r->src.pipe->n_threads > 1;
          if (par) {
            if (r->run_qid.size() < run_end - k0) r->run_qid.resize(run_end - k0);
            r->qids.resolve_run(recs + k0, run_end - k0, base, r->run_qid.data(), r->src.pipe->n_threads);
          }
          const double e0 = BgzfPipe::now();
          for (; k < run_end; ++k) {
            const ParsedRec &F = recs[k];
            if (!par && k + 12 < pb->n_recs) r->qids.prefetch(recs[k + 12].hash);
            const bool fstrand = (F.flags & PR_STRAND) != 0;
            const bool strand = (r->strand_compat == MM_STRAND_COMPAT_PLUS) ? true : (r->strand_compat == MM_STRAND_COMPAT_MINUS ? false : fstrand);
            const bool reaches_scan = !r->unique_only || F.nh == 1;   // mmannot.cpp:1773
            int32_t tid = r->bam_tid[F.ref];
            if (tid == -2 || (tid < 0 && reaches_scan)) { tid = tid_of(r, r->bam_chr[F.ref], reaches_scan); r->bam_tid[F.ref] = tid; }
            aln[n].start = F.start;
            aln[n].end = F.end;
            aln[n].tsn = MM_ALN_PACK(tid, strand, F.nh);
            aln[n].qid = par ? r->run_qid[k - k0] : r->qids.get((const char *) (base + F.off + 36), F.name_len, F.hash);
            put_nh(n, F.nh);
            if (F.flags & PR_HAS_NM) r->nmismatches = F.nm;          // (survives from record to record, like the reference's)
            ++n; ++r->n_reads;
          }
          r->t_emit += BgzfPipe::now() - e0;
          // state the general path and the end-of-file replay look at: the last record, which has nothing to complain about
          const ParsedRec &l = recs[k - 1];
          r->cigar.clear(); r->alts.clear(); r->alt_next = 0; r->have_record = true;
          r->cur_ref = l.ref; r->rec_start = (uint64_t) (int64_t) l.start; r->rec_strand = (l.flags & PR_STRAND) != 0; r->nhits = l.nh;
          r->cur_end = (uint64_t) (int64_t) l.end;
          r->name_len = std::min((size_t) l.name_len, sizeof(r->name_buf) - 1); memcpy(r->name_buf, base + l.off + 36, r->name_len); r->name_buf[r->name_len] = 0;
          r->rec_cur = k;
          continue;
        }
        // a record for the general parser: it reads the bytes where they lie; the framing says where it must end
        const ParsedRec &R = recs[r->rec_cur++];
        int32_t size;
        memcpy(&size, pb->data.data() + R.off, 4);
        r->src.p = pb->data.data(); r->src.len = pb->len; r->src.pos = R.off;
        r->slow_end = (size_t) R.off + 4 + (size_t) size; r->slow_batch = pb; r->check_slow = true;
      } else if (pb->framed_end < pb->len && !pb->tail_carried) {
        // unframed bytes behind the last record (the framing broke there, or the file ends inside a record): sequential from here
        r->parsed_mode = false;
        r->src.p = pb->data.data(); r->src.len = pb->len; r->src.pos = pb->framed_end;
        continue;
      } else {
        if (!r->src.next_batch()) { r->parsed_mode = false; continue; }       // end of file (or a corrupt block): the general path ends the stream
        r->rec_cur = 0;
        if (!r->src.cur->parsed) r->parsed_mode = false;                      // (cannot follow a carried tail; its prefix, if any, is unread)
        continue;
      }
    }
    // Runs of ordinary BAM records: parsed in place a few at a time, so that the read-name dictionary's cache misses
    // (requested while parsing) overlap instead of adding up; then finished one by one in file order.
    if (r->is_bam && r->allow_fast && !r->parsed_mode && !(r->have_record && r->alt_next < r->alts.size())) {
      FastRec fr[32];
      size_t m = 0;
      while (m < 32 && n + m < cap && fast_bam_record(r, fr[m])) ++m;
      if (m) {
        for (size_t k = 0; k < m; ++k) {
          const FastRec &f = fr[k];
          if (f.start > 0x7FFFFFF0ull || (f.end > 0x7FFFFFF0ull && f.end != f.start - 1)) {
            r->err = std::string("alignment coordinate outside the supported range [0, 2^31) for read ") + std::string(f.name, f.name_len);
            *n_out = n; *n_big = nb; if (max_nh) *max_nh = mx; return MM_ERR_RANGE;
          }
          const bool strand = (r->strand_compat == MM_STRAND_COMPAT_PLUS) ? true : (r->strand_compat == MM_STRAND_COMPAT_MINUS ? false : f.strand);
          const bool reaches_scan = !r->unique_only || f.nhits == 1;   // mmannot.cpp:1773
          int32_t tid = r->bam_tid[f.ref];
          if (tid == -2 || (tid < 0 && reaches_scan)) { tid = tid_of(r, r->bam_chr[f.ref], reaches_scan); r->bam_tid[f.ref] = tid; }
          aln[n].start = (int32_t) f.start;
          aln[n].end = (int32_t) (int64_t) f.end;
          aln[n].tsn = MM_ALN_PACK(tid, strand, f.nhits);
          aln[n].qid = r->qids.get(f.name, f.name_len, f.hash);
          put_nh(n, f.nhits);
          ++n; ++r->n_reads;
        }
        // state the general path and the end-of-file replay look at: the last record, which has nothing to complain about
        const FastRec &l = fr[m - 1];
        r->cigar.clear(); r->alts.clear(); r->alt_next = 0; r->have_record = true;
        r->cur_ref = l.ref; r->rec_start = l.start; r->rec_strand = l.strand; r->nhits = l.nhits; r->cur_end = l.end;
        r->name_len = std::min((size_t) l.name_len, sizeof(r->name_buf) - 1); memcpy(r->name_buf, l.name, r->name_len); r->name_buf[r->name_len] = 0;
        continue;
      }
    }
    std::string chr;
    bool strand, is_alt = false;
    uint64_t start, end;
    if (r->have_record && r->alt_next < r->alts.size()) {
      const AltHit &ah = r->alts[r->alt_next++];
      is_alt = true;
      chr = ah.chromosome; strand = ah.strand; start = ah.start;
      end = cigar_end(ah.cigar, start, cur_end, r->log);
    } else {
      const bool ok = r->is_bam ? next_bam_record(r) : next_sam_record(r);
      if (r->check_slow) {   // the record came out of a framed batch: did the general parser end where the framing says the next one starts?
        r->check_slow = false;
        if (!(ok && r->src.cur == r->slow_batch && r->src.pos == r->slow_end)) r->parsed_mode = false;   // no: read on sequentially, as the reference would
      }
      if (!ok) {
        r->over = true;
        if (r->err.empty() && !r->src.err.empty()) r->err = r->src.err + " in '" + r->path + "'";
        if (!r->err.empty()) { r->have_record = false; *n_out = n; *n_big = nb; if (max_nh) *max_nh = mx; return MM_ERR_PARSE; }
        // At end of file the reference calls read.reset() once more on the stale last record (mmannot.cpp:1414-1415,
        // 1444-1447, 1527-1532): its CIGAR is parsed again and any complaint about it is printed a second time.
        if (r->have_record) (void) cigar_end(r->cigar, r->rec_start, r->rec_start, r->log);
        r->have_record = false;
        break;
      }
      r->have_record = true; r->alt_next = 0;
      start = r->rec_start;
      strand = (r->strand_compat == MM_STRAND_COMPAT_PLUS) ? true : (r->strand_compat == MM_STRAND_COMPAT_MINUS ? false : r->rec_strand);
      end = cigar_end(r->cigar, start, start, r->log);
    }
    cur_end = end;
    if (start > 0x7FFFFFF0ull || (end > 0x7FFFFFF0ull && end != start - 1)) {
      r->err = std::string("alignment coordinate outside the supported range [0, 2^31) for read ") + r->name_buf; *n_out = n; *n_big = nb; if (max_nh) *max_nh = mx; return MM_ERR_RANGE;
    }
    const bool reaches_scan = !r->unique_only || r->nhits == 1;   // mmannot.cpp:1773
    int32_t tid;
    if (!is_alt && r->is_bam && r->cur_ref >= 0) {                // primary BAM record: chromosome known by reference id
      tid = r->bam_tid[r->cur_ref];
      if (tid == -2 || (tid < 0 && reaches_scan)) { tid = tid_of(r, r->bam_chr[r->cur_ref], reaches_scan); r->bam_tid[r->cur_ref] = tid; }
    } else tid = tid_of(r, is_alt ? chr : r->chromosome, reaches_scan);
    const uint32_t q = r->qids.get(r->name_buf, r->name_len);
    aln[n].start = (int32_t) start;
    aln[n].end = (int32_t) (int64_t) end;
    aln[n].tsn = MM_ALN_PACK(tid, strand, r->nhits);
    aln[n].qid = q;
    put_nh(n, r->nhits);
    ++n; ++r->n_reads;
  }
  *n_out = n; *n_big = nb;
  if (max_nh) *max_nh = mx;
  return MM_OK;
}

extern "C" uint64_t mm_reader_qid_bound(const mm_reader *r) { return r ? (uint64_t) r->qids.size() : 0; }
extern "C" uint64_t mm_reader_qid_lo(const mm_reader *r) { return !r ? 0 : (r->batch_qlo == ~0ull ? (uint64_t) r->qids.size() : r->batch_qlo); }
extern "C" uint64_t mm_reader_n_reads(const mm_reader *r) { return r ? r->n_reads : 0; }
extern "C" const char *mm_reader_name(const mm_reader *r, uint32_t qid) {
  return r ? r->qids.name(qid) : "";
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
  if (const char *e = getenv("MM_READER_TIMING")) if (*e && *e != '0')
    fprintf(stderr, "[reader] read-name ids of the parsed runs: sorting by sub-dictionary %.3f s, lookups %.3f, numbering + translation %.3f; finishing the tuples %.3f\n",
            r->qids.t_a, r->qids.t_b, r->qids.t_c, r->t_emit);
  r->src.close();
  delete r;
}

// One raw DEFLATE stream (the payload of a BGZF block) -> out[0..out_len), by the reader's own block decoder (mm_inflate.h).
// 0 = decoded, ending exactly at out_len; anything else = not taken (the reader then uses zlib).  Exported for the differential tests.
extern "C" int mm_inflate_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
  if ((!in && in_len) || (!out && out_len)) return -1;
  std::unique_ptr<mminf::Tables> t(new mminf::Tables);
  return mminf::inflate_block(in, in_len, out, out_len, *t);
}

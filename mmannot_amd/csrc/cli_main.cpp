// cli_main.cpp — `mmannot-amd`: same command line, count table and stderr statistics as `mmannot` v1.1
// (main, mmannot.cpp:1931-2149), with the overlap-and-count path running on the GPU through the C ABI.
// Host work kept here: argv, file names, printing.  -m (per-read statistics, incl. the -e rescue that only
// acts together with -m) is the reference's "slow" side channel: the per-hit element sets come
// back from the device and the multiset bookkeeping of printReadStats/rescue (mmannot.cpp:474-509) is done here.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "mmannot_amd.h"

static const char VERSION[] = "1.1";

static std::string commas(uint64_t v) {
  std::string d = std::to_string(v), o;
  for (size_t i = 0; i < d.size(); ++i) {
    if (i && (d.size() - i) % 3 == 0) o += ',';
    o += d[i];
  }
  return o;
}

static void usage(float overlap, unsigned up, unsigned down, float rescue, unsigned threads) {
  std::cerr << "Usage: mmannot [options]\n"
            << "\tCompulsory options:\n"
            << "\t\t-a file: annotation file in GTF format\n"
            << "\t\t-r file1 [file2 ...]: reads in BAM/SAM format\n"
            << "\tMain options:\n"
            << "\t\t-o output: output file (default: stdout)\n"
            << "\t\t-c config_file: configuration file (default: config.txt)\n"
            << "\t\t-n name1 name2...: short name for each of the reads files\n"
            << "\t\t-s strand: string (U, F, R, FR, RF, FF, defaut: F) (use several strand types if the library strategies differ)\n"
            << "\t\t-f format (SAM or BAM): format of the read files (default: guess from file extension)\n"
            << "\t\t-l integer: overlap type (<0: read is included, <1: % overlap, otherwise: # nt, default: " << overlap << ")\n"
            << "\t\t-d integer: upstream region size (default: " << up << ")\n"
            << "\t\t-D integer: downstream region size (default: " << down << ")\n"
            << "\t\t-y string: quantification strategy, valid values are: default, unique, random, ratio (default: default)\n"
            << "\t\t-e integer: attribute a read to a feature if at least N% of the hits map to the feature (default: "
            << static_cast<unsigned int>(rescue * 100) << "%)\n"
            << "\tOutput options:\n"
            << "\t\t-p: print progress\n"
            << "\t\t-m file: print mapping statistics for each read (slow, only work with 1 input file)\n"
            << "\t\t-M file: print mapping statistics for each interval (slow, only work with 1 input file)\n"
            << "\t\t-t integer: # threads (default: " << threads << ")\n"
            << "\t\t-h: this help" << std::endl;
}

// printStats, mmannot.cpp:139-143 (numbers under the ',' grouping locale of mmannot.cpp:111-115)
static std::string stat_line(uint64_t n, const char *label, uint64_t total) {
  unsigned size = (unsigned) (std::log10((double) total) + 1);
  size += (unsigned) (size / 3.0);
  std::string num = commas(n);
  while (num.size() < size) num = " " + num;
  char pc[64];
  snprintf(pc, sizeof pc, "%5.1f", (double) ((float) n / total * 100));
  return std::string("\t") + label + num + " (" + pc + "%)\n";
}

// glibc's rand() for an unseeded process (= srand(1)): the TYPE_3 additive feedback generator r[i] = r[i-31] + r[i-3]
// seeded by the minimal-standard LCG, first 310 outputs discarded, result >> 1.  Replayed here explicitly so that
// -y random (mmannot.cpp:1711) does not depend on who else in the process calls rand().
struct GlibcRand {
  uint32_t r[34];
  int k = 0;
  std::vector<uint32_t> st;
  GlibcRand() {
    st.resize(344);
    int64_t x = 1;
    st[0] = 1;
    for (int i = 1; i < 31; ++i) { x = (16807 * x) % 2147483647; if (x < 0) x += 2147483647; st[i] = (uint32_t) x; }
    for (int i = 31; i < 34; ++i) st[i] = st[i - 31];
    for (int i = 34; i < 344; ++i) st[i] = st[i - 31] + st[i - 3];
  }
  uint32_t next() {
    const size_t n = st.size();
    const uint32_t v = st[n - 31] + st[n - 3];
    st.push_back(v);
    if (st.size() > 4096) st.erase(st.begin(), st.begin() + 2048);
    return v >> 1;
  }
};

// Host staging of the tuple stream: three pinned buffers (hipHostMalloc through the C ABI).  The decoder fills one while the
// library's copy stream drains the two submitted before it (a host buffer is free again once the second-next
// mm_annotate_batch has returned), so H2D copies, kernels and decoding overlap and nothing waits per batch.
// closes a reads decoder when the scope that opened it ends, however it ends
struct ReaderGuard {
  mm_reader **rd;
  ~ReaderGuard() { if (rd && *rd) { mm_reader_close(*rd); *rd = nullptr; } }
};

struct Staging {
  // tuples per batch (2^21: the decoder, not the device, sets the pace of a run from files, and hands over smaller batches sooner -
  // measured on a 16.8 M-record BAM against 2^22: 0.53 s instead of 0.60); MM_CLI_BATCH overrides it (tests: many small batches
  // through the asynchronous pipeline)
  static size_t batch() {
    static const size_t b = [] { const char *e = getenv("MM_CLI_BATCH"); const long v = e ? atol(e) : 0; return v >= 64 ? (size_t) v : (size_t) 1 << 21; }();
    return b;
  }
  mm_aln *buf[3] = {nullptr, nullptr, nullptr};
  std::vector<mm_nh_big> big;
  std::vector<uint64_t> hm;
  int n_buf;
  bool ok() const { for (int k = 0; k < n_buf; ++k) if (!buf[k]) return false; return true; }
  explicit Staging(int n = 3) : n_buf(n) { for (int k = 0; k < n_buf; ++k) buf[k] = (mm_aln *) mm_host_alloc(batch() * sizeof(mm_aln)); big.resize(1 << 16); }
  Staging(const Staging &) = delete;
  Staging &operator=(const Staging &) = delete;
  ~Staging() { for (auto &b : buf) mm_host_free(b); }   // (mm_host_free(NULL) is a no-op)
};

// -y random across input files (sequential runs): the reference's Counter keeps `seen`, `chosenId` and `numberSeen` - all keyed
// by read name - when it moves on to the next file (Counter::clear, mmannot.cpp:1742-1747, does not touch them), and rand() is
// process-global, so file 2 continues file 1's sequence and a name already output in file 1 is never output again.
struct RandomState {
  GlibcRand rng;
  std::unordered_map<std::string, std::pair<uint32_t, uint32_t>> choice;   // name -> (chosenId, numberSeen) still open at the end of earlier files
  std::unordered_map<std::string, bool> seen;                              // names output in earlier files
};

struct PendingRead {  // readCounts + rawCounts + readsIntervals entry, mmannot.cpp:1656-1660
  uint32_t remaining, raw;
  std::vector<uint8_t> regions;
  std::vector<uint32_t> intervals;
};

int main(int argc, char **argv) {
  float overlap = -1.0f, rescue_threshold = 1.0f;
  bool use_rescue = false;
  int strandedness = MM_STRANDED_F, strategy = MM_STRAT_DEFAULT, format = MM_FMT_AUTO, strand_compat = MM_STRAND_COMPAT_NONE;
  unsigned upstream = 1000, downstream = 1000, n_threads = 1;
  bool progress = false;
  int device = 0;
  bool device_given = false;
  std::vector<int> devices;
  std::string gtf, output, config = "config.txt", read_stats_file, interval_stats_file;
  std::vector<std::string> reads, names;
  if (argc == 1) { usage(overlap, upstream, downstream, rescue_threshold, n_threads); return EXIT_SUCCESS; }
  for (int i = 1; i < argc; i++) {
    std::string s(argv[i]);
    if (s.empty()) continue;
    auto next = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : (++i, ""); };
    if (s == "-a") gtf = next();
    else if (s == "-r") { for (++i; i < argc; ++i) { std::string t(argv[i]); if (t[0] == '-') { --i; break; } reads.push_back(t); } }
    else if (s == "-n") { for (++i; i < argc; ++i) { std::string t(argv[i]); if (t[0] == '-') { --i; break; } names.push_back(t); } }
    else if (s == "-c") config = next();
    else if (s == "-o") output = next();
    else if (s == "-l") overlap = std::stof(next());
    else if (s == "-s") {
      for (++i; i < argc; ++i) {
        s = argv[i];
        if (s == "U") strandedness = MM_STRANDED_U;
        else if (s == "F") strandedness = MM_STRANDED_F;
        else if (s == "R") strandedness = MM_STRANDED_R;
        else if (s[0] == '-' || s.empty()) { --i; break; }
        else {
          std::cerr << "Do not understand strandedness " << s << "\n" << "Exiting." << std::endl;
          usage(overlap, upstream, downstream, rescue_threshold, n_threads);
          return EXIT_FAILURE;
        }
      }
    }
    else if (s == "-p") progress = true;
    else if (s == "-t") n_threads = (unsigned) std::stoi(next());
    else if (s == "-m") read_stats_file = next();
    else if (s == "-M") interval_stats_file = next();
    else if (s == "-f") {
      for (++i; i < argc; ++i) {
        s = argv[i];
        std::transform(s.begin(), s.end(), s.begin(), ::tolower);
        if (s == "sam") format = MM_FMT_SAM;
        else if (s == "bam") format = MM_FMT_BAM;
        else if (s.empty() || s[0] == '-') { --i; break; }
        else {
          std::cerr << "Do not understand reads format " << s << "\n" << "Exiting." << std::endl;
          usage(overlap, upstream, downstream, rescue_threshold, n_threads);
          return EXIT_FAILURE;
        }
      }
    }
    else if (s == "-e") { rescue_threshold = std::stof(next()) / 100.0f; if (rescue_threshold < 1.0) use_rescue = true; }
    else if (s == "-d") upstream = (unsigned) std::stoul(next());
    else if (s == "-D") downstream = (unsigned) std::stoul(next());
    else if (s == "-y") {
      s = next();
      std::transform(s.begin(), s.end(), s.begin(), ::tolower);
      if (s == "default") strategy = MM_STRAT_DEFAULT;
      else if (s == "unique") strategy = MM_STRAT_UNIQUE;
      else if (s == "random") strategy = MM_STRAT_RANDOM;
      else if (s == "ratio") strategy = MM_STRAT_RATIO;
      else {
        std::cerr << "Do not understand strategy " << s << "\n" << "Exiting." << std::endl;
        usage(overlap, upstream, downstream, rescue_threshold, n_threads);
        return EXIT_FAILURE;
      }
    }
    else if (s == "-v") { std::cerr << "mmannot v" << VERSION << std::endl; return EXIT_SUCCESS; }
    else if (s == "-h") { usage(overlap, upstream, downstream, rescue_threshold, n_threads); return EXIT_SUCCESS; }
    // extensions (not in the reference): device selection and the verbatim-strand compatibility switch
    else if (s == "--selftest-rand") { GlibcRand g; const int n = std::stoi(next()); for (int k = 0; k < n; ++k) std::cout << g.next() << "\n"; return EXIT_SUCCESS; }
    else if (s == "--device") { device = std::stoi(next()); device_given = true; }
    else if (s == "--devices") { std::stringstream ss(next()); std::string tok; while (std::getline(ss, tok, ',')) if (!tok.empty()) devices.push_back(std::stoi(tok)); }
    else if (s == "--compat-strand") {
      s = next();
      if (s == "plus" || s == "const1") strand_compat = MM_STRAND_COMPAT_PLUS;
      else if (s == "minus" || s == "const0") strand_compat = MM_STRAND_COMPAT_MINUS;
      else strand_compat = MM_STRAND_COMPAT_NONE;
    }
    else {
      std::cerr << "Error: wrong parameter '" << s << "'.\nExiting." << std::endl;
      usage(overlap, upstream, downstream, rescue_threshold, n_threads);
      return EXIT_FAILURE;
    }
  }
  if (gtf.empty()) { std::cerr << "Missing input GTF file.\nExiting." << std::endl; usage(overlap, upstream, downstream, rescue_threshold, n_threads); return EXIT_FAILURE; }
  if (reads.empty()) { std::cerr << "Missing input BAM file.\nExiting." << std::endl; usage(overlap, upstream, downstream, rescue_threshold, n_threads); return EXIT_FAILURE; }
  const size_t n_inputs = reads.size();
  if (names.empty()) {
    for (const std::string &f : reads) {
      std::string n = f;
      size_t p = n.find_last_of("/");
      if (p != std::string::npos) n = n.substr(p + 1);
      p = n.find_last_of(".");
      if (p != std::string::npos) n = n.substr(0, p);
      names.push_back(n);
    }
  } else if (names.size() != n_inputs) {
    std::cerr << "Number of names is not equal to number of file names.\nExiting." << std::endl;
    usage(overlap, upstream, downstream, rescue_threshold, n_threads);
    return EXIT_FAILURE;
  }
  const bool read_stats = !read_stats_file.empty(), interval_stats = !interval_stats_file.empty();
  if ((read_stats || interval_stats) && n_inputs != 1) {
    std::cerr << "Only one reads file when providing reads or interval statistics.\nExiting." << std::endl;
    usage(overlap, upstream, downstream, rescue_threshold, n_threads);
    return EXIT_FAILURE;
  }
  // -m, -M and -y random are the reference's sequential side channels (per-read multisets, per-read interval lists, libc
  // rand() in file order): the device delivers the per-hit element sets / interval lists, the bookkeeping happens here.
  const bool host_book = read_stats || interval_stats || strategy == MM_STRAT_RANDOM;
  std::ofstream of, read_stats_out, interval_stats_out;
  if (!output.empty()) of.open(output.c_str());
  std::ostream &out = output.empty() ? std::cout : of;
  if (read_stats) read_stats_out.open(read_stats_file.c_str());
  if (interval_stats) interval_stats_out.open(interval_stats_file.c_str());

  // MM_CLI_TIMING=1: one extra stderr line with the wall time of the phases (not part of the reference's output)
  const bool timing = getenv("MM_CLI_TIMING") && *getenv("MM_CLI_TIMING") && *getenv("MM_CLI_TIMING") != '0';
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  double t_annot = 0, t_index = 0, t_decode = 0, t_device = 0, t_host = 0;
  uint64_t n_records = 0;
  const auto t_start = now();
  const bool teardown = getenv("MM_CLI_TEARDOWN") && *getenv("MM_CLI_TEARDOWN") && *getenv("MM_CLI_TEARDOWN") != '0';   // (see the end of main)
  // The device side needs no annotation to come up: the HIP runtime's start and the pinned staging buffers of the plain path (one
  // device, files one after the other) are requested now, on a thread of their own, while the GTF is parsed.
  const bool plain_path = devices.size() <= 1 && (n_threads <= 1 || n_inputs <= 1);
  std::unique_ptr<Staging> stg_plain;
  std::future<void> bringup = std::async(std::launch::async, [&]() {
    if (plain_path) stg_plain.reset(new Staging);
    else mm_host_free(mm_host_alloc(4096));
  });
  mm_annotation *an = nullptr;
  if (mm_annotation_load(&an, config.c_str(), gtf.c_str(), upstream, downstream) != MM_OK) {
    std::cerr << mm_annotation_error() << std::endl;
    return EXIT_FAILURE;
  }
  std::cerr << mm_annotation_log(an);
  t_annot = secs(t_start, now());
  bringup.get();
  // devices: --device d (one), --devices a,b,.. (several); with -t n > 1 and neither given, every visible device takes files
  if (devices.empty()) {
    if (device_given || n_threads <= 1 || n_inputs <= 1) devices.push_back(device);
    else { const int nd = std::max(1, mm_device_count()); for (int d = 0; d < nd; ++d) devices.push_back(d); }
  }
  // several devices for ONE input file: its reads are sharded over them (not with the sequential side channels -m / -M / -y random)
  const bool shard_one_input = n_inputs == 1 && devices.size() > 1 && !host_book;
  if (!shard_one_input && devices.size() > n_inputs) devices.resize(std::max<size_t>(1, n_inputs));
  mm_options opt;
  opt.overlap_mode = overlap < 0.0 ? MM_OVL_INCLUSION : (overlap < 1.0 ? MM_OVL_PC : MM_OVL_NT);
  opt.overlap_value = overlap; opt.strandedness = strandedness; opt.strategy = strategy; opt.keep_hit_masks = host_book ? 1 : 0; opt.defer_ratio = 0;
  std::vector<mm_index *> indices;
  // plain path: index and context are built on a thread of their own while the first batch of reads is decoded (the decoder needs the
  // annotation's chromosome names only); the first mm_annotate_batch waits for them
  const size_t n_workers = std::max<size_t>(1, std::min<size_t>((size_t) n_threads, n_inputs));
  const bool late_device = plain_path && devices.size() == 1 && !shard_one_input && n_workers == 1;
  mm_ctx *plain_ctx = nullptr;
  std::string late_err;
  std::future<void> late_job;
  auto build_indices = [&]() -> bool {
    const auto t0 = now();
    for (int d : devices) {
      mm_index *ix = nullptr;
      if (mm_index_from_annotation(&ix, an, d, 0) != MM_OK) { late_err = mm_last_error(); return false; }
      indices.push_back(ix);
    }
    t_index = secs(t0, now());
    return true;
  };
  if (late_device) {
    late_job = std::async(std::launch::async, [&]() {
      if (!stg_plain || !stg_plain->ok()) { late_err = "cannot allocate pinned host memory (is there a HIP device?)"; return; }
      if (!build_indices()) return;
      if (mm_ctx_create(&plain_ctx, indices[0], &opt, nullptr) != MM_OK) { late_err = mm_last_error(); plain_ctx = nullptr; }
    });
  } else if (!build_indices()) { std::cerr << "mmannot-amd: " << late_err << std::endl; return EXIT_FAILURE; }
  double t_wait_device = 0;
  // the plain path's context, waiting for the background build the first time (nullptr: it failed; the message is printed here)
  std::function<mm_ctx *()> late_ctx = [&]() -> mm_ctx * {
    if (late_job.valid()) {
      const auto t0 = now();
      late_job.get();
      t_wait_device = secs(t0, now());
      if (!plain_ctx) std::cerr << "mmannot-amd: " << late_err << std::endl;
    }
    return plain_ctx;
  };
  const int32_t E = mm_annotation_n_elements(an);
  std::vector<std::string> elem_names;
  for (int32_t e = 0; e < E; ++e) elem_names.push_back(mm_annotation_element_name(an, e));

  std::map<uint64_t, std::vector<unsigned>> table;  // element set -> one column per input (TableCount::regionCounts, :1858)
  const size_t BATCH = Staging::batch();
  // One input file on one context.  `err` is std::cerr itself when files are taken one after the other, and a buffer that is
  // printed in command-line order afterwards when several are in flight (-t): the reference's threads interleave their
  // messages as they come; here the text is always the one a -t 1 run prints.
  std::mutex table_mu;
  // Counter::dump (:1806-1818, the -M file :1819-1850) and TableCount::addCounter (:1861-1876) for one input file
  auto report = [&](size_t fi, const uint64_t *st, const std::vector<uint64_t> &masks, const std::vector<double> &sums, std::ostream &err,
                    std::map<std::vector<uint32_t>, unsigned> *interval_counts) {
    err << "Results for " << reads[fi] << ":" << std::endl;
    if (st[MM_STAT_HITS] == 0) err << "\tNo hit." << std::endl;
    else {
      err << "\t# reads:                       " << commas(st[MM_STAT_READS]) << "\n"
                << stat_line(st[MM_STAT_UNIQUE], "# uniquely mapped reads:       ", st[MM_STAT_READS])
                << stat_line(st[MM_STAT_RESCUED], "# multi-mapping rescued reads: ", st[MM_STAT_READS])
                << "\t# hits:                        " << commas(st[MM_STAT_HITS]) << "\n"
                << stat_line(st[MM_STAT_AMBIGUOUS], "# ambiguous hits:              ", st[MM_STAT_HITS])
                << stat_line(st[MM_STAT_UNASSIGNED], "# unassigned hits:             ", st[MM_STAT_HITS]);
    }
    if (interval_stats && interval_counts) {             // Counter::dump, :1819-1850
      const uint8_t *iv_type = nullptr;
      mm_annotation_intervals(an, nullptr, nullptr, nullptr, &iv_type, nullptr);
      std::vector<std::pair<std::string, unsigned>> lines;
      for (auto &p : *interval_counts) {
        std::vector<std::string> nm;
        for (uint32_t i : p.first) nm.push_back(std::string(mm_annotation_interval_id(an, i)) + " (" + elem_names[iv_type[i]] + ")");
        std::sort(nm.begin(), nm.end());
        std::string joined;
        for (size_t k = 0; k < nm.size(); ++k) joined += (k ? " -- " : "") + nm[k];
        lines.emplace_back(joined, p.second);
      }
      std::sort(lines.begin(), lines.end());
      std::string cur;
      unsigned count = 0;
      for (auto &ln : lines) {
        if (ln.first == cur) count += ln.second;
        else { if (!cur.empty()) interval_stats_out << cur << "\t" << count << "\n"; cur = ln.first; count = ln.second; }
      }
      if (!cur.empty()) interval_stats_out << cur << "\t" << count << "\n";
    }
    std::lock_guard<std::mutex> lk(table_mu);
    for (size_t r = 0; r < masks.size(); ++r) {          // TableCount::addCounter, :1861-1876 (column = file index, not completion order)
      auto &col = table[masks[r]];
      if (col.empty()) col.assign(n_inputs, 0);
      col[fi] = mm_round_to_uint(sums[r]);
    }
  };
  auto process_file = [&](size_t fi, mm_ctx *ctx, std::ostream &err, Staging &stg, RandomState &rnd) -> int {
    std::vector<uint64_t> &hm = stg.hm;
    // ctx == nullptr: the plain path's context, still being built when the first file is opened (late_ctx)
    if (ctx && mm_reset(ctx) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
    mm_reader *rd = nullptr;
    int rc = mm_reader_open(&rd, reads[fi].c_str(), format, an, strand_compat, strategy == MM_STRAT_UNIQUE, read_stats);
    ReaderGuard rd_guard{&rd};                               // (the reader's threads and its mapping go on every way out of here)
    if (rc != MM_OK) { err << mm_reader_take_log(rd) << mm_reader_error(rd) << std::endl; return EXIT_FAILURE; }
    // host bookkeeping for -m / -M / -y random
    std::unordered_map<uint32_t, PendingRead> pending;
    std::map<uint64_t, double> host_counts;
    std::map<std::vector<uint32_t>, unsigned> interval_counts;      // intervalCounts, mmannot.cpp:1659
    std::unordered_map<uint32_t, std::pair<uint32_t, uint32_t>> rnd_choice;   // chosenId / numberSeen, mmannot.cpp:1661 (this file's name ids)
    std::vector<uint8_t> rnd_seen;                                   // seen, mmannot.cpp:1662: bit 0 = seen, bit 1 = earlier files consulted
    GlibcRand &grand = rnd.rng;
    const bool rnd_carry = !rnd.choice.empty() || !rnd.seen.empty();
    uint64_t h_reads = 0, h_rescued = 0;
    auto print_read_stats = [&](uint32_t qid, unsigned nhits, std::vector<uint8_t> &regions) -> void {  // :474-493 + rescue :497-509
      if (!read_stats) return;
      std::sort(regions.begin(), regions.end());
      read_stats_out << mm_reader_name(rd, qid) << " \t" << nhits;
      for (size_t a = 0; a < regions.size();) {
        size_t b = a;
        while (b < regions.size() && regions[b] == regions[a]) ++b;
        read_stats_out << "\t" << elem_names[regions[a]] << ": " << (b - a);
        a = b;
      }
      bool rescued = false;
      if (use_rescue && regions.size() != 1) {
        const size_t n = regions.size();
        const size_t t = (size_t) std::ceil(n * rescue_threshold);
        std::vector<size_t> c((size_t) E, 0);
        for (uint8_t r : regions) if (++c[r] >= t) { regions.assign(1, r); rescued = true; break; }
      }
      if (rescued) read_stats_out << "\tRescued";
      read_stats_out << "\n";
    };
    auto count_regions = [&](std::vector<uint8_t> &regions, double w) -> size_t {
      uint64_t m = 0;
      for (uint8_t r : regions) m |= 1ull << r;
      host_counts[m] += w;
      return (size_t) __builtin_popcountll(m);
    };
    auto count_intervals = [&](std::vector<uint32_t> &iv) {
      if (!interval_stats || iv.empty()) return;
      std::sort(iv.begin(), iv.end());
      ++interval_counts[iv];
    };
    std::vector<uint32_t> hi_cnt, hi_idx;
    uint64_t cpt = 0;
    for (size_t batch_no = 0;; ++batch_no) {
      size_t n = 0, n_big = 0;
      uint32_t max_nh = 0;
      mm_aln *aln = stg.buf[batch_no % 3];
      const auto t0 = now();
      rc = mm_reader_next(rd, aln, BATCH, &n, stg.big.data(), stg.big.size(), &n_big, &max_nh);
      err << mm_reader_take_log(rd);
      if (rc != MM_OK) { err << mm_reader_error(rd) << std::endl; return EXIT_FAILURE; }
      const auto t1 = now();
      { std::lock_guard<std::mutex> lk(table_mu); t_decode += secs(t0, t1); n_records += n; }
      if (!ctx) {   // (before the end-of-file test: an empty file still needs its context for the report)
        ctx = late_ctx();
        if (!ctx) return EXIT_FAILURE;
        if (mm_reset(ctx) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
      }
      if (n == 0) break;
      // asynchronous: the copy and the kernels of this batch run while the next one is decoded; only the sequential side
      // channels (-m / -M / -y random), which read per-hit results back, wait for the batch
      if (mm_annotate_batch(ctx, aln, n, stg.big.data(), n_big, mm_reader_qid_lo(rd), mm_reader_qid_bound(rd), max_nh <= 1 ? MM_BATCH_SINGLE_HIT : 0) != MM_OK ||
          (host_book && mm_sync(ctx) != MM_OK)) {
        err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE;
      }
      { std::lock_guard<std::mutex> lk(table_mu); t_device += secs(t1, now()); }
      if (host_book) {
        hm.resize(n);
        if (mm_get_hit_masks(ctx, hm.data(), n) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
        size_t hi_total = 0;
        if (interval_stats) {
          hi_cnt.resize(n);
          if (mm_get_hit_intervals(ctx, hi_cnt.data(), nullptr, 0, &hi_total) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
          hi_idx.resize(hi_total ? hi_total : 1);
          if (hi_total && mm_get_hit_intervals(ctx, hi_cnt.data(), hi_idx.data(), hi_total, &hi_total) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
        }
        size_t hi_off = 0, big_at = 0;
        for (size_t i = 0; i < n; ++i) {
          if (hm[i] == ~0ull) continue;            // skipped by -y unique
          std::vector<uint8_t> regions;
          for (int e = 0; e < E; ++e) if (hm[i] >> e & 1) regions.push_back((uint8_t) e);
          std::vector<uint32_t> intervals;
          if (interval_stats) { intervals.assign(hi_idx.begin() + hi_off, hi_idx.begin() + hi_off + hi_cnt[i]); hi_off += hi_cnt[i]; }
          uint32_t h = MM_ALN_NH_FIELD(aln[i].tsn);
          if (h == MM_ALN_NH_ESCAPE) { while (big_at < n_big && stg.big[big_at].index < i) ++big_at; h = stg.big[big_at].nh; }
          const uint32_t q = aln[i].qid;
          if (h > 1 && strategy == MM_STRAT_DEFAULT) {   // :1669-1702
            auto it = pending.find(q);
            if (it == pending.end()) { pending[q] = PendingRead{h - 1, h, regions, intervals}; ++h_reads; }
            else {
              PendingRead &p = it->second;
              --p.remaining;
              p.regions.insert(p.regions.end(), regions.begin(), regions.end());
              p.intervals.insert(p.intervals.end(), intervals.begin(), intervals.end());
              if (p.remaining == 0) {
                if (!p.regions.empty()) {
                  print_read_stats(q, h, p.regions);
                  if (count_regions(p.regions, 1.0) == 1) ++h_rescued;
                  count_intervals(p.intervals);
                }
                pending.erase(it);
              }
            }
          } else {                                       // :1703-1738
            if (!regions.empty()) {
              bool output = true;
              if (strategy == MM_STRAT_RANDOM) {         // :1706-1726, libc rand() without srand, as the reference
                output = false;
                if (q >= rnd_seen.size()) rnd_seen.resize((size_t) q + 1024, 0);
                if (rnd_carry && !(rnd_seen[q] & 2)) {   // first time this file meets the name: what did earlier files leave for it?
                  rnd_seen[q] |= 2;
                  const std::string nm = mm_reader_name(rd, q);
                  if (rnd.seen.count(nm)) rnd_seen[q] |= 1;
                  auto pc = rnd.choice.find(nm);
                  if (pc != rnd.choice.end()) { rnd_choice[q] = pc->second; rnd.choice.erase(pc); }
                }
                if (!(rnd_seen[q] & 1)) {
                  auto pc = rnd_choice.find(q);
                  uint32_t chosen, seen_n;
                  if (pc == rnd_choice.end()) { chosen = h ? (uint32_t) (grand.next() % h) : 0u; seen_n = 0; rnd_choice[q] = std::make_pair(chosen, seen_n); }
                  else { chosen = pc->second.first; seen_n = ++pc->second.second; }
                  if (seen_n == chosen) { output = true; rnd_choice.erase(q); rnd_seen[q] |= 1; }
                }
              }
              if (output) {
                print_read_stats(q, h, regions);
                count_regions(regions, strategy == MM_STRAT_RATIO ? 1.0 / h : 1.0);
                count_intervals(intervals);
              }
            }
            ++h_reads;
          }
        }
      }
      cpt += n;
      if (progress && n_threads == 1) err << "\t" << commas(cpt) << " lines read.\r" << std::flush;
    }
    err << "\t" << commas(mm_reader_n_reads(rd)) << " lines read, done." << std::endl;
    if (mm_finalize(ctx) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
    uint64_t st[MM_N_STATS];
    mm_get_stats(ctx, st);
    std::vector<uint64_t> masks;
    std::vector<double> sums;
    if (host_book) {
      // (tests: a marker line between the reads printed in file order and the ones still pending here, which the reference prints in
      // its hash map's iteration order - only the first part has an order to compare)
      if (read_stats && getenv("MM_CLI_MARK_FLUSH")) read_stats_out << "#pending at end of file\n";
      for (auto &e : pending) {                          // EOF flush :1783-1800
        if (!e.second.regions.empty()) {
          print_read_stats(e.first, e.second.raw, e.second.regions);
          if (count_regions(e.second.regions, 1.0) == 1 && e.second.raw > 1) ++h_rescued;
        }
        count_intervals(e.second.intervals);
      }
    }
    if (read_stats || strategy == MM_STRAT_RANDOM) {     // counts that depend on the host bookkeeping (-e rescue, rand())
      st[MM_STAT_READS] = h_reads; st[MM_STAT_RESCUED] = h_rescued;
      for (auto &e : host_counts) { masks.push_back(e.first); sums.push_back(e.second); }
    } else {
      size_t nr = 0;
      mm_get_rows(ctx, nullptr, nullptr, 0, &nr);
      masks.resize(nr); sums.resize(nr);
      if (nr && mm_get_rows(ctx, masks.data(), sums.data(), nr, &nr) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
    }
    report(fi, st, masks, sums, err, &interval_counts);
    if (strategy == MM_STRAT_RANDOM && n_inputs > 1) {   // hand this file's name-keyed state on to the next file
      for (size_t q = 0; q < rnd_seen.size(); ++q) if (rnd_seen[q] & 1) rnd.seen[mm_reader_name(rd, (uint32_t) q)] = true;
      for (auto &e : rnd_choice) rnd.choice[mm_reader_name(rd, e.first)] = e.second;
    }
    mm_reader_close(rd); rd = nullptr;
    return 0;
  };
  // One input file on several devices (--devices a,b,... with a single -r file).  -y default / unique: the tuples
  // of every decoded batch are dealt out by read name (mm_partition), so all hits of a name meet on one device; -y ratio: whole
  // batches go round-robin and the file-order double sums are chained when the file ends (mm_merge).  Same table and the same
  // stderr block as one device.
  auto process_file_multi = [&](size_t fi, std::vector<mm_ctx *> &ctxs, std::ostream &err) -> int {
    const size_t N = ctxs.size();
    for (mm_ctx *c : ctxs) if (mm_reset(c) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
    mm_reader *rd = nullptr;
    int rc = mm_reader_open(&rd, reads[fi].c_str(), format, an, strand_compat, strategy == MM_STRAT_UNIQUE, false);
    ReaderGuard rd_guard{&rd};
    if (rc != MM_OK) { err << mm_reader_take_log(rd) << mm_reader_error(rd) << std::endl; return EXIT_FAILURE; }
    std::vector<std::unique_ptr<Staging>> stg_own;  // per device: three pinned buffers of a full batch
    for (size_t d = 0; d < N; ++d) stg_own.emplace_back(new Staging(3));
    auto stg = [&](size_t d) -> Staging & { return *stg_own[d]; };
    Staging decode(1);                              // where the decoder writes before the batch is dealt out
    for (size_t d = 0; d < N; ++d) if (!stg(d).ok()) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
    if (!decode.ok()) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
    std::vector<size_t> submitted(N, 0), part_n(N), part_nbig(N);
    std::vector<uint32_t> part_max(N);
    std::vector<uint64_t> part_qlo(N);
    std::vector<mm_aln *> part_out(N);
    std::vector<mm_nh_big *> part_big(N);
    const bool ratio = strategy == MM_STRAT_RATIO;
    uint64_t cpt = 0;
    for (size_t batch_no = 0;; ++batch_no) {
      size_t n = 0, n_big = 0;
      uint32_t max_nh = 0;
      const size_t dr = batch_no % N;              // -y ratio: the device this whole batch goes to
      mm_aln *aln = ratio ? stg(dr).buf[submitted[dr] % 3] : decode.buf[0];
      std::vector<mm_nh_big> &big = ratio ? stg(dr).big : decode.big;
      const auto t0 = now();
      rc = mm_reader_next(rd, aln, BATCH, &n, big.data(), big.size(), &n_big, &max_nh);
      err << mm_reader_take_log(rd);
      if (rc != MM_OK) { err << mm_reader_error(rd) << std::endl; return EXIT_FAILURE; }
      const auto t1 = now();
      t_decode += secs(t0, t1); n_records += n;
      if (n == 0) break;
      if (ratio) {
        if (mm_annotate_batch(ctxs[dr], aln, n, big.data(), n_big, 0, 0, 0) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
        ++submitted[dr];
      } else {
        for (size_t d = 0; d < N; ++d) { part_out[d] = stg(d).buf[submitted[d] % 3]; part_big[d] = stg(d).big.data(); }
        if (mm_partition(aln, n, big.data(), n_big, (int) N, part_out.data(), part_n.data(), part_big.data(), stg(0).big.size(), part_nbig.data(),
                         part_max.data(), part_qlo.data()) != MM_OK) {
          const int prc = mm_partition(aln, n, big.data(), n_big, (int) N, part_out.data(), part_n.data(), part_big.data(), stg(0).big.size(), part_nbig.data(), nullptr, nullptr);
          err << "mmannot-amd: cannot deal the batch out to the devices: "
              << (prc == MM_ERR_RANGE ? "a device's share of the NH side list (NH >= 2047) does not fit its staging list"
                                      : (prc == MM_ERR_ARG ? "a tuple's NH field says 'see the side list' but the list has no entry for it" : "internal error")) << std::endl;
          return EXIT_FAILURE;
        }
        const uint64_t qb = (mm_reader_qid_bound(rd) + N - 1) / N;
        for (size_t d = 0; d < N; ++d) {
          if (part_n[d] == 0) continue;
          if (mm_annotate_batch(ctxs[d], part_out[d], part_n[d], part_big[d], part_nbig[d], part_qlo[d], qb, part_max[d] <= 1 ? MM_BATCH_SINGLE_HIT : 0) != MM_OK) {
            err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE;
          }
          ++submitted[d];
        }
      }
      t_device += secs(t1, now());
      cpt += n;
      if (progress && n_threads == 1) err << "\t" << commas(cpt) << " lines read.\r" << std::flush;
    }
    err << "\t" << commas(mm_reader_n_reads(rd)) << " lines read, done." << std::endl;
    for (mm_ctx *c : ctxs) if (mm_finalize(c) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
    uint64_t st[MM_N_STATS];
    size_t nr = 0;
    if (mm_merge(ctxs.data(), (int) N, strategy, nullptr, nullptr, 0, &nr, st) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
    std::vector<uint64_t> masks(nr ? nr : 1);
    std::vector<double> sums(nr ? nr : 1);
    if (nr && mm_merge(ctxs.data(), (int) N, strategy, masks.data(), sums.data(), nr, &nr, st) != MM_OK) { err << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
    masks.resize(nr); sums.resize(nr);
    report(fi, st, masks, sums, err, nullptr);
    mm_reader_close(rd); rd = nullptr;
    return 0;
  };
  int rc_all = 0;
  if (shard_one_input) {
    mm_options mopt = opt;
    mopt.defer_ratio = strategy == MM_STRAT_RATIO ? 1 : 0;
    std::vector<mm_ctx *> ctxs;
    for (mm_index *ix : indices) {
      mm_ctx *ctx = nullptr;
      if (mm_ctx_create(&ctx, ix, &mopt, nullptr) != MM_OK) { std::cerr << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
      ctxs.push_back(ctx);
    }
    rc_all = process_file_multi(0, ctxs, std::cerr);
    for (mm_ctx *c : ctxs) mm_ctx_destroy(c);
  } else if (n_workers == 1) {
    RandomState rnd;
    if (late_device) {
      if (!stg_plain || !stg_plain->ok()) { late_ctx(); return EXIT_FAILURE; }   // (no device: the background job says why)
      for (size_t fi = 0; fi < n_inputs && !rc_all; ++fi) rc_all = process_file(fi, fi == 0 ? nullptr : plain_ctx, std::cerr, *stg_plain, rnd);
      if (late_job.valid()) late_job.get();          // (the first file failed before it needed the device)
      if (plain_ctx && teardown) mm_ctx_destroy(plain_ctx);
    } else {
      std::unique_ptr<Staging> own(stg_plain ? nullptr : new Staging);
      Staging &stg = stg_plain ? *stg_plain : *own;
      mm_ctx *ctx = nullptr;
      if (!stg.ok()) { std::cerr << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
      if (mm_ctx_create(&ctx, indices[0], &opt, nullptr) != MM_OK) { std::cerr << "mmannot-amd: " << mm_last_error() << std::endl; return EXIT_FAILURE; }
      for (size_t fi = 0; fi < n_inputs && !rc_all; ++fi) rc_all = process_file(fi, ctx, std::cerr, stg, rnd);
      if (teardown) mm_ctx_destroy(ctx);
    }
  } else {
    // -t n: the reference runs n threads over the input files (mmannot.cpp:2117-2143); here n files are in flight, each on its
    // own context, spread over the devices
    std::vector<std::ostringstream> errs(n_inputs);
    std::vector<int> rcs(n_inputs, 0);
    std::atomic<size_t> next_file(0);
    std::vector<std::thread> workers;
    for (size_t w = 0; w < n_workers; ++w)
      workers.emplace_back([&, w]() {
        // (-y random with several files in flight: every worker replays its own rand() sequence per file - the reference's threads
        // race on one process-global rand() there, so no particular interleaving is "the" answer)
        Staging stg;
        mm_ctx *ctx = nullptr;
        const bool ok = stg.ok() && mm_ctx_create(&ctx, indices[w % indices.size()], &opt, nullptr) == MM_OK;
        for (;;) {
          const size_t fi = next_file.fetch_add(1);
          if (fi >= n_inputs) break;
          if (!ok) { errs[fi] << "mmannot-amd: " << mm_last_error() << std::endl; rcs[fi] = EXIT_FAILURE; continue; }
          RandomState rnd;
          rcs[fi] = process_file(fi, ctx, errs[fi], stg, rnd);
        }
        if (ctx) mm_ctx_destroy(ctx);
      });
    for (auto &t : workers) t.join();
    for (size_t fi = 0; fi < n_inputs; ++fi) { std::cerr << errs[fi].str(); if (rcs[fi]) rc_all = rcs[fi]; }
  }
  if (rc_all) return rc_all;
  // TableCount::dump, :1877-1900
  std::vector<uint64_t> keys;
  for (auto &e : table) keys.push_back(e.first);
  mm_sort_rows(keys.data(), nullptr, keys.size());
  out << "Type";
  for (const std::string &s : names) out << "\t" << s;
  out << "\n";
  for (uint64_t k : keys) {
    bool first = true;
    for (int e = 0; e < E; ++e) if (k >> e & 1) { out << (first ? "" : "--") << elem_names[e]; first = false; }
    for (unsigned v : table[k]) out << "\t" << v;
    out << "\n";
  }
  out.flush();
  if (read_stats) read_stats_out.close();
  if (interval_stats) interval_stats_out.close();
  std::cerr << "Successfully done." << std::endl;
  // Everything is written.  Tearing the process down piece by piece (three pinned buffers, the index, the annotation's strings, the
  // HIP runtime's own exit handlers) took a fifth of a second - a quarter of a 17 M-record run - for memory the kernel reclaims
  // anyway, so the process ends here; MM_CLI_TEARDOWN=1 keeps the orderly exit (leak checkers, sanitizers).
  if (teardown) {
    for (mm_index *ix : indices) mm_index_destroy(ix);
    mm_annotation_free(an);
  }
  if (timing) {
    const double total = secs(t_start, now());
    // (plain path: index and context are built beside the first batch's decoding; only what the main thread still waited for is its own)
    t_host = total - t_annot - (late_device ? t_wait_device : t_index) - t_decode - t_device;
    char ix[96];
    if (late_device) snprintf(ix, sizeof ix, "index %.3f (beside the first batch; waited %.3f for index + context)", t_index, t_wait_device);
    else snprintf(ix, sizeof ix, "index %.3f", t_index);
    fprintf(stderr, "[mmannot-amd timing] total %.3f s: annotation %.3f, %s, waiting for the decoder %.3f, device %.3f, other host %.3f; %llu records, %.2f M records/s\n",
            total, t_annot, ix, t_decode, t_device, t_host, (unsigned long long) n_records, n_records / total / 1e6);
  }
  if (!teardown) {
    if (!output.empty()) of.close();
    std::cout.flush(); std::cerr.flush(); fflush(nullptr);
    std::_Exit(0);
  }
  return 0;
}

// mm_annot.cpp — host side, cold path (once per run): configuration file + GTF -> the sorted, typed
// interval vector the device index is built from.  Replaces Config::parse (mmannot.cpp:237-382),
// GtfLineParser (:515-580), Gene/Transcript (:708-990) and the IntervalList constructor (:1094-1290)
// of the reference.  Written from the behavioural description in SURVEY.md section 5.6; pinned by the
// interval dumps of the compiled reference (tests/golden/*/index_*.tsv.gz), including the order of
// equal-start intervals that the reference's unstable std::sort leaves behind.
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <regex>
#include <sstream>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/mmannot_amd.h"
#include "mm_hostutil.h"

namespace mmh {

static const uint64_t UNSET = ~0ull;

// ------------------------------------------------------------------ configuration
struct OrderEntry {
  std::string source, type;
  std::regex re;
  int strand;
};

struct ConfigModel {
  std::vector<std::pair<std::regex, std::string>> synonyms;
  std::vector<std::vector<OrderEntry>> lines;
  struct Rule { std::string source, type; long a = -1, b = -1; };
  std::vector<Rule> introns, vicinity;
  std::vector<std::pair<int, int>> flat;  // element -> (line, position in line)

  static std::string star(const std::string &s) {  // only the first '*' becomes ".*" (mmannot.cpp:273, 314, 325)
    std::string r = s;
    size_t p = r.find('*');
    if (p != std::string::npos) r.replace(p, 1, ".*");
    return r;
  }

  // first flattened Order entry whose source regex fully matches and whose type is empty or equal.  A GTF has millions of
  // lines but a few dozen distinct (source, type) pairs: the regex matches run once per pair.
  mutable std::unordered_map<std::string, long> order_cache;
  mutable std::unordered_map<std::string, std::string> translate_cache;
  long order_of(const std::string &source, const std::string &type) const {
    std::string key;
    key.reserve(source.size() + type.size() + 1);
    key += source; key += '\t'; key += type;
    auto it = order_cache.find(key);
    if (it != order_cache.end()) return it->second;
    long found = -1, n = 0;
    for (const auto &ln : lines) {
      for (size_t i = 0; i < ln.size() && found < 0; ++i)
        if (std::regex_match(source, ln[i].re) && (ln[i].type.empty() || ln[i].type == type)) found = n + (long) i;
      if (found >= 0) break;
      n += (long) ln.size();
    }
    order_cache.emplace(std::move(key), found);
    return found;
  }
  const std::string &translate(const std::string &s) const {
    auto it = translate_cache.find(s);
    if (it != translate_cache.end()) return it->second;
    std::string out = s;
    for (const auto &sy : synonyms)
      if (std::regex_match(s, sy.first)) { out = sy.second; break; }
    return translate_cache.emplace(s, std::move(out)).first->second;
  }
  static long rule_of(const std::vector<Rule> &rs, const std::string &source, const std::string &type, bool second) {
    for (const auto &r : rs)
      if ((r.source == "*" || r.source == source) && (r.type == "*" || r.type == type)) return second ? r.b : r.a;
    return -1;
  }
  long intron_of(const std::string &s, const std::string &t) const { return rule_of(introns, s, t, false); }
  long upstream_of(const std::string &s, const std::string &t) const { return rule_of(vicinity, s, t, false); }
  long downstream_of(const std::string &s, const std::string &t) const { return rule_of(vicinity, s, t, true); }

  const OrderEntry &entry(size_t e) const { return lines[flat[e].first][flat[e].second]; }
  std::string name(size_t e) const {
    const OrderEntry &a = entry(e);
    std::string s = a.source;
    if (!a.type.empty()) s += ":" + a.type;
    if (a.strand == MM_STRAND_F) s += " (+)";
    else if (a.strand == MM_STRAND_R) s += " (-)";
    return s;
  }

  // returns "" on success, else the message the reference prints before exit(EXIT_FAILURE)
  std::string parse(const std::string &path, std::string &log) {
    std::ifstream f(path.c_str());
    if (!f.good()) return "Error, configuration file '" + path + "' does not exists!";
    enum { NONE, SYN, INTR, VIC, ORD } sec = NONE;
    std::string line;
    while (std::getline(f, line)) {
      trim(line);
      if (line.empty() || line[0] == '#') continue;
      if (line == "Synonyms:") { sec = SYN; continue; }
      if (line == "Introns:") { sec = INTR; continue; }
      if (line == "Vicinity:") { sec = VIC; continue; }
      if (line == "Order:") { sec = ORD; continue; }
      std::string key, value;
      if (sec == SYN) {
        if (!split_first(line, ':', key, value))
          return "Error, cannot parse line '" + line + "' in the 'Synonyms' section of the configuration file!";
        key = star(key);
        try { synonyms.emplace_back(std::regex(key), value); }
        catch (const std::regex_error &) {
          return "Error, cannot parse regular expression '" + key + "' in line '" + line + "' in the 'Synonyms' section of the configuration file!";
        }
      } else if (sec == INTR || sec == VIC) {
        if (!split_first(line, ':', key, value))
          return "Error, cannot parse line '" + line + "' in the '" + (sec == INTR ? "Introns" : "Vicinity") + "' section of the configuration file!";
        Rule r; r.source = key; r.type = value;
        (sec == INTR ? introns : vicinity).push_back(r);
      } else if (sec == ORD) {
        std::stringstream ls(line);
        std::string field;
        std::vector<OrderEntry> entries;
        while (std::getline(ls, field, ',')) {
          int strand = MM_STRAND_ALL;
          std::string rest, ss;
          if (split_first(field, ' ', rest, ss)) {
            if (ss == "+") strand = MM_STRAND_F;
            else if (ss == "-") strand = MM_STRAND_R;
            else return "Error, cannot parse line '" + line + "' in the 'Order' section of the configuration file (last item item should be the strand: '+' or '-')!";
            field = rest;
          }
          OrderEntry oe;
          oe.strand = strand;
          std::string pattern;
          if (split_first(field, ':', key, value)) { oe.source = key; oe.type = value; pattern = star(key); }
          else { oe.source = field; oe.type = ""; pattern = star(field); }
          try { oe.re = std::regex(pattern); }
          catch (const std::regex_error &) {
            return "Error, cannot parse regular expression '" + key + "' in line '" + line + "' in the 'Order' section of the configuration file!";
          }
          entries.push_back(oe);
        }
        lines.push_back(entries);
      } else {
        return "Error, line '" + line + "' is not in the 'Synonyms', 'Introns', 'Vicinity', nor 'Order' section !";
      }
    }
    if (lines.empty()) return "Error, the 'Order' section is empty!  Please provide annotations.";
    for (auto &r : introns) {
      r.a = order_of(r.source, "intron");
      if (r.a < 0) return "Error, type '" + r.source + ":intron' (of '" + r.source + ":" + r.type + "') should be included in the 'Order:' section.";
    }
    for (auto &r : vicinity) {
      r.a = order_of(r.source, "upstream");
      if (r.a < 0) return "Error, type '" + r.source + ":upstream' (of '" + r.source + ":" + r.type + "') should be included in the 'Order:' section.";
      r.b = order_of(r.source, "downstream");
      if (r.b < 0) return "Error, type '" + r.source + ":downstream' (of '" + r.source + ":" + r.type + "') should be included in the 'Order:' section.";
    }
    for (size_t i = 0; i < lines.size(); ++i)
      for (size_t j = 0; j < lines[i].size(); ++j) flat.emplace_back((int) i, (int) j);
    // the echo of the Order the reference writes to stderr (mmannot.cpp:375-381)
    log += "Order:\n";
    for (const auto &ln : lines) {
      for (const auto &a : ln) {
        log += a.source + ":" + a.type + " ";
        if (a.strand == MM_STRAND_F) log += "(+)";
        else if (a.strand == MM_STRAND_R) log += "(-)";
        log += "\t";
      }
      log += "\n";
    }
    return "";
  }
};

// ------------------------------------------------------------------ GTF record
struct GtfRecord {
  // Views into the line being parsed (valid while that line is): a GTF has millions of lines, and the map of
  // vector-of-strings this used to build per line was most of the loading time.
  std::string chromosome, source, type;
  uint64_t start = 0, end = 0;
  int strand = MM_STRAND_R;
  struct Tag { std::string_view key, value; };
  std::vector<Tag> tags;       // a later occurrence of a key replaces an earlier one (map assignment in the reference)
  bool ok = false;

  const Tag *find(const char *k) const {
    const std::string_view kv(k);
    for (size_t i = tags.size(); i-- > 0;) if (tags[i].key == kv) return &tags[i];
    return nullptr;
  }
  bool has(const char *k) const { return find(k) != nullptr; }
  // first element of the comma-separated value ("" when the value is empty)
  std::string first(const char *k) const {
    const Tag *t = find(k);
    if (!t) return std::string();
    return std::string(t->value.substr(0, t->value.find(',')));
  }
  static void ltrim_v(std::string_view &v) { size_t i = 0; while (i < v.size() && std::isspace((unsigned char) v[i])) ++i; v.remove_prefix(i); }
  static void rtrim_v(std::string_view &v) { size_t n = v.size(); while (n > 0 && std::isspace((unsigned char) v[n - 1])) --n; v = v.substr(0, n); }

  explicit GtfRecord(const std::string &line) {
    // getline-style split at tabs: a trailing empty field is not produced
    std::string_view col[10];
    size_t nc = 0, p0 = 0;
    const std::string_view lv(line);
    while (p0 < lv.size() && nc < 10) {
      const size_t t = lv.find('\t', p0);
      if (t == std::string_view::npos) { col[nc++] = lv.substr(p0); p0 = lv.size(); break; }
      col[nc++] = lv.substr(p0, t - p0);
      p0 = t + 1;
    }
    if (nc != 9 || p0 < lv.size()) return;  // the reference asserts here (mmannot.cpp:526)
    chromosome.assign(col[0]); source.assign(col[1]); type.assign(col[2]);
    strand = (col[6] == "+") ? MM_STRAND_F : MM_STRAND_R;
    try { start = std::stoul(std::string(col[3])); end = std::stoul(std::string(col[4])); } catch (...) { return; }
    std::string_view rem = col[8];
    ltrim_v(rem); rtrim_v(rem);
    // attribute grammar: key, then blank or '=', then a "quoted" or bare value, up to ';'
    while (!rem.empty()) {
      const size_t sp = rem.find(' '), eq = rem.find('=');
      const size_t cut = (eq == std::string_view::npos) ? sp : ((sp == std::string_view::npos) ? eq : std::min(sp, eq));
      std::string_view tag = rem.substr(0, cut), value;
      rtrim_v(tag);
      if (cut != std::string_view::npos) rem.remove_prefix(cut + 1);  // no separator at all: the text is kept, as in the reference
      ltrim_v(rem);
      if (!rem.empty() && rem[0] == '"') {
        rem.remove_prefix(1);
        const size_t q = rem.find('"');
        value = rem.substr(0, q);
        if (q != std::string_view::npos) rem.remove_prefix(q + 1);   // an unterminated quote keeps the text, as in the reference
      } else {
        size_t sc = rem.find(';');
        if (sc == std::string_view::npos) sc = rem.size();
        value = rem.substr(0, sc);
        rtrim_v(value);
      }
      tags.push_back(Tag{tag, value});
      const size_t semi = rem.find(';');
      if (semi == std::string_view::npos) rem = std::string_view();
      else { rem.remove_prefix(semi + 1); ltrim_v(rem); }
    }
    ok = true;
  }
};

// ------------------------------------------------------------------ gene model
struct Seg {
  uint64_t s = UNSET, e = UNSET;
  bool set() const { return s != UNSET && e != UNSET; }
};

struct GeneModel {
  std::string id, source, type;
  uint64_t start, end;     // the gene's own span (line span merged with every exon)
  int strand;
  uint32_t chr;
  Seg hull;                // span of the merged transcript
  std::vector<Seg> exons;  // merged, sorted
  Seg cds;                 // hull of the CDS lines

  GeneModel(const GtfRecord &r, uint32_t c) : source(r.source), type(r.type), start(r.start), end(r.end), strand(r.strand), chr(c) {
    if (r.has("gene_id")) id = r.first("gene_id");
    else if (r.has("ID")) id = r.first("ID");
    else if (r.has("transcript_id")) id = r.first("transcript_id");
    else { id = r.first("Parent"); size_t p = id.find('.'); if (p != std::string::npos) id = id.substr(0, p); }
    hull.s = r.start; hull.e = r.end;
  }
  void add_exon(uint64_t s, uint64_t e) {
    start = std::min(start, s); end = std::max(end, e);
    hull.s = std::min(hull.s, s); hull.e = std::max(hull.e, e);
    Seg x; x.s = s; x.e = e;
    exons.push_back(x);
    std::sort(exons.begin(), exons.end(), [](const Seg &a, const Seg &b) { return a.s < b.s; });
    std::vector<Seg> merged;
    Seg cur;
    for (const Seg &ex : exons) {
      if (!cur.set()) cur = ex;
      else if (cur.e < ex.s) { merged.push_back(cur); cur = ex; }   // abutting exons stay separate
      else { cur.s = std::min(cur.s, ex.s); cur.e = std::max(cur.e, ex.e); }
    }
    merged.push_back(cur);
    exons.swap(merged);
  }
  void add_cds(uint64_t s, uint64_t e) {
    add_exon(s, e);
    if (cds.set()) { cds.s = std::min(cds.s, s); cds.e = std::max(cds.e, e); }
    else { cds.s = s; cds.e = e; }
  }
  static void clip(const std::vector<Seg> &src, uint64_t lo, uint64_t hi, std::vector<Seg> &out) {
    out.clear();
    for (const Seg &x : src) {
      const uint64_t a = std::max(x.s, lo), b = std::min(x.e, hi);
      if (a <= b) { Seg y; y.s = a; y.e = b; out.push_back(y); }
    }
  }
};

struct IvOut {
  uint32_t chr;
  uint64_t start, end;
  uint32_t type;
  uint8_t strand;
  std::string id;
};

}  // namespace mmh

using namespace mmh;

struct mm_annotation {
  ConfigModel cfg;
  std::vector<std::string> chromosomes;
  std::vector<IvOut> iv;
  // flat views
  std::vector<int32_t> line_of, a_chr, a_start, a_end;
  std::vector<uint8_t> e_strand, e_up, e_down, a_type, a_strand;
  std::vector<std::string> names;
  std::vector<uint8_t> chr_has_interval;
  std::string log;       // what the reference writes to stderr while loading
  uint64_t n_lines = 0, n_genes = 0;
};

static thread_local std::string g_annot_err;
extern "C" const char *mm_annotation_error(void) { return g_annot_err.c_str(); }

extern "C" int mm_annotation_load(mm_annotation **out, const char *config_path, const char *gtf_path, uint32_t upstream,
                                  uint32_t downstream) {
  if (!out || !config_path || !gtf_path) { g_annot_err = "mm_annotation_load: null argument"; return MM_ERR_ARG; }
  *out = nullptr;
  mm_annotation *A = new mm_annotation();
  std::string err = A->cfg.parse(config_path, A->log);
  if (!err.empty()) { g_annot_err = err; delete A; return MM_ERR_PARSE; }
  const ConfigModel &C = A->cfg;
  std::ifstream f(gtf_path);
  if (!f.good()) { g_annot_err = A->log + "Error, Annotation file '" + gtf_path + "' does not exists!"; delete A; return MM_ERR_IO; }
  A->log += "Reading GTF file\n";
  std::unordered_map<std::string, size_t> gene_of;  // any id (gene, transcript, ...) -> gene index; reset per chromosome block
  std::unordered_set<std::string> unused;
  std::vector<GeneModel> genes;
  genes.reserve(1 << 16);
  std::string line, chromosome;
  uint32_t chr_id = 0xFFFFFFFFu;
  uint64_t cpt = 0;
  for (; std::getline(f, line); ++cpt) {
    if (line.empty() || line[0] == '#') continue;
    GtfRecord r(line);
    if (!r.ok) {
      g_annot_err = "GTF line " + std::to_string(cpt) + " does not have 9 tab-separated fields with numeric coordinates (the reference aborts here, mmannot.cpp:526)";
      delete A; return MM_ERR_PARSE;
    }
    r.source = C.translate(r.source);
    r.type = C.translate(r.type);
    if (r.chromosome != chromosome) {
      gene_of.clear(); unused.clear();
      chromosome = r.chromosome;
      auto it = std::find(A->chromosomes.begin(), A->chromosomes.end(), chromosome);
      if (it == A->chromosomes.end()) { chr_id = (uint32_t) A->chromosomes.size(); A->chromosomes.push_back(chromosome); }
      else chr_id = (uint32_t) (it - A->chromosomes.begin());
    }
    if (r.type == "gene") {
      std::string gid;
      if (r.has("ID")) gid = r.first("ID");
      else if (r.has("gene_id")) gid = r.first("gene_id");
      else A->log += "Warning, cannot deduce gene id at line " + commas(cpt) + ": '" + line + "'.\n";
      gene_of[gid] = genes.size();
      genes.emplace_back(r, chr_id);
    } else if (r.type == "transcript") {
      std::string id, gid;
      if (r.has("ID")) id = r.first("ID");
      else if (r.has("transcript_id")) id = r.first("transcript_id");
      else A->log += "Warning, cannot deduce transcript id at line " + commas(cpt) + ": '" + line + "'.\n";
      if (r.has("Parent")) gid = r.first("Parent");
      else if (r.has("gene_id")) gid = r.first("gene_id");
      else A->log += "Warning, cannot deduce transcript parent id at line " + commas(cpt) + ": '" + line + "'.\n";
      if (unused.find(gid) == unused.end()) {
        auto it = gene_of.find(gid);
        if (it != gene_of.end()) { const size_t g = it->second; gene_of[id] = g; }
      }
    } else if (r.type == "exon") {
      std::string gid;
      if (r.has("Parent")) gid = r.first("Parent");
      else if (r.has("gene_id")) gid = r.first("gene_id");
      else if (r.has("transcript_id")) gid = r.first("transcript_id");
      else A->log += "Warning, cannot deduce exon id at line " + commas(cpt) + ": '" + line + "'.\n";
      if (unused.find(gid) == unused.end()) {
        auto it = gene_of.find(gid);
        if (it == gene_of.end()) {
          GeneModel g(r, chr_id);
          g.add_exon(r.start, r.end);
          gene_of[gid] = genes.size();
          genes.push_back(g);
        } else genes[it->second].add_exon(r.start, r.end);
      }
    } else if (r.type == "CDS") {
      std::string gid;
      if (r.has("gene_id")) gid = r.first("gene_id");
      else if (r.has("Parent")) gid = r.first("Parent");
      else if (r.has("transcript_id")) gid = r.first("transcript_id");
      else A->log += "Warning, cannot deduce CDS parent id at line " + commas(cpt) + ": '" + line + "'.\n";
      auto it = gene_of.find(gid);
      if (it == gene_of.end()) {
        GeneModel g(r, chr_id);
        g.add_cds(r.start, r.end);
        gene_of[gid] = genes.size();
        genes.push_back(g);
      } else genes[it->second].add_cds(r.start, r.end);
    } else if (r.type == "5'UTR" || r.type == "3'UTR") {
      // derived from exons and CDS instead
    } else if (C.order_of(r.source, r.type) >= 0) {
      std::string id;
      if (r.has("ID")) id = r.first("ID");
      else if (r.has("gene_id")) id = r.first("gene_id");
      else if (r.has("transcript_id")) id = r.first("transcript_id");
      else if (r.has("Parent")) id = r.first("Parent") + "_" + r.type;
      else A->log += "Warning, cannot deduce id at line " + commas(cpt) + ": '" + line + "'.\n";
      gene_of[id] = genes.size();
      genes.emplace_back(r, chr_id);
    } else {
      if (r.has("gene_id")) unused.insert(r.first("gene_id"));
      if (r.has("transcript_id")) unused.insert(r.first("transcript_id"));
      if (r.has("ID")) unused.insert(r.first("ID"));
    }
  }
  A->n_lines = cpt; A->n_genes = genes.size();
  A->log += "\t" + commas(cpt) + " lines read, done.  " + commas(genes.size()) + " genes found.\n";
  // gene -> typed intervals, in the reference's emission order
  std::vector<IvOut> &iv = A->iv;
  auto push = [&](const GeneModel &g, uint64_t s, uint64_t e, long t, const char *suffix) {
    iv.emplace_back();
    IvOut &o = iv.back();
    o.chr = g.chr; o.start = s; o.end = e; o.type = (uint32_t) t; o.strand = (uint8_t) g.strand;
    o.id.reserve(g.id.size() + 12); o.id = g.id; o.id += suffix;
  };
  std::vector<Seg> piece, u_left, u_right;
  iv.reserve(genes.size() * 6);
  for (GeneModel &g : genes) {
    std::sort(g.exons.begin(), g.exons.end(), [](const Seg &a, const Seg &b) { return a.s < b.s; });
    if (g.exons.empty()) g.exons.push_back(g.hull);
    g.start = g.hull.s; g.end = g.hull.e;
    long t;
    bool have_cds = false;
    Seg cds_hull;
    if (g.cds.set()) {
      GeneModel::clip(g.exons, g.cds.s, g.cds.e, piece);
      if (!piece.empty()) { have_cds = true; cds_hull.s = piece.front().s; cds_hull.e = piece.back().e; }
    } else piece.clear();
    if ((t = C.order_of(g.source, "CDS")) >= 0)
      for (const Seg &x : piece) push(g, x.s, x.e, t, "-CDS");
    u_left.clear(); u_right.clear();
    if (have_cds) {
      GeneModel::clip(g.exons, g.start, cds_hull.s - 1, u_left);
      GeneModel::clip(g.exons, cds_hull.e + 1, g.end, u_right);
      if (cds_hull.s - 1 < g.start) u_left.clear();   // an inverted window selects nothing
      if (g.end < cds_hull.e + 1) u_right.clear();
    }
    const std::vector<Seg> &u5 = (g.strand == MM_STRAND_R) ? u_right : u_left, &u3 = (g.strand == MM_STRAND_R) ? u_left : u_right;
    if ((t = C.order_of(g.source, "5'UTR")) >= 0)
      for (const Seg &x : u5) push(g, x.s, x.e, t, "-5UTR");
    if ((t = C.order_of(g.source, "3'UTR")) >= 0)
      for (const Seg &x : u3) push(g, x.s, x.e, t, "-3UTR");
    if ((t = C.intron_of(g.source, g.type)) >= 0)
      for (size_t k = 0; k + 1 < g.exons.size(); ++k) push(g, g.exons[k].e + 1, g.exons[k + 1].s - 1, t, "-intron");
    uint64_t up_s, up_e, dn_s, dn_e;
    if (g.strand == MM_STRAND_F) {
      up_s = (g.start <= upstream) ? 1 : g.start - upstream; up_e = g.start - 1;
      dn_s = g.end + 1; dn_e = g.end + downstream;
    } else {
      dn_s = (g.start <= downstream) ? 1 : g.start - downstream; dn_e = g.start - 1;
      up_s = g.end + 1; up_e = g.end + upstream;
    }
    if ((t = C.upstream_of(g.source, g.type)) >= 0) push(g, up_s, up_e, t, "-upstream");
    if ((t = C.downstream_of(g.source, g.type)) >= 0) push(g, dn_s, dn_e, t, "-downstream");
    if ((t = C.order_of(g.source, g.type)) >= 0)
      for (const Seg &x : g.exons) push(g, x.s, x.e, t, "");
  }
  // The reference sorts with std::sort on (chromosome, start) only (mmannot.cpp:697-699, 1267): the order it
  // leaves equal-start intervals in is observable (last writer wins).  Same libstdc++ algorithm, same input
  // order, same comparator => same permutation.
  // (sorted through 16-byte keys and then permuted: the algorithm's comparisons and moves do not depend on what else an element
  // carries, and moving the records themselves - each owns its id string - was a fifth of the loading time)
  {
    struct Key { uint32_t chr; uint32_t idx; uint64_t start; };
    std::vector<Key> keys(iv.size());
    for (size_t i = 0; i < iv.size(); ++i) { keys[i].chr = iv[i].chr; keys[i].idx = (uint32_t) i; keys[i].start = iv[i].start; }
    std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) { return a.chr < b.chr || (a.chr == b.chr && a.start < b.start); });
    std::vector<IvOut> sorted;
    sorted.reserve(iv.size());
    for (const Key &k : keys) sorted.push_back(std::move(iv[k.idx]));
    iv.swap(sorted);
  }
  if (iv.empty()) {
    g_annot_err = "Error, the annotation file has not been parsed properly!\nPlease check that your annotation file is not empty, and that your configuration file matches your annotation file.\nIf you have trouble designing a configuration file, please use the companion tool 'createConfigFile'.";
    g_annot_err = A->log + g_annot_err;  // the log written so far comes first, as on the reference's stderr
    delete A; return MM_ERR_PARSE;
  }
  A->log += "\t" + commas(iv.size()) + " intervals found.\n";
  // flat views
  const size_t E = C.flat.size();
  A->line_of.resize(E); A->e_strand.resize(E); A->e_up.resize(E); A->e_down.resize(E); A->names.resize(E);
  for (size_t e = 0; e < E; ++e) {
    A->line_of[e] = C.flat[e].first; A->e_strand[e] = (uint8_t) C.entry(e).strand;
    A->e_up[e] = C.entry(e).type == "upstream"; A->e_down[e] = C.entry(e).type == "downstream";
    A->names[e] = C.name(e);
  }
  A->chr_has_interval.assign(A->chromosomes.size(), 0);
  const size_t I = iv.size();
  A->a_chr.resize(I); A->a_start.resize(I); A->a_end.resize(I); A->a_type.resize(I); A->a_strand.resize(I);
  for (size_t i = 0; i < I; ++i) {
    if (iv[i].start > 0x7FFFFFF0ull || iv[i].end > 0x7FFFFFF0ull || E > 255) {
      g_annot_err = "annotation coordinate or element count outside the supported range"; delete A; return MM_ERR_RANGE;
    }
    A->a_chr[i] = (int32_t) iv[i].chr; A->a_start[i] = (int32_t) iv[i].start; A->a_end[i] = (int32_t) iv[i].end;
    A->a_type[i] = (uint8_t) iv[i].type; A->a_strand[i] = iv[i].strand;
    A->chr_has_interval[iv[i].chr] = 1;
  }
  *out = A;
  return MM_OK;
}

extern "C" void mm_annotation_free(mm_annotation *a) { delete a; }
extern "C" const char *mm_annotation_log(const mm_annotation *a) { return a ? a->log.c_str() : ""; }
extern "C" int32_t mm_annotation_n_elements(const mm_annotation *a) { return a ? (int32_t) a->cfg.flat.size() : 0; }
extern "C" int32_t mm_annotation_n_chromosomes(const mm_annotation *a) { return a ? (int32_t) a->chromosomes.size() : 0; }
extern "C" size_t mm_annotation_n_intervals(const mm_annotation *a) { return a ? a->iv.size() : 0; }
extern "C" const char *mm_annotation_element_name(const mm_annotation *a, int32_t e) {
  return (a && e >= 0 && (size_t) e < a->names.size()) ? a->names[e].c_str() : "";
}
extern "C" const char *mm_annotation_chromosome(const mm_annotation *a, int32_t c) {
  return (a && c >= 0 && (size_t) c < a->chromosomes.size()) ? a->chromosomes[c].c_str() : "";
}
extern "C" int mm_annotation_chromosome_has_intervals(const mm_annotation *a, int32_t c) {
  return (a && c >= 0 && (size_t) c < a->chr_has_interval.size()) ? a->chr_has_interval[c] : 0;
}
extern "C" const char *mm_annotation_interval_id(const mm_annotation *a, size_t i) {
  return (a && i < a->iv.size()) ? a->iv[i].id.c_str() : "";
}
extern "C" int mm_annotation_elements(const mm_annotation *a, mm_elements *out) {
  if (!a || !out) return MM_ERR_ARG;
  out->n_elements = (int32_t) a->cfg.flat.size();
  out->line_of = a->line_of.data(); out->elem_strand = a->e_strand.data(); out->is_up = a->e_up.data(); out->is_down = a->e_down.data();
  return MM_OK;
}
extern "C" int mm_annotation_intervals(const mm_annotation *a, const int32_t **chr, const int32_t **start, const int32_t **end,
                                       const uint8_t **type, const uint8_t **strand) {
  if (!a) return MM_ERR_ARG;
  if (chr) *chr = a->a_chr.data();
  if (start) *start = a->a_start.data();
  if (end) *end = a->a_end.data();
  if (type) *type = a->a_type.data();
  if (strand) *strand = a->a_strand.data();
  return MM_OK;
}
// convenience: annotation -> device index
extern "C" int mm_index_from_annotation(mm_index **out, const mm_annotation *a, int device, int bin_shift) {
  if (!a || !out) return MM_ERR_ARG;
  mm_elements el;
  mm_annotation_elements(a, &el);
  return mm_index_create(out, &el, (int32_t) a->chromosomes.size(), a->iv.size(), a->a_chr.data(), a->a_start.data(), a->a_end.data(),
                         a->a_type.data(), a->a_strand.data(), device, bin_shift);
}

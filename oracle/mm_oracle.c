/*
 * mm_oracle.c — TEST INFRASTRUCTURE ONLY (never linked, imported or called by the
 * product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it, and only as the checker).
 *
 * A plain-C restatement of the overlap-and-count hot path of mzytnicki/mmannot
 * v1.1 (reference file: mmannot.cpp).  Every function cites the reference lines
 * it follows.  It deliberately follows the reference's *algorithm* (binned walk,
 * per-element last-writer table, sequential name-keyed group map, file-order
 * double accumulation), not the closed forms the HIP kernels use, so that the
 * two are independent derivations of the same behaviour.
 *
 * PARITY PIN: this restatement is checked (tests/test_oracle_golden.py) against
 * outputs of the reference itself compiled in the build container
 * (oracle/build_ref.sh -> oracle/_ref/mmannot*), committed as derived fixtures
 * under tests/golden/ (count tables, stderr statistics, per-hit region sets,
 * edge-case known-answer BAMs).  The reference ships no tests of its own
 * (SURVEY.md section 4).
 *
 * Domain restrictions (asserted): number of Order elements E <= 64 so that a set
 * of element ids is a 64-bit mask; coordinates fit int64.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>

#define MMO_BIN_SIZE 16384 /* mmannot.cpp:67 binSize */

enum { MMO_STRAND_ALL = 0, MMO_STRAND_F = 1, MMO_STRAND_R = 2 }; /* mmannot.cpp:49 */
enum { MMO_OVL_INCLUSION = 0, MMO_OVL_PC = 1, MMO_OVL_NT = 2 };  /* mmannot.cpp:992-1002, 1972-1977 */
enum { MMO_STRANDED_U = 0, MMO_STRANDED_F = 1, MMO_STRANDED_R = 2 }; /* mmannot.cpp:836-844 */
enum { MMO_STRAT_DEFAULT = 0, MMO_STRAT_UNIQUE = 1, MMO_STRAT_RANDOM = 2, MMO_STRAT_RATIO = 3 }; /* :50 */

typedef struct {
  int32_t n_elements;          /* Config::nElements                      mmannot.cpp:226 */
  const int32_t *line_of;      /* RegionType.id of element e             mmannot.cpp:370-374 */
  const uint8_t *elem_strand;  /* AnnotationType.strand of element e     mmannot.cpp:302-305 */
  const uint8_t *is_up;        /* Config::isUpstream(e)                  mmannot.cpp:463-466 */
  const uint8_t *is_down;      /* Config::isDownstream(e)                mmannot.cpp:467-470 */
  int32_t overlap_mode;        /* which intervalOverlapFunction          mmannot.cpp:1972-1977 */
  float overlap_value;         /* Globals::overlap (float)               mmannot.cpp:70 */
  int32_t strandedness;        /* Globals::strandednessFunction          mmannot.cpp:1981-1983 */
  int32_t strategy;            /* Globals::strategy                      mmannot.cpp:2036-2039 */
} mmo_config;

typedef struct {               /* sorted TypedInterval vector, mmannot.cpp:1088, 1267 */
  int64_t n_intervals;
  int32_t n_chromosomes;
  const int32_t *chr;          /* chromosomeId   */
  const int64_t *start;        /* 1-based inclusive */
  const int64_t *end;
  const int32_t *type;         /* element id     */
  const uint8_t *strand;       /* MMO_STRAND_F / MMO_STRAND_R (GTF col 7, mmannot.cpp:530) */
} mmo_index;

typedef struct {               /* one entry per Read handed to scan(), mmannot.cpp:846-903 */
  int64_t n;
  const int32_t *tid;          /* index into the GTF chromosome table; <0 = chromosome unknown to the GTF (mmannot.cpp:1293-1302) */
  const int64_t *start;        /* record.start = pos0+1   mmannot.cpp:1537 */
  const int64_t *end;          /* after parseCigar        mmannot.cpp:852-875 */
  const uint8_t *strand;       /* record.strand: 1 = flag 0x10 clear  mmannot.cpp:608 */
  const uint32_t *nh;          /* record.nHits            mmannot.cpp:1560, 1637-1641 */
  const int64_t *qid;          /* dense id of the read *name* (the reference keys on the string, mmannot.cpp:1671) */
} mmo_reads;

typedef struct {
  /* count table: distinct element sets (as masks) with the raw double and the printed unsigned */
  int64_t n_rows, cap_rows;
  uint64_t *row_mask;          /* sorted as TableCount::dump sorts   mmannot.cpp:1883 */
  double *row_sum;             /* regionCounts value                 mmannot.cpp:1658 */
  uint32_t *row_count;         /* round() -> unsigned int            mmannot.cpp:1868 */
  uint64_t stats[7];           /* nHits nReads nUnique nAmbiguous nMultiple nUnassigned nRescued  mmannot.cpp:1663 */
  uint64_t *hit_mask;          /* optional [n]: regions of each hit handed to addCount (all-ones = hit skipped by -y unique) */
  int64_t walked;              /* total intervals visited by the second loop of scan (profiling aid) */
} mmo_result;

/* ------------------------------------------------------------------ bins */
/* mmannot.cpp:1277-1284: per chromosome, entry b = global index of the first
 * interval (in sorted order) whose end / binSize >= b. */
typedef struct { int64_t *first; int64_t n; } mmo_bins;

static mmo_bins *build_bins(const mmo_index *ix) {
  mmo_bins *bins = (mmo_bins *) calloc((size_t) ix->n_chromosomes, sizeof(mmo_bins));
  for (int64_t i = 0; i < ix->n_intervals; ++i) {
    mmo_bins *cb = &bins[ix->chr[i]];
    int64_t bin = (int64_t) ((uint32_t) (ix->end[i] / MMO_BIN_SIZE)); /* 'unsigned bin' :1279 */
    if (cb->n <= bin) {                                                /* :1280 */
      cb->first = (int64_t *) realloc(cb->first, (size_t) (bin + 1) * sizeof(int64_t));
      for (int64_t b = cb->n; b <= bin; ++b) cb->first[b] = i;         /* insert(end, bin-size+1, i) :1282 */
      cb->n = bin + 1;
    }
  }
  return bins;
}

/* ------------------------------------------------------------------ overlap functions */
/* Interval::overlaps mmannot.cpp:632-636 (unsigned arithmetic: s >= e -> 0, else e - s; no +1) */
static inline uint64_t iv_overlaps(int64_t s1, int64_t e1, int64_t s2, int64_t e2) {
  uint64_t s, e;
  if ((uint64_t) s1 > (uint64_t) s2) s = (uint64_t) s1; else s = (uint64_t) s2;
  if ((uint64_t) e1 < (uint64_t) e2) e = (uint64_t) e1; else e = (uint64_t) e2;
  if (s >= e) return 0;
  return e - s;
}
/* intervalInclusion :992-994 / intervalOverlapPc :995-998 / intervalOverlap :999-1002.
 * (i = annotation interval, r = read).  Globals::overlap is a float, so the
 * products/compares happen in float exactly as C++ usual arithmetic conversions
 * give them: size_t * float -> float; size_t >= float -> float compare. */
static inline uint64_t overlap_fn(const mmo_config *c, int64_t is, int64_t ie, int64_t rs, int64_t re) {
  if (c->overlap_mode == MMO_OVL_INCLUSION) {
    return ((uint64_t) rs >= (uint64_t) is && (uint64_t) re <= (uint64_t) ie) ? 1 : 0; /* Interval::includes :637-639 */
  }
  uint64_t o = iv_overlaps(is, ie, rs, re);
  if (c->overlap_mode == MMO_OVL_PC) {
    uint64_t size = (uint64_t) re - (uint64_t) rs + 1;                 /* Interval::getSize :629-631 */
    volatile float lhs = (float) size * c->overlap_value;
    return (lhs <= (float) o) ? o : 0;
  }
  return ((float) o >= c->overlap_value) ? o : 0;
}
/* Read::getDistance(p) — Interval::getDistance mmannot.cpp:661-665 */
static inline uint64_t rd_distance(int64_t rs, int64_t re, int64_t p) {
  if ((uint64_t) p < (uint64_t) rs) return (uint64_t) rs - (uint64_t) p;
  if ((uint64_t) p > (uint64_t) re) return (uint64_t) p - (uint64_t) re;
  return 0;
}
/* Config::checkStrand mmannot.cpp:438-443 */
static inline int check_strand(uint8_t elem_strand, uint8_t iv_strand, int read_strand) {
  if (elem_strand == MMO_STRAND_ALL) return 1;
  if (elem_strand == MMO_STRAND_F)
    return ((iv_strand == MMO_STRAND_F) && read_strand) || ((iv_strand == MMO_STRAND_R) && !read_strand);
  return ((iv_strand == MMO_STRAND_F) && !read_strand) || ((iv_strand == MMO_STRAND_R) && read_strand);
}
/* strandF / strandR / strandU mmannot.cpp:836-844 applied in Read::reset :884 */
static inline int read_strand(const mmo_config *c, uint8_t record_strand) {
  if (c->strandedness == MMO_STRANDED_U) return 1;
  if (c->strandedness == MMO_STRANDED_F) return record_strand != 0;
  return record_strand == 0;
}

/* ------------------------------------------------------------------ scan + getFirst */
/* IntervalList::scan mmannot.cpp:1291-1332 with EvaluationStructure :1012-1076.
 * Returns the region set as a mask; *n_regions gets its size.  ev_o/ev_d are
 * caller-provided scratch of E entries (the two heap vectors of :1018). */
static uint64_t scan_one(const mmo_config *c, const mmo_index *ix, const mmo_bins *bins,
                         int32_t tid, int64_t rs, int64_t re, int rstrand,
                         uint64_t *ev_o, uint64_t *ev_d, int64_t *walked,
                         int64_t *sel_iv, int64_t *n_sel_iv /* optional -M support: ids[type] lists */) {
  const int E = c->n_elements;
  (void) sel_iv; (void) n_sel_iv;
  if (tid < 0 || tid >= ix->n_chromosomes || bins[tid].n == 0) return 0;  /* :1293-1302 chromosome not in bins */
  uint32_t bin = (uint32_t) ((uint64_t) rs / MMO_BIN_SIZE);               /* :1292 */
  const mmo_bins *cb = &bins[tid];
  if ((int64_t) bin > cb->n - 1) bin = (uint32_t) (cb->n - 1);            /* :1303 */
  int64_t id = cb->first[bin];                                            /* :1304 */
  int32_t chr = ix->chr[id];                                              /* :1305 */
  while (id < ix->n_intervals && ix->chr[id] == chr &&
         (uint64_t) ix->end[id] < (uint64_t) rs) ++id;                    /* isBefore :1306-1308 */
  for (int e = 0; e < E; ++e) { ev_o[e] = 0; ev_d[e] = 0; }               /* :1018-1022 */
  unsigned n_set = 0;
  while (id < ix->n_intervals && ix->chr[id] == chr &&
         !((uint64_t) re < (uint64_t) ix->start[id])) {                   /* !isAfter :1311 */
    int32_t t = ix->type[id];
    if (check_strand(c->elem_strand[t], ix->strand[id], rstrand)) {       /* :1313 */
      uint64_t o = overlap_fn(c, ix->start[id], ix->end[id], rs, re);     /* :1315 */
      if (o > 0) {
        uint64_t d = 0;
        if (c->is_up[t]) d = rd_distance(rs, re, ix->end[id]);            /* :1317-1319 */
        else if (c->is_down[t]) d = rd_distance(rs, re, ix->start[id]);   /* :1320-1322 */
        ev_o[t] = o; ev_d[t] = d; ++n_set;                                /* set() :1023-1028 (overwrite) */
      }
    }
    ++id; if (walked) ++*walked;
  }
  /* getFirst :1029-1076 */
  if (n_set == 0) return 0;
  int good = -1;
  uint64_t selected = 0, max_o = 0;
  for (int i = 0; i < E; ++i) {
    if (good != -1 && c->line_of[i] != good) break;                       /* :1037-1039 */
    uint64_t o = 0;
    if (ev_o[i] > 0) { o = ev_o[i]; good = c->line_of[i]; }               /* :1042-1045 */
    if (o > 0) {
      if (o > max_o) { selected = 1ull << i; max_o = o; }                 /* :1050-1053 */
      else if (o == max_o) selected |= 1ull << i;                         /* :1054-1056 */
    }
  }
  if (selected == 0) return 0;                                            /* :1059-1061 */
  if ((selected & (selected - 1)) == 0) return selected;                  /* :1062-1065 */
  uint64_t min_d = UINT64_MAX, regions = 0;
  for (int i = 0; i < E; ++i) {
    if (!(selected >> i & 1)) continue;
    if (ev_d[i] < min_d) { min_d = ev_d[i]; regions = 1ull << i; }        /* :1068-1071 */
    else if (ev_d[i] == min_d) regions |= 1ull << i;                      /* :1072-1074 */
  }
  return regions;
}

/* ------------------------------------------------------------------ small hash maps */
/* regionCounts: unordered_map<vector<size_t>,double> mmannot.cpp:1658 -> mask keyed. */
typedef struct { uint64_t *key; double *val; uint8_t *used; size_t cap, n; } rc_map;
static void rc_init(rc_map *m, size_t cap) {
  m->cap = cap; m->n = 0;
  m->key = (uint64_t *) calloc(cap, 8); m->val = (double *) calloc(cap, 8); m->used = (uint8_t *) calloc(cap, 1);
}
static void rc_free(rc_map *m) { free(m->key); free(m->val); free(m->used); }
static size_t rc_slot(rc_map *m, uint64_t k);
static void rc_grow(rc_map *m) {
  rc_map o = *m; rc_init(m, o.cap * 2);
  for (size_t i = 0; i < o.cap; ++i) if (o.used[i]) { size_t s = rc_slot(m, o.key[i]); m->val[s] = o.val[i]; }
  rc_free(&o);
}
static size_t rc_slot(rc_map *m, uint64_t k) {
  if ((m->n + 1) * 2 > m->cap) rc_grow(m);
  uint64_t h = k * 0x9E3779B97F4A7C15ull; h ^= h >> 29;
  size_t s = (size_t) (h & (m->cap - 1));
  while (m->used[s] && m->key[s] != k) s = (s + 1) & (m->cap - 1);
  if (!m->used[s]) { m->used[s] = 1; m->key[s] = k; m->val[s] = 0.0; ++m->n; }
  return s;
}

/* the slot must be resolved before m->val is read: rc_slot may grow (reallocate) the table */
static void rc_add(rc_map *m, uint64_t k, double x) {
  size_t s = rc_slot(m, k);
  m->val[s] += x;
}

/* readCounts / rawCounts: unordered_map<string, pair<unsigned, vector<size_t>>> mmannot.cpp:1656-1657,
 * keyed here by the dense name id.  'present' distinguishes "in the map". */
typedef struct { uint32_t remaining; uint32_t raw; uint64_t set; uint8_t present; } grp_state;

/* chosenId / numberSeen / seen for -y random, mmannot.cpp:1661-1662 */
typedef struct { uint8_t seen, has_choice; uint64_t chosen, number_seen; } rnd_state;

/* TableCount::dump ordering, mmannot.cpp:1883: std::sort over vector<size_t> keys =
 * lexicographic order of the ascending element-id lists. */
static int cmp_lists(uint64_t a, uint64_t b) {
  while (a && b) {
    int ea = __builtin_ctzll(a), eb = __builtin_ctzll(b);
    if (ea != eb) return ea < eb ? -1 : 1;
    a &= a - 1; b &= b - 1;
  }
  if (!a && !b) return 0;
  return a ? 1 : -1;   /* shorter list (proper prefix) sorts first */
}
typedef struct { uint64_t mask; double sum; } row_t;
static int cmp_rows(const void *x, const void *y) {
  return cmp_lists(((const row_t *) x)->mask, ((const row_t *) y)->mask);
}

/* TableCount::addCounter mmannot.cpp:1868/1872: `unsigned int v = round(double)`.
 * On the reference's platform (x86-64, g++ -O3) double->unsigned int is cvttsd2si
 * to 64 bit followed by truncation to 32 bit; out-of-range/inf/nan give
 * 0x8000000000000000 -> 0 (observed: KAT E4, NH:c tag -> nHits 0 -> 1.0/0). */
uint32_t mmo_round_to_uint(double v) {
  double r = round(v);
  if (!(r > -9223372036854775808.0 && r < 9223372036854775808.0)) return 0u;
  return (uint32_t) (uint64_t) (int64_t) r;
}

/* ------------------------------------------------------------------ Counter::read + addCount */
/* Counter::read mmannot.cpp:1767-1792 driving IntervalList::scan and
 * Counter::addCount mmannot.cpp:1665-1739.  Returns 0 on success. */
int mmo_run(const mmo_config *c, const mmo_index *ix, const mmo_reads *rd, mmo_result *res) {
  const int E = c->n_elements;
  if (E < 1 || E > 64) return -1;
  mmo_bins *bins = build_bins(ix);
  uint64_t *ev_o = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) E);
  uint64_t *ev_d = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) E);
  int64_t max_qid = -1;
  for (int64_t i = 0; i < rd->n; ++i) if (rd->qid && rd->qid[i] > max_qid) max_qid = rd->qid[i];
  grp_state *grp = NULL; rnd_state *rnd = NULL;
  if (c->strategy == MMO_STRAT_DEFAULT) grp = (grp_state *) calloc((size_t) (max_qid + 2), sizeof(grp_state));
  if (c->strategy == MMO_STRAT_RANDOM) { rnd = (rnd_state *) calloc((size_t) (max_qid + 2), sizeof(rnd_state)); srand(1); }
  rc_map rc; rc_init(&rc, 1024);
  uint64_t nHits = 0, nReads = 0, nUnique = 0, nAmbiguous = 0, nMultiple = 0, nUnassigned = 0, nRescued = 0; /* clear() :1746 */
  res->walked = 0;

  for (int64_t i = 0; i < rd->n; ++i) {                                   /* :1772 */
    uint32_t nh = rd->nh ? rd->nh[i] : 1u;
    if (res->hit_mask) res->hit_mask[i] = ~0ull;
    if (!((c->strategy != MMO_STRAT_UNIQUE) || (nh == 1))) continue;     /* :1773 */
    ++nHits;                                                              /* :1774 */
    uint64_t regions = scan_one(c, ix, bins, rd->tid[i], rd->start[i], rd->end[i],
                                read_strand(c, rd->strand ? rd->strand[i] : 1), ev_o, ev_d, &res->walked, NULL, NULL);
    if (res->hit_mask) res->hit_mask[i] = regions;
    int n_regions = __builtin_popcountll(regions);
    /* addCount :1665 */
    if (n_regions == 0) nUnassigned++;                                    /* :1666 */
    else if (n_regions > 1) nAmbiguous++;                                 /* :1667 */
    else if (nh == 1) nUnique++;                                          /* :1668 */
    if (nh > 1 && c->strategy == MMO_STRAT_DEFAULT) {                     /* :1669 */
      nMultiple++;
      grp_state *g = &grp[rd->qid[i]];
      if (!g->present) {                                                  /* :1672-1677 */
        g->present = 1; g->remaining = nh - 1; g->set = regions; g->raw = nh;
        ++nReads;
      } else {                                                            /* :1678-1701 */
        g->remaining--; g->set |= regions;
        if (g->remaining == 0) {
          if (g->set) {                                                   /* :1685 */
            rc_add(&rc, g->set, 1.0);                                     /* ++regionCounts[...] :1688 */
            if (__builtin_popcountll(g->set) == 1) nRescued++;            /* :1689 */
          }
          g->present = 0; g->set = 0;                                     /* erase :1698-1699 */
        }
      }
    } else {                                                              /* :1703 */
      if (regions) {
        int output = 0;
        if (c->strategy == MMO_STRAT_RANDOM) {                            /* :1706-1726 */
          rnd_state *r = &rnd[rd->qid[i]];
          if (!r->seen) {
            uint64_t ch;
            if (!r->has_choice) { ch = (uint64_t) (rand() % nh); r->chosen = ch; r->has_choice = 1; r->number_seen = 0; }
            else { ch = r->chosen; ++r->number_seen; }
            if (r->number_seen == ch) { output = 1; r->has_choice = 0; r->number_seen = 0; r->seen = 1; }
          }
        }
        if (c->strategy != MMO_STRAT_RANDOM || output) {                  /* :1727-1730 */
          rc_add(&rc, regions, (c->strategy == MMO_STRAT_RATIO) ? 1.0 / (double) nh : 1.0);
        }
      }
      ++nReads;                                                           /* :1737 */
    }
  }
  /* flush of still-pending multi-hit reads :1783-1792 (map iteration order is irrelevant
   * for the sums: only DEFAULT fills readCounts, and there every flush adds exactly 1) */
  if (grp) {
    for (int64_t q = 0; q <= max_qid; ++q) {
      grp_state *g = &grp[q];
      if (g->present && g->set) {
        rc_add(&rc, g->set, 1.0);
        if (g->raw > 1 && __builtin_popcountll(g->set) == 1) nRescued++;  /* :1789 */
      }
    }
  }
  /* TableCount::addCounter :1861-1876 + dump ordering :1883 */
  row_t *rows = (row_t *) malloc(sizeof(row_t) * (rc.n ? rc.n : 1));
  size_t nr = 0;
  for (size_t s = 0; s < rc.cap; ++s) if (rc.used[s]) { rows[nr].mask = rc.key[s]; rows[nr].sum = rc.val[s]; ++nr; }
  qsort(rows, nr, sizeof(row_t), cmp_rows);
  if ((int64_t) nr > res->cap_rows) { free(rows); rc_free(&rc); free(grp); free(rnd); free(ev_o); free(ev_d); return -2; }
  for (size_t r = 0; r < nr; ++r) {
    res->row_mask[r] = rows[r].mask; res->row_sum[r] = rows[r].sum; res->row_count[r] = mmo_round_to_uint(rows[r].sum);
  }
  res->n_rows = (int64_t) nr;
  res->stats[0] = nHits; res->stats[1] = nReads; res->stats[2] = nUnique; res->stats[3] = nAmbiguous;
  res->stats[4] = nMultiple; res->stats[5] = nUnassigned; res->stats[6] = nRescued;
  free(rows); rc_free(&rc); free(grp); free(rnd); free(ev_o); free(ev_d);
  for (int32_t k = 0; k < ix->n_chromosomes; ++k) free(bins[k].first);
  free(bins);
  return 0;
}

/* Comparison helper exported for the tests of the product's row ordering. */
int mmo_cmp_lists(uint64_t a, uint64_t b) { return cmp_lists(a, b); }

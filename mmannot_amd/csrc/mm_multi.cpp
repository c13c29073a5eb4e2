// mm_multi.cpp — one input on several GPUs: the host-side partitioner and the table merge of a process
// that owns all the contexts.  Replaces what the reference cannot do at all: its thread pool only spreads whole files
// (mmannot.cpp:2117-2143).
//   * -y default / unique: every hit of one read name has to meet on one device (the multi-hit countdown, mmannot.cpp:1669-1701),
//     so mm_partition deals the tuples of a batch out by qid mod N, file order kept inside each part; names are renumbered
//     qid / N so that every device sees dense ids.  Tables and counters then add up (groups are never split).
//   * -y ratio: regionCounts[set] += 1.0/nHits is a file-order double sum (mmannot.cpp:1730) - whole batches go round-robin to
//     contexts created with defer_ratio, and mm_merge replays the parked accumulations in file order, handing the row sums
//     (bit patterns) from context to context.
// Across processes (one rank per GPU) the same merge is an all_gather of keys + all_reduce of counts over RCCL:
// mmannot_amd/dist.py.
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/mmannot_amd.h"

extern "C" int mm_partition(const mm_aln *aln, size_t n, const mm_nh_big *big, size_t n_big, int n_parts, mm_aln *const *out, size_t *out_n,
                            mm_nh_big *const *out_big, size_t big_cap, size_t *out_n_big, uint32_t *out_max_nh, uint64_t *out_qid_lo) {
  if (n_parts < 1 || !out || !out_n || (n && !aln) || (n_big && (!big || !out_big || !out_n_big))) return MM_ERR_ARG;
  for (int p = 0; p < n_parts; ++p) { out_n[p] = 0; if (out_n_big) out_n_big[p] = 0; if (out_max_nh) out_max_nh[p] = 0; if (out_qid_lo) out_qid_lo[p] = ~0ull; }
  size_t bi = 0;
  const uint32_t N = (uint32_t) n_parts;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t q = aln[i].qid, p = q % N;
    mm_aln a = aln[i];
    a.qid = q / N;
    const size_t at = out_n[p]++;
    out[p][at] = a;
    uint32_t nh = MM_ALN_NH_FIELD(a.tsn);
    if (nh == MM_ALN_NH_ESCAPE) {
      while (bi < n_big && big[bi].index < i) ++bi;
      if (bi >= n_big || big[bi].index != i) return MM_ERR_ARG;          // escaped tuple without an entry
      if (out_n_big[p] >= big_cap) return MM_ERR_RANGE;
      mm_nh_big &e = out_big[p][out_n_big[p]++];
      e.index = (uint32_t) at; e.nh = big[bi].nh;
      nh = big[bi].nh;
    }
    if (out_max_nh && nh > out_max_nh[p]) out_max_nh[p] = nh;
    if (out_qid_lo && nh > 1 && a.qid < out_qid_lo[p]) out_qid_lo[p] = a.qid;
  }
  if (out_qid_lo) for (int p = 0; p < n_parts; ++p) if (out_qid_lo[p] == ~0ull) out_qid_lo[p] = 0;
  return MM_OK;
}

extern "C" int mm_merge(mm_ctx *const *ctxs, int n_ctx, int strategy, uint64_t *masks, double *sums, size_t cap, size_t *n_rows,
                        uint64_t stats[MM_N_STATS]) {
  if (!ctxs || n_ctx < 1 || !n_rows || !stats) return MM_ERR_ARG;
  for (int s = 0; s < MM_N_STATS; ++s) stats[s] = 0;
  for (int k = 0; k < n_ctx; ++k) {
    uint64_t st[MM_N_STATS];
    const int rc = mm_get_stats(ctxs[k], st);
    if (rc) return rc;
    for (int s = 0; s < MM_N_STATS; ++s) stats[s] += st[s];
  }
  std::vector<uint64_t> m;
  std::vector<double> v;
  auto rows_of = [&](mm_ctx *c) -> int {
    size_t nr = 0;
    int rc = mm_get_rows(c, nullptr, nullptr, 0, &nr);
    if (rc) return rc;
    m.assign(nr ? nr : 1, 0); v.assign(nr ? nr : 1, 0.0);
    rc = nr ? mm_get_rows(c, m.data(), v.data(), nr, &nr) : MM_OK;
    m.resize(nr); v.resize(nr);
    return rc;
  };
  if (strategy == MM_STRAT_RATIO && n_ctx > 1) {
    // batch j of the input went to ctxs[j % n_ctx]: replay the parked accumulations in that order, each one starting from the
    // sums its predecessor ended with
    bool first = true;
    for (size_t j = 0;; ++j) {
      mm_ctx *c = ctxs[j % (size_t) n_ctx];
      if (mm_ratio_pending(c) == 0) break;
      if (!first && !m.empty()) { const int rc = mm_set_row_sums(c, m.data(), v.data(), m.size()); if (rc) return rc; }
      int rc = mm_ratio_resolve_n(c, 1);
      if (rc) return rc;
      rc = rows_of(c);
      if (rc) return rc;
      first = false;
    }
    for (int k = 0; k < n_ctx; ++k) if (mm_ratio_pending(ctxs[k]) != 0) return MM_ERR_STATE;   // batches were not dealt round-robin from ctxs[0]
    if (first) { const int rc = rows_of(ctxs[0]); if (rc) return rc; }   // nothing parked (any more): every context holds the final rows
    else for (int k = 0; k < n_ctx; ++k) {   // leave the input's final sums in every context (a second call - sizes first, rows then - finds them)
      const int rc = m.empty() ? MM_OK : mm_set_row_sums(ctxs[k], m.data(), v.data(), m.size());
      if (rc) return rc;
    }
  } else if (n_ctx == 1) {
    const int rc = rows_of(ctxs[0]);
    if (rc) return rc;
  } else {
    std::map<uint64_t, double> all;     // counts are integers here: exact in double up to 2^53
    for (int k = 0; k < n_ctx; ++k) {
      const int rc = rows_of(ctxs[k]);
      if (rc) return rc;
      for (size_t r = 0; r < m.size(); ++r) all[m[r]] += v[r];
    }
    m.clear(); v.clear();
    for (auto &e : all) { m.push_back(e.first); v.push_back(e.second); }
  }
  *n_rows = m.size();
  if (!masks) return MM_OK;
  if (m.size() > cap) return MM_ERR_ARG;
  for (size_t r = 0; r < m.size(); ++r) { masks[r] = m[r]; if (sums) sums[r] = v[r]; }
  return MM_OK;
}

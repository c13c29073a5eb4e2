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
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "../../include/mmannot_amd.h"

// a 16-byte tuple written past the caches (the pinned staging buffer it goes to is read by the copy engine)
static inline void stream_store(mm_aln *dst, const mm_aln &a) {
#if defined(__SSE2__)
  __m128i v;
  memcpy(&v, &a, 16);
  _mm_stream_si128((__m128i *) dst, v);
#else
  *dst = a;
#endif
}
static inline void stream_fence() {
#if defined(__SSE2__)
  _mm_sfence();
#endif
}

// ---- a few persistent worker threads for mm_partition (a batch is dealt out in ~1 ms: starting threads per call would cost as much)
namespace {
struct PartPool {
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::vector<std::thread> workers;
  const std::function<void(int)> *job = nullptr;
  uint64_t generation = 0;
  int pending = 0, n_workers = 0;
  bool stop = false;
  std::mutex call_mu;                       // one mm_partition at a time uses the pool (others run their slices themselves)
  void ensure(int n) {
    std::lock_guard<std::mutex> lk(mu);
    while ((int) workers.size() < n) {
      const int id = (int) workers.size() + 1;       // worker ids 1..n; the caller is 0
      workers.emplace_back([this, id] {
        uint64_t seen = 0;
        for (;;) {
          const std::function<void(int)> *j;
          {
            std::unique_lock<std::mutex> lk2(mu);
            cv_go.wait(lk2, [&] { return stop || (generation != seen && id <= n_workers); });
            if (stop) return;
            seen = generation; j = job;
          }
          (*j)(id);
          { std::lock_guard<std::mutex> lk2(mu); if (--pending == 0) cv_done.notify_all(); }
        }
      });
    }
  }
  // runs f(0..n-1): f(0) on the calling thread, the rest on the workers
  void run(int n, const std::function<void(int)> &f) {
    if (n <= 1) { f(0); return; }
    ensure(n - 1);
    { std::lock_guard<std::mutex> lk(mu); job = &f; n_workers = n - 1; pending = n - 1; ++generation; }
    cv_go.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
    n_workers = 0;
  }
  ~PartPool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_go.notify_all();
    for (auto &w : workers) w.join();
  }
};
PartPool &part_pool() { static PartPool p; return p; }

std::atomic<int> g_part_threads{0};       // 0 = not set: the cores this process may use, 16 at most
std::atomic<bool> g_part_threads_forced{false};
int partition_threads(size_t n) {
  int t = g_part_threads.load();
  if (t <= 0) {
    t = (int) std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    cpu_set_t set;                                          // (a GPU box hands a job a share of the host's cores)
    if (sched_getaffinity(0, sizeof set, &set) == 0) t = std::min(t, std::max(1, CPU_COUNT(&set)));
    if (const char *e = getenv("MM_PARTITION_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) { t = v; g_part_threads_forced = true; } }
    g_part_threads = t;
  }
  if (g_part_threads_forced.load()) return (int) std::max<size_t>(1, std::min<size_t>((size_t) t, n));
  return (int) std::max<size_t>(1, std::min<size_t>((size_t) t, n >> 16));   // below 64 K tuples per thread the hand-over costs more than it saves
}

// the part and the renumbered name of read-name id q; N is a compile-time constant for the usual device counts (a division by a
// run-time value costs ~25 cycles, twice per tuple)
template <uint32_t NC> struct Deal {
  uint32_t N;
  explicit Deal(uint32_t n) : N(n) {}
  inline uint32_t part(uint32_t q) const { return NC ? q % NC : q % N; }
  inline uint32_t name(uint32_t q) const { return NC ? q / NC : q / N; }
};

struct SliceStat { std::vector<size_t> cnt, cnt_big; std::vector<uint32_t> max_nh; std::vector<uint64_t> qid_lo; int err = MM_OK; };

template <uint32_t NC>
int partition_impl(const mm_aln *aln, size_t n, const mm_nh_big *big, size_t n_big, uint32_t N, mm_aln *const *out, size_t *out_n,
                   mm_nh_big *const *out_big, size_t big_cap, size_t *out_n_big, uint32_t *out_max_nh, uint64_t *out_qid_lo) {
  const Deal<NC> D(N);
  const int T = partition_threads(n);
  std::vector<SliceStat> st((size_t) T);
  auto lo_of = [&](int t) { return n * (size_t) t / (size_t) T; };
  // pass 1: every thread counts its slice's tuples (and side-list entries) per part
  std::function<void(int)> count = [&](int t) {
    SliceStat &S = st[(size_t) t];
    S.cnt.assign(N, 0); S.cnt_big.assign(N, 0); S.max_nh.assign(N, 0); S.qid_lo.assign(N, ~0ull);
    const size_t lo = lo_of(t), hi = lo_of(t + 1);
    size_t bi = (size_t) (std::lower_bound(big, big + n_big, lo, [](const mm_nh_big &e, size_t v) { return e.index < v; }) - big);
    for (size_t i = lo; i < hi; ++i) {
      const uint32_t q = aln[i].qid, p = D.part(q);
      ++S.cnt[p];
      uint32_t nh = MM_ALN_NH_FIELD(aln[i].tsn);
      if (nh == MM_ALN_NH_ESCAPE) {
        while (bi < n_big && big[bi].index < i) ++bi;
        if (bi >= n_big || big[bi].index != i) { S.err = MM_ERR_ARG; return; }     // escaped tuple without an entry
        nh = big[bi].nh;
        ++S.cnt_big[p];
      }
      if (nh > S.max_nh[p]) S.max_nh[p] = nh;
      if (nh > 1) { const uint64_t qq = D.name(q); if (qq < S.qid_lo[p]) S.qid_lo[p] = qq; }
    }
  };
  std::unique_lock<std::mutex> pool_lock(part_pool().call_mu, std::try_to_lock);
  auto run = [&](const std::function<void(int)> &f) {
    if (pool_lock.owns_lock()) part_pool().run(T, f);
    else for (int t = 0; t < T; ++t) f(t);                 // (another thread is dealing a batch out through the pool right now)
  };
  run(count);
  for (int t = 0; t < T; ++t) if (st[(size_t) t].err) return st[(size_t) t].err;
  // offsets: part p's tuples of slice t start behind those of the slices before it - file order is kept
  std::vector<size_t> off((size_t) T * N), off_big((size_t) T * N);
  for (uint32_t p = 0; p < N; ++p) {
    size_t a = 0, b = 0;
    uint32_t mx = 0;
    uint64_t ql = ~0ull;
    for (int t = 0; t < T; ++t) {
      off[(size_t) t * N + p] = a; off_big[(size_t) t * N + p] = b;
      a += st[(size_t) t].cnt[p]; b += st[(size_t) t].cnt_big[p];
      mx = std::max(mx, st[(size_t) t].max_nh[p]); ql = std::min(ql, st[(size_t) t].qid_lo[p]);
    }
    out_n[p] = a;
    if (out_n_big) out_n_big[p] = b;
    if (b > big_cap) return MM_ERR_RANGE;                  // side list full
    if (out_max_nh) out_max_nh[p] = mx;
    if (out_qid_lo) out_qid_lo[p] = ql == ~0ull ? 0 : ql;
  }
  // pass 2: stable scatter.  Every thread collects the tuples of a part in a 64-byte line of its own and writes whole lines with
  // streaming stores (the pinned staging buffer they go to is read next by the copy engine, not by this core; single 16-byte streaming
  // stores to N interleaved destinations would thrash the core's few write-combining buffers).  The lines at the two ends of a thread's
  // range of a part are shared with its neighbours: those entries are written one by one.
  bool aligned = !getenv("MM_PARTITION_PLAIN");
  for (uint32_t p = 0; p < N; ++p) if (((uintptr_t) out[p]) & 63u) aligned = false;
  std::function<void(int)> scatter = [&](int t) {
    const size_t lo = lo_of(t), hi = lo_of(t + 1);
    std::vector<size_t> at(off.begin() + (long) ((size_t) t * N), off.begin() + (long) ((size_t) (t + 1) * N));
    std::vector<size_t> at_big(off_big.begin() + (long) ((size_t) t * N), off_big.begin() + (long) ((size_t) (t + 1) * N));
    struct alignas(64) Line { mm_aln v[4]; };
    std::vector<Line> stage(N);
    std::vector<uint8_t> from(N);                          // first entry of the current line that is this thread's
    for (uint32_t p = 0; p < N; ++p) from[p] = (uint8_t) (at[p] & 3u);
    auto flush = [&](uint32_t p, size_t line_first, uint32_t a0, uint32_t a1) {   // entries [a0, a1) of the line starting at line_first
      mm_aln *dst = out[p] + line_first;
      if (aligned && a0 == 0 && a1 == 4) { for (int k = 0; k < 4; ++k) stream_store(dst + k, stage[p].v[k]); }
      else for (uint32_t k = a0; k < a1; ++k) dst[k] = stage[p].v[k];
    };
    size_t bi = (size_t) (std::lower_bound(big, big + n_big, lo, [](const mm_nh_big &e, size_t v) { return e.index < v; }) - big);
    for (size_t i = lo; i < hi; ++i) {
      mm_aln a = aln[i];
      const uint32_t q = a.qid, p = D.part(q);
      a.qid = D.name(q);
      const size_t w = at[p]++;
      stage[p].v[w & 3u] = a;
      if ((w & 3u) == 3u) { flush(p, w - 3, from[p], 4); from[p] = 0; }
      if (MM_ALN_NH_FIELD(a.tsn) == MM_ALN_NH_ESCAPE) {
        while (bi < n_big && big[bi].index < i) ++bi;
        mm_nh_big &e = out_big[p][at_big[p]++];
        e.index = (uint32_t) w; e.nh = big[bi].nh;
      }
    }
    for (uint32_t p = 0; p < N; ++p) if ((at[p] & 3u) > from[p] || ((at[p] & 3u) && from[p] == 0)) flush(p, at[p] & ~(size_t) 3, from[p], (uint32_t) (at[p] & 3u));
    stream_fence();
  };
  run(scatter);
  return MM_OK;
}
}  // namespace

// number of host threads mm_partition deals a batch out with: n >= 1 sets it (used as given, also for small batches), 0 asks
extern "C" int mm_partition_threads(int n) {
  if (n >= 1) { g_part_threads = std::min(n, 64); g_part_threads_forced = true; }
  return partition_threads((size_t) 1 << 30);
}

// Round 3: dealt out by all the host cores the process may use (per-thread slices, counted, then scattered behind per-part prefix
// offsets - the same stable order as the sequential loop it replaces, which at ~0.25 G tuples/s could not feed one device).
extern "C" int mm_partition(const mm_aln *aln, size_t n, const mm_nh_big *big, size_t n_big, int n_parts, mm_aln *const *out, size_t *out_n,
                            mm_nh_big *const *out_big, size_t big_cap, size_t *out_n_big, uint32_t *out_max_nh, uint64_t *out_qid_lo) {
  if (n_parts < 1 || !out || !out_n || (n && !aln) || (n_big && (!big || !out_big || !out_n_big))) return MM_ERR_ARG;
  if (n == 0) {
    for (int p = 0; p < n_parts; ++p) { out_n[p] = 0; if (out_n_big) out_n_big[p] = 0; if (out_max_nh) out_max_nh[p] = 0; if (out_qid_lo) out_qid_lo[p] = 0; }
    return MM_OK;
  }
  const uint32_t N = (uint32_t) n_parts;
#define MM_DEAL(K) case K: return partition_impl<K>(aln, n, big, n_big, N, out, out_n, out_big, big_cap, out_n_big, out_max_nh, out_qid_lo);
  switch (N) {
    MM_DEAL(1) MM_DEAL(2) MM_DEAL(3) MM_DEAL(4) MM_DEAL(5) MM_DEAL(6) MM_DEAL(7) MM_DEAL(8) MM_DEAL(16)
    default: return partition_impl<0>(aln, n, big, n_big, N, out, out_n, out_big, big_cap, out_n_big, out_max_nh, out_qid_lo);
  }
#undef MM_DEAL
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
    // batch j of the input went to ctxs[j % n_ctx]: replay the parked accumulations in that order, each one starting from the sums its
    // predecessor ended with.  Round 3: the table goes from context to context on the devices (mm_ratio_handoff: a peer copy ordered by
    // events on both streams) and nothing waits for the host until the last batch is enqueued - one mm_get_rows / mm_set_row_sums round
    // trip per batch before.
    std::vector<size_t> cursor((size_t) n_ctx, 0);
    mm_ctx *prev = nullptr;
    for (size_t j = 0;; ++j) {
      const size_t k = j % (size_t) n_ctx;
      mm_ctx *c = ctxs[k];
      if (cursor[k] >= mm_ratio_pending(c)) break;
      if (prev) { const int rc = mm_ratio_handoff(prev, c, 0, 1); if (rc) return rc; }
      const int rc = mm_ratio_replay(c, cursor[k]++, 1, 0, 1);
      if (rc) return rc;
      prev = c;
    }
    for (int k = 0; k < n_ctx; ++k) if (cursor[(size_t) k] != mm_ratio_pending(ctxs[k])) return MM_ERR_STATE;   // batches were not dealt round-robin from ctxs[0]
    for (int k = 0; k < n_ctx; ++k) { const int rc = mm_ratio_release(ctxs[k]); if (rc) return rc; }
    if (!prev) { const int rc = rows_of(ctxs[0]); if (rc) return rc; }   // nothing parked (any more): every context holds the final rows
    else {
      // leave the input's final sums in every context (a second call - sizes first, rows then - finds them)
      for (int k = 0; k < n_ctx; ++k) if (ctxs[k] != prev) { const int rc = mm_ratio_handoff(prev, ctxs[k], 0, 1); if (rc) return rc; }
      for (int k = 0; k < n_ctx; ++k) { const int rc = mm_sync(ctxs[k]); if (rc) return rc; }
      const int rc = rows_of(prev);
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

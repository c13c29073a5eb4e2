// mm_device.hip — gfx950 (MI355X) device path of the overlap-and-count engine + its C ABI.
//
// Reference behaviour (mzytnicki/mmannot v1.1, mmannot.cpp):
//   IntervalList::scan            :1291-1332   -> k_probe (candidate enumeration + strand + -l filter)
//   EvaluationStructure::getFirst :1029-1076   -> priority selection inside k_probe
//   Counter::addCount             :1665-1739   -> k_probe (per-hit branch), k_leaf / k_chunk_default (multi-hit
//                                                 countdown / union), k_ratio_tiles + k_ratio_rows (-y ratio)
//   Counter::read flush           :1783-1792   -> k_flush_pending
//   Counter::getCounts / dump     :1803-1818   -> mm_get_rows / mm_get_stats
//
// This is not a translation of the reference's walk.  The reference sorts intervals by start, jumps through a 16 kb
// bin table keyed on interval *ends* and walks forward; long intervals make that walk long and lane-divergent.  Here the
// index is a "checkpointed sweep": the candidates of a read starting in bin b (width W) are
//   act(b)  = the intervals open at the bin's left edge (start < b*W <= end+1), in index order, followed by
//   iv(b..) = the intervals starting at or after b*W, in index order, while start <= read.end,
// visited in the reference's index order (which matters: EvaluationStructure::set is last-writer-wins).  Two encodings:
//   * compact slots (probe_slots): one 64-B sector per bin with up to 7 packed 8-B records - every read with end >= start;
//   * full 16-B records + bins + active lists (probe_one): reads without CIGAR operations (they need the "leading
//     isBefore run" rule, which uses each record's prefix-max of ends) and the -M interval listing.
// Downstream of the probe: multi-hit records are brought together by bucket of 4096 read names with stable partition passes
// (k_rs_*; one pass for a key range of 21 bits) and each bucket is reduced per name in LDS (k_leaf) - no full sort, no
// per-name gathers; name-grouped input with a wide key range is only compacted and reduced by k_chunk_default.  -y ratio sorts
// by element set and accumulates each row's doubles in file order, exactly but in parallel (k_ratio_tiles, k_ratio_rows).
// See DESIGN.md.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mmannot_amd.h"

typedef unsigned long long u64;
typedef unsigned int u32;

// The input tuples are read exactly once: loading them non-temporally keeps 0.3 GB per 2^24-tuple batch from evicting the
// interval index out of the caches (measured: probe -7 %).  The sort's buffers are NOT streamed this way: consecutive
// passes re-read them out of L2 / Infinity Cache (non-temporal there: sort +30 %).
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 ld_stream(const int4 *p) {
  const v4i_t v = __builtin_nontemporal_load((const v4i_t *) p);
  return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ u32 ld_stream(const u32 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(u32 *p, u32 v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_stream(u64 *p, u64 v) { __builtin_nontemporal_store(v, p); }

// ----------------------------------------------------------------------------- errors
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail(MM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
  } while (0)

extern "C" const char *mm_last_error(void) { return g_err.c_str(); }
extern "C" const char *mm_version(void) { return MM_VERSION; }
extern "C" void *mm_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable) != hipSuccess) { g_err = "mm_host_alloc: hipHostMalloc failed"; return nullptr; }
  return p;
}
extern "C" void mm_host_free(void *p) { if (p) (void) hipHostFree(p); }
extern "C" int mm_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// ----------------------------------------------------------------------------- device structures
#define GT_CAP (1u << 16)          // global element-set table (distinct rows of the count table)
#define GT_NEAR_END 512u           // = RS_BINS: slots below it are handed out first (see gt_find_or_insert)
#define GT_NEAR_PROBES 16u
#define LT_CAP 512u                // per-workgroup LDS table (with 1024 entries the merge kernel could not hold 8 waves per SIMD)
#define LT_MAXPROBE 8u             // a set that does not find a slot this fast goes to the global table instead
#define NO_KEY 0xFFFFFFFFu
// A multi-hit record's element set travels to the merge kernels in 16 bits of its payload: e < 64 = the single element e (the
// bulk: no table access on either side), GT_NO_SLOT = no element, anything else = the set's slot in the global table.
// gt_find_or_insert therefore never hands out the slots below 64 or the last one.
#define GT_NO_SLOT (GT_CAP - 1u)
#define GT_FIRST_SLOT 64u
#define MM_LDS_CHR 256               // chromosome tables up to this size are staged in LDS by k_probe
// k_probe is a chain of dependent gathers (tuple -> slot line -> rest of the line), i.e. latency bound: it runs at the
// hardware's maximum of 8 waves per SIMD (<= 64 VGPRs, < 20 KB LDS per 256 threads).  Measured on MI355X against the
// compiler's own choice of 73 VGPRs / 6 waves: -9 % (c3), -20 % (c2), -18 % (c4, where the -l arithmetic then spills
// 88 B per lane to scratch and is still faster).  Prefetching the next tuple instead did not help.
#define MM_PROBE_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))

// per-batch scalars kept on the device (no host round trip between the kernels of one batch)
enum { INFO_M = 0,      // number of records in the grouping pipeline
       INFO_QINV = 1,   // max over ~qid of the multi-hit records = ~(smallest qid)
       INFO_QMAX = 2,   // largest qid of the multi-hit records
       INFO_BIGNH = 3,  // some multi-hit record has NH >= 65535: payloads carry the record index instead of (slot, NH)
       INFO_UNSORTED = 4,  // the multi-hit records' qids are not already non-decreasing in file order (name-grouped input is)
       INFO_NLIST = 5,  // number of non-empty buckets listed by k_bucket_bounds (leaf layout with more than one partition pass)
       INFO_WIDEKEY = 6,  // -y ratio: some sort key (table slot) of the batch is not below RS_BINS: two sort passes instead of one
       INFO_WORDS = 7 };

__host__ __device__ inline int bits_for_range(unsigned int span) {  // bits needed for keys in [0, span]
  int b = 1;
  while (b < 32 && (1ull << b) <= (unsigned long long) span) ++b;
  return b;
}
// Grouping layout of this batch.
//   * -y default ("leaf" layout): the multi-hit records, keyed by qid - qmin, are brought together by *bucket* = key >> LEAF_BITS
//     (4096 consecutive read names) with stable partition passes over the bucket id, 9 bits or fewer at a time, lowest digit first:
//     ONE pass when the key range fits LEAF_BITS + 9 = 21 bits (a 2^24-tuple batch of reads with ~10 hits each spans 21), two up
//     to 30 bits, three beyond.  k_leaf then reduces each bucket's names in LDS.  The low LEAF_BITS of the key are never sorted
//     on, and no per-name state is gathered from HBM.
//   * -y ratio sorts its (table row, NH) pairs on a fixed 16 bits: two 8-bit LSD passes.
#define LEAF_BITS 12
#define LEAF_NAMES (1u << LEAF_BITS)
#define LEAF_DIGIT_BITS 9
#define RS_BINS (1u << LEAF_DIGIT_BITS)   // digit rows of the per-tile histograms (a pass uses the first 2^width of them)
struct SortLayout { int passes; u32 width; u32 shift0; bool leaf; };
__host__ __device__ inline SortLayout leaf_layout(int key_bits) {
  const int bb = key_bits > LEAF_BITS ? key_bits - LEAF_BITS : 1;    // bits of the bucket id
  SortLayout L;
  L.passes = (bb + LEAF_DIGIT_BITS - 1) / LEAF_DIGIT_BITS;
  L.width = (u32) ((bb + L.passes - 1) / L.passes);
  L.shift0 = LEAF_BITS; L.leaf = true;
  return L;
}
__device__ __forceinline__ SortLayout batch_layout(const unsigned int *info, int fixed_bits) {
  if (!fixed_bits) {
    const unsigned int qmin = ~info[INFO_QINV], qmax = info[INFO_QMAX];
    return leaf_layout(qmax >= qmin ? bits_for_range(qmax - qmin) : 1);
  }
  SortLayout L;
  L.shift0 = 0; L.leaf = false;
  // -y ratio sorts by table slot.  New element sets are placed in the table's first RS_BINS slots while there is room
  // (gt_find_or_insert), and few runs ever see more sets than that: one pass of one RS_BINS-wide digit then sorts the batch
  if (fixed_bits == 16 && !info[INFO_WIDEKEY]) { L.passes = 1; L.width = LEAF_DIGIT_BITS; return L; }
  L.passes = (fixed_bits + 7) / 8;
  L.width = (u32) ((fixed_bits + L.passes - 1) / L.passes);
  return L;
}
// passes this batch really runs: one (the compacting first pass) when the records are already grouped by name
__device__ __forceinline__ int batch_passes(const unsigned int *info, int fixed_bits) {
  if (!fixed_bits && !info[INFO_UNSORTED]) return 1;
  return batch_layout(info, fixed_bits).passes;
}
// is this batch reduced by k_leaf (else - name-grouped input - by k_chunk_default behind the plain compaction)
__device__ __forceinline__ bool batch_is_leaf(const unsigned int *info) { return info[INFO_UNSORTED] != 0; }
// digit of key k in the pass the host launched with shift = 8 * pass
__device__ __forceinline__ u32 batch_digit(u32 k, const SortLayout L, int shift) { return (k >> (L.shift0 + (u32) (shift >> 3) * L.width)) & ((1u << L.width) - 1u); }

struct GlobalTable {
  u64 *key;      // 0 = empty (an element set is never empty)
  u64 *cnt;      // integer count (default / unique)
  double *sum;   // -y ratio accumulator (file-order double adds, mmannot.cpp:1730)
  u32 *err;      // ERR_* of the first problem a kernel met (0 = none); read by mm_sync
  u32 nearProbes;  // -y ratio: GT_NEAR_PROBES slots of the table's first RS_BINS are tried first (gt_find_or_insert); else 0 - counts
                   // that are added to with atomics are better off spread over the whole table (hot rows side by side in one cache
                   // line cost k_leaf a quarter of its time)
};
enum { ERR_TABLE_FULL = 1, ERR_QID_BOUND = 2, ERR_SINGLE_HIT = 3, ERR_NH_LIST = 4, ERR_QID_LO = 5 };

// mm_aln.tsn = strand:1 | NH:11 | (chromosome + 1):20 (include/mmannot_amd.h)
#define TSN_TID1(z) ((u32) (z) >> (MM_ALN_NH_BITS + 1))
// record.nHits of tuple i: the tuple's NH field, or - for the rare value that does not fit - the batch's side list
// (ascending by tuple index).  A tuple that escapes without an entry is the caller's error.
__device__ __forceinline__ u32 tuple_nh(u32 tsn, size_t i, const uint2 *__restrict__ big, u32 nbig, u32 *err) {
  const u32 f = (tsn >> 1) & MM_ALN_NH_ESCAPE;
  if (f != MM_ALN_NH_ESCAPE) return f;
  u32 lo = 0, hi = nbig;
  while (lo < hi) {
    const u32 mid = (lo + hi) >> 1;
    if (big[mid].x < (u32) i) lo = mid + 1; else hi = mid;
  }
  if (lo < nbig && big[lo].x == (u32) i) return big[lo].y;
  atomicCAS(err, 0u, (u32) ERR_NH_LIST);
  return f;
}

struct ProbeParams {
  const int4 *iv;          // {start, end, type | F<<8, exclusive prefix max of end within the chromosome}
  const int4 *act;         // same record format
  const uint2 *bins;       // per chromosome nbins+1 entries {first, actOff}
  const uint4 *chrInfo;     // per chromosome {first bin record, number of bins, end of its interval range, 0}
  // compact "slot" index used for every read with end >= start (see slot_walk)
  const uint4 *slots;       // 64 B per bin: {nAct | nIv<<16, offset into flat, rec0} {rec1, rec2} ... 7 records
  const uint2 *flat;        // every bin's full record list (only read when a bin holds more than 7 records)
  const uint2 *chrSlot;     // per chromosome {first slot, number of slots}
  const u32 *slotBits;      // the slots' map, one word per 8 bins: bits 0-15 = 2 bits per bin (FS_EMPTY / FS_GENERAL / FS_COVERED), bits 16-23 =
                            // element:6 | ok('+'):1 | ok('-'):1 of the interval covering the group's FS_COVERED bins (code and class in ONE
                            // L2 access); NULL when the map would save too few fetches
  const int *longEnd;       // true end of the few intervals too long for the 22-bit length field
  const uint4 *fslots;      // fast slots: 32 B per bin (see mm_index_create); NULL = none
  const u32 *fbits;         // the map, 2 bits per fast slot: FS_EMPTY, FS_GENERAL (fetch the slot), FS_COVERED (the answer is in fcls)
  const u32 *fcls;          // one byte per 8 fast slots (four to a word): element:6 | ok('+'):1 | ok('-'):1 of the covering interval.
                            // (Separate from the map here, unlike slotBits: k_probe_fast WAITS for the map word and only requests the class
                            // word, so the denser map is what counts - merged, HS38-shaped batches took 335 instead of 301 us;
                            // profiles/r03g_ab_merged_map.txt)
  u32 fextClass[3];         // extension blocks numbered below these hold at most 11 / 15 / 23 records (3 / 4 / 6 quads to request)
  const uint4 *fext;        // extension blocks: 128 B = count + up to 31 records, for the bins whose slot says FS_OVERFLOW | block
  const uint2 *fchr;        // per chromosome {first fast slot, number of fast slots}
  u32 fLmax;                // longest read the fast slots answer exactly
  u32 fshift, freach;       // the fast slots' own bin width and reach
  u32 sshift;
  u32 reach;               // a slot also lists the intervals starting in the first `reach` bases of the next bin
  int useSlots;
  u32 nchr;
  u32 shift;
  u64 elemF, elemR, elemUp, elemDown;  // bit e set: element e is declared '+', '-', is "upstream", is "downstream"
  int overlapMode;
  float overlapValue;
  int strandedness;
  int strategy;
  int keep;
  const u64 *lineMask;     // [64] lineMask[e] = elements on the same Order line as e (device memory: indexed per lane)
};

__device__ __forceinline__ u32 hash64(u64 k) {
  k *= 0x9E3779B97F4A7C15ull;
  return (u32) (k >> 40);
}

// a word of the workgroup's LDS tables that other lanes may have changed: a relaxed atomic load (a `volatile` access is fenced with
// s_waitcnt vmcnt(0) lgkmcnt(0), which drains every global load the probe has in flight)
template <typename T>
__device__ __forceinline__ T lds_peek(T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <u32 CAP>
struct LdsTableT {
  u64 key[CAP];
  u32 cnt[CAP];
  u32 slot[CAP];   // cached slot of the set in the global table (NO_KEY = not looked up yet)
  u32 single[64];     // counts of the single-element sets {e} (by far the most frequent): no hashing
  u32 singleSlot[64];
};
typedef LdsTableT<LT_CAP> LdsTable;

template <u32 CAP>
__device__ __forceinline__ void lds_table_init(LdsTableT<CAP> &t) {
  for (u32 i = threadIdx.x; i < CAP; i += blockDim.x) { t.key[i] = 0; t.cnt[i] = 0; t.slot[i] = NO_KEY; }
  if (threadIdx.x < 64) { t.single[threadIdx.x] = 0; t.singleSlot[threadIdx.x] = NO_KEY; }
}

template <u32 CAP>
__device__ __forceinline__ bool lds_table_add(LdsTableT<CAP> &t, u64 m, u32 inc) {
  u32 h = hash64(m) & (CAP - 1);
  for (u32 p = 0; p < LT_MAXPROBE; ++p) {
    u64 k = lds_peek(&t.key[h]);
    if (k == 0) {
      k = atomicCAS(&t.key[h], 0ull, m);
      if (k == 0) k = m;
    }
    if (k == m) { atomicAdd(&t.cnt[h], inc); return true; }
    h = (h + 1) & (CAP - 1);
  }
  return false;
}

__device__ __forceinline__ u32 gt_find_or_insert(const struct GlobalTable &g, u64 m);

// slot of set m in the global table, through the workgroup's LDS table (one global lookup per set and workgroup)
template <u32 CAP>
__device__ __forceinline__ u32 lds_cached_slot(LdsTableT<CAP> &t, const struct GlobalTable &g, u64 m) {
  u32 h = hash64(m) & (CAP - 1);
  for (u32 p = 0; p < LT_MAXPROBE; ++p) {
    u64 k = lds_peek(&t.key[h]);
    if (k == 0) {
      k = atomicCAS(&t.key[h], 0ull, m);
      if (k == 0) k = m;
    }
    if (k == m) {
      u32 sl = lds_peek(&t.slot[h]);
      if (sl == NO_KEY) { sl = gt_find_or_insert(g, m); t.slot[h] = sl; }
      return sl;
    }
    h = (h + 1) & (CAP - 1);
  }
  return gt_find_or_insert(g, m);
}

// per-lane: count one occurrence of set m / get the global slot of set m (no wave-level serialisation)
__device__ __forceinline__ void gt_add(const struct GlobalTable &g, u64 m, u64 inc);
template <u32 CAP>
__device__ __forceinline__ void count_set(LdsTableT<CAP> &t, const struct GlobalTable &g, u64 m) {
  if ((m & (m - 1)) == 0) atomicAdd(&t.single[__ffsll((long long) m) - 1], 1u);
  else if (!lds_table_add(t, m, 1)) gt_add(g, m, 1);
}
template <u32 CAP>
__device__ __forceinline__ u32 slot_of_set(LdsTableT<CAP> &t, const struct GlobalTable &g, u64 m) {
  if ((m & (m - 1)) == 0) {
    const int e = __ffsll((long long) m) - 1;
    u32 sl = lds_peek(&t.singleSlot[e]);
    if (sl == NO_KEY) { sl = gt_find_or_insert(g, m); t.singleSlot[e] = sl; }
    return sl;
  }
  return lds_cached_slot(t, g, m);
}

// returns slot or NO_KEY when full (error flag raised); slots [GT_FIRST_SLOT, GT_NO_SLOT) only (see the payload encoding)
__device__ __forceinline__ u32 gt_find_or_insert(const GlobalTable &g, u64 m) {
  // near region first (-y ratio): a set's probe sequence starts with GT_NEAR_PROBES slots of [GT_FIRST_SLOT, GT_NEAR_END) and goes on in the rest
  // of the table.  Slots are never emptied, so a lookup walks the same sequence as the insertion did.  While a run knows fewer than a
  // few hundred sets - nearly every run - all its slots are below RS_BINS, which is what lets -y ratio sort its batches in one pass.
  const u32 hh = hash64(m);
  u32 h = GT_FIRST_SLOT + hh % (GT_NEAR_END - GT_FIRST_SLOT);
#pragma unroll 1
  for (u32 p = 0; p < g.nearProbes; ++p) {
    u64 k = *(volatile u64 *) &g.key[h];
    if (k == 0) {
      k = atomicCAS(&g.key[h], 0ull, m);
      if (k == 0) k = m;
    }
    if (k == m) return h;
    if (++h == GT_NEAR_END) h = GT_FIRST_SLOT;
  }
  h = hh & (GT_CAP - 1);
#pragma unroll 1
  for (u32 p = 0; p < GT_CAP; ++p) {
    if (h < (g.nearProbes ? GT_NEAR_END : GT_FIRST_SLOT) || h == GT_NO_SLOT) { h = g.nearProbes ? GT_NEAR_END : GT_FIRST_SLOT; continue; }
    u64 k = *(volatile u64 *) &g.key[h];
    if (k == 0) {
      k = atomicCAS(&g.key[h], 0ull, m);
      if (k == 0) k = m;
    }
    if (k == m) return h;
    h = (h + 1) & (GT_CAP - 1);
  }
  atomicCAS(g.err, 0u, (u32) ERR_TABLE_FULL);
  return NO_KEY;
}

// element set behind the 16-bit payload field of a multi-hit record
__device__ __forceinline__ u64 payload_set(const GlobalTable &g, u32 slot) {
  if (slot < GT_FIRST_SLOT) return 1ull << slot;
  return slot == GT_NO_SLOT ? 0ull : g.key[slot];
}

__device__ __forceinline__ void gt_add(const GlobalTable &g, u64 m, u64 inc) {
  u32 s = gt_find_or_insert(g, m);
  if (s != NO_KEY) atomicAdd(&g.cnt[s], inc);
}

template <u32 CAP>
__device__ __forceinline__ void lds_table_flush(LdsTableT<CAP> &t, const GlobalTable &g) {
  if (threadIdx.x < 64 && t.single[threadIdx.x]) {
    const u32 sl = t.singleSlot[threadIdx.x];
    if (sl != NO_KEY) atomicAdd(&g.cnt[sl], (u64) t.single[threadIdx.x]);
    else gt_add(g, 1ull << threadIdx.x, (u64) t.single[threadIdx.x]);
  }
  for (u32 i = threadIdx.x; i < CAP; i += blockDim.x) {
    const u64 k = t.key[i];
    const u32 c = t.cnt[i];
    if (k && c) {
      const u32 sl = t.slot[i];
      if (sl != NO_KEY) atomicAdd(&g.cnt[sl], (u64) c);
      else gt_add(g, k, (u64) c);
    }
  }
}

// ----------------------------------------------------------------------------- candidate test
// checkStrand (mmannot.cpp:438-443) + the selected intervalOverlapFunction (mmannot.cpp:992-1002).
// r = interval record, [s,e] = read.  o receives the overlap value handed to EvaluationStructure::set.
template <int OVL>
__device__ __forceinline__ bool iv_pass(const ProbeParams &P, const int4 r, const int s, const int e, const int rstrand, u32 &o) {
  const u32 type = (u32) r.z & 0xffu;
  const int ivF = (r.z >> 8) & 1;
  const u32 eF = (u32) (P.elemF >> type) & 1u, eR = (u32) (P.elemR >> type) & 1u;
  bool ok = true;
  if (eF | eR) {
    const bool same = (ivF == rstrand);
    ok = eF ? same : !same;
  }
  // An interval that lies before the read (end < read.start) is only ever tested by the reference when
  // an earlier interval of the chromosome reaches the read: the first loop of scan() swallows the whole
  // leading isBefore run (mmannot.cpp:1306-1308).  r.w = max end over all earlier intervals of the chromosome.
  if (r.y < s && r.w < s) ok = false;
  if (OVL == MM_OVL_INCLUSION) {  // Interval::includes mmannot.cpp:637-639
    o = 1;
    return ok && r.x <= s && e <= r.y;
  }
  const int lo = max(s, r.x), hi = min(e, r.y);  // Interval::overlaps mmannot.cpp:632-636 (no +1)
  if (lo >= hi) return false;
  const u32 ov = (u32) (hi - lo);
  o = ov;
  if (OVL == MM_OVL_PC) {  // intervalOverlapPc mmannot.cpp:995-998: float product, float compare
    const u32 size = (u32) (e - s + 1);
    const float lhs = __fmul_rn((float) size, P.overlapValue);
    return ok && (lhs <= (float) ov);
  }
  return ok && ((float) ov >= P.overlapValue);  // intervalOverlap mmannot.cpp:999-1002
}

// Read::getDistance(p), mmannot.cpp:661-665
__device__ __forceinline__ u32 rd_distance(int s, int e, int p) {
  if (p < s) return (u32) (s - p);
  if (p > e) return (u32) (p - e);
  return 0;
}

struct Cand {
  u32 a0, a1;   // active-list range
  u32 i0, i1;   // sorted-interval range start / chromosome end
};

__device__ __forceinline__ bool locate(const ProbeParams &P, u32 tid1, int s, Cand &c) {
  if (tid1 == 0 || tid1 > P.nchr) return false;    // chromosome unknown to the annotation (mmannot.cpp:1293-1302)
  const uint4 ci = P.chrInfo[tid1 - 1];
  const u32 nb = ci.y;
  if (nb == 0) return false;                       // chromosome without any interval: not in `bins` either
  u32 b = (u32) s >> P.shift;
  b = min(b, nb - 1);
  const uint2 *B = P.bins + ci.x + b;
  const uint2 b0 = B[0];
  c.a0 = b0.y; c.a1 = B[1].y; c.i0 = b0.x; c.i1 = ci.z;
  return true;
}

// last passing interval of element `t` in index order -> its (overlap, distance): what
// EvaluationStructure::set leaves in evaluation[t] (mmannot.cpp:1023-1028, 1316-1323)
template <int OVL>
__device__ __forceinline__ void last_of_type(const ProbeParams &P, const Cand &c, int s, int e, int rstrand, u32 t, u32 &o_out, u32 &d_out) {
  const bool up = (P.elemUp >> t) & 1, down = (P.elemDown >> t) & 1;
  for (u32 k = c.a0; k < c.a1; ++k) {
    const int4 r = P.act[k];
    u32 o;
    if (((u32) r.z & 0xffu) == t && iv_pass<OVL>(P, r, s, e, rstrand, o)) {
      o_out = o; d_out = up ? rd_distance(s, e, r.y) : (down ? rd_distance(s, e, r.x) : 0u);
    }
  }
  for (u32 i = c.i0; i < c.i1; ++i) {
    const int4 r = P.iv[i];
    if (r.x > e) break;
    u32 o;
    if (((u32) r.z & 0xffu) == t && iv_pass<OVL>(P, r, s, e, rstrand, o)) {
      o_out = o; d_out = up ? rd_distance(s, e, r.y) : (down ? rd_distance(s, e, r.x) : 0u);
    }
  }
}

// ---- compact index --------------------------------------------------------------------------------------
// 8-byte candidate record: x = start; y = len:22 | type:6 | ok('-' read):1 | ok('+' read):1 | long:1 | 0.
// The two "ok" bits are checkStrand (mmannot.cpp:438-443) evaluated at build time for both read strands.
// A bin's slot lists, in index order, the intervals open at the bin's left edge followed by the intervals
// starting inside the bin: exactly act(b) ++ iv[first(b)..] of the 16-byte structure, one 64-B sector.
#define REC_LEN_BITS 22
#define REC_LEN_MASK ((1u << REC_LEN_BITS) - 1u)
#define REC_TYPE(y) (((y) >> REC_LEN_BITS) & 63u)
#define REC_OKBIT 28
#define REC_LONG (1u << 30)
// fast slots (4-byte bin-relative records, see mm_index_create): usable up to this bin shift - beyond it the saturating length field
// (24 - shift - 1 bits) is shorter than a bin
#define FS_MAX_SHIFT 10
#define FS_MIN_SHIFT 7
#define FS_OVERFLOW_SHARE 0.01      // bins holding more than 7 records: above this share the fast table takes half the bin width
#define FS_OVERFLOW 0x80000000u     // header: more than 7 records - the tuple takes the 8-byte slots
#define FS_EXT_NONE 0x7FFFFu         // ... | extension block (19 bits: it travels beside a 13-bit tile-local index), this value = none
#define FS_EXT_RECORDS 31u
enum { FS_EMPTY = 0, FS_GENERAL = 1, FS_COVERED = 2 };   // codes of the fast slots' map

template <int OVL>
__device__ __forceinline__ bool rec_pass(const ProbeParams &P, u32 rx, u32 ry, int s, int e, int rstrand, u32 &o, int &ie) {
  const int is = (int) rx;
  const u32 len = ry & REC_LEN_MASK;
  ie = (ry & REC_LONG) ? P.longEnd[len] : is + (int) len;
  const bool ok = (ry >> (REC_OKBIT + rstrand)) & 1u;
  if (OVL == MM_OVL_INCLUSION) { o = 1; return ok && is <= s && e <= ie; }
  const int lo = max(s, is), hi = min(e, ie);
  if (lo >= hi) return false;
  const u32 ov = (u32) (hi - lo);
  o = ov;
  if (OVL == MM_OVL_PC) {
    const u32 size = (u32) (e - s + 1);
    const float lhs = __fmul_rn((float) size, P.overlapValue);
    return ok && (lhs <= (float) ov);
  }
  return ok && ((float) ov >= P.overlapValue);
}

// rec_pass for a record whose end is already known, as a 0/1 value (no branch)
template <int OVL>
__device__ __forceinline__ u32 span_pass(const ProbeParams &P, int is, int ie, u32 ok, int s, int e) {
  if (OVL == MM_OVL_INCLUSION) return ok & (u32) (is <= s) & (u32) (e <= ie);
  const int lo = max(s, is), hi = min(e, ie);
  const u32 ov = (u32) (hi - lo);
  if (OVL == MM_OVL_PC) {
    const u32 size = (u32) (e - s + 1);
    const float lhs = __fmul_rn((float) size, P.overlapValue);
    return ok & (u32) (lo < hi) & (u32) (lhs <= (float) ov);
  }
  return ok & (u32) (lo < hi) & (u32) ((float) ov >= P.overlapValue);
}

// number of reach records behind a slot's own (second header word: the count itself while the whole list fits the 7 inline records,
// else the list's offset in flat, >= 8, with the count in the record before it)
__device__ __forceinline__ u32 slot_reach_count(const ProbeParams &P, u32 w1) {
  return w1 < 8u ? w1 : P.flat[w1 - 1].x;
}

// The slot walk, written once as a statement macro and expanded with two visitors (plain straight-line code: callbacks made
// the compiler keep closures and a copy of the parameter block in scratch).  VISIT(rx, ry) sees, in index order, every
// record that can concern the read [s,e] (e >= s): the whole slot of the bin of s, then only the "starts here" part of
// the following bins up to the bin of e - or, for a read that ends within `reach` bases of the next bin, the reach records
// at the end of the first slot instead (the same records in the same order, without a second line).  The slot's first 32 bytes (header + 3 records) are requested first, its
// second 32 bytes only when the header says they hold records - and nothing at all when the
// slot is empty, which a bitmap small enough to stay in L2 (1 bit per bin) tells: the kernel's limit is the number of
// random 64-B sector fetches it can keep in flight, and about half of the bins of a genome hold no record.  The test is
// a dependent L2 access of its own, so the index only carries the bitmap when at least a third of the slots are empty
// (measured, end of round 2: HS38-shaped 50 % empty: 0.281 instead of 0.405 ms per 2^24 tuples; TAIR10-shaped, where every slot line
// is a 128-B fetch out of the Infinity Cache and the kernel runs at that fabric's rate: 34.7 % empty -11 %, 32.3 % (256-base bins)
// -3 %, 30.1 % (512-base bins) +3 %.  Round 1's kernel, bound elsewhere, broke even at 42 %.)
#define MM_SLOT_WALK(VISIT)                                                                      \
  for (u32 b_ = b0; b_ <= bl; ++b_) {                                                            \
    if (P.slotBits && !((P.slotBits[(cs.x + b_) >> 3] >> (((cs.x + b_) & 7u) << 1)) & 3u)) continue; /* empty bin: no line to fetch */ \
    const uint4 *S_ = P.slots + (size_t) (cs.x + b_) * 4;                                        \
    const uint4 q0 = S_[0], q1 = S_[1];                                                          \
    const u32 nAct_ = q0.x & 0xffffu;                                                            \
    u32 n_ = nAct_ + (q0.x >> 16);                                                               \
    if (reach_) n_ += slot_reach_count(P, q0.y);                                                 \
    const u32 first_ = (b_ == b0) ? 0u : nAct_;                                                  \
    walked_ += n_ - min(first_, n_);               /* records this bin shows the read (profiling aid) */ \
    if (first_ == 0 && n_ > 0) VISIT(q0.z, q0.w)                                                 \
    if (first_ <= 1 && n_ > 1) VISIT(q1.x, q1.y)                                                 \
    if (first_ <= 2 && n_ > 2) VISIT(q1.z, q1.w)                                                 \
    if (n_ > 3) {                                                                                \
      const uint4 q2 = S_[2], q3 = S_[3];                                                        \
      if (first_ <= 3) VISIT(q2.x, q2.y)                                                         \
      if (first_ <= 4 && n_ > 4) VISIT(q2.z, q2.w)                                               \
      if (first_ <= 5 && n_ > 5) VISIT(q3.x, q3.y)                                               \
      if (first_ <= 6 && n_ > 6) VISIT(q3.z, q3.w)                                               \
      if (n_ > 7) {                                                                              \
        for (u32 k_ = max(7u, first_); k_ < n_; k_ += 4) { /* rare long lists: four loads in flight */ \
          const uint2 *F_ = P.flat + q0.y;                                                       \
          const uint2 r0 = F_[k_], r1 = F_[min(k_ + 1, n_ - 1)], r2 = F_[min(k_ + 2, n_ - 1)], r3 = F_[min(k_ + 3, n_ - 1)]; \
          VISIT(r0.x, r0.y)                                                                      \
          if (k_ + 1 < n_) VISIT(r1.x, r1.y)                                                     \
          if (k_ + 2 < n_) VISIT(r2.x, r2.y)                                                     \
          if (k_ + 3 < n_) VISIT(r3.x, r3.y)                                                     \
        }                                                                                        \
      }                                                                                          \
    }                                                                                            \
  }

// scan() + getFirst() through the compact index, for a read with e >= s.
// EAGER (k_probe_list: every tuple is a heavy one and few are in flight - what counts is the length of the chain of dependent loads):
// no look at the bitmap of non-empty slots, the whole 64-byte slot requested at once, long lists eight records at a time
template <int OVL, bool EAGER = false>
__device__ __forceinline__ u64 probe_slots(const ProbeParams &P, const uint2 *chrSlot, const u64 *lineMask, int s, int e, u32 tid1, int rstrand,
                                            u32 &walked) {
  if (tid1 == 0 || tid1 > P.nchr) return 0;
  const uint2 cs = chrSlot[tid1 - 1];
  const u32 nb = cs.y;
  if (nb == 0) return 0;
  const u32 eb = (u32) e >> P.sshift;
  const u32 b0 = min((u32) s >> P.sshift, nb - 1), b1 = min(eb, nb - 1);
  // a read that ends less than `reach` bases into the next bin finds that bin's candidates at the end of its own slot
  const bool reach_ = b1 == b0 + 1 && eb == b1 && (u32) e - (b1 << P.sshift) < P.reach;
  const u32 bl = reach_ ? b0 : b1;
  // pass 1: which elements have a passing interval
  // The kernel is bound by instruction issue, not by the sector fetches (SQ counters, profiles/r02e_c3_sq_counters.txt),
  // so the seven records of a slot line are tested without a branch each: a record outside [first, n) or failing the
  // test ORs in a zero.  Ends of the rare intervals too long for the 22-bit length field are patched in beforehand.
  u64 hit = 0;
  u32 walked_ = 0;
#define MM_TEST(RX, RY, IE, VALID)                                                                                   \
  hit |= (u64) ((u32) (VALID) & span_pass<OVL>(P, (int) (RX), (IE), ((RY) >> (REC_OKBIT + rstrand)) & 1u, s, e)) << REC_TYPE(RY);
#define MM_VISIT1(RX, RY)                                                                  \
  {                                                                                        \
    u32 o_; int ie_;                                                                       \
    if (rec_pass<OVL>(P, (RX), (RY), s, e, rstrand, o_, ie_)) hit |= 1ull << REC_TYPE(RY); \
  }
  for (u32 b_ = b0; b_ <= bl; ++b_) {
    if (!EAGER && P.slotBits) {
      const u32 sl_ = cs.x + b_, mw_ = P.slotBits[sl_ >> 3], code_ = (mw_ >> ((sl_ & 7u) << 1)) & 3u;
      if (code_ == FS_EMPTY) continue;                      // empty bin: no line to fetch
      if (code_ == FS_COVERED && bl == b0) {                // one interval covers all this slot can be asked about: its element, if the strands allow
        const u32 cb_ = (mw_ >> 16) & 0xFFu;
        hit |= (u64) span_pass<OVL>(P, s, e, (cb_ >> rstrand) & 1u, s, e) << (cb_ >> 2);
        ++walked_;
        continue;
      }
    }
    const uint4 *S_ = P.slots + (size_t) (cs.x + b_) * 4;
    const uint4 q0 = S_[0], q1 = S_[1];
    uint4 q2 = make_uint4(0u, 0u, 0u, 0u), q3 = q2;
    if (EAGER) { q2 = S_[2]; q3 = S_[3]; }
    const u32 nAct_ = q0.x & 0xffffu;
    u32 n_ = nAct_ + (q0.x >> 16);
    if (reach_) n_ += slot_reach_count(P, q0.y);
    const u32 first_ = (b_ == b0) ? 0u : nAct_;
    walked_ += n_ - min(first_, n_);
    int e0 = (int) q0.z + (int) (q0.w & REC_LEN_MASK), e1 = (int) q1.x + (int) (q1.y & REC_LEN_MASK),
        e2 = (int) q1.z + (int) (q1.w & REC_LEN_MASK);
    if ((q0.w | q1.y | q1.w) & REC_LONG) {
      if (q0.w & REC_LONG) e0 = P.longEnd[q0.w & REC_LEN_MASK];
      if (q1.y & REC_LONG) e1 = P.longEnd[q1.y & REC_LEN_MASK];
      if (q1.w & REC_LONG) e2 = P.longEnd[q1.w & REC_LEN_MASK];
    }
    MM_TEST(q0.z, q0.w, e0, first_ == 0 && n_ > 0)
    MM_TEST(q1.x, q1.y, e1, first_ <= 1 && n_ > 1)
    MM_TEST(q1.z, q1.w, e2, first_ <= 2 && n_ > 2)
    if (n_ > 3) {
      if (!EAGER) { q2 = S_[2]; q3 = S_[3]; }
      int e3 = (int) q2.x + (int) (q2.y & REC_LEN_MASK), e4 = (int) q2.z + (int) (q2.w & REC_LEN_MASK),
          e5 = (int) q3.x + (int) (q3.y & REC_LEN_MASK), e6 = (int) q3.z + (int) (q3.w & REC_LEN_MASK);
      if ((q2.y | q2.w | q3.y | q3.w) & REC_LONG) {
        if (q2.y & REC_LONG) e3 = P.longEnd[q2.y & REC_LEN_MASK];
        if (q2.w & REC_LONG) e4 = P.longEnd[q2.w & REC_LEN_MASK];
        if (q3.y & REC_LONG) e5 = P.longEnd[q3.y & REC_LEN_MASK];
        if (q3.w & REC_LONG) e6 = P.longEnd[q3.w & REC_LEN_MASK];
      }
      MM_TEST(q2.x, q2.y, e3, first_ <= 3)
      MM_TEST(q2.z, q2.w, e4, first_ <= 4 && n_ > 4)
      MM_TEST(q3.x, q3.y, e5, first_ <= 5 && n_ > 5)
      MM_TEST(q3.z, q3.w, e6, first_ <= 6 && n_ > 6)
      if (n_ > 7) {
        const uint2 *F_ = P.flat + q0.y;
        u32 k_ = max(7u, first_);
        if (EAGER) for (; k_ + 4 < n_; k_ += 8) {            // eight loads in flight
          const uint2 r0 = F_[k_], r1 = F_[k_ + 1], r2 = F_[k_ + 2], r3 = F_[k_ + 3], r4 = F_[k_ + 4], r5 = F_[min(k_ + 5, n_ - 1)],
                      r6 = F_[min(k_ + 6, n_ - 1)], r7 = F_[min(k_ + 7, n_ - 1)];
          MM_VISIT1(r0.x, r0.y) MM_VISIT1(r1.x, r1.y) MM_VISIT1(r2.x, r2.y) MM_VISIT1(r3.x, r3.y) MM_VISIT1(r4.x, r4.y)
          if (k_ + 5 < n_) MM_VISIT1(r5.x, r5.y)
          if (k_ + 6 < n_) MM_VISIT1(r6.x, r6.y)
          if (k_ + 7 < n_) MM_VISIT1(r7.x, r7.y)
        }
        for (; k_ < n_; k_ += 4) {   // rare long lists: four loads in flight
          const uint2 r0 = F_[k_], r1 = F_[min(k_ + 1, n_ - 1)], r2 = F_[min(k_ + 2, n_ - 1)], r3 = F_[min(k_ + 3, n_ - 1)];
          MM_VISIT1(r0.x, r0.y)
          if (k_ + 1 < n_) MM_VISIT1(r1.x, r1.y)
          if (k_ + 2 < n_) MM_VISIT1(r2.x, r2.y)
          if (k_ + 3 < n_) MM_VISIT1(r3.x, r3.y)
        }
      }
    }
  }
#undef MM_VISIT1
#undef MM_TEST
  walked += walked_;
  if (!hit) return 0;
  // getFirst (mmannot.cpp:1029-1076): only the first Order line holding a hit element counts
  const int low = __ffsll((long long) hit) - 1;
  const u64 cand = hit & lineMask[low];
  if ((cand & (cand - 1)) == 0) return cand;
  // inclusion counts every passing interval as overlap 1, and only "upstream" / "downstream" elements have a distance: all tie
  if (OVL == MM_OVL_INCLUSION && !(cand & (P.elemUp | P.elemDown))) return cand;
  // pass 2 (several elements of the winning line were hit): each element's last writer (overlap, distance), then max
  // overlap, then min distance, ties kept (mmannot.cpp:1023-1028, 1049-1075); the slot lines are in L1 by now
  u64 sel = 0;
  u32 maxO = 0, minD = 0;
  for (u64 cm = cand; cm; cm &= cm - 1) {
    const u32 t = (u32) (__ffsll((long long) cm) - 1);
    const bool up = (P.elemUp >> t) & 1, down = (P.elemDown >> t) & 1;
    u32 o_last = 0, d_last = 0;
#define MM_VISIT2(RX, RY)                                                                                   \
  if (REC_TYPE(RY) == t) {                                                                                  \
    u32 o_; int ie_;                                                                                        \
    if (rec_pass<OVL>(P, (RX), (RY), s, e, rstrand, o_, ie_)) {                                             \
      o_last = o_; d_last = up ? rd_distance(s, e, ie_) : (down ? rd_distance(s, e, (int) (RX)) : 0u);      \
    }                                                                                                       \
  }
    MM_SLOT_WALK(MM_VISIT2)
#undef MM_VISIT2
    if (o_last > maxO) { maxO = o_last; minD = d_last; sel = 1ull << t; }
    else if (o_last == maxO) {
      if (d_last < minD) { minD = d_last; sel = 1ull << t; }
      else if (d_last == minD) sel |= 1ull << t;
    }
  }
  return sel;
}

// scan() + getFirst() for one read -> element set
template <int OVL>
__device__ __forceinline__ u64 probe_one(const ProbeParams &P, int s, int e, u32 tid1, int rstrand, u32 &walked) {
  Cand c;
  if (!locate(P, tid1, s, c)) return 0;
  u64 hit = 0;
  u32 o;
  // The loads are issued ahead of their use so that one memory round trip serves most reads: the first two
  // sorted-range records are requested together with the active list (a read rarely sees more than two).
  const int4 none = make_int4(0x7fffffff, 0, 0, 0);
  u32 i = c.i0;
  int4 f0 = (i < c.i1) ? P.iv[i] : none;
  int4 f1 = (i + 1 < c.i1) ? P.iv[i + 1] : none;
  u32 k = c.a0;
  for (; k + 1 < c.a1; k += 2) {
    const int4 r0 = P.act[k], r1 = P.act[k + 1];
    if (iv_pass<OVL>(P, r0, s, e, rstrand, o)) hit |= 1ull << ((u32) r0.z & 0xffu);
    if (iv_pass<OVL>(P, r1, s, e, rstrand, o)) hit |= 1ull << ((u32) r1.z & 0xffu);
    walked += 2;
  }
  if (k < c.a1) {
    const int4 r0 = P.act[k];
    if (iv_pass<OVL>(P, r0, s, e, rstrand, o)) hit |= 1ull << ((u32) r0.z & 0xffu);
    ++walked;
  }
  for (;;) {
    if (f0.x > e) break;                           // isAfter, mmannot.cpp:1311 (also the end-of-chromosome sentinel)
    if (iv_pass<OVL>(P, f0, s, e, rstrand, o)) hit |= 1ull << ((u32) f0.z & 0xffu);
    ++walked;
    if (f1.x > e) break;
    if (iv_pass<OVL>(P, f1, s, e, rstrand, o)) hit |= 1ull << ((u32) f1.z & 0xffu);
    ++walked;
    i += 2;
    f0 = (i < c.i1) ? P.iv[i] : none;
    f1 = (i + 1 < c.i1) ? P.iv[i + 1] : none;
  }
  if (!hit) return 0;
  // getFirst (mmannot.cpp:1029-1076): only the first Order line holding a hit element counts
  const int low = __ffsll((long long) hit) - 1;
  const u64 cand = hit & P.lineMask[low];
  if ((cand & (cand - 1)) == 0) return cand;
  // inclusion counts every passing interval as overlap 1, and only "upstream" / "downstream" elements have a distance: all tie
  if (OVL == MM_OVL_INCLUSION && !(cand & (P.elemUp | P.elemDown))) return cand;
  // several elements of the winning line were hit: max overlap, then min distance, ties kept
  u64 sel = 0;
  u32 maxO = 0, minD = 0;
  for (u64 cm = cand; cm; cm &= cm - 1) {
    const u32 t = (u32) (__ffsll((long long) cm) - 1);
    u32 o = 0, d = 0;
    last_of_type<OVL>(P, c, s, e, rstrand, t, o, d);
    if (o > maxO) { maxO = o; minD = d; sel = 1ull << t; }
    else if (o == maxO) {
      if (d < minD) { minD = d; sel = 1ull << t; }
      else if (d == minD) sel |= 1ull << t;
    }
  }
  return sel;
}

// ----------------------------------------------------------------------------- tiles of the grouping pipeline (k_probe fills the first histogram itself when it can)
#define RS_THREADS 512
#define RS_NW (RS_THREADS / 64)
#define RS_ROUNDS 16
#define RS_TILE (RS_THREADS * RS_ROUNDS)

// ----------------------------------------------------------------------------- stable radix passes (digits of up to 9 bits)
// First pass (FIRST = true) reads the batch's records where k_probe left them: key gkey[i] (NO_KEY = not in the
// grouping pipeline) and payload gpay[i], i in file order.  Filtering, key rebasing (qid - qmin) and the choice of the
// payload happen on the fly, so the ordered compaction costs no pass of its own; the pass leaves the pairs compacted,
// in file order within each digit, and k_rs_scan publishes their number (INFO_M) for the following passes.
// Histogram update of one wave.  Read-name ids spread evenly over the digits and plain LDS atomics are best (the
// aggregation below costs 6 % of the sort there).  -y ratio sorts by table slot, where a handful of rows take most
// records: then (skewed) the digit of the first pending lane is counted once for all the lanes that share it, twice,
// before the rest use atomics - this removes the same-address serialisation SQ_WAIT_INST_LDS showed in these kernels.
// SH = 0: h holds one u32 counter per digit; SH = 1: two 16-bit counters per u32 (digit d in half d & 1 of word d >> 1;
// a wave adds at most 1024 per digit and tile, so a half never carries into its neighbour).
template <int SH>
__device__ __forceinline__ void wave_hist_add(u32 *h, u32 d, bool ok, bool skewed) {
  const u32 idx = SH ? (d >> 1) : d, sh = SH ? ((d & 1u) << 4) : 0u;
  if (!skewed) { if (ok) atomicAdd(&h[idx], 1u << sh); return; }
  const u32 lane = threadIdx.x & 63;
  u64 pend = __ballot(ok);
  for (int it = 0; it < 2 && pend; ++it) {
    const int l0 = __ffsll((long long) pend) - 1;
    const u32 d0 = (u32) __builtin_amdgcn_readlane((int) d, l0);
    const u64 same = __ballot(ok && d == d0) & pend;
    if ((int) lane == l0) atomicAdd(&h[idx], (u32) __popcll(same) << sh);
    pend &= ~same;
  }
  if ((pend >> lane) & 1ull) atomicAdd(&h[idx], 1u << sh);
}

// scan() + getFirst() of one tuple: read strand from -s (strandF / strandR / strandU, mmannot.cpp:836-844), then the index
template <int OVL, bool EAGER = false>
__device__ __forceinline__ u64 probe_tuple(const ProbeParams &P, bool useSlots, const uint2 *chrSlot, const u64 *sLine, const int4 a, u32 &walked) {
  const int rec_strand = a.z & 1;
  const int rstrand = (P.strandedness == MM_STRANDED_U) ? 1 : ((P.strandedness == MM_STRANDED_F) ? rec_strand : (rec_strand ^ 1));
  // reads without CIGAR operations (end < start) need the 'leading isBefore run' rule and take the full structure
  return (useSlots && a.y >= a.x) ? probe_slots<OVL, EAGER>(P, chrSlot, sLine, a.x, a.y, TSN_TID1(a.z), rstrand, walked)
                                    : probe_one<OVL>(P, a.x, a.y, TSN_TID1(a.z), rstrand, walked);
}

// 12-byte link format of a tuple (mm_aln12, include/mmannot_amd.h) -> the 16-byte tuple every kernel reads.  Lossless; what does
// not fit the narrower fields never gets packed (mm_pack12 refuses the batch).
__global__ __launch_bounds__(256) void k_expand12(const u32 *__restrict__ pk, int4 *__restrict__ out, size_t n, u32 qid_base) {
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
    const u32 w0 = __builtin_nontemporal_load(pk + 3 * i), w1 = __builtin_nontemporal_load(pk + 3 * i + 1), w2 = __builtin_nontemporal_load(pk + 3 * i + 2);
    const int start = (int) (w0 & 0x7FFFFFFFu);
    const u32 len = w1 & 0x3FFFu, nh8 = (w1 >> 14) & 0xFFu, tid1 = (w1 >> 22) | ((w2 >> 24) << 10);
    const u32 nh = nh8 == 255u ? (u32) MM_ALN_NH_ESCAPE : nh8;
    out[i] = make_int4(start, start + (int) len - 1, (int) ((w0 >> 31) | (nh << 1) | (tid1 << (MM_ALN_NH_BITS + 1))), (int) (qid_base + (w2 & 0xFFFFFFu)));
  }
}

// ----------------------------------------------------------------------------- what every probe kernel does with a tuple's element set
// addCount's per-hit half (mmannot.cpp:1666-1668, 1703-1738) and the hand-over of multi-hit records to the grouping pipeline
// (mmannot.cpp:1669-1702), shared by k_probe (the general walk), k_probe_fast (round 3: pipelined walk over the 32-byte slots) and
// k_probe_list (the tuples k_probe_fast passes on).  Wave-level ballots inside: every lane of the wave has to come through here,
// lanes without a tuple with processed = false.
struct ProbeAcc {
  // the six per-hit counters of addCount, two 16-bit fields to a register, kept per lane and summed once at the end (a lane sees at
  // most 2^32 / (2048 x 256) = 8192 tuples): a ballot and a population count per counter and tuple was a tenth of the kernel's
  // instructions, and the probe is bound by instruction issue (SQ_ACTIVE_INST_VALU = 73 % of the SIMDs' issue slots, round 2)
  u32 cHU = 0, cAM = 0, cRN = 0;   // hits | unassigned << 16;  ambiguous | multiple << 16;  reads | unique << 16
  u32 qinv = 0, qmax = 0, bigNh = 0;
};
enum { HIST_NONE = 0, HIST_LDS = 1, HIST_GLOBAL = 2 };   // where the first partition pass's digit counts of a tile go: nowhere / the workgroup's LDS row / straight to hist[] with atomics (k_probe_list)
template <u32 CAP>
__device__ __forceinline__ void tuple_tail(const bool grouped_strategy, const bool ratio, const bool randomStrategy, const int histMode, LdsTableT<CAP> &T,
                                           u32 *sHist, u32 *__restrict__ hist, const u32 nblk, const GlobalTable &G, const u32 qlo, const u32 lane,
                                           const size_t i, const int4 a, const u32 nh, const bool processed, const u64 mask, ProbeAcc &A, u32 &key, u32 &pay) {
  const int np = __popcll(mask);
  bool grouped = processed && grouped_strategy && nh > 1;
  // a multi-hit tuple below the caller's qid_lo: the launches (and, fused, the histogram's origin) were sized on that promise -
  // the batch is refused, and the tuple stays out of the pipeline so that nothing is written out of bounds
  if (grouped && (u32) a.w < qlo) { atomicCAS(G.err, 0u, (u32) ERR_QID_LO); grouped = false; }
  A.cHU += (processed ? 1u : 0u) | ((processed && np == 0) ? 0x10000u : 0u);
  A.cAM += ((processed && np > 1) ? 1u : 0u) | (grouped ? 0x10000u : 0u);
  if (histMode == HIST_LDS && !ratio) {   // (wave-uniform) the wave's records mostly share one bucket: count those once, the others one by one
    const u64 gm = __ballot(grouped);
    if (gm) {
      const u32 d = min(((u32) a.w - qlo) >> LEAF_BITS, RS_BINS - 1u);   // (an id outside the promised range is an error anyway)
      const int l0 = __ffsll((long long) gm) - 1;
      const u32 d0 = (u32) __builtin_amdgcn_readlane((int) d, l0);
      const u64 same = __ballot(grouped && d == d0);
      if ((int) lane == l0) atomicAdd(&sHist[d0], (u32) __popcll(same));
      else if (grouped && d != d0) atomicAdd(&sHist[d], 1u);
    }
  } else if (histMode == HIST_GLOBAL && !ratio && grouped)
    atomicAdd(&hist[(size_t) min(((u32) a.w - qlo) >> LEAF_BITS, RS_BINS - 1u) * nblk + (u32) (i / RS_TILE)], 1u);
  A.cRN += ((processed && !grouped) ? 1u : 0u) | ((processed && np == 1 && nh == 1) ? 0x10000u : 0u);   // mmannot.cpp:1737, 1668
  const bool counted = processed && !grouped && mask != 0 && !randomStrategy;   // mmannot.cpp:1704 (random: the caller decides)
  key = NO_KEY; pay = GT_NO_SLOT;
  if (grouped) {
    key = (u32) a.w;
    A.qinv = max(A.qinv, ~key); A.qmax = max(A.qmax, key);
    if (nh >= 0xFFFFu) A.bigNh = 1;
  }
  // counting / slot lookup, per lane through the workgroup's LDS tables:
  //   per-hit branch (default/unique): the set is counted now;
  //   -y ratio: the set's table slot becomes the sort key of the ordered accumulation;
  //   multi-hit branch: the slot travels with the record so that k_chunk_default streams.
  if (counted && !ratio) count_set(T, G, mask);
  if (ratio && counted) { key = slot_of_set(T, G, mask); if (key != NO_KEY && key >= RS_BINS) A.bigNh = 1; }   // (-y ratio: the flag means "wide key")
  // -y ratio: the one-pass sort's digit is the slot itself (a handful of rows take most hits: equal digits are counted together);
  // should a slot of the batch turn out not to be below RS_BINS, this histogram is void and k_rs_hist counts 8-bit digits instead
  if (histMode == HIST_LDS && ratio) wave_hist_add<0>(sHist, key & (RS_BINS - 1u), key != NO_KEY, true);
  else if (histMode == HIST_GLOBAL && ratio) { if (key != NO_KEY) atomicAdd(&hist[(size_t) (key & (RS_BINS - 1u)) * nblk + (u32) (i / RS_TILE)], 1u); }
  else if (grouped && mask != 0) {
    if ((mask & (mask - 1)) == 0) pay = (u32) (__ffsll((long long) mask) - 1);   // one element: the payload names it
    else pay = slot_of_set(T, G, mask) & 0xFFFFu;   // (NO_KEY, table full: error flag raised, the result is refused)
  }
}

// end of a probe kernel: the per-lane counters and key range -> the context's counters and the batch scalars; the workgroup's counts -> the table
template <u32 CAP>
__device__ __forceinline__ void probe_finish(const bool grouped_strategy, const bool ratio, const bool haveKey, ProbeAcc &A, u32 *sstat, u32 *sq, LdsTableT<CAP> &T,
                                             const GlobalTable &G, u64 *__restrict__ stats, u32 *__restrict__ info) {
  const u32 lane = threadIdx.x & 63;
  if (grouped_strategy && haveKey) {
    for (int off = 32; off > 0; off >>= 1) { A.qinv = max(A.qinv, __shfl_down(A.qinv, off)); A.qmax = max(A.qmax, __shfl_down(A.qmax, off)); A.bigNh |= __shfl_down(A.bigNh, off); }
    if (lane == 0) { atomicMax(&sq[0], A.qinv); atomicMax(&sq[1], A.qmax); if (A.bigNh) atomicOr(&sq[2], 1u); }
  }
  if (ratio && haveKey && __ballot(A.bigNh != 0) && lane == 0) atomicOr(&sq[2], 1u);
  {
    u32 c[6] = {A.cHU & 0xFFFFu, A.cHU >> 16, A.cAM & 0xFFFFu, A.cAM >> 16, A.cRN & 0xFFFFu, A.cRN >> 16};
#pragma unroll
    for (int k = 0; k < 6; ++k) for (int off = 32; off > 0; off >>= 1) c[k] += __shfl_down(c[k], off);
    if (lane == 0) {
      atomicAdd(&sstat[MM_STAT_HITS], c[0]); atomicAdd(&sstat[MM_STAT_UNASSIGNED], c[1]);
      atomicAdd(&sstat[MM_STAT_AMBIGUOUS], c[2]); atomicAdd(&sstat[MM_STAT_MULTIPLE], c[3]);
      atomicAdd(&sstat[MM_STAT_READS], c[4]); atomicAdd(&sstat[MM_STAT_UNIQUE], c[5]);
    }
  }
  __syncthreads();
  if (threadIdx.x < MM_N_STATS && sstat[threadIdx.x]) atomicAdd(&stats[threadIdx.x], (u64) sstat[threadIdx.x]);
  if (grouped_strategy && haveKey && threadIdx.x == 0) {
    if (sq[0]) atomicMax(&info[INFO_QINV], sq[0]);
    if (sq[1]) atomicMax(&info[INFO_QMAX], sq[1]);
    if (sq[2]) atomicOr(&info[INFO_BIGNH], 1u);
  }
  if (ratio && haveKey && threadIdx.x == 0 && sq[2]) atomicOr(&info[INFO_WIDEKEY], 1u);
  lds_table_flush(T, G);
}

// ----------------------------------------------------------------------------- k_probe
// One lane per alignment, grid-stride over the batch.  Per alignment: scan+getFirst, the
// three per-hit counters of addCount (mmannot.cpp:1666-1668), then either
//   * the per-hit branch (mmannot.cpp:1703-1738): element set counted through a wave-aggregated,
//     LDS-privatised table (default with NH<=1, unique), or its table slot recorded for the
//     ordered -y ratio accumulation, or
//   * the multi-hit branch (mmannot.cpp:1669-1702): set + name id recorded for k_chunk_default.
// what a k_probe instance takes for granted (the general one reads everything from its arguments)
enum { PV_GENERAL = 0, PV_COUNT = 1,        // per-hit counting only: default with the single-hit promise, unique
       PV_GROUPS_HIST = 2, PV_GROUPS = 3,   // default with multi-hit records, with / without the fused bucket histogram
       PV_RATIO = 4, PV_RATIO_HIST = 5 };   // -y ratio, without / with the fused histogram of the one-pass sort by table slot
template <int OVL, int VAR>
__global__ __launch_bounds__(256) MM_PROBE_ATTR void k_probe(const ProbeParams P, const int4 *__restrict__ aln, const uint2 *__restrict__ big, u32 nbig,
                                               int single, u32 qlo, size_t n, u64 *__restrict__ hm, u32 *__restrict__ gkey, u32 *__restrict__ gpay,
                                               u32 *__restrict__ info, GlobalTable G, u64 *__restrict__ stats,
                                               u64 *__restrict__ walkstat, u32 *__restrict__ hist, u32 nblk) {
  // hist != NULL (the host knows that the batch takes the leaf layout, rebased at qlo): one workgroup per RS_TILE tuples, in tile
  // order, which also counts the tile's multi-hit records per bucket - the partition pass's histogram, without a kernel of its own
  // re-reading the keys.  The probe waits on gathers most of the time; the LDS adds ride along.
  __shared__ LdsTable T;
  __shared__ u32 sHist[RS_BINS];
  __shared__ u32 sstat[8];
  __shared__ u32 sq[3];   // max(~qid) = ~min, max(qid), saw NH >= 0xFFFF
  __shared__ u64 sLine[64];          // lineMask and the per-chromosome slot table are read once per alignment:
  __shared__ uint2 sChr[MM_LDS_CHR]; // keep them in LDS instead of gathering them from global memory
  if (threadIdx.x < 64) sLine[threadIdx.x] = P.lineMask[threadIdx.x];
  for (u32 c = threadIdx.x; c < min(P.nchr, (u32) MM_LDS_CHR); c += blockDim.x) sChr[c] = P.chrSlot[c];
  // VAR != 0: what the launch knows is a compile-time constant here (see the PV_* list) - each run-time flag of the general kernel
  // (VAR == PV_GENERAL) costs a pair of scalar registers held across the loop, and the kernel already keeps some in VGPR lanes
  constexpr bool V = VAR != PV_GENERAL;
  const uint2 *chrSlot = (V || P.nchr <= (u32) MM_LDS_CHR) ? sChr : P.chrSlot;
  const bool useHist = V ? (VAR == PV_GROUPS_HIST || VAR == PV_RATIO_HIST) : (hist != nullptr);
  const bool haveKey = V ? (VAR != PV_COUNT) : (gkey != nullptr);
  const bool keepSets = V ? false : (hm != nullptr && P.keep);
  const bool useSlots = V ? true : (P.useSlots != 0);
  lds_table_init(T);
  if (threadIdx.x < 8) sstat[threadIdx.x] = 0;
  if (threadIdx.x < 3) sq[threadIdx.x] = 0;
  if (useHist) for (u32 d = threadIdx.x; d < RS_BINS; d += blockDim.x) sHist[d] = 0;
  __syncthreads();
  const u32 lane = threadIdx.x & 63;
  ProbeAcc A;
  u32 walked = 0;
  const bool grouped_strategy = V ? (VAR == PV_GROUPS || VAR == PV_GROUPS_HIST) : (P.strategy == MM_STRAT_DEFAULT);
  const bool ratio = V ? (VAR == PV_RATIO || VAR == PV_RATIO_HIST) : (P.strategy == MM_STRAT_RATIO);
  const bool uniqueOnly = (V && VAR != PV_COUNT) ? false : (P.strategy == MM_STRAT_UNIQUE);
  const bool randomStrategy = V ? false : (P.strategy == MM_STRAT_RANDOM);
  const size_t first = useHist ? (size_t) blockIdx.x * RS_TILE : (size_t) blockIdx.x * blockDim.x;
  const size_t last = useHist ? min(n, first + (size_t) RS_TILE) : n, step = useHist ? (size_t) blockDim.x : (size_t) gridDim.x * blockDim.x;
  for (size_t base = first; base < last; base += step) {
    const size_t i = base + threadIdx.x;
    const bool valid = i < last;
    int4 a = make_int4(0, 0, 0, 0);
    u32 nh = 1;
    if (valid) {
      a = ld_stream(aln + i);
      nh = tuple_nh((u32) a.z, i, big, nbig, G.err);
      if (single && nh > 1) { atomicCAS(G.err, 0u, (u32) ERR_SINGLE_HIT); nh = 1; }   // MM_BATCH_SINGLE_HIT was a promise
    }
    const bool processed = valid && !(uniqueOnly && nh != 1);  // mmannot.cpp:1773
    u64 mask = 0;
    if (processed) mask = probe_tuple<OVL>(P, useSlots, chrSlot, sLine, a, walked);
    u32 key, pay;
    tuple_tail(grouped_strategy, ratio, randomStrategy, useHist ? HIST_LDS : HIST_NONE, T, sHist, hist, nblk, G, qlo, lane, i, a, nh, processed, mask, A, key, pay);
    if (valid) {
      if (haveKey) {
        gkey[i] = key;                       // read again right away by the first sort pass: keep these cacheable
        gpay[i] = ratio ? nh : (pay | (min(nh, 0xFFFFu) << 16));
      }
      if (keepSets) st_stream(hm + i, processed ? mask : ~0ull);   // (parity / side channels only, see k_probe_sets)
    }
  }
  if (!V && P.keep) {
    // total candidates visited (profiling aid)
    u32 w = walked;
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off);
    if (lane == 0) atomicAdd(&walkstat[0], (u64) w);
  }
  // (the host has zeroed hist: a tile's names fall into a few of the 512 buckets in most inputs, and each count is a store of its own
  // into another row - writing all 512 was 6 % of the kernel)
  __syncthreads();
  if (useHist) for (u32 d = threadIdx.x; d < RS_BINS; d += blockDim.x) if (sHist[d]) hist[(size_t) d * nblk + blockIdx.x] = sHist[d];
  probe_finish(grouped_strategy, ratio, haveKey, A, sstat, sq, T, G, stats, info);
}

// The element set of a multi-hit record normally travels in 16 bits of its sort payload.  A batch holding an NH >= 65535 cannot
// spare them (INFO_BIGNH: the payload is the record's index instead) - only then, which no real aligner output does, this kernel
// runs the probe a second time and leaves every multi-hit record's set in hm[] for the merge kernels.  (Until round 2 k_probe
// stored 8 bytes per multi-hit record for this case in every batch.)
template <int OVL>
__global__ __launch_bounds__(256) MM_PROBE_ATTR void k_probe_sets(const ProbeParams P, const int4 *__restrict__ aln, const uint2 *__restrict__ big, u32 nbig,
                                                                  size_t n, u64 *__restrict__ hm, const u32 *__restrict__ info, u32 *__restrict__ err) {
  __shared__ u64 sLine[64];
  __shared__ uint2 sChr[MM_LDS_CHR];
  if (!info[INFO_BIGNH]) return;
  if (threadIdx.x < 64) sLine[threadIdx.x] = P.lineMask[threadIdx.x];
  for (u32 c = threadIdx.x; c < min(P.nchr, (u32) MM_LDS_CHR); c += blockDim.x) sChr[c] = P.chrSlot[c];
  const uint2 *chrSlot = (P.nchr <= (u32) MM_LDS_CHR) ? sChr : P.chrSlot;
  __syncthreads();
  u32 walked = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
    const int4 a = ld_stream(aln + i);
    if (tuple_nh((u32) a.z, i, big, nbig, err) > 1) hm[i] = probe_tuple<OVL>(P, P.useSlots != 0, chrSlot, sLine, a, walked);
  }
}

// ----------------------------------------------------------------------------- k_probe_fast (round 3)
// The same answers as k_probe for the tuples it takes, through the 32-byte fast slots (4-byte bin-relative records, mm_index_create)
// and a software pipeline per lane.  Why: k_probe's waves spend two thirds of their time in s_waitcnt on a chain of three dependent
// loads per tuple (tuple -> bitmap word -> slot line [-> rest of the line]) with nothing else in flight (profiles/r02e, r03a), and a
// dependent second access to a slot line costs as much as the first (profiles/r03a_exp_gather.txt: 0.31 -> 0.46-0.57 ms per 2^24).
// Here one workgroup takes a tile of RS_TILE tuples and every lane keeps three tuples in flight:
//     iteration k:  wait: slot(k) has landed in LDS, tuple(k+1) in registers
//                   request bitmap word(k+1), tuple(k+2);  when the word is there: request slot(k+1) - the whole 32 bytes, by LDS-DMA
//                   (global_load_lds_dwordx4: no destination registers while in flight; each lane reads back its own 32 bytes only)
//                   test the records of slot(k) - straight-line, no further load
// so the slot fetch of tuple k+1 (the long pole: an L2 miss) and the stream of tuple k+2 overlap the arithmetic on tuple k, and the
// loop has no load that depends on another load of the same iteration except the bitmap word (L2).  Everything that would need one -
// more than 7 records in the bin, a read that spans further than its slot's reach, is longer than the saturating fields allow or has
// no CIGAR operations, the second selection pass (several elements of the winning Order line: mmannot.cpp:1049-1075) - is not done
// here: the tuple's index goes to a list (a region per wave: no atomics) and k_probe_list walks it with k_probe's code right after.
#define FS_LT_CAP 64u
// LDS-DMA: 16 bytes per lane from base + off (bytes, per lane) to the LDS address LDS + 16 x lane; MM_GLDS32: the next 16 bytes too, to
// LDS2 + 16 x lane.  M0 (the LDS base of the transfer) is the compiler's: saved and put back inside the statement that uses it.
#define MM_GLDS16(BASE, OFF, LDS, CACHE)                                                                                       \
  {                                                                                                                            \
    u32 keep_;                                                                                                                 \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" CACHE "\n\ts_mov_b32 m0, %0"  \
                 : "=&s"(keep_) : "v"(OFF), "s"(BASE), "s"(LDS) : "memory");                                                   \
  }
#define MM_GLDS4(BASE, OFF, LDS)                                                                                               \
  {                                                                                                                            \
    u32 keep_;                                                                                                                 \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"             \
                 : "=&s"(keep_) : "v"(OFF), "s"(BASE), "s"(LDS) : "memory");                                                   \
  }
#define MM_GLDS32(BASE, OFF, LDS, LDS2)                                                                                        \
  {                                                                                                                            \
    u32 keep_;                                                                                                                 \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"                      \
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:16\n\ts_mov_b32 m0, %0"                  \
                 : "=&s"(keep_) : "v"(OFF), "s"(BASE), "s"(LDS), "s"((LDS2) - 16u) : "memory");                                \
  }
template <int OVL, int VAR>
__global__ __launch_bounds__(256) MM_PROBE_ATTR void k_probe_fast(const ProbeParams P, const int4 *__restrict__ aln, int single, u32 qlo, size_t n,
                                                                  u32 *__restrict__ gkey, u32 *__restrict__ gpay, u32 *__restrict__ info, GlobalTable G,
                                                                  u64 *__restrict__ stats, u32 *__restrict__ hist, u32 nblk, u32 *__restrict__ dlist,
                                                                  u32 *__restrict__ dlist2, u32 *__restrict__ dcnt) {
  constexpr bool useHist = VAR == PV_GROUPS_HIST || VAR == PV_RATIO_HIST;
  constexpr bool haveKey = VAR != PV_COUNT;
  constexpr bool grouped_strategy = VAR == PV_GROUPS || VAR == PV_GROUPS_HIST;
  constexpr bool ratio = VAR == PV_RATIO || VAR == PV_RATIO_HIST;
  __shared__ LdsTableT<FS_LT_CAP> T;
  __shared__ u32 sHist[useHist ? RS_BINS : 1];
  __shared__ u32 sstat[8];
  __shared__ u32 sq[3];
  __shared__ u64 sLine[64];
  __shared__ uint2 sChr[MM_LDS_CHR];
  // landing zones of the loads in flight (LDS-DMA: no destination registers, and invisible to the compiler's vmcnt bookkeeping - every
  // wait on them is written out below): lane t's slot is S[0][t], S[1][t]; its next tuple is TU[t]
  __shared__ uint4 S[2][256];
  __shared__ uint4 TU[256];
  __shared__ u32 CW[256];               // the word of fcls holding the class byte of a covered bin
  const u32 t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t < 64) sLine[t] = P.lineMask[t];
  for (u32 c = t; c < (u32) MM_LDS_CHR; c += 256) sChr[c] = c < P.nchr ? P.fchr[c] : make_uint2(0u, 0u);   // (beyond the last chromosome: no bins)
  lds_table_init(T);
  if (t < 8) sstat[t] = 0;
  if (t < 3) sq[t] = 0;
  if (useHist) for (u32 d = t; d < RS_BINS; d += 256) sHist[d] = 0;
  __syncthreads();
  const bool uniqueOnly = VAR == PV_COUNT && P.strategy == MM_STRAT_UNIQUE;
  const size_t first = (size_t) blockIdx.x * RS_TILE;
  const u32 cnt = (u32) (min(n, first + (size_t) RS_TILE) - first);      // tuples of this tile; everything below is tile-local (32-bit offsets)
  const int nit = (int) ((cnt + 255u) / 256u);
  const int4 *const alnT = aln + first;
  u32 *const gkeyT = haveKey ? gkey + first : nullptr, *const gpayT = haveKey ? gpay + first : nullptr;
  const u32 sshift = P.fshift, soMask = (2u << sshift) - 1u, Wm1 = (1u << sshift) - 1u, freach = P.freach;
  const u32 ldsS = (u32) __builtin_amdgcn_readfirstlane((int) (u32) (size_t) &S[0][t & ~63u]);   // this wave's KiB of S[0]; S[1] lies 4 KiB behind
  const u32 ldsT = (u32) __builtin_amdgcn_readfirstlane((int) (u32) (size_t) &TU[t & ~63u]);
  const u32 ldsC = (u32) __builtin_amdgcn_readfirstlane((int) (u32) (size_t) &CW[t & ~63u]);
  const u64 updown = P.elemUp | P.elemDown;
  ProbeAcc A;
  u32 dn = 0, en = 0;                                    // tuples this wave has passed on so far: to k_probe_list / to its own extension pass
  u32 *const dl = dlist + first + (size_t) w * (RS_TILE / 4);

  // stage "locate" of tuple li (tile-local index): which slot, may this kernel answer it, is the slot empty; requests the slot's 32
  // bytes and tuple li + 256.  FL: bit 0 = has a tuple, 1 = skipped by -y unique, 2 = passed on, 3 = slot requested, 4 = covered bin (class word
  // requested), bits 8-9 = which byte of that word
  int4 a0 = make_int4(0, 0, 0, 0), a1 = a0;            // tuple it (located, its slot requested), it + 1 (loaded)
  u32 fl0 = 0;
  int bs0 = 0;                                           // first base of tuple it's bin
#define MM_FAST_LOCATE(A_, LI_, FL_, BS_)                                                                             \
  {                                                                                                                  \
    /* straight-line up to the bitmap word: a load whose use sits in another branch than the load makes the compiler wait with \
       vmcnt(0) where the branches join - behind our requests, which that wait would then drain */                    \
    const bool val_ = (LI_) < cnt;                                                                                   \
    const u32 tid0_ = TSN_TID1((A_).z) - 1u;            /* chromosome; 0xFFFFFFFF = unknown to the annotation */     \
    const u32 nhf_ = ((u32) (A_).z >> 1) & MM_ALN_NH_ESCAPE;                                                         \
    const uint2 cs_ = sChr[min(tid0_, (u32) MM_LDS_CHR - 1u)];                                                       \
    const u32 nb_ = (val_ && tid0_ < P.nchr) ? cs_.y : 0u;                                                           \
    const u32 sb_ = (u32) (A_).x >> sshift, db_ = ((u32) (A_).y >> sshift) - sb_;                                    \
    /* one slot serves the read: it ends in its bin, or in the reach behind it, or the bin is the chromosome's last */ \
    const bool one_ = db_ == 0 || sb_ + 1 == nb_ || (db_ == 1 && ((u32) (A_).y & Wm1) < freach);                     \
    const bool skip_ = val_ && uniqueOnly && nhf_ != 1u;                                                             \
    const bool esc_ = nhf_ == MM_ALN_NH_ESCAPE;        /* NH in the side list: a dependent load - not in this loop */ \
    /* (a read that starts behind the chromosome's last bin starts behind every interval's end: no interval includes or overlaps it) */ \
    const bool look_ = nb_ != 0 && !skip_ && (sb_ < nb_ || (A_).y < (A_).x);                                         \
    const bool elig_ = look_ && !esc_ && (A_).y >= (A_).x && one_;                                                   \
    const bool pass_ = val_ && !skip_ && (esc_ || (look_ && !elig_));                                                \
    const u32 slot_ = elig_ ? cs_.x + sb_ : 0u;                                                                      \
    BS_ = (int) (sb_ << sshift);                                                                                     \
    const u32 mw_ = P.fbits[slot_ >> 4];                                                                             \
    const u32 code_ = elig_ ? (mw_ >> ((slot_ & 15u) << 1)) & 3u : 0u;                                              \
    FL_ = (val_ ? 1u : 0u) | (skip_ ? 2u : 0u) | (pass_ ? 4u : 0u) | (code_ == FS_GENERAL ? 8u : 0u) | (code_ == FS_COVERED ? 16u : 0u) | ((slot_ & 24u) << 5); \
    if (code_ == FS_GENERAL) {                                                                                       \
      const u32 o_ = slot_ << 5;                                                                                     \
      MM_GLDS32(P.fslots, o_, ldsS, ldsS + 4096u)                                                                    \
    }                                                                                                                \
    if (code_ == FS_COVERED) { const u32 o_ = (slot_ >> 3) & ~3u; MM_GLDS4(P.fcls, o_, ldsC) }                       \
    if ((LI_) + 256u < cnt) { const u32 o_ = ((LI_) + 256u) << 4; MM_GLDS16(alnT, o_, ldsT, " nt") }                 \
  }
  // prologue: tuple 0 located and its slot requested, tuple 1 requested
  if (t < cnt) a0 = ld_stream(alnT + t);
  MM_FAST_LOCATE(a0, t, fl0, bs0)
  u32 pkey = 0, ppay = 0, pli = 0;
  bool pend = false;                                     // outputs of the previous tuple, stored one iteration late (see below)
  for (int it = 0; it < nit; ++it) {
    const u32 li = (u32) it * 256u + t;
    // slot(it) and tuple(it+1) are in LDS: everything requested so far has arrived
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint4 q0 = S[0][t], q1 = S[1][t];
    const u32 cw = CW[t];
    { const uint4 u = TU[t]; a1 = make_int4((int) u.x, (int) u.y, (int) u.z, (int) u.w); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the next requests may land in S / TU right away)
    // the previous tuple's outputs go out now, ahead of this iteration's requests: at the loop's wait they are the oldest and long done
    // (stores count in vmcnt like loads, in issue order - issued last they would be what every wave waits for)
    if (haveKey && pend) { gkeyT[pli] = pkey; gpayT[pli] = ppay; }
    // tuple it+1: locate, bitmap word, slot request; tuple it+2: request
    u32 fl1; int bs1;
    MM_FAST_LOCATE(a1, li + 256u, fl1, bs1)
    // ---- tuple `it` against its slot: straight-line - seven records, the unused ones zero (no ok bit: they pass nothing)
    const bool valid = (fl0 & 1u) != 0;
    u32 nh = ((u32) a0.z >> 1) & MM_ALN_NH_ESCAPE;
    if (single && valid && nh > 1) { atomicCAS(G.err, 0u, (u32) ERR_SINGLE_HIT); nh = 1; }   // MM_BATCH_SINGLE_HIT was a promise
    const bool have = (fl0 & 8u) != 0;
    const bool ovf = have && (q0.x & FS_OVERFLOW) != 0;
    const int rec_strand = a0.z & 1;
    const u32 okbit = 24u + (u32) ((P.strandedness == MM_STRANDED_U) ? 1 : ((P.strandedness == MM_STRANDED_F) ? rec_strand : (rec_strand ^ 1)));
    const int sr = a0.x - bs0, er = a0.y - bs0;           // the read relative to its bin (sr >= 0)
    u64 hit = 0;
#define MM_FAST_TEST(R)                                                                                           \
  {                                                                                                              \
    const u32 r_ = (R);                                                                                          \
    hit |= (u64) span_pass<OVL>(P, (int) (r_ & soMask), (int) ((r_ & 0xFFFFFFu) >> (sshift + 1)), (r_ >> okbit) & 1u, sr, er) << (r_ >> 26); \
  }
    MM_FAST_TEST(q0.y) MM_FAST_TEST(q0.z) MM_FAST_TEST(q0.w) MM_FAST_TEST(q1.x) MM_FAST_TEST(q1.y) MM_FAST_TEST(q1.z) MM_FAST_TEST(q1.w)
#undef MM_FAST_TEST
    if (!have || ovf) hit = 0;                             // (a lane without a slot has read stale LDS)
    {                                                      // a covered bin: one interval, open at the bin's left edge, reaching past every read served here
      const u32 cb = (fl0 & 16u) ? (cw >> ((fl0 >> 5) & 24u)) & 0xFFu : 0u;
      const u32 r_ = (cb << 24) | (0xFFFFFFu & ~soMask);
      hit |= (u64) span_pass<OVL>(P, 0, (int) (0xFFFFFFu >> (sshift + 1)), (r_ >> okbit) & 1u, sr, er) << (r_ >> 26);
    }
    // getFirst (mmannot.cpp:1029-1076): only the first Order line holding a hit element counts; one element, or inclusion without
    // "upstream" / "downstream" elements (every passing interval has overlap 1, distance 0: all tie) is settled here - max overlap /
    // min distance over last writers is k_probe_list's
    const u64 cand = hit & sLine[(__ffsll((long long) hit) - 1) & 63];
    const bool settled = (cand & (cand - 1)) == 0 || (OVL == MM_OVL_INCLUSION && !(cand & updown));
    const u64 mask = settled ? cand : 0ull;
    const bool ext = ovf && (q0.x & FS_EXT_NONE) != FS_EXT_NONE;      // the bin's records are in an extension block: this kernel's own later pass
    const u32 xblk = q0.x & FS_EXT_NONE;
    const bool defer = (fl0 & 4u) != 0 || ovf || !settled;              // not answered in this iteration
    const bool processed = valid && !(fl0 & 2u) && !defer;
    {                                                      // pass the tuple on: this wave's region, no atomics - k_probe_list's from the front, ours from the back
      const u64 dm = __ballot(defer && !ext), em = __ballot(ext);
      if (dm | em) {
        const u64 below = (1ull << lane) - 1ull;
        if (defer && !ext) dl[dn + (u32) __popcll(dm & below)] = (u32) (first + li);
        if (ext) dl[RS_TILE / 4 - 1 - (en + (u32) __popcll(em & below))] = li | (xblk << 13);
        dn += (u32) __popcll(dm); en += (u32) __popcll(em);
      }
    }
    u32 key, pay;
    tuple_tail(grouped_strategy, ratio, false, useHist ? HIST_LDS : HIST_NONE, T, sHist, hist, nblk, G, qlo, lane, first + li, a0, nh, processed, mask, A, key, pay);
    pend = valid && !defer;                                // (k_probe_list writes the outputs of the tuples it is handed)
    pli = li; pkey = key; ppay = ratio ? nh : (pay | (min(nh, 0xFFFFu) << 16));
    a0 = a1; fl0 = fl1; bs0 = bs1;
  }
#undef MM_FAST_LOCATE
  if (haveKey && pend) { gkeyT[pli] = pkey; gpayT[pli] = ppay; }
  // ---- the tuples on bins with 8 to 31 records: one more pass per 64 of them - tuple and extension block requested together, the block's
  // records tested like a slot's.  (A chain of two loads at the end of the workgroup's life, where k_probe_list would walk the 8-byte
  // slots and their lists behind a kernel boundary: five to ten dependent loads.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (this wave's writes to its region)
  for (u32 base = 0; base < en; base += 64) {
    const bool valid = base + lane < en;
    // (from the region's lower end: what this pass still hands to k_probe_list is appended to the front part, which may then grow over
    // entries of this pass and of earlier ones only)
    const u32 ent = valid ? __builtin_nontemporal_load(dl + (RS_TILE / 4 - en + base + lane)) : 0u;
    const u32 li = ent & (RS_TILE - 1u);
    int4 a = make_int4(0, 0, 0, 0);
    uint4 x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = make_uint4(0u, 0u, 0u, 0u);
    const u32 xb = ent >> 13;
    const u32 nq = !valid ? 0u : (xb < P.fextClass[0] ? 3u : (xb < P.fextClass[1] ? 4u : (xb < P.fextClass[2] ? 6u : 8u)));   // quads of the block that hold records
    if (valid) {
      a = alnT[li];
      const uint4 *E = P.fext + (size_t) xb * 8;
      x[0] = E[0]; x[1] = E[1]; x[2] = E[2];
      if (nq > 3) x[3] = E[3];
      if (nq > 4) { x[4] = E[4]; x[5] = E[5]; }
      if (nq > 6) { x[6] = E[6]; x[7] = E[7]; }
    }
    const u32 nh = ((u32) a.z >> 1) & MM_ALN_NH_ESCAPE;   // (not the escape value, and within the MM_BATCH_SINGLE_HIT promise: checked in the main loop)
    const int bs = (int) (((u32) a.x >> sshift) << sshift);
    const int rec_strand = a.z & 1;
    const u32 okbit = 24u + (u32) ((P.strandedness == MM_STRANDED_U) ? 1 : ((P.strandedness == MM_STRANDED_F) ? rec_strand : (rec_strand ^ 1)));
    const int sr = a.x - bs, er = a.y - bs;
    u64 hit = 0;
#define MM_FAST_TEST(R)                                                                                           \
  {                                                                                                              \
    const u32 r_ = (R);                                                                                          \
    hit |= (u64) span_pass<OVL>(P, (int) (r_ & soMask), (int) ((r_ & 0xFFFFFFu) >> (sshift + 1)), (r_ >> okbit) & 1u, sr, er) << (r_ >> 26); \
  }
    MM_FAST_TEST(x[0].y) MM_FAST_TEST(x[0].z) MM_FAST_TEST(x[0].w)
#pragma unroll
    for (int k = 1; k < 3; ++k) { MM_FAST_TEST(x[k].x) MM_FAST_TEST(x[k].y) MM_FAST_TEST(x[k].z) MM_FAST_TEST(x[k].w) }
    if (__ballot(nq > 3)) {                                // (wave-uniform; quads that were not requested are zero: they pass nothing)
      MM_FAST_TEST(x[3].x) MM_FAST_TEST(x[3].y) MM_FAST_TEST(x[3].z) MM_FAST_TEST(x[3].w)
      if (__ballot(nq > 4)) {
#pragma unroll
        for (int k = 4; k < 8; ++k) { MM_FAST_TEST(x[k].x) MM_FAST_TEST(x[k].y) MM_FAST_TEST(x[k].z) MM_FAST_TEST(x[k].w) }
      }
    }
#undef MM_FAST_TEST
    const u64 cand = hit & sLine[(__ffsll((long long) hit) - 1) & 63];
    const bool settled = (cand & (cand - 1)) == 0 || (OVL == MM_OVL_INCLUSION && !(cand & updown));
    const u64 mask = settled ? cand : 0ull;
    const bool defer = valid && !settled;
    {
      const u64 dm = __ballot(defer);
      if (dm) {
        if (defer) dl[dn + (u32) __popcll(dm & ((1ull << lane) - 1ull))] = (u32) (first + li);
        dn += (u32) __popcll(dm);
      }
    }
    u32 key, pay;
    tuple_tail(grouped_strategy, ratio, false, useHist ? HIST_LDS : HIST_NONE, T, sHist, hist, nblk, G, qlo, lane, first + li, a, nh, valid && !defer, mask, A, key, pay);
    if (haveKey && valid && !defer) { gkeyT[li] = key; gpayT[li] = ratio ? nh : (pay | (min(nh, 0xFFFFu) << 16)); }
  }
  // the four waves' regions -> one list for the whole batch (one atomic per workgroup), so that k_probe_list runs on full wavefronts
  __shared__ u32 wcnt[4], wbase[4];
  if (lane == 0) wcnt[w] = dn;
  __syncthreads();                                         // (also: the waves' writes to their regions are done)
  if (t == 0) {
    const u32 tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    u32 b = tot ? atomicAdd(dcnt, tot) : 0u;
    for (u32 q = 0; q < 4; ++q) { wbase[q] = b; b += wcnt[q]; }
  }
  __syncthreads();
  for (u32 j = lane; j < dn; j += 64) dlist2[wbase[w] + j] = __builtin_nontemporal_load(dl + j);
  if (useHist) for (u32 d = t; d < RS_BINS; d += 256) if (sHist[d]) hist[(size_t) d * nblk + blockIdx.x] = sHist[d];
  probe_finish(grouped_strategy, ratio, haveKey, A, sstat, sq, T, G, stats, info);
}

// The tuples k_probe_fast passed on (dlist[0 .. *dcnt)), through k_probe's walk (8-byte slots, full index, second selection pass) and
// the same tail.  The partition histogram of the batch (hist != NULL: k_probe_fast has stored its tiles' counts, the host zeroed the
// rest) gets these tuples' digits by atomics.  The list is a few per cent of a batch, heavy tuples mostly (bins with many records), and a
// chain of dependent loads each: what matters is that all of them are in flight at once - one tuple per lane, as many workgroups as the
// list needs (the others leave at once).
template <int OVL>
__global__ __launch_bounds__(256) MM_PROBE_ATTR void k_probe_list(const ProbeParams P, const int4 *__restrict__ aln, const uint2 *__restrict__ big, u32 nbig,
                                                                  int single, u32 qlo, u32 *__restrict__ gkey, u32 *__restrict__ gpay, u32 *__restrict__ info,
                                                                  GlobalTable G, u64 *__restrict__ stats, u32 *__restrict__ hist, u32 nblk,
                                                                  const u32 *__restrict__ dlist, const u32 *__restrict__ dcnt, u32 *__restrict__ dcnt_next,
                                                                  u64 *__restrict__ walkstat) {
  __shared__ LdsTable T;
  __shared__ u32 sstat[8];
  __shared__ u32 sq[3];
  __shared__ u64 sLine[64];
  __shared__ uint2 sChr[MM_LDS_CHR];
  const u32 total = *dcnt;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *dcnt_next = 0;                                        // the next batch's counter (nothing touches it during this launch)
    if (total) atomicAdd(&walkstat[2], (u64) total);       // profiling aid: tuples passed on (mm_get_probe_stats)
  }
  if ((size_t) blockIdx.x * 256 >= total) return;          // (uniform)
  if (threadIdx.x < 64) sLine[threadIdx.x] = P.lineMask[threadIdx.x];
  for (u32 c = threadIdx.x; c < min(P.nchr, (u32) MM_LDS_CHR); c += blockDim.x) sChr[c] = P.chrSlot[c];
  const uint2 *chrSlot = P.nchr <= (u32) MM_LDS_CHR ? sChr : P.chrSlot;
  lds_table_init(T);
  if (threadIdx.x < 8) sstat[threadIdx.x] = 0;
  if (threadIdx.x < 3) sq[threadIdx.x] = 0;
  __syncthreads();
  const u32 lane = threadIdx.x & 63;
  const bool grouped_strategy = P.strategy == MM_STRAT_DEFAULT, ratio = P.strategy == MM_STRAT_RATIO, uniqueOnly = P.strategy == MM_STRAT_UNIQUE;
  const bool haveKey = gkey != nullptr;
  ProbeAcc A;
  u32 walked = 0;
  for (size_t base = (size_t) blockIdx.x * 256; base < total; base += (size_t) gridDim.x * 256) {
    const size_t j = base + threadIdx.x;
    const bool valid = j < total;
    size_t i = 0;
    int4 a = make_int4(0, 0, 0, 0);
    u32 nh = 1;
    if (valid) {
      i = dlist[j];
      a = aln[i];
      nh = tuple_nh((u32) a.z, i, big, nbig, G.err);
      if (single && nh > 1) { atomicCAS(G.err, 0u, (u32) ERR_SINGLE_HIT); nh = 1; }
    }
    const bool processed = valid && !(uniqueOnly && nh != 1);
    u64 mask = 0;
    if (processed) mask = probe_tuple<OVL, true>(P, true, chrSlot, sLine, a, walked);
    u32 key, pay;
    tuple_tail(grouped_strategy, ratio, false, hist ? HIST_GLOBAL : HIST_NONE, T, (u32 *) nullptr, hist, nblk, G, qlo, lane, i, a, nh, processed, mask, A, key, pay);
    if (valid && haveKey) { gkey[i] = key; gpay[i] = ratio ? nh : (pay | (min(nh, 0xFFFFu) << 16)); }
  }
  probe_finish(grouped_strategy, ratio, haveKey, A, sstat, sq, T, G, stats, info);
}

// ----------------------------------------------------------------------------- grouping pipeline: tiles of (key, payload) pairs
template <bool FIRST>
__device__ __forceinline__ bool rs_load(const u32 *__restrict__ kin, const u32 *__restrict__ vin, const u32 *__restrict__ info, size_t i, size_t n,
                                        int rebase, u32 &k, u32 &v) {
  if (FIRST) {
    if (i >= n) return false;
    const u32 g = kin[i];
    if (g == NO_KEY) return false;
    k = rebase ? g - ~info[INFO_QINV] : g;
    v = (rebase && info[INFO_BIGNH]) ? (u32) i : vin[i];
    return true;
  }
  if (i >= info[INFO_M]) return false;
  if (rebase) { const u64 r = ((const u64 *) kin)[i]; k = (u32) (r >> 32); v = (u32) r; }   // (leaf layout: 8-byte pairs between the passes)
  else { k = kin[i]; v = vin[i]; }
  return true;
}

// Rows of `hist` (nblk entries each): 0..RS_BINS-1 the digit counts per tile (a pass writes the first 2^width of them); for the
//   first pass of the name sort also RS_ROW_COUNT the tile's number of pairs, RS_ROW_FIRST / RS_ROW_LAST its first / last key
//   (NO_KEY: none) - see the order check below.
#define RS_ROW_COUNT RS_BINS
#define RS_ROW_FIRST (RS_BINS + 1)
#define RS_ROW_LAST (RS_BINS + 2)
#define RS_ROWS (RS_BINS + 3)

// Order check (first pass of the name sort only): an aligner writes the hits of one read one after the other, and read-name
// ids are handed out in order of first appearance, so the qids of such a file are already non-decreasing and the "sort"
// only has to compact.  Every record is compared with the nearest earlier record of the pipeline: inside a wave by ballot,
// between the 64 wave-segments of a tile through LDS, between tiles by k_rs_scan.  Any inversion sets INFO_UNSORTED
// and the batch takes the general path.
template <bool FIRST>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const u32 *__restrict__ keys, const u32 *__restrict__ vals, u32 *__restrict__ info,
                                                        size_t n, int rebase, int shift, int fixed_bits, u32 *__restrict__ hist, u32 nblk, int only_if_wide) {
  __shared__ u32 h[RS_BINS];
  if (FIRST && only_if_wide && !info[INFO_WIDEKEY]) return;      // k_probe's own count of the slots stands (-y ratio, one pass)
  __shared__ u32 segF[RS_ROUNDS * RS_NW], segL[RS_ROUNDS * RS_NW];
  __shared__ u32 tileCnt, tileBad;
  if (!FIRST && (shift >> 3) >= batch_passes(info, fixed_bits)) return;   // this pass is not needed for this batch's key range
  const SortLayout L = batch_layout(info, fixed_bits);
  const bool order = FIRST && rebase;
  // (the check stops at the first inversion: this wave's, or one another tile has already published)
  bool chk = order && ((volatile u32 *) info)[INFO_UNSORTED] == 0;
  const u32 lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  h[threadIdx.x] = 0;                                    // (RS_THREADS == RS_BINS)
  if (threadIdx.x == 0) { tileCnt = 0; tileBad = chk ? 0u : 1u; }
  __syncthreads();
  const size_t lo = (size_t) blockIdx.x * RS_TILE;
  u32 wcnt = 0;
  u32 kk[RS_ROUNDS], okm = 0;        // all of the thread's keys are requested before the first is used
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    u32 v;
    kk[r] = 0;
    if (rs_load<FIRST>(keys, vals, info, lo + (size_t) r * RS_THREADS + threadIdx.x, n, rebase, kk[r], v)) okm |= 1u << r;
  }
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const u32 k = kk[r];
    const bool ok = (okm >> r) & 1u;
    // (leaf layout: the digit is the bucket, 4096 consecutive names - the hits of a read lie close together in most inputs, so a tile
    // sees a handful of buckets: same-address LDS atomics unless the wave's equal digits are counted together)
    wave_hist_add<0>(h, batch_digit(k, L, shift), ok, fixed_bits != 0 || L.leaf);
    if (chk) {   // wave-uniform
      const u64 vm = __ballot(ok);
      const u64 below = vm & ((1ull << lane) - 1ull);
      const u32 pk = __shfl(k, below ? 63 - __clzll((long long) below) : 0);
      const u32 kf = __shfl(k, vm ? __ffsll((long long) vm) - 1 : 0), kl = __shfl(k, vm ? 63 - __clzll((long long) vm) : 0);
      if (lane == 0) { segF[r * RS_NW + w] = vm ? kf : NO_KEY; segL[r * RS_NW + w] = vm ? kl : NO_KEY; wcnt += (u32) __popcll(vm); }
      if (__ballot(ok && below && pk > k)) { chk = false; if (lane == 0) tileBad = 1; }
    }
  }
  if (chk && lane == 0) atomicAdd(&tileCnt, wcnt);
  __syncthreads();
  if (threadIdx.x < (1u << L.width)) hist[(size_t) threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
  if (order && w == 0 && tileBad == 0) {   // the tile's segments in file order (segment = round * waves + wave), 64 at a time
    u32 carryL = NO_KEY, tfirst = NO_KEY, tlast = NO_KEY;
    bool sbad = false;
    for (u32 half = 0; half < RS_ROUNDS * RS_NW / 64; ++half) {
      const u32 F = segF[half * 64 + lane], Lk = segL[half * 64 + lane];
      const bool has = F != NO_KEY;
      const u64 vm = __ballot(has);
      const u64 below = vm & ((1ull << lane) - 1ull);
      const u32 pl = __shfl(Lk, below ? 63 - __clzll((long long) below) : 0);
      const u32 prevL = below ? pl : carryL;
      if (__ballot(has && prevL != NO_KEY && prevL > F)) sbad = true;
      if (vm) {
        const u32 tf = __shfl(F, __ffsll((long long) vm) - 1), tl = __shfl(Lk, 63 - __clzll((long long) vm));
        if (tfirst == NO_KEY) tfirst = tf;
        tlast = tl; carryL = tl;
      }
    }
    if (lane == 0) {
      hist[(size_t) RS_ROW_COUNT * nblk + blockIdx.x] = tileCnt;
      hist[(size_t) RS_ROW_FIRST * nblk + blockIdx.x] = tfirst;
      hist[(size_t) RS_ROW_LAST * nblk + blockIdx.x] = tlast;
      if (sbad) tileBad = 1;
    }
  }
  // one flag for the whole batch: on unsorted input nearly every tile finds an inversion, so look before the atomic
  if (order && threadIdx.x == 0 && tileBad && ((volatile u32 *) info)[INFO_UNSORTED] == 0) atomicOr(&info[INFO_UNSORTED], 1u);
}

// One WAVE per digit row (four rows to a workgroup): exclusive scan of its row of per-tile counts - every lane takes 32 consecutive
// tiles, all its loads in flight together, then a local scan and one wave scan; no workgroup barrier on the way.  The workgroup that
// finishes last then turns the row totals into the digits' global bases (and, for the first pass of the name sort, finishes the order
// check between tiles and publishes the number of pairs) - one launch instead of two on a path made of short kernels.  (Until round 2
// a workgroup per row: 513 arrivals on the one "done" counter, 33 us for the leaf layout's 513 rows; now 129.)
__global__ __launch_bounds__(256) void k_rs_scan(u32 *__restrict__ hist, u32 nblk, u32 nrows, u32 *__restrict__ rowTotal, u32 *__restrict__ done,
                                                 u32 *__restrict__ info, int shift, int fixed_bits, int order, u64 *__restrict__ walkstat) {
  __shared__ u32 wsum[4];
  __shared__ u32 isLast;
  if (shift > 0 && (shift >> 3) >= batch_passes(info, fixed_bits)) return;
  const u32 lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const u32 rowId = blockIdx.x * 4 + w;
  if (rowId < nrows) {                                    // (wave-uniform)
    u32 *row = hist + (size_t) rowId * nblk;
    // digit rows this pass did not write (its digits are narrower than RS_BINS) total zero
    const bool unused = rowId < RS_BINS && rowId >= (1u << batch_layout(info, fixed_bits).width);
    u32 carry = 0;
    for (u32 base = 0; base < nblk && !unused; base += 64 * 32) {
      const u32 i0 = base + lane * 32;
      u32 v[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) v[k] = (i0 + k < nblk) ? row[i0 + k] : 0u;
      u32 tot = 0;
#pragma unroll
      for (int k = 0; k < 32; ++k) { const u32 t = v[k]; v[k] = tot; tot += t; }
      u32 x = tot;
      for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(x, off);
        if ((int) lane >= off) x += y;
      }
      const u32 pre = carry + x - tot;
#pragma unroll
      for (int k = 0; k < 32; ++k) if (i0 + k < nblk) row[i0 + k] = pre + v[k];
      carry += (u32) __builtin_amdgcn_readlane((int) x, 63);
    }
    if (lane == 0) rowTotal[rowId] = carry;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    isLast = (atomicAdd(done, 1u) == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!isLast) return;
  __threadfence();
  if (threadIdx.x == 0) *done = 0;   // ready for the next pass
  if (order && ((volatile u32 *) info)[INFO_UNSORTED] == 0) {   // between tiles: a tile's first key may not be below the largest last key before it
    const u32 *first = hist + (size_t) RS_ROW_FIRST * nblk, *last = hist + (size_t) RS_ROW_LAST * nblk;
    u32 cmax = 0, bad = 0;
    for (u32 base = 0; base < nblk; base += 256) {
      const u32 i = base + threadIdx.x;
      const u32 F = i < nblk ? first[i] : NO_KEY;
      const u32 v = F != NO_KEY ? last[i] : 0u;
      u32 x = v;
      for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(x, off);
        if ((int) lane >= off) x = max(x, y);
      }
      if (lane == 63) wsum[w] = x;
      __syncthreads();
      u32 before = cmax, all = cmax;
      for (u32 q = 0; q < 4; ++q) { if (q < w) before = max(before, wsum[q]); all = max(all, wsum[q]); }
      const u32 prev = __shfl_up(x, 1);
      if (lane > 0) before = max(before, prev);
      if (F != NO_KEY && F < before) bad = 1;
      cmax = all;
      __syncthreads();
    }
    if (bad) atomicOr(&info[INFO_UNSORTED], 1u);
  }
  {  // exclusive scan of the RS_BINS row totals, two per thread
    const u32 v0 = ((volatile u32 *) rowTotal)[2 * threadIdx.x], v1 = ((volatile u32 *) rowTotal)[2 * threadIdx.x + 1];
    const u32 v = v0 + v1;
    u32 x = v;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 y = __shfl_up(x, off);
      if ((int) lane >= off) x += y;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    u32 woff = 0;
    for (u32 k = 0; k < w; ++k) woff += wsum[k];
    rowTotal[2 * threadIdx.x] = woff + x - v;
    rowTotal[2 * threadIdx.x + 1] = woff + x - v + v0;
    if (threadIdx.x == 255) {
      const u32 acc = woff + x;
      if (shift == 0) info[INFO_M] = acc;   // the first pass compacts: this is the number of pairs in the pipeline
      if (order && acc && !((volatile u32 *) info)[INFO_UNSORTED]) atomicAdd(&walkstat[1], 1ull);   // (reported by mm_get_walk_stats)
    }
  }
}

// One stable pass for one tile of RS_TILE pairs (8192: 512 threads, 74 KB of LDS, two workgroups per CU;
// measured against 4096-pair tiles of 256 threads: sort -3 %, ordered compaction -7 % - longer digit runs, half the tiles).
//   A. every wave owns a contiguous eighth of the tile, keeps its 16x64 pairs in registers and counts digits
//      per wave in LDS (16-bit counters, two to a word);
//   B. per-digit exclusive offsets (tile-local and global) from those counts;
//   C. each wave ranks its pairs sub-round by sub-round (one ballot per digit bit gives the lanes holding the same
//      digit) against its private running offsets - no workgroup barrier inside the loop - and drops the pairs
//      at their tile-local sorted position in LDS;
//   D. the sorted tile is copied out: consecutive lanes write consecutive addresses inside each digit run, so
//      the scatter reaches memory as 64-B-and-longer segments instead of isolated 4-B stores.  The partition pass of the
//      "leaf" layout writes each pair as one 8-byte word (key << 32 | payload): k_leaf streams those.
template <bool FIRST>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const u32 *__restrict__ kin, const u32 *__restrict__ vin, u32 *__restrict__ kout,
                                                           u32 *__restrict__ vout, const u32 *__restrict__ info, size_t n, int rebase, int shift,
                                                           int fixed_bits, const u32 *__restrict__ hist, const u32 *__restrict__ base, u32 nblk, int forced_leaf,
                                                           u32 *__restrict__ kfin, u32 *__restrict__ vfin) {
  __shared__ u32 lkey[RS_TILE];
  __shared__ u32 lval[RS_TILE];
  __shared__ __attribute__((aligned(16))) unsigned short cntw[RS_NW][RS_BINS];   // A: digit counts per wave; C: running tile-local offset per wave (< 8192)
  __shared__ u32 goff[RS_BINS];      // global position of digit d's first pair of this tile, minus its tile-local offset
  __shared__ u32 wsum[RS_NW];
  __shared__ u32 tcountS;
  if (!FIRST && (shift >> 3) >= batch_passes(info, fixed_bits)) return;
  const SortLayout L = batch_layout(info, fixed_bits);
  // the host took the one-pass leaf layout for granted (k_probe counted buckets) but a read-name id lies outside the range it was
  // promised: the histogram does not fit this layout - write nothing; k_leaf reports the batch
  if (forced_leaf && L.passes != 1) return;
  // -y ratio, one pass (batch_layout): this pass's output is the sorted batch - it goes where the second pass would have left it
  if (FIRST && fixed_bits == 16 && L.passes == 1) { kout = kfin; vout = vfin; }
  const size_t lo = (size_t) blockIdx.x * RS_TILE;
  const size_t lim = FIRST ? n : (size_t) info[INFO_M];
  if (lo >= lim) return;
  // name-grouped input (see k_rs_hist): one "digit" for everything, i.e. a stable compaction in file order
  const bool presorted = FIRST && rebase && !info[INFO_UNSORTED];
  const u32 dmask = presorted ? 0u : (RS_BINS - 1u);   // (presorted: every pair in "digit" 0)
  const bool pairs64 = rebase && !presorted;                      // the leaf layout's partition passes write 8-byte pairs
  const u32 t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (u32 q = 0; q < RS_NW; ++q) cntw[q][t] = 0;
  __syncthreads();
  u32 kreg[RS_ROUNDS], vreg[RS_ROUNDS];
  u32 vmask = 0;                     // which of this lane's 16 pairs exist (the first pass filters on the fly)
  const u32 wbase = w * (RS_TILE / RS_NW);
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const u32 li = wbase + (u32) r * 64u + lane;
    kreg[r] = 0; vreg[r] = 0;
    const bool ok = rs_load<FIRST>(kin, vin, info, lo + li, n, rebase, kreg[r], vreg[r]);
    if (ok) vmask |= 1u << r;
    if (presorted) {   // (wave-uniform) all pairs count for digit 0: one add per wave instead of 64 colliding atomics
      const u32 c = (u32) __popcll(__ballot(ok));
      if (lane == 0) cntw[w][0] = (unsigned short) (cntw[w][0] + c);
    } else wave_hist_add<1>((u32 *) cntw[w], batch_digit(kreg[r], L, shift), ok, fixed_bits != 0 || L.leaf);
  }
  __syncthreads();
  if (presorted) {   // plain ordered compaction: the pairs go straight from the registers to their place, in file order
    u32 off = hist[(size_t) RS_ROW_COUNT * nblk + blockIdx.x];
    for (u32 q = 0; q < w; ++q) off += cntw[q][0];
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
      const bool valid = (vmask >> r) & 1u;
      const u64 bal = __ballot(valid);
      if (valid) { const u32 pos = off + (u32) __popcll(bal & lt); kout[pos] = kreg[r]; vout[pos] = vreg[r]; }
      off += (u32) __popcll(bal);
    }
    return;
  }
  {  // B: thread t owns digit t
    u32 tot = 0;
    for (u32 q = 0; q < RS_NW; ++q) tot += cntw[q][t];
    u32 x = tot;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 y = __shfl_up(x, off);
      if ((int) lane >= off) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    u32 loff = x - tot;
    for (u32 q = 0; q < w; ++q) loff += wsum[q];
    if (t == RS_BINS - 1) tcountS = loff + tot;                  // number of pairs this tile holds
    // (a digit this tile does not hold needs no offset: its row is another 4-byte gather)
    goff[t] = base[t] + ((tot && t < (1u << L.width)) ? hist[(size_t) t * nblk + blockIdx.x] : 0u) - loff;
    u32 run = loff;
    for (u32 q = 0; q < RS_NW; ++q) { const u32 c = cntw[q][t]; cntw[q][t] = (unsigned short) run; run += c; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {  // C: wave-private, LDS operations of one wave execute in program order
    const bool valid = (vmask >> r) & 1u;
    const u32 d = batch_digit(kreg[r], L, shift) & dmask;
    u64 peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < LEAF_DIGIT_BITS; ++b) {   // (skipping these when the wave's digits are all equal was measured: no gain)
      const u64 bb = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bb : ~bb;
    }
    const u32 rank = (u32) __popcll(peers & ((1ull << lane) - 1));
    u32 pos = 0;
    if (valid) pos = (u32) cntw[w][d] + rank;
    if (valid && rank == 0) cntw[w][d] = (unsigned short) (pos + (u32) __popcll(peers));
    if (valid) { lkey[pos] = kreg[r]; lval[pos] = vreg[r]; }
  }
  __syncthreads();
  const u32 tcount = tcountS;
  if (pairs64) {
    u64 *__restrict__ rout = (u64 *) kout;      // (kout and vout are the two halves of one allocation)
    for (u32 i = t; i < tcount; i += RS_THREADS) {
      const u32 k = lkey[i];
      rout[goff[batch_digit(k, L, shift)] + i] = ((u64) k << 32) | lval[i];
    }
    return;
  }
  for (u32 i = t; i < tcount; i += RS_THREADS) {  // D
    const u32 k = lkey[i];
    const u32 g = goff[batch_digit(k, L, shift) & dmask] + i;
    kout[g] = k; vout[g] = lval[i];
  }
}

// ----------------------------------------------------------------------------- multi-hit countdown / union (-y default, NH > 1)
// Input: the batch's multi-hit records sorted by read-name id, file order kept inside a name.
// One lane per name walks that name's records exactly as addCount does (mmannot.cpp:1669-1701):
// the first record of a group sets remaining = NH-1 (mmannot.cpp:1673), later ones count down and
// union their element sets; at 0 the group is closed, counted (mmannot.cpp:1688) and erased, so a
// further record of the same name opens a new group.  State (remaining, union) is carried across
// batches in grpRem/grpU, the dense replacement of readCounts (mmannot.cpp:1656).
// (compiled for 8 waves per SIMD like k_probe: the compiler then spends the whole 64-VGPR budget on keeping the next round's
// gathers in flight - measured 0.20 -> 0.16 ms per 2^24 records against its default schedule of 54 VGPRs)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_chunk_default(const u32 *__restrict__ kA, const u32 *__restrict__ vA, const u32 *__restrict__ kB,
                                                       const u32 *__restrict__ vB, const u32 *__restrict__ info,
                                                       const int4 *__restrict__ aln, const uint2 *__restrict__ big, u32 nbig, const u64 *__restrict__ hm, u32 *__restrict__ grpRem,
                                                       u64 *__restrict__ grpU, u64 qcap, GlobalTable G, u64 *__restrict__ stats, int nelem) {
  __shared__ LdsTable T;
  __shared__ u32 sOpened, sRescued;
  __shared__ u32 sKey[256 + 64], sNh[256 + 64];   // this round's records and the 64 behind them
  __shared__ u64 sM[256 + 64];
  if (batch_is_leaf(info)) return;                       // this batch is k_leaf's
  // a read-name id at or beyond the caller's qid_bound would index the per-name state out of bounds: refuse the batch
  if (info[INFO_M] != 0 && (u64) info[INFO_QMAX] >= qcap) { if (threadIdx.x == 0) atomicCAS(G.err, 0u, (u32) ERR_QID_BOUND); return; }
  lds_table_init(T);
  if (threadIdx.x == 0) { sOpened = 0; sRescued = 0; }
  __syncthreads();
  const u32 M = info[INFO_M];
  const u32 qmin = ~info[INFO_QINV];
  const bool by_index = info[INFO_BIGNH] != 0;
  const int passes = batch_passes(info, 0);                    // the sort ping-pongs A -> B -> A ...
  const u32 *__restrict__ skeys = (passes & 1) ? kB : kA;
  const u32 *__restrict__ svals = (passes & 1) ? vB : vA;
  const u32 lane = threadIdx.x & 63;
  u32 opened = 0, rescued = 0;
  // payload of a record of the sorted stream -> (element set, NH)
  auto decode = [&](u32 pv, u64 &m, u32 &nhv) __attribute__((always_inline)) {
    if (by_index) { m = hm[pv]; nhv = tuple_nh((u32) aln[pv].z, pv, big, nbig, G.err); }
    else { m = payload_set(G, pv & 0xFFFFu); nhv = pv >> 16; }
  };
  auto fetch = [&](size_t k, u64 &m, u32 &nhv) __attribute__((always_inline)) { decode(svals[k], m, nhv); };
  auto close_group = [&](u64 U) __attribute__((always_inline)) {
    if (U) {
      count_set(T, G, U);
      if ((U & (U - 1)) == 0) ++rescued;
    }
  };
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  const u32 w = threadIdx.x >> 6;
  // The kernel is a chain of dependent gathers (key -> per-name state; payload -> element set): each round requests the
  // next round's records - its own, and the first wave also the 64 behind the workgroup's 256 - before it works on its own.
  // (A name has one head record per batch, so its state is read and written by exactly one lane: reading it early is safe.)
  u32 nKey = 0, nPrev = 0, nNext = 0, nNh = 0, nRem = 0, nHk = NO_KEY, nHn = 0;
  u64 nM = 0, nU = 0, nHm = 0;
  auto preload = [&](size_t b) __attribute__((always_inline)) {
    const size_t j = b + w * 64 + (63 - lane);
    nKey = 0; nNh = 0; nM = 0; nPrev = 0; nNext = 0;
    if (j < M) {
      nKey = skeys[j];
      nPrev = j ? skeys[j - 1] : ~nKey;
      nNext = (j + 1 < M) ? skeys[j + 1] : ~nKey;
      if (nPrev != nKey) { nRem = grpRem[nKey + qmin]; nU = grpU[nKey + qmin]; }
      fetch(j, nM, nNh);
    }
    if (threadIdx.x < 64) {
      const size_t jh = b + 256 + threadIdx.x;
      nHk = NO_KEY; nHn = 0; nHm = 0;
      if (jh < M) { nHk = skeys[jh]; fetch(jh, nHm, nHn); }
    }
  };
  preload((size_t) blockIdx.x * blockDim.x);
  for (size_t base = (size_t) blockIdx.x * blockDim.x; base < M; base += stride) {   // (workgroup-uniform trip count)
    // Every lane loads its own record and both neighbours' keys; a name's records are consecutive.  Lanes take the
    // wave's 64 records in REVERSE order (lane 63 = the earliest record): the per-name reduction then runs towards
    // higher lanes, which is the direction the DPP row-shift / row-broadcast network moves data in.
    // The workgroup's 256 records plus the 64 that follow them are also staged in LDS: a name that runs past the end
    // of its wave is completed from there instead of through another round of dependent global loads.
    const u32 pos = w * 64 + (63 - lane);                 // position inside the workgroup's 256 records
    const size_t j = base + pos;
    const bool valid = j < M;
    // this round's record was requested during the previous round (see preload)
    const u32 key = nKey, nhv = nNh;
    const u64 m = nM;
    const bool head = valid && nPrev != key, last = !valid || nNext != key;
    u32 rem = head ? nRem : 0u;
    u64 U = head ? nU : 0ull;
    const u32 hk = nHk, hn = nHn;
    const u64 hmk = nHm;
    preload(base + stride);
    sKey[pos] = valid ? key : NO_KEY; sM[pos] = m; sNh[pos] = nhv;
    if (threadIdx.x < 64) { sKey[256 + threadIdx.x] = hk; sM[256 + threadIdx.x] = hmk; sNh[256 + threadIdx.x] = hn; }
    __syncthreads();
    // Segmented inclusive scan towards the head lane (a name's last record starts a segment): union of the sets,
    // record count, min / max NH.  Kogge-Stone inside the rows of 16 lanes, then the two row broadcasts - all DPP moves
    // on the vector ALU.  (The same reduction written with __shfl_down costs ~36 LDS-crossbar operations per round,
    // and SQ_WAIT_INST_LDS showed this kernel waiting on exactly those.)
    u32 ulo = (u32) m, uhi = (u32) (m >> 32), tcnt = valid ? 1u : 0u, tmin = valid ? nhv : 0xFFFFFFFFu, tmax = valid ? nhv : 0u;
    u32 f = (last || lane == 0) ? 1u : 0u;
#define MM_DPP(OLD, SRC, CTRL, RMASK) ((u32) __builtin_amdgcn_update_dpp((int) (OLD), (int) (SRC), (CTRL), (RMASK), 0xF, false))
#define MM_SEG_STEP(CTRL, RMASK)                                                                               \
    {                                                                                                          \
      const u32 sf = MM_DPP(f, f, CTRL, RMASK); /* a lane without a source keeps its own flag and gets identities */ \
      const u32 a0 = MM_DPP(0u, ulo, CTRL, RMASK), a1 = MM_DPP(0u, uhi, CTRL, RMASK), ac = MM_DPP(0u, tcnt, CTRL, RMASK); \
      const u32 an = MM_DPP(0xFFFFFFFFu, tmin, CTRL, RMASK), ax = MM_DPP(0u, tmax, CTRL, RMASK);               \
      if (!f) { ulo |= a0; uhi |= a1; tcnt += ac; tmin = min(tmin, an); tmax = max(tmax, ax); f = sf; }        \
    }
    MM_SEG_STEP(0x111, 0xF) MM_SEG_STEP(0x112, 0xF) MM_SEG_STEP(0x114, 0xF) MM_SEG_STEP(0x118, 0xF)   // row_shr:1,2,4,8
    MM_SEG_STEP(0x142, 0xA)                                                                            // row_bcast:15 -> rows 1, 3
    MM_SEG_STEP(0x143, 0xC)                                                                            // row_bcast:31 -> rows 2, 3
#undef MM_SEG_STEP
    u64 tU = ((u64) uhi << 32) | ulo;
    // Does the wave's last name (at lane 0) continue behind the wave?  Then the whole wave looks at the following
    // records together and the matching prefix joins that name's head lane - if its head is in this wave at all.
    const u64 heads = __ballot(head);
    const bool spill = (__ballot(valid && !last) & 1ull) != 0 && heads != 0;   // wave-uniform
    if (spill) {
      const int hl = __ffsll((long long) heads) - 1;                      // head lane of the name that reaches lane 0
      const u32 skey = (u32) __builtin_amdgcn_readlane((int) key, 0);
      u64 xU = 0;
      u32 xcnt = 0, xmin = 0xFFFFFFFFu, xmax = 0;
      size_t nb = base + (size_t) (w + 1) * 64;
      for (int step = 0;; ++step) {
        u32 x0 = 0, x1 = 0, xn = 0xFFFFFFFFu, xx = 0;
        bool match;
        if (step == 0) {   // the 64 records behind this wave: from LDS (NO_KEY marks the end of the batch)
          const u32 li = (w + 1) * 64 + lane;
          match = sKey[li] == skey;
          if (match) { const u64 mv = sM[li]; x0 = (u32) mv; x1 = (u32) (mv >> 32); xn = xx = sNh[li]; }
        } else {
          const size_t k = nb + lane;
          match = k < M && skeys[k] == skey;
          if (match) { u64 mv; u32 n2; fetch(k, mv, n2); x0 = (u32) mv; x1 = (u32) (mv >> 32); xn = xx = n2; }
        }
        const u64 mm = __ballot(match);
        const u32 fl = (~mm == 0ull) ? 64u : (u32) (__ffsll((long long) ~mm) - 1);   // length of the matching prefix
        if (lane >= fl) { x0 = 0; x1 = 0; xn = 0xFFFFFFFFu; xx = 0; }                  // (a later record of the same key cannot exist, but stay exact)
#define MM_RED_STEP(CTRL, RMASK)                                                                     \
        { x0 |= MM_DPP(0u, x0, CTRL, RMASK); x1 |= MM_DPP(0u, x1, CTRL, RMASK);                      \
          xn = min(xn, MM_DPP(0xFFFFFFFFu, xn, CTRL, RMASK)); xx = max(xx, MM_DPP(0u, xx, CTRL, RMASK)); }
        MM_RED_STEP(0x111, 0xF) MM_RED_STEP(0x112, 0xF) MM_RED_STEP(0x114, 0xF) MM_RED_STEP(0x118, 0xF)
        MM_RED_STEP(0x142, 0xA) MM_RED_STEP(0x143, 0xC)                                              // lane 63 now holds the wave's reduction
#undef MM_RED_STEP
        xU |= ((u64) (u32) __builtin_amdgcn_readlane((int) x1, 63) << 32) | (u32) __builtin_amdgcn_readlane((int) x0, 63);
        xmin = min(xmin, (u32) __builtin_amdgcn_readlane((int) xn, 63)); xmax = max(xmax, (u32) __builtin_amdgcn_readlane((int) xx, 63));
        xcnt += fl;
        if (fl < 64) break;
        nb += 64;
      }
      if ((int) lane == hl) { tU |= xU; tcnt += xcnt; tmin = min(tmin, xmin); tmax = max(tmax, xmax); }
    }
#undef MM_DPP
    __syncthreads();   // LDS is rewritten by the next round
    if (!head) continue;
    const u32 q = key + qmin;
    if (rem > 0 && tcnt <= rem) {
      // all records fall into the pending group of this name (their own NH is irrelevant, mmannot.cpp:1681-1684)
      U |= tU; rem -= tcnt;
      if (rem == 0) { close_group(U); U = 0; }
    } else if (rem == 0 && tmin == tmax && tcnt <= tmin) {
      // one new group, opened by the first record: remaining = NH-1 (mmannot.cpp:1673), order of the others is irrelevant
      ++opened; U = tU; rem = tmin - tcnt;
      if (rem == 0) { close_group(U); U = 0; }
    } else {
      // order matters (a group closes inside the run: over-complete names, mixed NH): replay in file order
      for (size_t r = j; r < j + tcnt; ++r) {
        u64 m2; u32 n2;
        fetch(r, m2, n2);
        if (rem == 0) { rem = n2 - 1; U = m2; ++opened; }              // mmannot.cpp:1672-1677
        else {                                                          // mmannot.cpp:1678-1701
          --rem; U |= m2;
          if (rem == 0) { close_group(U); U = 0; }
        }
      }
    }
    grpRem[q] = rem; grpU[q] = U;
  }
  if (opened) atomicAdd(&sOpened, opened);
  if (rescued) atomicAdd(&sRescued, rescued);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sOpened) atomicAdd(&stats[MM_STAT_READS], (u64) sOpened);   // ++nReads per new group, mmannot.cpp:1675
    if (sRescued) atomicAdd(&stats[MM_STAT_RESCUED], (u64) sRescued);
  }
  lds_table_flush(T, G);
}

// per-batch scalars before k_probe: all zero - or, when the host already knows that the batch's key range fits the leaf layout
// (qid_lo / qid_bound of mm_annotate_batch), the lower bound as the rebasing origin and "take the leaf path" (no order check:
// within 21 key bits one partition pass + k_leaf cost no more than the compaction + k_chunk_default of name-grouped input)
__global__ void k_set_info(u32 *__restrict__ info, u32 qinv, u32 unsorted) {
  if (threadIdx.x < INFO_WORDS) info[threadIdx.x] = threadIdx.x == INFO_QINV ? qinv : (threadIdx.x == INFO_UNSORTED ? unsorted : 0u);
}

// Leaf layout with more than one partition pass: the records are sorted by bucket now, but where a bucket starts is not a by-product
// of one pass's digit bases any more.  One streaming pass finds every non-empty bucket's range and lists the buckets for k_leaf
// (the list is unordered - buckets do not depend on each other).
__global__ __launch_bounds__(256) void k_bucket_bounds(const u64 *__restrict__ recsA, const u64 *__restrict__ recsB, u32 *__restrict__ info,
                                                       u32 *__restrict__ bLo, u32 *__restrict__ bHi, u32 *__restrict__ list, u64 nbcap, u32 *__restrict__ err) {
  if (!batch_is_leaf(info)) return;
  const SortLayout L = batch_layout(info, 0);
  if (L.passes < 2) return;
  const u64 *__restrict__ recs = (L.passes & 1) ? recsB : recsA;      // the passes ping-pong A <- B <- A ...; the first one writes B
  const u32 M = info[INFO_M];
  if ((u64) ((info[INFO_QMAX] - ~info[INFO_QINV]) >> LEAF_BITS) >= nbcap) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicCAS(err, 0u, (u32) ERR_QID_BOUND); return; }
  for (size_t j = (size_t) blockIdx.x * blockDim.x + threadIdx.x; j < M; j += (size_t) gridDim.x * blockDim.x) {
    const u32 b = (u32) (recs[j] >> (32 + LEAF_BITS));
    const bool first = j == 0 || (u32) (recs[j - 1] >> (32 + LEAF_BITS)) != b;
    const bool last = j + 1 == M || (u32) (recs[j + 1] >> (32 + LEAF_BITS)) != b;
    if (first) { bLo[b] = (u32) j; list[atomicAdd(&info[INFO_NLIST], 1u)] = b; }
    if (last) bHi[b] = (u32) j + 1;
  }
}

// ----------------------------------------------------------------------------- k_leaf: multi-hit countdown / union per bucket, in LDS
// "Leaf" layout (see batch_layout): the partition pass has brought the batch's multi-hit records together by bucket = 4096
// consecutive read names, file order kept inside a bucket.  One workgroup takes a bucket:
//   A. streams its records once (8 B each) and reduces them per name in LDS - union of the element sets, record count,
//      OR and AND of the NH values (all NH of a name equal <=> OR == AND);
//   B. per name, reads the state carried from earlier batches (remaining, union) - consecutive names, so coalesced - and
//      applies the countdown of addCount (mmannot.cpp:1669-1701) in closed form where the order of the records cannot
//      matter: they all fall into the pending group, or they open exactly one new group;
//   C. names where the order does matter (a group closes inside the run: over-complete names such as both mates under one
//      name, mixed NH) are replayed in file order: a second sweep over the bucket, workgroup-chunk by chunk, wave by wave,
//      lanes of one wave that hold the same name one after the other.  Every other lane skips it.
// Exact for every input; the common case costs one streaming read of the records and no per-name gather from HBM.
#define LEAF_THREADS 1024
#define LEAF_ILL 0xFFFFFFFFu   // sOA value marking an order-dependent name (no OR/AND pair of 16-bit NH values looks like this)
__global__ __launch_bounds__(LEAF_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_leaf(
    const u64 *__restrict__ recsA, const u64 *__restrict__ recsB, const u32 *__restrict__ info, const u32 *__restrict__ base, const u32 *__restrict__ bLo,
    const u32 *__restrict__ bHi, const u32 *__restrict__ list, const int4 *__restrict__ aln,
    const uint2 *__restrict__ big, u32 nbig, const u64 *__restrict__ hm, u32 *__restrict__ grpRem, u64 *__restrict__ grpU, u64 qcap,
    GlobalTable G, u64 *__restrict__ stats, int forced) {
  __shared__ LdsTable T;
  __shared__ u64 sU[LEAF_NAMES];     // A: union of the name's sets in this batch; C: the name's union so far
  __shared__ u32 sCnt[LEAF_NAMES];   // A: records of the name in this batch;      C: the name's remaining count
  __shared__ u32 sOA[LEAF_NAMES];    // A: AND << 16 | OR of the name's NH values;  C: LEAF_ILL marks the names being replayed
  __shared__ u32 sOpened, sRescued, sIll;
  if (!batch_is_leaf(info)) return;                      // name-grouped input: k_chunk_default's
  const SortLayout L = batch_layout(info, 0);
  // (forced: the caller's qid_lo / qid_bound promised a key range of at most 21 bits, the launches and k_probe's histogram were made
  // for ONE partition pass; a wider range means some read-name id lies outside the promise)
  if (forced && L.passes != 1) { if (threadIdx.x == 0) atomicCAS(G.err, 0u, (u32) ERR_QID_BOUND); return; }
  const u32 M = info[INFO_M];
  if (M == 0) return;
  const bool listed = L.passes > 1;                      // buckets from k_bucket_bounds' list; else bucket = digit of the one pass
  const u64 *__restrict__ recs = (L.passes & 1) ? recsB : recsA;
  // a read-name id at or beyond the caller's qid_bound would index the per-name state out of bounds: refuse the batch
  if ((u64) info[INFO_QMAX] >= qcap) { if (threadIdx.x == 0) atomicCAS(G.err, 0u, (u32) ERR_QID_BOUND); return; }
  const u32 qmin = ~info[INFO_QINV];
  const u32 nbk = listed ? info[INFO_NLIST] : ((info[INFO_QMAX] - qmin) >> LEAF_BITS) + 1u;      // (one pass: <= RS_BINS by the layout's definition)
  const bool by_index = info[INFO_BIGNH] != 0;
  const u32 t = threadIdx.x, lane = t & 63, w = t >> 6;
  lds_table_init(T);
  if (t == 0) { sOpened = 0; sRescued = 0; }
  u32 opened = 0, rescued = 0;
  auto decode = [&](u32 pv, u64 &m, u32 &nhv) __attribute__((always_inline)) {
    if (by_index) { m = hm[pv]; nhv = tuple_nh((u32) aln[pv].z, pv, big, nbig, G.err); }
    else { m = payload_set(G, pv & 0xFFFFu); nhv = pv >> 16; }
  };
  auto close_group = [&](u64 U) __attribute__((always_inline)) {
    if (U) {
      count_set(T, G, U);
      if ((U & (U - 1)) == 0) ++rescued;
    }
  };
  for (u32 bi = blockIdx.x; bi < nbk; bi += gridDim.x) {    // (workgroup-uniform)
    const u32 b = listed ? list[bi] : bi;
    const u32 lo = listed ? bLo[b] : base[b], hi = listed ? bHi[b] : ((b + 1 < RS_BINS) ? base[b + 1] : M);
    __syncthreads();                                     // the previous bucket's LDS state is no longer read
    if (hi <= lo) continue;
    for (u32 i = t; i < LEAF_NAMES; i += LEAF_THREADS) { sU[i] = 0; sCnt[i] = 0; sOA[i] = 0xFFFF0000u; }
    if (t == 0) sIll = 0;
    __syncthreads();
    // ---- A: one streaming pass, four records per lane in flight (a bucket is ~40 K records: ten round trips to memory).
    // Consecutive lanes hold consecutive records.  Where an aligner's output keeps the hits of a read together, they also hold the
    // same name, and 64 lanes would queue on a handful of LDS words: such a wave first reduces each run of equal names towards its
    // last lane (segmented Kogge-Stone scan on DPP moves, vector ALU only - the network of k_chunk_default) and only that lane goes
    // to LDS.  A wave without two equal neighbours skips the scan.
#define LEAF_UNROLL 4
#define MM_DPP(OLD, SRC, CTRL, RMASK) ((u32) __builtin_amdgcn_update_dpp((int) (OLD), (int) (SRC), (CTRL), (RMASK), 0xF, false))
    for (u64 i0w = (u64) lo + (t - lane); i0w < hi; i0w += LEAF_UNROLL * LEAF_THREADS) {   // (wave-uniform trip count)
      const u64 i0 = i0w + lane;
      u64 r[LEAF_UNROLL];
#pragma unroll
      for (int k = 0; k < LEAF_UNROLL; ++k) { const u64 i = i0 + (u64) k * LEAF_THREADS; r[k] = (i < hi) ? recs[i] : ~0ull; }
#pragma unroll
      for (int k = 0; k < LEAF_UNROLL; ++k) {
        const bool valid = i0 + (u64) k * LEAF_THREADS < hi;
        const u32 loc = valid ? ((u32) (r[k] >> 32) & (LEAF_NAMES - 1u)) : (0x40000000u | lane);   // (no two invalid lanes look alike)
        u64 m = 0;
        u32 nhv = 0;
        if (valid && !by_index) decode((u32) r[k], m, nhv);   // (by index: every name is replayed, only the count is used)
        u32 ulo = (u32) m, uhi = (u32) (m >> 32), cnt = valid ? 1u : 0u, nor = valid ? nhv : 0u, nand = valid ? nhv : 0xFFFFu;
        const u32 ploc = __shfl_up(loc, 1), nloc = __shfl_down(loc, 1);
        u32 f = (lane == 0 || ploc != loc) ? 1u : 0u;       // a run starts here
        if (__ballot(f == 0u)) {                             // (wave-uniform) some lane continues its neighbour's name
#define MM_SEG_STEP(CTRL, RMASK)                                                                               \
          {                                                                                                    \
            const u32 sf = MM_DPP(f, f, CTRL, RMASK); /* a lane without a source keeps its own flag and gets identities */ \
            const u32 a0 = MM_DPP(0u, ulo, CTRL, RMASK), a1 = MM_DPP(0u, uhi, CTRL, RMASK), ac = MM_DPP(0u, cnt, CTRL, RMASK); \
            const u32 ao = MM_DPP(0u, nor, CTRL, RMASK), aa = MM_DPP(0xFFFFu, nand, CTRL, RMASK);             \
            if (!f) { ulo |= a0; uhi |= a1; cnt += ac; nor |= ao; nand &= aa; f = sf; }                       \
          }
          MM_SEG_STEP(0x111, 0xF) MM_SEG_STEP(0x112, 0xF) MM_SEG_STEP(0x114, 0xF) MM_SEG_STEP(0x118, 0xF)   // row_shr:1,2,4,8
          MM_SEG_STEP(0x142, 0xA)                                                                            // row_bcast:15 -> rows 1, 3
          MM_SEG_STEP(0x143, 0xC)                                                                            // row_bcast:31 -> rows 2, 3
#undef MM_SEG_STEP
        }
        if (valid && (lane == 63 || nloc != loc)) {          // the last lane of a run holds the run's reduction
          atomicAdd(&sCnt[loc], cnt);
          if (!by_index) {
            const u64 um = ((u64) uhi << 32) | ulo;
            if (um) atomicOr(&sU[loc], um);
            atomicOr(&sOA[loc], nor);                        // OR in the low half (NH < 65535 here) ...
            atomicAnd(&sOA[loc], (nand << 16) | 0xFFFFu);    // ... AND in the high half
          }
        }
      }
    }
#undef MM_DPP
#undef LEAF_UNROLL
    __syncthreads();
    // ---- B: closed form per name
    const u32 q0 = qmin + (b << LEAF_BITS);
    for (u32 nm = t; nm < LEAF_NAMES; nm += LEAF_THREADS) {
      const u32 c = sCnt[nm];
      if (c == 0) { sOA[nm] = 0; continue; }
      const u32 q = q0 + nm;                              // (a name with records has key <= qmax, so q < qcap)
      u32 rem = grpRem[q];
      u64 U = grpU[q];
      const u64 tU = sU[nm];
      const u32 oa = sOA[nm], nor = oa & 0xFFFFu, nand = oa >> 16;
      bool ill = false;
      if (by_index) ill = true;
      else if (rem > 0 && c <= rem) {
        // all records fall into the pending group of this name (their own NH is irrelevant, mmannot.cpp:1681-1684)
        U |= tU; rem -= c;
        if (rem == 0) { close_group(U); U = 0; }
      } else if (rem == 0 && nor == nand && c <= nor) {
        // one new group, opened by the first record: remaining = NH-1 (mmannot.cpp:1673), order of the others is irrelevant
        ++opened; U = tU; rem = nor - c;
        if (rem == 0) { close_group(U); U = 0; }
      } else ill = true;
      if (ill) { sU[nm] = U; sCnt[nm] = rem; sOA[nm] = LEAF_ILL; atomicAdd(&sIll, 1u); }   // C starts from the carried state
      else { grpRem[q] = rem; grpU[q] = U; sOA[nm] = 0; }
    }
    __syncthreads();
    if (sIll == 0) continue;                              // (workgroup-uniform)
    // ---- C: order-dependent names, replayed in file order
    for (u64 cb = lo; cb < hi; cb += LEAF_THREADS) {
      const u64 i = cb + t;
      u32 loc = 0, nhv = 0;
      u64 m = 0;
      bool mine = false;
      if (i < hi) {
        const u64 r = recs[i];
        loc = (u32) (r >> 32) & (LEAF_NAMES - 1u);
        mine = ((volatile u32 *) sOA)[loc] == LEAF_ILL;
        if (mine) decode((u32) r, m, nhv);
      }
      if (!__syncthreads_or(mine ? 1 : 0)) continue;      // nothing to replay in these 1024 records
      // lanes of this wave holding the same name, in lane (= file) order
      u64 peers = __ballot(mine);
#pragma unroll
      for (int bit = 0; bit < LEAF_BITS; ++bit) {
        const u64 bb = __ballot((loc >> bit) & 1u);
        peers &= ((loc >> bit) & 1u) ? bb : ~bb;
      }
      const u32 rank = (u32) __popcll(peers & ((1ull << lane) - 1ull));
      for (u32 ww = 0; ww < LEAF_THREADS / 64; ++ww) {    // the workgroup's waves one after the other
        if (w == ww) {
          for (u32 rr = 0; __ballot(mine && rank >= rr); ++rr) {
            if (mine && rank == rr) {                     // (LDS operations of one wave execute in program order)
              u32 rem = ((volatile u32 *) sCnt)[loc];
              u64 U = ((volatile u64 *) sU)[loc];
              if (rem == 0) { rem = nhv - 1; U = m; ++opened; }          // mmannot.cpp:1672-1677
              else {                                                     // mmannot.cpp:1678-1701
                --rem; U |= m;
                if (rem == 0) { close_group(U); U = 0; }
              }
              ((volatile u32 *) sCnt)[loc] = rem;
              ((volatile u64 *) sU)[loc] = U;
            }
          }
        }
        __syncthreads();
      }
    }
    for (u32 nm = t; nm < LEAF_NAMES; nm += LEAF_THREADS)
      if (sOA[nm] == LEAF_ILL) { grpRem[q0 + nm] = sCnt[nm]; grpU[q0 + nm] = sU[nm]; }
  }
  if (opened) atomicAdd(&sOpened, opened);
  if (rescued) atomicAdd(&sRescued, rescued);
  __syncthreads();
  if (t == 0) {
    if (sOpened) atomicAdd(&stats[MM_STAT_READS], (u64) sOpened);   // ++nReads per new group, mmannot.cpp:1675
    if (sRescued) atomicAdd(&stats[MM_STAT_RESCUED], (u64) sRescued);
  }
  lds_table_flush(T, G);
}

// End-of-file flush of the still-pending groups, mmannot.cpp:1783-1792
__global__ __launch_bounds__(256) void k_flush_pending(u32 *__restrict__ grpRem, u64 *__restrict__ grpU, u64 Q, GlobalTable G,
                                                       u64 *__restrict__ stats) {
  __shared__ LdsTable T;
  __shared__ u32 sRescued;
  lds_table_init(T);
  if (threadIdx.x == 0) sRescued = 0;
  __syncthreads();
  u32 rescued = 0;
  for (u64 q = (u64) blockIdx.x * blockDim.x + threadIdx.x; q < Q; q += (u64) gridDim.x * blockDim.x) {
    if (grpRem[q] != 0) {
      const u64 U = grpU[q];
      if (U) {
        count_set(T, G, U);
        if ((U & (U - 1)) == 0) ++rescued;    // rawCounts > 1 always holds for a pending group, mmannot.cpp:1789
      }
      grpRem[q] = 0; grpU[q] = 0;
    }
  }
  if (rescued) atomicAdd(&sRescued, rescued);
  __syncthreads();
  if (threadIdx.x == 0 && sRescued) atomicAdd(&stats[MM_STAT_RESCUED], (u64) sRescued);
  lds_table_flush(T, G);
}

// -y ratio: regionCounts[set] += 1.0/nHits, as double, in file order (mmannot.cpp:1730); the table prints
// round(sum) (mmannot.cpp:1868).  Half-integer sums are common, so the accumulation has to be bit-faithful to the
// reference's order of additions - yet a hot row receives millions of terms per batch.  Exact parallel form:
// while the running sum S stays inside one binade [2^k, 2^(k+1)), S = N * u with u = 2^(k-52) and an integer N in
// [2^52, 2^53); adding x = fl(1/nh) gives fl(S + x) = (N + inc) * u with
//      inc = round-to-nearest(x / u)        (an integer that does not depend on N)
// unless x / u lies exactly half-way between two integers (then round-half-even looks at N's parity), which for a
// given x can happen in exactly one binade: the one where u/2 is x's lowest set bit.  So a chunk of terms is reduced
// with integer adds (increments of small NH values come from an LDS table rebuilt when the binade changes); a chunk
// that contains a tie-capable term, an infinite term (NH = 0) or that would leave the binade (the crossing add rounds
// on the coarser grid) is replayed by one lane with real IEEE adds.  Input: (slot, NH) pairs sorted by slot, file
// order kept inside a slot; segLo/segHi give each slot's run.
#define RATIO_SENTINEL 0xFFFFFFFFFFFFFFFFull
#define RATIO_LUT 4096
#define RATIO_THREADS 1024           // (measured: 512-thread workgroups, two per CU, with 8192-term tiles: accumulation 0.155 -> 0.185 ms per batch)
#define RATIO_PER_THREAD 16
#define RATIO_SERIAL_STEP 1024      // a chunk that cannot be reduced is replayed this many terms at a time
#define RATIO_CHUNK (RATIO_THREADS * RATIO_PER_THREAD)
#define RATIO_SLOTS_PER_WG 128
#define RATIO_STAGE 1024            // tile summaries staged in LDS at a time (a 2^24-tuple batch has 1024 tiles)
#define RATIO_SHORT_ROW 96          // rows with at most this many terms in a batch are summed by a single lane

__global__ __launch_bounds__(256) void k_seg_bounds(const u32 *__restrict__ skeys, const u32 *__restrict__ info, u32 *__restrict__ segLo,
                                                    u32 *__restrict__ segHi) {
  const u32 M = info[INFO_M];
  for (size_t j = (size_t) blockIdx.x * blockDim.x + threadIdx.x; j < M; j += (size_t) gridDim.x * blockDim.x) {
    const u32 k = skeys[j];
    if (j == 0 || skeys[j - 1] != k) segLo[k] = (u32) j;
    if (j + 1 == M || skeys[j + 1] != k) segHi[k] = (u32) j + 1;
  }
}

// increment of N when fl(1/nh) is added to a sum in binade k; RATIO_SENTINEL = "take the serial path"
__device__ __forceinline__ u64 ratio_inc(u32 nh, int k) {
  if (nh == 0) return RATIO_SENTINEL;                       // 1.0/0 = inf
  const double x = __ddiv_rn(1.0, (double) nh);
  const u64 bx = (u64) __double_as_longlong(x);
  const u64 Mx = (bx & 0xFFFFFFFFFFFFFull) | (1ull << 52);  // x = Mx * 2^ex (x is normal: nh < 2^32)
  const int ex = (int) ((bx >> 52) & 0x7FF) - 1075;
  const int sh = (k - 52) - ex;                             // x / u = Mx / 2^sh
  if (sh <= 0) return RATIO_SENTINEL;                       // x >= 2^52 u: N + x/u >= 2^53, the add leaves the binade
  if (sh > 54) return 0;                                    // below u/2: absorbed
  const u64 q = Mx >> sh;
  const u64 rem = Mx & ((1ull << sh) - 1), half = 1ull << (sh - 1);
  if (rem == half) return RATIO_SENTINEL;                   // exact tie: round-half-even depends on N
  return q + (rem > half ? 1ull : 0ull);
}

// Tile summaries for long rows: every 16384-term tile of the sorted stream that lies inside one row gets the sum of
// its integer increments, computed for the binade the row's sum is in when the batch starts.  The whole GPU works on
// this; k_ratio_rows then consumes a tile in O(1) as long as the binade is still the same and the tile neither holds a
// tie-capable / infinite term nor leaves the binade, and falls back to its own exact chunk path otherwise.
// Rows can be accumulated in any order relative to each other (only the terms of ONE row are ordered), so a replay may be cut into
// "row chunks": chunk c of n holds the rows whose element set hashes to c.  Ranks chaining their sums hand
// chunk c on while they work on chunk c + 1.
__device__ __forceinline__ bool row_in_chunk(u64 mask, u32 chunk, u32 nchunks) { return nchunks <= 1 || hash64(mask) % nchunks == chunk; }

__global__ __launch_bounds__(256) void k_ratio_tiles(const u32 *__restrict__ skeys, const u32 *__restrict__ svals, const u32 *__restrict__ info,
                                                     GlobalTable G, u64 *__restrict__ tileTot, int *__restrict__ tileK, u32 chunk, u32 nchunks) {
  // tileTot[2T], tileK[2T]: summary for the binade k the row is in now; [2T+1]: for binade k+1 (a long row crosses
  // into the next binade at most once per batch, and the tiles behind the crossing then still have a summary)
  __shared__ u64 wsum[2][4];
  __shared__ u32 wbad[2][4];
  const u32 M = info[INFO_M];
  const size_t lo = (size_t) blockIdx.x * RATIO_CHUNK;
  if (lo >= M) return;
  const u32 t = threadIdx.x, lane = t & 63, w = t >> 6;
  int k = 0x7ffffffe;                                        // "no summary"
  if (lo + RATIO_CHUNK <= M) {
    const u32 key0 = skeys[lo], key1 = skeys[lo + RATIO_CHUNK - 1];
    if (key0 == key1 && row_in_chunk(G.key[key0], chunk, nchunks)) {
      const u64 bs = (u64) __double_as_longlong(G.sum[key0]);
      const int E = (int) ((bs >> 52) & 0x7FF);
      if (E != 0 && E != 0x7FF && !(bs >> 63)) k = E - 1023;
    }
  }
  if (k == 0x7ffffffe) { if (t < 2) tileK[2 * blockIdx.x + t] = k; return; }   // block-uniform
  u64 acc0 = 0, acc1 = 0;
  u32 bad0 = 0, bad1 = 0;
  for (u32 i = t; i < RATIO_CHUNK; i += 256) {
    const u32 nh = svals[lo + i];
    const u64 i0 = ratio_inc(nh, k), i1 = ratio_inc(nh, k + 1);
    if (i0 == RATIO_SENTINEL) bad0 = 1; else acc0 += i0;    // inc < 2^52, 64 terms per lane, 64 lanes: below 2^64
    if (i1 == RATIO_SENTINEL) bad1 = 1; else acc1 += i1;
  }
  for (int off = 32; off > 0; off >>= 1) {
    acc0 += __shfl_down(acc0, off); bad0 |= __shfl_down(bad0, off);
    acc1 += __shfl_down(acc1, off); bad1 |= __shfl_down(bad1, off);
  }
  if (lane == 0) { wsum[0][w] = acc0; wbad[0][w] = bad0; wsum[1][w] = acc1; wbad[1][w] = bad1; }
  __syncthreads();
  if (t < 2) {
    u64 tot = 0; u32 b = 0;
    for (int q = 0; q < 4; ++q) { if (wsum[t][q] >= (1ull << 53)) b = 1; else tot += wsum[t][q]; b |= wbad[t][q]; }
    tileTot[2 * blockIdx.x + t] = tot;
    tileK[2 * blockIdx.x + t] = b ? 0x7ffffffe : k + (int) t;
  }
}

__global__ __launch_bounds__(RATIO_THREADS) void k_ratio_rows(const u32 *__restrict__ svals, const u32 *__restrict__ segLo,
                                                              const u32 *__restrict__ segHi, GlobalTable G, const u64 *__restrict__ tileTot,
                                                              const int *__restrict__ tileK, u32 chunk, u32 nchunks) {
  __shared__ u64 lut[RATIO_LUT];
  __shared__ u64 wsum[RATIO_THREADS / 64];
  __shared__ u32 wflag[RATIO_THREADS / 64];
  __shared__ double sS;
  __shared__ int sK, sSerial;
  __shared__ int sTileK[RATIO_STAGE], sTileK2[RATIO_STAGE];   // (48 KB of LDS in all: every workgroup of the launch is resident at once;
  __shared__ u64 sTot2[RATIO_STAGE];                          //  with 4096-entry staging it was 96 KB, one workgroup per CU, two rounds)
  __shared__ u32 sUsed;
  __shared__ u64 sBefore;
  __shared__ u32 sLo[RATIO_SLOTS_PER_WG], sHi[RATIO_SLOTS_PER_WG];
  const u32 t = threadIdx.x, lane = t & 63, w = t >> 6;
  int kLut = 0x7fffffff;
  // this workgroup's slots: one coalesced read of their segment bounds instead of a dependent load per slot
  // (interleaved: workgroup b takes the slots b, b + gridDim.x, ... - the table hands out its first few hundred slots first, and the
  // hot rows of a run sit among them: a workgroup per row, not 128 of them in a row)
#define RATIO_SLOT(I) (blockIdx.x + (u32) (I) * gridDim.x)
  if (t < RATIO_SLOTS_PER_WG) {
    u32 lo = segLo[RATIO_SLOT(t)], hi = segHi[RATIO_SLOT(t)];
    // Short rows first, each by ONE lane, all of them at once: most rows of a table are element sets that collect a handful of hits
    // per batch (ambiguous combinations), and the file-order sum of a few terms is just that - a loop of IEEE adds
    // (mmannot.cpp:1730).  The workgroup-wide machinery below is for the rows that collect thousands.
    if (hi > lo && hi - lo <= RATIO_SHORT_ROW && row_in_chunk(G.key[RATIO_SLOT(t)], chunk, nchunks)) {
      double S = G.sum[RATIO_SLOT(t)];
      for (u32 j = lo; j < hi; ++j) S = __dadd_rn(S, __ddiv_rn(1.0, (double) svals[j]));
      G.sum[RATIO_SLOT(t)] = S;
      hi = lo;                                               // done
    }
    sLo[t] = lo; sHi[t] = hi;
  }
  __syncthreads();
  for (u32 si = 0; si < RATIO_SLOTS_PER_WG; ++si) {
    const u32 slot = RATIO_SLOT(si);
    const u32 lo = sLo[si], hi = sHi[si];
    if (hi <= lo) continue;                                  // block-uniform
    if (!row_in_chunk(G.key[slot], chunk, nchunks)) continue;  // (block-uniform) another chunk's row
    if (t == 0) sS = G.sum[slot];
    __syncthreads();
    for (u32 cb = lo; cb < hi;) {
      u32 cn = min(hi - cb, (u32) RATIO_CHUNK - (cb % RATIO_CHUNK));   // chunks end on tile boundaries
      if (cn == RATIO_CHUNK) {
        // Full tiles ahead: consume their precomputed summaries (k_ratio_tiles) while they stay valid.
        const u32 T0 = cb / RATIO_CHUNK, nT = min((hi - cb) / RATIO_CHUNK, (u32) RATIO_STAGE);
        __syncthreads();
        for (u32 i = t; i < nT; i += RATIO_THREADS) {           // lut doubles as staging; tot2 holds the next binade's sums
          lut[i] = tileTot[2 * (T0 + i)]; sTileK[i] = tileK[2 * (T0 + i)];
          sTot2[i] = tileTot[2 * (T0 + i) + 1]; sTileK2[i] = tileK[2 * (T0 + i) + 1];
        }
        kLut = 0x7fffffff;
        __syncthreads();
        if (t == 0) {
          double S = sS;
          u64 bs = (u64) __double_as_longlong(S);
          int E = (int) ((bs >> 52) & 0x7FF);
          u32 used = 0;
          if (E != 0 && E != 0x7FF && !(bs >> 63)) {
            const int k = E - 1023;
            u64 N = (bs & 0xFFFFFFFFFFFFFull) | (1ull << 52);
            for (; used < nT; ++used) {
              const u64 add = (sTileK[used] == k) ? lut[used] : ((sTileK2[used] == k) ? sTot2[used] : ~0ull);
              if (add == ~0ull || N + add >= (1ull << 53)) break;
              N += add;
            }
            sS = __longlong_as_double((long long) ((((u64) E) << 52) | (N & 0xFFFFFFFFFFFFFull)));
          }
          sUsed = used;
        }
        __syncthreads();
        const u32 used = sUsed;
        __syncthreads();
        if (used) { cb += used * RATIO_CHUNK; continue; }     // block-uniform
      }
      // Exact chunk step.  Every thread takes 16 consecutive terms; a block-wide prefix scan of the integer increments
      // finds the first term that cannot be applied as an integer step: one that would leave the binade, one whose
      // rounding is an exact tie, or an infinite one (and the very first term of a row, S = 0).  Everything before it is
      // applied at once; that single term is added with a real IEEE add; the next step starts behind it.
      if (t == 0) {
        const u64 bs = (u64) __double_as_longlong(sS);
        const int E = (int) ((bs >> 52) & 0x7FF);
        sSerial = (E == 0 || (bs >> 63)) ? 1 : (E == 0x7FF ? 2 : 0);   // 1: no binade yet (S = 0), 2: inf/nan absorbs everything
        sK = E - 1023;
      }
      __syncthreads();
      const int mode = sSerial, k = sK;
      if (mode == 2) { cb = hi; __syncthreads(); continue; }          // inf + x = inf
      if (mode == 0 && k != kLut) {
        for (u32 v = t; v < RATIO_LUT; v += RATIO_THREADS) lut[v] = ratio_inc(v, k);
        kLut = k;
        __syncthreads();
      }
      const u64 CAP = 1ull << 54;                                       // saturating sums: only "< 2^53" matters
      u64 pre[RATIO_PER_THREAD];
      u64 mine = 0;
      u32 firstBad = 0xFFFFFFFFu;                                       // first term of mine that needs the serial add
      const u32 e0 = t * RATIO_PER_THREAD;
      if (mode == 0) {
        u32 nhv[RATIO_PER_THREAD];
#pragma unroll
        for (int r = 0; r < RATIO_PER_THREAD; r += 4) {
          if (e0 + r + 3 < cn) { const uint4 q = *(const uint4 *) (svals + cb + e0 + r); nhv[r] = q.x; nhv[r + 1] = q.y; nhv[r + 2] = q.z; nhv[r + 3] = q.w; }
          else { for (int z = 0; z < 4; ++z) nhv[r + z] = (e0 + r + z < cn) ? svals[cb + e0 + r + z] : 0xFFFFFFFFu; }
        }
#pragma unroll
        for (int r = 0; r < RATIO_PER_THREAD; ++r) {
          u64 inc = 0;
          if (e0 + r < cn) {
            const u32 nh = nhv[r];
            inc = (nh < RATIO_LUT) ? lut[nh] : ratio_inc(nh, k);
            if (inc == RATIO_SENTINEL) { if (firstBad == 0xFFFFFFFFu) firstBad = e0 + r; inc = 0; }
          }
          mine = min(mine + inc, CAP);
          pre[r] = mine;                                                // inclusive prefix inside my 16 terms
        }
      }
      // block exclusive scan of the thread totals (saturating)
      u64 x = mine;
      for (int off = 1; off < 64; off <<= 1) {
        const u64 y = __shfl_up(x, off);
        if ((int) lane >= off) x = min(x + y, CAP);
      }
      if (lane == 63) wsum[w] = x;
      __syncthreads();
      u64 base = x - mine;                                              // exact: x < CAP unless saturated, then base >= 2^53 anyway
      if (x >= CAP) base = CAP;
      for (u32 q = 0; q < w; ++q) base = min(base + wsum[q], CAP);
      // my first term that cannot be an integer step
      const u64 N0 = ((u64) __double_as_longlong(sS) & 0xFFFFFFFFFFFFFull) | (1ull << 52);
      u32 stopAt = 0xFFFFFFFFu;
      if (mode == 1) { if (t == 0) stopAt = 0; }
      else {
#pragma unroll
        for (int r = 0; r < RATIO_PER_THREAD; ++r) {
          if (stopAt == 0xFFFFFFFFu && e0 + r < cn) {
            const bool cross = min(N0 + min(base + pre[r], CAP), CAP) >= (1ull << 53);
            if (cross || firstBad == e0 + r) stopAt = e0 + r;
          }
        }
      }
      u32 m = stopAt;
      for (int off = 32; off > 0; off >>= 1) m = min(m, (u32) __shfl_down(m, off));
      if (lane == 0) wflag[w] = m;
      __syncthreads();
      u32 stop = 0xFFFFFFFFu;
      for (u32 q = 0; q < RATIO_THREADS / 64; ++q) stop = min(stop, wflag[q]);
      // the owner of the stop term publishes the increment sum of everything before it
      if (stop != 0xFFFFFFFFu) {
        if (stop >= e0 && stop < e0 + RATIO_PER_THREAD) {
          const u32 r = stop - e0;
          u64 before = base;
#pragma unroll
          for (int z = 0; z < RATIO_PER_THREAD; ++z) if ((u32) z + 1 == r) before = base + pre[z];
          sBefore = (mode == 1) ? 0 : before;
        }
      } else if (t == RATIO_THREADS - 1) sBefore = min(base + mine, CAP);   // whole chunk applies
      __syncthreads();
      if (t == 0) {
        double S = sS;
        if (mode == 0) {
          const u64 N2 = N0 + sBefore;                                  // < 2^53 by construction
          S = __longlong_as_double((long long) ((((u64) (k + 1023)) << 52) | (N2 & 0xFFFFFFFFFFFFFull)));
        }
        if (stop != 0xFFFFFFFFu) S = __dadd_rn(S, __ddiv_rn(1.0, (double) svals[cb + stop]));   // the reference's own add
        sS = S;
      }
      cb += (stop != 0xFFFFFFFFu) ? stop + 1 : cn;
      __syncthreads();
    }
    if (t == 0) G.sum[slot] = sS;
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------- -M: intervals behind each hit
// EvaluationStructure::getIds (mmannot.cpp:1077-1081): for every element of the hit's final set, all intervals of that
// element that passed the strand / overlap test during scan().  COUNT pass writes how many, FILL pass writes their
// indices (in the reference's sorted interval vector) at the offsets an exclusive scan of the counts gives.
template <int OVL, bool FILL>
__global__ __launch_bounds__(256) void k_hit_intervals(const ProbeParams P, const u32 *__restrict__ actIdx, const int4 *__restrict__ aln, size_t n,
                                                       const u64 *__restrict__ hm, u32 *__restrict__ cnt, const u32 *__restrict__ off,
                                                       u32 *__restrict__ out) {
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
    const u64 mask = hm[i];
    u32 c = 0;
    if (mask != 0 && mask != ~0ull) {
      const int4 a = aln[i];
      const int s = a.x, e = a.y;
      const int rec_strand = a.z & 1;
      const int rstrand = (P.strandedness == MM_STRANDED_U) ? 1 : ((P.strandedness == MM_STRANDED_F) ? rec_strand : (rec_strand ^ 1));
      Cand cd;
      if (locate(P, TSN_TID1(a.z), s, cd)) {
        u32 w = FILL ? off[i] : 0u;
        for (u32 k = cd.a0; k < cd.a1; ++k) {
          const int4 r = P.act[k];
          u32 o;
          if (((mask >> ((u32) r.z & 0xffu)) & 1ull) && iv_pass<OVL>(P, r, s, e, rstrand, o)) { if (FILL) out[w++] = actIdx[k]; ++c; }
        }
        for (u32 k = cd.i0; k < cd.i1; ++k) {
          const int4 r = P.iv[k];
          if (r.x > e) break;
          u32 o;
          if (((mask >> ((u32) r.z & 0xffu)) & 1ull) && iv_pass<OVL>(P, r, s, e, rstrand, o)) { if (FILL) out[w++] = k; ++c; }
        }
      }
    }
    if (!FILL) cnt[i] = c;
  }
}

// ----------------------------------------------------------------------------- synthetic workload (bench / tests)
struct SynthDev {
  u64 seed, first;
  int read_len, nh_max, nchr;
  u32 scatter_log2;
  u32 pair_insert;   // > 0: tuples 2p, 2p+1 are the mates of pair p (same name, opposite strands, starts this far apart + 0..63)
  u64 cum[65];   // cumulative chromosome lengths
};

__host__ __device__ inline u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  u64 z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_synth(const SynthDev S, int4 *__restrict__ aln, size_t n) {
  for (size_t j = (size_t) blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t) gridDim.x * blockDim.x) {
    const u64 i = S.first + j;
    // physical index -> logical hit index: affine permutation inside windows of 2^scatter_log2
    u64 L = i;
    if (S.scatter_log2) {
      const u64 W = 1ull << S.scatter_log2;
      const u64 r = i & (W - 1);
      L = (i & ~(W - 1)) | ((r * 0x9E3779B1ull + (S.seed & (W - 1))) & (W - 1));
    }
    u32 nh = 1;
    u64 qid = L;
    if (S.nh_max > 1) {
      const u64 K = (u64) S.nh_max, T = K * (K + 1) / 2;
      const u64 blk = L / T, off = L % T;
      u64 k = (u64) ((sqrt(8.0 * (double) off + 1.0) - 1.0) * 0.5);
      while (k * (k + 1) / 2 > off) --k;
      while ((k + 1) * (k + 2) / 2 <= off) ++k;
      nh = (u32) k + 1;
      qid = blk * K + k;
    }
    u64 mate = 0, foot = (u64) S.read_len;   // foot: what has to fit on the chromosome
    if (S.pair_insert) { mate = L & 1ull; L >>= 1; qid = L; foot += (u64) S.pair_insert + 63; }
    const u64 h = splitmix64(S.seed ^ (L * 0xD1342543DE82EF95ull));
    const u64 h2 = splitmix64(h);
    const u64 g = h % S.cum[S.nchr];
    int c = 0;
    while (g >= S.cum[c + 1]) ++c;
    const u64 len = S.cum[c + 1] - S.cum[c];
    const u64 span = len > foot ? len - foot + 1 : 1;
    const u64 pos0 = h2 % span + mate * ((u64) S.pair_insert + ((h >> 32) & 63ull));
    int4 a;
    a.x = (int) (pos0 + 1);
    a.y = a.x + S.read_len - 1;
    a.z = (int) MM_ALN_PACK(c, ((h >> 63) ^ mate) & 1ull, nh);
    a.w = (int) (u32) qid;
    aln[j] = a;
  }
}

// Rows of the table for the host (mm_get_rows), in slot order: interleaved (set, value) pairs behind a count, so that the read-back is
// one short copy instead of the three 512-KB arrays (the table holds 10^2..10^4 rows of its 65 536 slots).  Two small launches: every
// workgroup counts the rows among its 1024 slots; then it adds up the counts of the workgroups before it (at most 63 values), scans
// its own flags and writes its rows where they belong.
#define ROWS_WG (GT_CAP / 1024u)
__device__ __forceinline__ bool row_live(const GlobalTable &G, int ratio, u32 i) { return G.key[i] != 0 && (ratio || G.cnt[i] != 0); }
__global__ __launch_bounds__(1024) void k_rows_count(GlobalTable G, int ratio, u32 *__restrict__ wgCount) {
  __shared__ u32 n;
  if (threadIdx.x == 0) n = 0;
  __syncthreads();
  const u64 b = __ballot(row_live(G, ratio, blockIdx.x * 1024u + threadIdx.x));
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&n, (u32) __popcll(b));
  __syncthreads();
  if (threadIdx.x == 0) wgCount[blockIdx.x] = n;
}
__global__ __launch_bounds__(1024) void k_rows_write(GlobalTable G, int ratio, const u32 *__restrict__ wgCount, u64 *__restrict__ out) {
  __shared__ u32 wsum[16];
  const u32 t = threadIdx.x, lane = t & 63, w = t >> 6, i = blockIdx.x * 1024u + t;
  u32 base = 0;
  for (u32 q = 0; q < blockIdx.x; ++q) base += wgCount[q];
  if (blockIdx.x == ROWS_WG - 1 && t == 0) out[0] = (u64) (base + wgCount[blockIdx.x]);
  const bool live = row_live(G, ratio, i);
  const u64 b = __ballot(live);
  if (lane == 0) wsum[w] = (u32) __popcll(b);
  __syncthreads();
  u32 pos = base + (u32) __popcll(b & ((1ull << lane) - 1ull));
  for (u32 q = 0; q < w; ++q) pos += wsum[q];
  if (live) {
    out[1 + 2 * (size_t) pos] = G.key[i];
    out[2 + 2 * (size_t) pos] = ratio ? (u64) __double_as_longlong(G.sum[i]) : (u64) __double_as_longlong((double) G.cnt[i]);
  }
}

// Tables of other ranks (mm_table_export_counts_device layout: sets, counts, counters - `stride` words per table) added into this
// context's own table and counters; table `skip` is this context's own.
__global__ __launch_bounds__(256) void k_table_merge(const u64 *__restrict__ tabs, u32 n_tables, u32 stride, u32 skip, GlobalTable G, u64 *__restrict__ stats) {
  const u32 total = n_tables * GT_CAP;
  for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
    const u32 r = j / GT_CAP, i = j % GT_CAP;
    if (r == skip) continue;
    const u64 k = tabs[(size_t) r * stride + i], n = tabs[(size_t) r * stride + GT_CAP + i];
    if (k && n) gt_add(G, k, n);
  }
  if (blockIdx.x == 0 && threadIdx.x < n_tables * 8u) {
    const u32 r = threadIdx.x >> 3, q = threadIdx.x & 7u;
    if (r != skip && q < MM_N_STATS) atomicAdd(&stats[q], tabs[(size_t) r * stride + 2 * (size_t) GT_CAP + q]);
  }
}

// ============================================================================= host side
struct mm_index {
  int device = 0;
  int nchr = 0;
  size_t n_iv = 0, n_act = 0, n_bins = 0;
  int shift = 0;
  int E = 0;
  int4 *d_iv = nullptr, *d_act = nullptr;
  uint2 *d_bins = nullptr;
  uint4 *d_chrInfo = nullptr;
  u64 *d_lineMask = nullptr;
  u32 *d_actIdx = nullptr;
  uint4 *d_slots = nullptr;
  uint2 *d_flat = nullptr, *d_chrSlot = nullptr;
  u32 *d_slotBits = nullptr;   // the slots' map: per 8 bins one word of 2-bit codes + the covering intervals' class byte
  size_t n_slotBits = 0;
  int *d_longEnd = nullptr;
  int sshift = 0;
  u32 reach = 0;
  size_t n_slots = 0, n_flat = 0;
  // fast index (round 3): the same bins as 32-byte slots of 4-byte bin-relative records + a bitmap of the non-empty ones; NULL when the
  // bin width does not allow the encoding (see FS_* below)
  uint4 *d_fslots = nullptr;
  u32 *d_fbits = nullptr;  // the map: 2 bits per fast slot
  u32 *d_fcls = nullptr;   // one byte per 8 fast slots: element and ok bits of the interval covering them (FS_COVERED)
  u32 *d_fext = nullptr;   // extension blocks (32 words each) of the bins with 8 to 31 records
  u32 fextClass[4] = {0, 0, 0, 0};   // blocks numbered below fextClass[c] hold at most 11 / 15 / 23 / 31 records
  uint2 *d_fchr = nullptr; // per chromosome {first fast slot, number of fast slots}
  int fshift = 0;          // log2 of the fast slots' bin width (may be one less than sshift)
  u32 freach = 0;
  size_t n_fslots = 0;
  u32 fLmax = 0;           // longest read the fast slots answer exactly
  u64 elemF = 0, elemR = 0, elemUp = 0, elemDown = 0;
  u64 lineMask[64];
};

struct mm_ctx {
  const mm_index *ix = nullptr;
  mm_options opt{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool finalized = false;
  // batch buffers
  size_t cap = 0;
  int4 *d_aln = nullptr;   // = d_alnBuf[cur]
  // host-tuple path: two staging sets filled by a dedicated copy stream, so the copy of batch k+1 overlaps the kernels of batch k
  int4 *d_alnBuf[2] = {nullptr, nullptr};
  u32 *d_pkBuf[2] = {nullptr, nullptr};   // 12-byte tuples as they cross the link (mm_annotate_batch12), expanded into d_alnBuf
  size_t pk_cap = 0;
  // NH values too large for the tuple's NH field: the caller's (index, NH) list, staged through pinned memory (two sets)
  uint2 *d_big[2] = {nullptr, nullptr}, *h_big[2] = {nullptr, nullptr};
  size_t big_cap[2] = {0, 0};
  hipEvent_t ev_big[2] = {nullptr, nullptr};
  bool big_valid[2] = {false, false};
  int big_cur = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
  bool consumed_valid[2] = {false, false}, copied_valid[2] = {false, false};
  int cur = 0;
  u64 *d_hm = nullptr;
  u32 *d_gkey = nullptr, *d_gpay = nullptr, *d_kA = nullptr, *d_vA = nullptr, *d_kB = nullptr, *d_vB = nullptr;
  u32 *d_hist = nullptr, *d_rowTot = nullptr, *d_M = nullptr, *d_seg = nullptr;
  int dcnt_cur = 0;
  // which probe this context uses where both apply: 0 = still measuring, 1 = the pipelined one, 2 = the walk
  int tune_choice = 0;
  unsigned tune_turn = 0;
  struct TuneSample { int path; size_t n; hipEvent_t a, b; };
  std::vector<TuneSample> tune_pending;
  double tune_ns_per_tuple[2] = {0, 0};
  int tune_samples[2] = {0, 0};      // (the first sample of each path is not counted: code load, cold caches)
  bool env_fast = false;
  u32 *d_dlist = nullptr, *d_dcnt = nullptr;   // tuples k_probe_fast passes on to k_probe_list: one region per wave of a tile, and the regions' lengths
  // A/B switches of the environment, read once (mm_ctx_create): MMANNOT_AMD_NO_FUSED_HIST, MMANNOT_AMD_GENERAL_PROBE, MMANNOT_AMD_FULL_INDEX_ONLY,
  // MMANNOT_AMD_NO_FAST (the round-2 walk instead of the pipelined one)
  bool env_no_fused_hist = false, env_general_probe = false, env_full_index_only = false, env_no_fast = false;
  u64 *d_tileTot = nullptr;
  int *d_tileK = nullptr;
  size_t last_n = 0;
  const int4 *last_aln = nullptr;   // device tuples of the last batch (valid until the next mm_annotate_batch)
  // persistent state
  GlobalTable G{};
  u64 *d_stats = nullptr, *d_walk = nullptr;
  u64 *d_rows = nullptr, *h_rows = nullptr;   // mm_get_rows: count + (set, value) pairs, device and pinned host
  u64 qcap = 0;
  u32 *d_grpRem = nullptr;
  u64 *d_grpU = nullptr;
  u32 *d_bLo = nullptr, *d_bHi = nullptr, *d_bList = nullptr;   // per bucket of 4096 names: record range of the batch, and the non-empty buckets
  u64 nbcap = 0;
  // -y ratio with defer_ratio: each batch's slot-sorted (slot, NH) pairs are parked here until mm_ratio_resolve
  // (round 3: a parked batch keeps the buffer the sort left it in - the context takes another one from a pool for its next batch - and
  // buffers come back to the pool when the runs are released: no allocation, no copy, no implicit device synchronisation per batch once
  // the pool is warm; mm_ratio_reserve warms it ahead of time)
  struct Run { u32 *k = nullptr, *v = nullptr, *info = nullptr, *seg = nullptr; size_t n = 0, cap = 0; };
  std::vector<Run> runs;
  std::vector<Run> run_pool;          // free buffers: k = base of cap x 8 bytes (v = k + cap), then seg (2 x GT_CAP x 4), then info
  u64 *d_xkeys = nullptr;             // staging of a table handed over from another device (mm_ratio_handoff)
  double *d_xsums = nullptr;
  hipEvent_t ev_handoff = nullptr;
  // optional live timing (HIP events on the context's stream), see mm_set_profiling
  int profiling = 0;   // 0 off, 1 every phase, 2 the probe phase only
  struct Span { int phase; hipEvent_t a, b; };
  std::vector<Span> spans;
};

static void free_runs(mm_ctx *c);
static void free_run_pool(mm_ctx *c);
static void tune_collect(mm_ctx *c);

struct PhaseTimer {  // records an event pair around a group of launches when profiling is on
  mm_ctx *c; int phase; hipEvent_t a = nullptr, b = nullptr;
  hipStream_t stream_override = nullptr;
  PhaseTimer(mm_ctx *c_, int ph) : c(c_), phase(ph) {
    if ((c->profiling == 1 || (c->profiling == 2 && (ph == MM_PHASE_PROBE || ph == MM_PHASE_H2D))) && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess)
      (void) hipEventRecord(a, c->stream);
  }
  void restart() { if (a) (void) hipEventRecord(a, stream_override ? stream_override : c->stream); }
  void stop() {
    if (a && b) { (void) hipEventRecord(b, stream_override ? stream_override : c->stream); c->spans.push_back({phase, a, b}); a = b = nullptr; }
  }
  ~PhaseTimer() { stop(); }
};

// timed batches that have finished -> per-path averages; decides once both paths have two counted samples
static void tune_collect(mm_ctx *c) {
  for (size_t k = 0; k < c->tune_pending.size();) {
    mm_ctx::TuneSample &t = c->tune_pending[k];
    if (hipEventQuery(t.b) != hipSuccess) { ++k; continue; }
    float ms = 0;
    if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess && c->tune_samples[t.path]++ > 0) c->tune_ns_per_tuple[t.path] += (double) ms * 1e6 / (double) t.n;
    (void) hipEventDestroy(t.a); (void) hipEventDestroy(t.b);
    c->tune_pending.erase(c->tune_pending.begin() + (long) k);
  }
  if (c->tune_choice == 0 && c->tune_samples[0] >= 2 && c->tune_samples[1] >= 2) {
    const double f = c->tune_ns_per_tuple[0] / (c->tune_samples[0] - 1), w = c->tune_ns_per_tuple[1] / (c->tune_samples[1] - 1);
    c->tune_choice = f <= w ? 1 : 2;
  }
}

static int device_check(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(MM_ERR_HIP, std::string("no HIP device available (") + hipGetErrorString(e) + "); the HIP path is the only path");
  if (device < 0 || device >= n) return fail(MM_ERR_ARG, "device index out of range");
  HIPCHK(hipSetDevice(device));
  return MM_OK;
}

// The maps as the kernels read them: per group of 8 bins one word - the bins' 2-bit codes in bits 0-15, the group's class byte in
// bits 16-23 (the builders above fill the two as separate arrays: 16 codes to a word, four class bytes to a word).
static std::vector<u32> merge_map(const std::vector<u32> &bits, const std::vector<u32> &cls, u64 ns) {
  const unsigned char *cb = (const unsigned char *) cls.data();
  const u64 ng = (ns + 7) / 8;
  std::vector<u32> m(ng + 2, 0u);                       // (+ the slack the old layout had: a lane without a slot reads word 0, nothing reads behind ng)
  for (u64 g = 0; g < ng; ++g) m[g] = ((bits[g >> 1] >> ((g & 1) * 16)) & 0xFFFFu) | ((u32) cb[g] << 16);
  return m;
}

extern "C" int mm_index_create(mm_index **out, const mm_elements *el, int32_t n_chromosomes, size_t n_iv, const int32_t *chr,
                               const int32_t *start, const int32_t *end, const uint8_t *type, const uint8_t *strand, int device,
                               int bin_shift) {
  if (!out || !el || !chr || !start || !end || !type || !strand) return fail(MM_ERR_ARG, "mm_index_create: null argument");
  *out = nullptr;
  const int E = el->n_elements;
  if (E < 1 || E > MM_MAX_ELEMENTS) return fail(MM_ERR_RANGE, "mm_index_create: number of Order elements must be 1..64");
  if (n_chromosomes < 1) return fail(MM_ERR_ARG, "mm_index_create: no chromosome");
  if (n_chromosomes > MM_ALN_MAX_CHROMOSOMES) return fail(MM_ERR_RANGE, "mm_index_create: more chromosomes than the tuple's 20-bit chromosome field holds");
  if (n_iv == 0) return fail(MM_ERR_ARG, "mm_index_create: empty interval list (the reference exits here, mmannot.cpp:1285-1288)");
  if (n_iv >= 0xFFFFFFF0ull) return fail(MM_ERR_RANGE, "mm_index_create: too many intervals");
  int rc = device_check(device);
  if (rc) return rc;
  mm_index *ix = new mm_index();
  ix->device = device; ix->nchr = n_chromosomes; ix->n_iv = n_iv; ix->E = E;
  for (int e = 0; e < E; ++e) {
    if (el->elem_strand[e] == MM_STRAND_F) ix->elemF |= 1ull << e;
    if (el->elem_strand[e] == MM_STRAND_R) ix->elemR |= 1ull << e;
    if (el->is_up[e]) ix->elemUp |= 1ull << e;
    if (el->is_down[e]) ix->elemDown |= 1ull << e;
  }
  for (int e = 0; e < 64; ++e) {
    ix->lineMask[e] = 0;
    if (e < E) for (int f = 0; f < E; ++f) if (el->line_of[f] == el->line_of[e]) ix->lineMask[e] |= 1ull << f;
  }
  // validate + per-chromosome ranges
  std::vector<u32> cs(n_chromosomes, 0), ce(n_chromosomes, 0);
  std::vector<int64_t> maxEnd(n_chromosomes, -1);
  std::vector<bool> seen(n_chromosomes, false);
  for (size_t i = 0; i < n_iv; ++i) {
    const int c = chr[i];
    if (c < 0 || c >= n_chromosomes) { delete ix; return fail(MM_ERR_ARG, "mm_index_create: chromosome id out of range"); }
    if (type[i] >= E) { delete ix; return fail(MM_ERR_ARG, "mm_index_create: element id out of range"); }
    if (start[i] < 0 || end[i] < 0 || start[i] > 0x7FFFFFF0 || end[i] > 0x7FFFFFF0) { delete ix; return fail(MM_ERR_RANGE, "mm_index_create: coordinate outside [0, 2^31)"); }
    if (i > 0 && (chr[i - 1] > c || (chr[i - 1] == c && start[i - 1] > start[i]))) { delete ix; return fail(MM_ERR_ARG, "mm_index_create: intervals must be sorted by (chromosome, start)"); }
    if (!seen[c]) { seen[c] = true; cs[c] = (u32) i; }
    ce[c] = (u32) i + 1;
    if (end[i] > maxEnd[c]) maxEnd[c] = end[i];
  }
  // bin width
  int shift = bin_shift;
  auto total_bins = [&](int sh) { u64 t = 0; for (int c = 0; c < n_chromosomes; ++c) if (seen[c]) t += (u64) (maxEnd[c] >> sh) + 2; return t; };
  if (shift <= 0) { shift = 10; while (total_bins(shift) > (6ull << 20)) ++shift; }
  if (shift > 30) { delete ix; return fail(MM_ERR_ARG, "mm_index_create: bin_shift too large"); }
  ix->shift = shift;
  std::vector<u32> binBase(n_chromosomes, 0), nBins(n_chromosomes, 0), ivEnd(n_chromosomes, 0);
  u64 nbt = 0;
  for (int c = 0; c < n_chromosomes; ++c) {
    if (!seen[c]) continue;
    nBins[c] = (u32) (maxEnd[c] >> shift) + 1; binBase[c] = (u32) nbt; ivEnd[c] = ce[c];
    nbt += nBins[c] + 1;
  }
  ix->n_bins = nbt;
  std::vector<uint2> bins(nbt);
  // active-list sizes: interval i is open at boundary b*W for b in [(start>>sh)+1, min((end+1)>>sh, nb-1)]
  std::vector<u32> actCnt(nbt, 0);
  for (size_t i = 0; i < n_iv; ++i) {
    const int c = chr[i];
    const u32 b0 = ((u32) start[i] >> shift) + 1, b1 = std::min<u32>(((u32) end[i] + 1u) >> shift, nBins[c] - 1);
    for (u32 b = b0; b <= b1 && b1 != 0xFFFFFFFFu; ++b) ++actCnt[binBase[c] + b];
  }
  u64 na = 0;
  for (u64 k = 0; k < nbt; ++k) { bins[k].y = (u32) na; na += actCnt[k]; }
  if (na >= 0xFFFFFFF0ull) { delete ix; return fail(MM_ERR_RANGE, "mm_index_create: active lists too large; raise bin_shift"); }
  ix->n_act = na;
  // records
  std::vector<int4> iv(n_iv), act(na ? na : 1);
  std::vector<u32> actIdx(na ? na : 1);
  for (int c = 0; c < n_chromosomes; ++c) {
    if (!seen[c]) continue;
    int pm = -1;
    for (u32 i = cs[c]; i < ce[c]; ++i) {
      iv[i] = make_int4(start[i], end[i], (int) type[i] | ((strand[i] == MM_STRAND_F) ? 0x100 : 0), pm);
      if (end[i] > pm) pm = end[i];
    }
    // first[b] = first interval with start >= b*W
    u32 p = cs[c];
    for (u32 b = 0; b <= nBins[c]; ++b) {
      const int64_t x = (int64_t) b << shift;
      while (p < ce[c] && start[p] < x) ++p;
      bins[binBase[c] + b].x = p;
    }
  }
  {
    std::vector<u32> cur(nbt);
    for (u64 k = 0; k < nbt; ++k) cur[k] = bins[k].y;
    for (size_t i = 0; i < n_iv; ++i) {
      const int c = chr[i];
      const u32 b0 = ((u32) start[i] >> shift) + 1, b1 = std::min<u32>(((u32) end[i] + 1u) >> shift, nBins[c] - 1);
      for (u32 b = b0; b <= b1 && b1 != 0xFFFFFFFFu; ++b) { actIdx[cur[binBase[c] + b]] = (u32) i; act[cur[binBase[c] + b]++] = iv[i]; }
    }
  }
  // ---- compact slot index (see slot_walk)
  std::vector<uint2> chrSlot(n_chromosomes, make_uint2(0u, 0u));
  std::vector<uint4> slots;
  std::vector<u32> slotBits, slotCls;
  std::vector<uint2> flat;
  std::vector<int> longEnd;
  std::vector<uint4> fslots;
  std::vector<u32> fbits, fcls, fext;
  std::vector<uint2> fchr;
  {
    // Bin width.  Bins holding more than 7 records leave the one-sector fast path (their list continues in `flat`),
    // so the widest bin width is chosen whose share of such bins stays below 2 %, among the widths whose table (64 B
    // per bin) fits 256 MB.  Record counts per bin are heavy-tailed (gene-dense regions), which is why the mean is
    // not the criterion.  (Measured against 128-B slots holding 15 records at twice the width - the same table size:
    // probe time -3.5 % HS38-shaped, -4 % TAIR10-shaped, -12 % Flybase6-shaped.)
    int best = -1, fallback = -1;
    for (int sh = 20; sh >= 7; --sh) {
      u64 nb = 0;
      std::vector<u64> base(n_chromosomes + 1, 0);
      for (int c = 0; c < n_chromosomes; ++c) { base[c] = nb; if (seen[c]) nb += (u64) (maxEnd[c] >> sh) + 2; }
      if (nb * 64ull > (256ull << 20)) break;
      fallback = sh;
      std::vector<int32_t> diff(nb + 2, 0), cnt(nb + 1, 0);
      for (size_t i = 0; i < n_iv; ++i) {
        if (end[i] < start[i]) continue;
        const u64 b0 = base[chr[i]] + ((u32) start[i] >> sh), b1 = base[chr[i]] + (((u32) end[i] + 1u) >> sh);
        ++cnt[b0];
        if (b1 > b0) { ++diff[b0 + 1]; --diff[b1 + 1]; }
      }
      u64 over = 0;
      int32_t run = 0;
      for (u64 k = 0; k <= nb; ++k) { run += diff[k]; if (cnt[k] + run > 7) ++over; }
      if ((double) over <= 0.02 * (double) nb) { best = sh; break; }
    }
    if (best < 0) best = fallback < 0 ? 20 : fallback;
    const int sh = bin_shift > 0 ? std::min(std::max(bin_shift, 7), 20) : best;
    ix->sshift = sh;
    std::vector<u32> sBase(n_chromosomes, 0), sN(n_chromosomes, 0);
    u64 ns = 0;
    for (int c = 0; c < n_chromosomes; ++c) {
      if (!seen[c]) continue;
      sN[c] = (u32) (maxEnd[c] >> sh) + 1; sBase[c] = (u32) ns; ns += sN[c];
      chrSlot[c] = make_uint2(sBase[c], sN[c]);
    }
    // Reach: a slot also lists, after its own records, the intervals that start within the first `reach` bases of the next bin, so
    // that a read ending there (every short read that crosses a bin boundary) is served by the one slot of its start.  Without it
    // nearly every wave walked a second bin for the tenth of its lanes that cross - a quarter of the probe's instructions.
    const u32 W = 1u << sh, reach = std::min<u32>(W, 256u);
    ix->reach = reach;
    std::vector<u32> nAct(ns, 0), nIv(ns, 0), nOvh(ns, 0);
    for (size_t i = 0; i < n_iv; ++i) {
      if (end[i] < start[i]) continue;   // inverted zero-size intervals (e.g. -d 0) can only concern reads without CIGAR operations
      const int c = chr[i];
      const u32 bs = (u32) start[i] >> sh;
      ++nIv[sBase[c] + bs];
      if (bs > 0 && ((u32) start[i] & (W - 1)) < reach) ++nOvh[sBase[c] + bs - 1];
      const u32 b0 = bs + 1, b1 = std::min<u32>(((u32) end[i] + 1u) >> sh, sN[c] - 1);
      for (u32 b = b0; b <= b1; ++b) ++nAct[sBase[c] + b];
    }
    // flat: per slot one count record {nOvh, 0} and then act ++ iv ++ reach records; off[k] = the slot's first record, never below 8
    // (the slot header's second word is the offset when the list is longer than the 7 inline records, the reach count otherwise)
    std::vector<u64> off(ns + 1, 0);
    u64 running = 8;
    for (u64 k = 0; k < ns; ++k) { off[k] = running + 1; running = off[k] + nAct[k] + nIv[k] + nOvh[k]; }
    off[ns] = running;
    if (off[ns] >= 0xFFFFFFF0ull) { delete ix; return fail(MM_ERR_RANGE, "mm_index_create: slot lists too large; raise bin_shift"); }
    flat.assign(off[ns], make_uint2(0u, 0u));
    std::vector<u32> curA(ns), curI(ns), curO(ns);
    for (u64 k = 0; k < ns; ++k) { curA[k] = (u32) off[k]; curI[k] = curA[k] + nAct[k]; curO[k] = curI[k] + nIv[k]; flat[off[k] - 1] = make_uint2(nOvh[k], 0u); }
    for (size_t i = 0; i < n_iv; ++i) {
      if (end[i] < start[i]) continue;
      const int c = chr[i];
      const u32 t = type[i];
      const bool ivF = strand[i] == MM_STRAND_F;
      const bool eF = (ix->elemF >> t) & 1, eR = (ix->elemR >> t) & 1;
      const bool okPlus = !(eF | eR) || (eF ? ivF : !ivF), okMinus = !(eF | eR) || (eF ? !ivF : ivF);
      u32 len = (u32) (end[i] - start[i]), y = 0;
      if (len >= REC_LEN_MASK) { y |= REC_LONG; len = (u32) longEnd.size(); longEnd.push_back(end[i]); }
      y |= len | (t << REC_LEN_BITS) | ((okMinus ? 1u : 0u) << REC_OKBIT) | ((okPlus ? 1u : 0u) << (REC_OKBIT + 1));
      const uint2 rec = make_uint2((u32) start[i], y);
      const u32 bs = (u32) start[i] >> sh;
      flat[curI[sBase[c] + bs]++] = rec;
      if (bs > 0 && ((u32) start[i] & (W - 1)) < reach) flat[curO[sBase[c] + bs - 1]++] = rec;
      const u32 b0 = bs + 1, b1 = std::min<u32>(((u32) end[i] + 1u) >> sh, sN[c] - 1);
      for (u32 b = b0; b <= b1; ++b) flat[curA[sBase[c] + b]++] = rec;
    }
    if (longEnd.size() >= REC_LEN_MASK) { delete ix; return fail(MM_ERR_RANGE, "mm_index_create: too many very long intervals"); }
    slots.assign(ns * 4, make_uint4(0u, 0u, 0u, 0u));
    // the slots' map, 2 bits per bin: FS_EMPTY (no line to fetch), FS_GENERAL, or FS_COVERED - the bin holds nothing but one interval open at
    // its left edge that reaches past its reach, so every read the one slot serves lies inside that interval and the answer is the
    // interval's element if the strands allow: element and ok bits are the group's class byte (one per 8 bins; bins of a group covered by
    // different elements stay FS_GENERAL), and no slot is fetched (round 3; as for the fast slots below).  Built as two arrays here,
    // merged into one word per 8 bins for the device (merge_map)
    slotBits.assign((ns + 15) / 16 + 1, 0u);
    slotCls.assign(((ns + 7) / 8 + 3) / 4 + 1, 0u);
    u64 nCovered = 0;
    {
      unsigned char *cls = (unsigned char *) slotCls.data();
      std::vector<unsigned char> have_cls((ns + 7) / 8, 0);
      for (int c = 0; c < n_chromosomes; ++c) {
        if (!seen[c]) continue;
        for (u32 b = 0; b < sN[c]; ++b) {
          const u64 k = (u64) sBase[c] + b;
          const u32 total = nAct[k] + nIv[k] + nOvh[k];
          if (!total) continue;
          u32 code = FS_GENERAL;
          if (total == 1 && nAct[k] == 1) {
            const uint2 r = flat[off[k]];
            const int64_t ie = (r.y & REC_LONG) ? (int64_t) longEnd[r.y & REC_LEN_MASK] : (int64_t) r.x + (int64_t) (r.y & REC_LEN_MASK);
            const unsigned char cl = (unsigned char) ((REC_TYPE(r.y) << 2) | ((r.y >> REC_OKBIT) & 3u));
            if (ie >= ((int64_t) b << sh) + (int64_t) W + (int64_t) reach - 1 && b + 1 < sN[c]) {
              if (!have_cls[k >> 3]) { have_cls[k >> 3] = 1; cls[k >> 3] = cl; }
              if (cls[k >> 3] == cl) { code = FS_COVERED; ++nCovered; }
            }
          }
          slotBits[k >> 4] |= code << ((k & 15) * 2);
        }
      }
    }
    for (u64 k = 0; k < ns; ++k) {
      const u32 total = nAct[k] + nIv[k] + nOvh[k];
      u32 *w = (u32 *) &slots[k * 4];
      w[0] = nAct[k] | (nIv[k] << 16);
      w[1] = total <= 7 ? nOvh[k] : (u32) off[k];
      if (nAct[k] > 0xFFFFu || nIv[k] > 0xFFFFu) { delete ix; return fail(MM_ERR_RANGE, "mm_index_create: more than 65535 intervals in one bin; lower bin_shift"); }
      const u32 n = std::min<u32>(total, 7u);
      for (u32 r = 0; r < n; ++r) { w[2 + 2 * r] = flat[off[k] + r].x; w[3 + 2 * r] = flat[off[k] + r].y; }
    }
    u64 nonEmpty = 0;
    for (u64 k = 0; k < ns; ++k) nonEmpty += (nAct[k] + nIv[k] + nOvh[k]) ? 1 : 0;
    if ((double) (ns - nonEmpty + nCovered) < 0.33 * (double) ns) { slotBits.clear(); slotCls.clear(); }   // (the look at the map is a dependent L2 access: it has to save a third of the fetches)
    // (Measured and rejected, round 2: storing the non-empty bins only - slot found through a running count kept beside each bitmap
    // word, one 8-byte L2 access for "empty?" and "where?".  The HS38-scale table shrinks from 193 to 96 MB, yet the probe got 11 %
    // SLOWER (0.354 -> 0.394 ms per 2^24 tuples): the rank arithmetic sits on the dependent chain and costs the 64-VGPR budget a spill.)
    // code bits and class byte of a group of 8 bins in one word: what the kernels read (one L2 access, not two)
    if (!slotBits.empty()) slotBits = merge_map(slotBits, slotCls, ns);
    ix->n_slots = ns; ix->n_flat = off[ns]; ix->n_slotBits = slotBits.size();
  }
  // ---- fast slots (k_probe_fast): per bin 32 bytes = header + up to 7 records of 4 bytes, relative to the bin's first base.
  //   header = number of records | number of those that are open at the bin's left edge (the first ones) << 4, or FS_OVERFLOW | extension block
  //   record = element:6 << 26 | ok('+' read):1 << 25 | ok('-' read):1 << 24 | so : shift + 1 bits | eo above,
  //            so = start - binStart (0 for an interval open at the bin's left edge), eo = min(end - binStart, ESAT); unused records
  //            are zero and pass no test (neither ok bit set).
  // A saturated eo says "at least": the kernel sends only reads to these slots that end before a saturated end could matter (read
  // length <= fLmax = ESAT - 2 W, start inside the chromosome's bins), so every test stays exact.  Intervals that end before the
  // bin starts cannot pass for a read with end >= start and are left out; inverted zero-size intervals too (as above).  More than 7
  // records: the overflow mark (the tuple then takes the 8-byte slots above, as do reads that are too long, span more than the slot's
  // reach, have no CIGAR operations, or need the second selection pass).  The bin width is this table's own: half the width of the
  // 8-byte slots' where that cuts the share of overflowing bins below FS_OVERFLOW_SHARE (4-byte records: the table is no larger for
  // it), never below 128 bases (a read has to fit a bin and its reach), at most 256 MB.
  {
    auto count_fast = [&](int sh, std::vector<u32> &fBase, std::vector<u32> &fN, std::vector<u32> &cA, std::vector<u32> &cI, std::vector<u32> &cO) -> u64 {
      const u32 W = 1u << sh, reach = std::min<u32>(W, 256u);
      fBase.assign(n_chromosomes, 0); fN.assign(n_chromosomes, 0);
      u64 ns = 0;
      for (int c = 0; c < n_chromosomes; ++c) if (seen[c]) { fN[c] = (u32) (maxEnd[c] >> sh) + 1; fBase[c] = (u32) ns; ns += fN[c]; }
      cA.assign(ns + 1, 0); cI.assign(ns, 0); cO.assign(ns, 0);
      std::vector<int32_t> diff(ns + 2, 0);
      for (size_t i = 0; i < n_iv; ++i) {
        if (end[i] < start[i]) continue;
        const int c = chr[i];
        const u32 bs = (u32) start[i] >> sh;
        ++cI[fBase[c] + bs];
        if (bs > 0 && ((u32) start[i] & (W - 1)) < reach) ++cO[fBase[c] + bs - 1];
        // open at the left edge of bins bs+1 .. (end >> sh): the bin that starts at end + 1 would only see it end before its first base
        const u32 b1 = std::min<u32>((u32) end[i] >> sh, fN[c] - 1);
        if (b1 > bs) { ++diff[fBase[c] + bs + 1]; --diff[fBase[c] + b1 + 1]; }
      }
      int32_t run = 0;
      for (u64 k = 0; k < ns; ++k) { run += diff[k]; cA[k] = (u32) run; }
      return ns;
    };
    std::vector<u32> fBase, fN, cA, cI, cO;
    int fsh = -1;
    u64 fns = 0;
    const char *forced = getenv("MMANNOT_AMD_FAST_SHIFT");       // (A/B measurements)
    for (int sh = std::min(ix->sshift, FS_MAX_SHIFT); sh >= FS_MIN_SHIFT; --sh) {
      if (forced && atoi(forced) >= FS_MIN_SHIFT && atoi(forced) <= FS_MAX_SHIFT) sh = atoi(forced);
      const u64 ns = count_fast(sh, fBase, fN, cA, cI, cO);
      if (ns * 32ull > (256ull << 20) && fsh >= 0) break;       // (would be too large: keep the wider one)
      fsh = sh; fns = ns;
      u64 over = 0;
      for (u64 k = 0; k < ns; ++k) over += (cA[k] + cI[k] + cO[k] > 7) ? 1 : 0;
      if (forced || sh < ix->sshift || (double) over <= FS_OVERFLOW_SHARE * (double) ns) break;   // at most one halving
    }
    if (fsh >= 0 && fns * 32ull <= (1024ull << 20)) {
      const u64 ns = count_fast(fsh, fBase, fN, cA, cI, cO);   // (the loop may have ended on a wider table than it counted last)
      const int sh = fsh;
      const u32 W = 1u << sh, reach = std::min<u32>(W, 256u);
      const u32 eoBits = 24u - (u32) (sh + 1), ESAT = (1u << eoBits) - 1u;
      ix->fshift = sh; ix->freach = reach; ix->fLmax = ESAT - 2 * W; ix->n_fslots = ns;
      fslots.assign(ns * 2, make_uint4(0u, 0u, 0u, 0u));
      fbits.assign((ns + 15) / 16 + 1, 0u);               // the map: 2 bits per bin (FS_EMPTY / FS_GENERAL / FS_COVERED)
      fcls.assign(((ns + 7) / 8 + 3) / 4 + 1, 0u);        // one byte per 8 bins: the covering interval's element and ok bits
      fchr.assign(n_chromosomes, make_uint2(0u, 0u));
      for (int c = 0; c < n_chromosomes; ++c) if (seen[c]) fchr[c] = make_uint2(fBase[c], fN[c]);
      // headers, then the records through per-bin cursors: open ones first, then those starting in the bin, then the reach - each in index order.
      // A bin with 8 to FS_EXT_RECORDS records keeps them in an extension block of 128 bytes (count + records, same format), which the
      // kernel walks for its own tile's tuples when its main loop is through; its slot only points there.
      std::vector<unsigned char> curA(ns, 0), curI(ns, 0), curO(ns, 0);
      // (Blocks are numbered by size class - up to 11, 15, 23, 31 records = 3, 4, 6, 8 quads of 16 bytes - so that the kernel knows from
      // a block's number how much of it to request: most overflowing bins hold 8 to 11 records.)
      u64 n_ext = 0;
      const bool no_ext = getenv("MMANNOT_AMD_NO_EXT") != nullptr;       // (A/B: every bin with more than 7 records goes to k_probe_list)
      static const u32 cls_max[4] = {11u, 15u, 23u, FS_EXT_RECORDS};
      for (int cl = 0; cl < 4; ++cl) {
        for (u64 k = 0; k < ns && !no_ext; ++k) {
          const u32 tot = cA[k] + cI[k] + cO[k];
          if (tot > (cl ? cls_max[cl - 1] : 7u) && tot <= cls_max[cl] && n_ext < FS_EXT_NONE) ((u32 *) &fslots[k * 2])[0] = FS_OVERFLOW | (u32) n_ext++;
        }
        ix->fextClass[cl] = (u32) n_ext;                  // blocks below this number belong to class cl or a smaller one
      }
      for (u64 k = 0; k < ns; ++k) {
        const u32 tot = cA[k] + cI[k] + cO[k];
        u32 *w = (u32 *) &fslots[k * 2];
        if (tot > 7) { if (!(w[0] & FS_OVERFLOW)) w[0] = FS_OVERFLOW | FS_EXT_NONE; }
        else w[0] = tot | (cA[k] << 4);
        curI[k] = (unsigned char) cA[k]; curO[k] = (unsigned char) (cA[k] + cI[k]);
        if (tot) fbits[k >> 4] |= (u32) FS_GENERAL << ((k & 15) * 2);
      }
      fext.assign((n_ext + 1) * 32, 0u);
      for (u64 k = 0; k < ns; ++k) {
        const u32 w0 = ((const u32 *) &fslots[k * 2])[0];
        if ((w0 & FS_OVERFLOW) && (w0 & FS_EXT_NONE) != FS_EXT_NONE) fext[(size_t) (w0 & FS_EXT_NONE) * 32] = (cA[k] + cI[k] + cO[k]) | (cA[k] << 8);
      }
      for (size_t i = 0; i < n_iv; ++i) {
        if (end[i] < start[i]) continue;
        const int c = chr[i];
        const u32 t = type[i];
        const bool ivF = strand[i] == MM_STRAND_F;
        const bool eF = (ix->elemF >> t) & 1, eR = (ix->elemR >> t) & 1;
        const bool okPlus = !(eF | eR) || (eF ? ivF : !ivF), okMinus = !(eF | eR) || (eF ? !ivF : ivF);
        const u32 head = (t << 26) | ((okPlus ? 1u : 0u) << 25) | ((okMinus ? 1u : 0u) << 24);
        const u32 bs = (u32) start[i] >> sh;
        auto rec = [&](u32 b, bool open) -> u32 {            // the interval as bin b sees it
          const int64_t binStart = (int64_t) b << sh;
          const u32 so = open ? 0u : (u32) ((int64_t) start[i] - binStart);
          const u32 eo = (u32) std::min<int64_t>((int64_t) end[i] - binStart, (int64_t) ESAT);
          return head | so | (eo << (sh + 1));
        };
        auto put = [&](u64 k, unsigned char &cur, u32 r) {
          u32 *w = (u32 *) &fslots[k * 2];
          if (!(w[0] & FS_OVERFLOW)) w[1 + cur++] = r;
          else if ((w[0] & FS_EXT_NONE) != FS_EXT_NONE) fext[(size_t) (w[0] & FS_EXT_NONE) * 32 + 1 + cur++] = r;
        };
        { const u64 k = (u64) fBase[c] + bs; put(k, curI[k], rec(bs, false)); }
        if (bs > 0 && ((u32) start[i] & (W - 1)) < reach) { const u64 k = (u64) fBase[c] + bs - 1; put(k, curO[k], rec(bs - 1, false)); }
        const u32 b1 = std::min<u32>((u32) end[i] >> sh, fN[c] - 1);
        for (u32 b = bs + 1; b <= b1; ++b) { const u64 k = (u64) fBase[c] + b; put(k, curA[k], rec(b, true)); }
      }
      // Bins that one open interval covers from their first base to the end of their reach, and that hold nothing else: every read this
      // table serves there lies inside that interval (its end is below binStart + 2 W), so the answer follows from the interval's element
      // and ok bits alone - which the map keeps (code FS_COVERED + one byte per 8 bins, both small enough to stay in L2): no slot is
      // fetched for them.  A quarter of the reads of an HS38-shaped annotation fall on such bins (long introns, gene bodies): half of its
      // random 128-byte fetches.  Bins of one group of 8 that are covered by intervals of different elements keep their slots.
      unsigned char *cls = (unsigned char *) fcls.data();
      std::vector<unsigned char> have_cls((ns + 7) / 8, 0);
      for (u64 k = 0; k < ns; ++k) {
        const u32 *w = (const u32 *) &fslots[k * 2];
        if (w[0] != 1u || cA[k] != 1u) continue;
        const u32 r = w[1], eo = (r & 0xFFFFFFu) >> (sh + 1);
        if (eo < 2 * W - 1) continue;
        const unsigned char cl = (unsigned char) (r >> 24);
        if (!have_cls[k >> 3]) { have_cls[k >> 3] = 1; cls[k >> 3] = cl; }
        if (cls[k >> 3] != cl) continue;
        fbits[k >> 4] = (fbits[k >> 4] & ~(3u << ((k & 15) * 2))) | ((u32) FS_COVERED << ((k & 15) * 2));
      }
    }
  }
  std::vector<uint4> chrInfo(n_chromosomes);
  for (int c = 0; c < n_chromosomes; ++c) chrInfo[c] = make_uint4(binBase[c], nBins[c], ivEnd[c], 0u);
  auto up = [&](void **d, const void *h, size_t bytes) -> hipError_t {
    hipError_t e = hipMalloc(d, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    return bytes ? hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  hipError_t e;
  if ((e = up((void **) &ix->d_iv, iv.data(), n_iv * sizeof(int4))) != hipSuccess ||
      (e = up((void **) &ix->d_act, act.data(), na * sizeof(int4))) != hipSuccess ||
      (e = up((void **) &ix->d_actIdx, actIdx.data(), na * sizeof(u32))) != hipSuccess ||
      (e = up((void **) &ix->d_bins, bins.data(), nbt * sizeof(uint2))) != hipSuccess ||
      (e = up((void **) &ix->d_chrInfo, chrInfo.data(), (size_t) n_chromosomes * sizeof(uint4))) != hipSuccess ||
      (e = up((void **) &ix->d_lineMask, ix->lineMask, 64 * sizeof(u64))) != hipSuccess ||
      (e = up((void **) &ix->d_slots, slots.data(), slots.size() * sizeof(uint4))) != hipSuccess ||
      (e = up((void **) &ix->d_flat, flat.data(), flat.size() * sizeof(uint2))) != hipSuccess ||
      (!slotBits.empty() && (e = up((void **) &ix->d_slotBits, slotBits.data(), slotBits.size() * sizeof(u32))) != hipSuccess) ||
      (e = up((void **) &ix->d_chrSlot, chrSlot.data(), (size_t) n_chromosomes * sizeof(uint2))) != hipSuccess ||
      (e = up((void **) &ix->d_longEnd, longEnd.data(), longEnd.size() * sizeof(int))) != hipSuccess ||
      (!fslots.empty() && ((e = up((void **) &ix->d_fslots, fslots.data(), fslots.size() * sizeof(uint4))) != hipSuccess ||
                           (e = up((void **) &ix->d_fbits, fbits.data(), fbits.size() * sizeof(u32))) != hipSuccess ||
                           (e = up((void **) &ix->d_fcls, fcls.data(), fcls.size() * sizeof(u32))) != hipSuccess ||
                           (e = up((void **) &ix->d_fext, fext.data(), fext.size() * sizeof(u32))) != hipSuccess ||
                           (e = up((void **) &ix->d_fchr, fchr.data(), fchr.size() * sizeof(uint2))) != hipSuccess))) {
    mm_index_destroy(ix);
    return fail(MM_ERR_HIP, std::string("mm_index_create upload: ") + hipGetErrorString(e));
  }
  *out = ix;
  return MM_OK;
}

extern "C" void mm_index_destroy(mm_index *ix) {
  if (!ix) return;
  (void) hipFree(ix->d_iv); (void) hipFree(ix->d_act); (void) hipFree(ix->d_actIdx); (void) hipFree(ix->d_bins);
  (void) hipFree(ix->d_chrInfo); (void) hipFree(ix->d_lineMask); (void) hipFree(ix->d_slots); (void) hipFree(ix->d_flat); (void) hipFree(ix->d_chrSlot); (void) hipFree(ix->d_longEnd);
  (void) hipFree(ix->d_slotBits); (void) hipFree(ix->d_fslots); (void) hipFree(ix->d_fbits); (void) hipFree(ix->d_fcls); (void) hipFree(ix->d_fext); (void) hipFree(ix->d_fchr);
  delete ix;
}

extern "C" int mm_index_info(const mm_index *ix, uint64_t info[8]) {
  if (!ix || !info) return fail(MM_ERR_ARG, "mm_index_info: null argument");
  info[0] = ix->n_iv * sizeof(int4); info[1] = ix->n_bins * sizeof(uint2); info[2] = ix->n_act * sizeof(int4); info[3] = (u64) ix->shift;
  info[4] = ix->n_slots * 64ull; info[5] = ix->n_flat * sizeof(uint2); info[6] = (u64) ix->sshift; info[7] = ix->n_slotBits * sizeof(u32);
  return MM_OK;
}

static void ctx_free_batch(mm_ctx *c) {
  for (int b = 0; b < 2; ++b) { (void) hipFree(c->d_alnBuf[b]); c->d_alnBuf[b] = nullptr; c->consumed_valid[b] = false; c->copied_valid[b] = false; }
  for (int b = 0; b < 2; ++b) { (void) hipFree(c->d_pkBuf[b]); c->d_pkBuf[b] = nullptr; }
  c->pk_cap = 0;
  (void) hipFree(c->d_hm); (void) hipFree(c->d_gkey); (void) hipFree(c->d_gpay);
  (void) hipFree(c->d_kA); (void) hipFree(c->d_kB);
  (void) hipFree(c->d_hist); (void) hipFree(c->d_tileTot); (void) hipFree(c->d_tileK);
  (void) hipFree(c->d_dlist); (void) hipFree(c->d_dcnt);
  c->d_tileTot = nullptr; c->d_tileK = nullptr; c->d_dlist = nullptr; c->d_dcnt = nullptr;
  c->d_aln = nullptr; c->d_hm = nullptr; c->d_gkey = nullptr; c->d_gpay = nullptr;
  c->d_kA = c->d_vA = c->d_kB = c->d_vB = nullptr; c->d_hist = nullptr;
  c->cap = 0;
}

extern "C" int mm_ctx_create(mm_ctx **out, const mm_index *ix, const mm_options *opt, void *stream) {
  if (!out || !ix || !opt) return fail(MM_ERR_ARG, "mm_ctx_create: null argument");
  *out = nullptr;
  if (opt->strategy == MM_STRAT_RANDOM && !opt->keep_hit_masks)
    return fail(MM_ERR_ARG, "mm_ctx_create: -y random replays libc rand() in file order (mmannot.cpp:1706-1726): the device only provides "
                            "the per-hit element sets and counters for it, so keep_hit_masks is required");
  if (opt->strategy < 0 || opt->strategy > MM_STRAT_RATIO || opt->overlap_mode < 0 || opt->overlap_mode > MM_OVL_NT ||
      opt->strandedness < 0 || opt->strandedness > MM_STRANDED_R)
    return fail(MM_ERR_ARG, "mm_ctx_create: bad option value");
  int rc = device_check(ix->device);
  if (rc) return rc;
  mm_ctx *c = new mm_ctx();
  c->ix = ix; c->opt = *opt;
  {
    auto on = [](const char *name) { const char *e = getenv(name); return e && *e && *e != '0'; };
    c->env_no_fused_hist = on("MMANNOT_AMD_NO_FUSED_HIST"); c->env_general_probe = on("MMANNOT_AMD_GENERAL_PROBE");
    c->env_full_index_only = on("MMANNOT_AMD_FULL_INDEX_ONLY"); c->env_no_fast = on("MMANNOT_AMD_NO_FAST"); c->env_fast = on("MMANNOT_AMD_FAST");
  }
  if (stream) c->stream = (hipStream_t) stream;
  else { HIPCHK(hipStreamCreate(&c->stream)); c->own_stream = true; }
  HIPCHK(hipMalloc((void **) &c->G.key, GT_CAP * 8)); HIPCHK(hipMalloc((void **) &c->G.cnt, GT_CAP * 8));
  HIPCHK(hipMalloc((void **) &c->G.sum, GT_CAP * 8)); HIPCHK(hipMalloc((void **) &c->G.err, 4));
  c->G.nearProbes = c->opt.strategy == MM_STRAT_RATIO ? GT_NEAR_PROBES : 0u;
  HIPCHK(hipMalloc((void **) &c->d_stats, 8 * sizeof(u64))); HIPCHK(hipMalloc((void **) &c->d_walk, 4 * sizeof(u64)));
  HIPCHK(hipMalloc((void **) &c->d_rows, (1 + 2 * (size_t) GT_CAP + ROWS_WG) * 8));   // (+ the workgroups' row counts)
  HIPCHK(hipHostMalloc((void **) &c->h_rows, (1 + 2 * (size_t) GT_CAP) * 8, hipHostMallocDefault));
  HIPCHK(hipMalloc((void **) &c->d_rowTot, (RS_ROW_COUNT + 2) * 4));   // + the scan's "workgroups done" counter
  HIPCHK(hipMemset(c->d_rowTot, 0, (RS_ROW_COUNT + 2) * 4)); HIPCHK(hipMalloc((void **) &c->d_M, INFO_WORDS * 4));
  HIPCHK(hipMalloc((void **) &c->d_seg, 2 * GT_CAP * 4));
  *out = c;
  return mm_reset(c);
}

extern "C" void mm_ctx_destroy(mm_ctx *c) {
  if (!c) return;
  (void) hipSetDevice(c->ix->device);
  (void) hipStreamSynchronize(c->stream);
  for (auto &s : c->spans) { (void) hipEventDestroy(s.a); (void) hipEventDestroy(s.b); }
  for (auto &t : c->tune_pending) { (void) hipEventDestroy(t.a); (void) hipEventDestroy(t.b); }
  free_runs(c);
  free_run_pool(c);
  (void) hipFree(c->d_xkeys); (void) hipFree(c->d_xsums);
  if (c->ev_handoff) (void) hipEventDestroy(c->ev_handoff);
  ctx_free_batch(c);
  (void) hipFree(c->G.key); (void) hipFree(c->G.cnt); (void) hipFree(c->G.sum); (void) hipFree(c->G.err);
  (void) hipFree(c->d_rows); if (c->h_rows) (void) hipHostFree(c->h_rows);
  (void) hipFree(c->d_stats); (void) hipFree(c->d_walk); (void) hipFree(c->d_rowTot); (void) hipFree(c->d_M); (void) hipFree(c->d_seg);
  (void) hipFree(c->d_grpRem); (void) hipFree(c->d_grpU);
  (void) hipFree(c->d_bLo); (void) hipFree(c->d_bHi); (void) hipFree(c->d_bList);
  for (int b = 0; b < 2; ++b) { (void) hipFree(c->d_big[b]); (void) hipHostFree(c->h_big[b]); if (c->ev_big[b]) (void) hipEventDestroy(c->ev_big[b]); }
  if (c->copy_stream) {
    (void) hipStreamSynchronize(c->copy_stream);
    for (int b = 0; b < 2; ++b) { (void) hipEventDestroy(c->ev_copied[b]); (void) hipEventDestroy(c->ev_consumed[b]); }
    (void) hipStreamDestroy(c->copy_stream);
  }
  if (c->own_stream) (void) hipStreamDestroy(c->stream);
  delete c;
}

static int ensure_qids(mm_ctx *c, u64 qid_bound);
extern "C" int mm_reserve_qids(mm_ctx *c, uint64_t qid_bound) {
  if (!c) return fail(MM_ERR_ARG, "mm_reserve_qids: null context");
  HIPCHK(hipSetDevice(c->ix->device));
  return c->opt.strategy == MM_STRAT_DEFAULT ? ensure_qids(c, qid_bound ? qid_bound : 1) : MM_OK;
}

extern "C" int mm_reset(mm_ctx *c) {
  if (!c) return fail(MM_ERR_ARG, "mm_reset: null context");
  HIPCHK(hipSetDevice(c->ix->device));
  HIPCHK(hipMemsetAsync(c->G.key, 0, GT_CAP * 8, c->stream)); HIPCHK(hipMemsetAsync(c->G.cnt, 0, GT_CAP * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->G.sum, 0, GT_CAP * 8, c->stream)); HIPCHK(hipMemsetAsync(c->G.err, 0, 4, c->stream));
  HIPCHK(hipMemsetAsync(c->d_stats, 0, 8 * sizeof(u64), c->stream)); HIPCHK(hipMemsetAsync(c->d_walk, 0, 4 * sizeof(u64), c->stream));
  if (c->qcap) {
    HIPCHK(hipMemsetAsync(c->d_grpRem, 0, c->qcap * 4, c->stream)); HIPCHK(hipMemsetAsync(c->d_grpU, 0, c->qcap * 8, c->stream));
  }
  free_runs(c);
  c->finalized = false;
  c->last_n = 0;
  return MM_OK;
}

static int ensure_batch(mm_ctx *c, size_t n, bool need_copy) {
  if (n <= c->cap && (!need_copy || c->d_alnBuf[0])) return MM_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->copy_stream) HIPCHK(hipStreamSynchronize(c->copy_stream));
  size_t cap = std::max<size_t>(n, 1 << 16);
  if (n > c->cap) {
    ctx_free_batch(c);
    const size_t nblk = (cap + RS_TILE - 1) / RS_TILE;
    HIPCHK(hipMalloc((void **) &c->d_hm, cap * 8)); HIPCHK(hipMalloc((void **) &c->d_gkey, cap * 4)); HIPCHK(hipMalloc((void **) &c->d_gpay, cap * 4));
    // (keys and payloads of one buffer are the two halves of one allocation: the leaf layout's partition pass writes 8-byte pairs there)
    HIPCHK(hipMalloc((void **) &c->d_kA, cap * 8 + 2 * GT_CAP * 4 + 64)); c->d_vA = c->d_kA + cap;   // (laid out like a parked batch's buffer: -y ratio with defer_ratio hands it over)
    free_run_pool(c);                                       // (buffers of the old capacity)
    HIPCHK(hipMalloc((void **) &c->d_kB, cap * 8)); c->d_vB = c->d_kB + cap;
    HIPCHK(hipMalloc((void **) &c->d_hist, (size_t) RS_ROWS * nblk * 4));
    HIPCHK(hipMalloc((void **) &c->d_dlist, 2 * nblk * (size_t) RS_TILE * 4)); HIPCHK(hipMalloc((void **) &c->d_dcnt, 16)); HIPCHK(hipMemsetAsync(c->d_dcnt, 0, 16, c->stream));
    HIPCHK(hipMalloc((void **) &c->d_tileTot, 2 * (cap / RATIO_CHUNK + 2) * 8)); HIPCHK(hipMalloc((void **) &c->d_tileK, 2 * (cap / RATIO_CHUNK + 2) * 4));
    c->cap = cap;
  }
  if (need_copy && !c->d_alnBuf[0]) {
    for (int b = 0; b < 2; ++b) HIPCHK(hipMalloc((void **) &c->d_alnBuf[b], c->cap * sizeof(int4)));
    if (!c->copy_stream) {
      HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
      for (int b = 0; b < 2; ++b) { HIPCHK(hipEventCreateWithFlags(&c->ev_copied[b], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_consumed[b], hipEventDisableTiming)); }
    }
  }
  c->d_aln = c->d_alnBuf[0];
  return MM_OK;
}

static int ensure_qids(mm_ctx *c, u64 qid_bound) {
  if (qid_bound <= c->qcap) return MM_OK;
  if (qid_bound > 0xFFFFFFF0ull) return fail(MM_ERR_RANGE, "qid_bound exceeds 2^32");
  u64 ncap = std::max<u64>(qid_bound + qid_bound / 2, 1 << 16);
  ncap = std::min<u64>(ncap, 0xFFFFFFF0ull);
  u32 *nr = nullptr; u64 *nu = nullptr;
  HIPCHK(hipMalloc((void **) &nr, ncap * 4)); HIPCHK(hipMalloc((void **) &nu, ncap * 8));
  HIPCHK(hipMemsetAsync(nr, 0, ncap * 4, c->stream)); HIPCHK(hipMemsetAsync(nu, 0, ncap * 8, c->stream));
  if (c->qcap) {
    HIPCHK(hipMemcpyAsync(nr, c->d_grpRem, c->qcap * 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(nu, c->d_grpU, c->qcap * 8, hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  (void) hipFree(c->d_grpRem); (void) hipFree(c->d_grpU);
  c->d_grpRem = nr; c->d_grpU = nu; c->qcap = ncap;
  (void) hipFree(c->d_bLo); (void) hipFree(c->d_bHi); (void) hipFree(c->d_bList);
  c->nbcap = (ncap >> LEAF_BITS) + 2;
  HIPCHK(hipMalloc((void **) &c->d_bLo, c->nbcap * 4)); HIPCHK(hipMalloc((void **) &c->d_bHi, c->nbcap * 4)); HIPCHK(hipMalloc((void **) &c->d_bList, c->nbcap * 4));
  return MM_OK;
}

static void fill_params(const mm_ctx *c, ProbeParams &P) {
  const mm_index *ix = c->ix;
  P.iv = ix->d_iv; P.act = ix->d_act; P.bins = ix->d_bins;
  P.chrInfo = ix->d_chrInfo;
  P.fslots = ix->d_fslots; P.fbits = ix->d_fbits; P.fcls = ix->d_fcls; P.fext = (const uint4 *) ix->d_fext; P.fextClass[0] = ix->fextClass[0]; P.fextClass[1] = ix->fextClass[1]; P.fextClass[2] = ix->fextClass[2]; P.fchr = ix->d_fchr; P.fLmax = ix->fLmax; P.fshift = (u32) ix->fshift; P.freach = ix->freach;
  P.slots = ix->d_slots; P.flat = ix->d_flat; P.chrSlot = ix->d_chrSlot; P.slotBits = ix->d_slotBits; P.longEnd = ix->d_longEnd; P.sshift = (u32) ix->sshift; P.reach = ix->reach;
  P.useSlots = c->env_full_index_only ? 0 : 1;
  P.nchr = (u32) ix->nchr; P.shift = (u32) ix->shift;
  P.elemF = ix->elemF; P.elemR = ix->elemR; P.elemUp = ix->elemUp; P.elemDown = ix->elemDown;
  P.overlapMode = c->opt.overlap_mode; P.overlapValue = c->opt.overlap_value;
  P.strandedness = c->opt.strandedness; P.strategy = c->opt.strategy; P.keep = c->opt.keep_hit_masks;
  P.lineMask = ix->d_lineMask;
}

static int bits_for(u64 bound) {
  int b = 1;
  while (b < 32 && (1ull << b) < bound) ++b;
  return b;
}

// The stable passes over the batch's (key, payload) pairs: `passes` of them are launched; those the batch's actual key range does
// not need return immediately (device-side decision, no host round trip) and the consumer works out from the same device
// scalars whether the result sits in A or B.  rebase = 1: -y default (keys qid - qmin, leaf layout: 8-byte pairs between the
// passes); 0: -y ratio's fixed 16 key bits.
static int sort_pairs(mm_ctx *c, size_t n, int passes, int fixed_bits, int rebase, bool have_hist = false) {
  const u32 nblk = (u32) ((n + RS_TILE - 1) / RS_TILE);
  // first pass: straight from k_probe's (gkey, gpay), filtering and compacting on the fly -> (kB, vB)
  // (have_hist: k_probe has counted the tiles' buckets itself - one-pass leaf layout taken for granted, no order check, no count row)
  // (-y ratio with k_probe's histogram: valid for the one-pass layout only - the kernel returns at once unless the batch has a wide key)
  if (!have_hist || fixed_bits)
    hipLaunchKernelGGL(k_rs_hist<true>, dim3(nblk), dim3(RS_THREADS), 0, c->stream, c->d_gkey, c->d_gpay, c->d_M, n, rebase, 0, fixed_bits, c->d_hist, nblk,
                       have_hist ? 1 : 0);
  const u32 nrows = (rebase && !have_hist) ? RS_ROW_COUNT + 1 : RS_BINS;      // (with the order check the scan also covers row RS_ROW_COUNT, the offsets of a plain compaction)
  hipLaunchKernelGGL(k_rs_scan, dim3((nrows + 3) / 4), dim3(256), 0, c->stream, c->d_hist, nblk, nrows, c->d_rowTot, c->d_rowTot + RS_ROW_COUNT + 1,
                     c->d_M, 0, fixed_bits, rebase, c->d_walk);
  hipLaunchKernelGGL(k_rs_scatter<true>, dim3(nblk), dim3(RS_THREADS), 0, c->stream, c->d_gkey, c->d_gpay, c->d_kB, c->d_vB, c->d_M, n, rebase, 0,
                     fixed_bits, c->d_hist, c->d_rowTot, nblk, (have_hist && !fixed_bits) ? 1 : 0, c->d_kA, c->d_vA);
  u32 *kin = c->d_kB, *vin = c->d_vB, *kout = c->d_kA, *vout = c->d_vA;
  for (int pass = 1; pass < passes; ++pass) {
    const int shift = 8 * pass;
    hipLaunchKernelGGL(k_rs_hist<false>, dim3(nblk), dim3(RS_THREADS), 0, c->stream, kin, vin, c->d_M, n, rebase, shift, fixed_bits, c->d_hist, nblk, 0);
    hipLaunchKernelGGL(k_rs_scan, dim3(RS_BINS / 4), dim3(256), 0, c->stream, c->d_hist, nblk, (u32) RS_BINS, c->d_rowTot, c->d_rowTot + RS_ROW_COUNT + 1, c->d_M, shift,
                       fixed_bits, 0, c->d_walk);
    hipLaunchKernelGGL(k_rs_scatter<false>, dim3(nblk), dim3(RS_THREADS), 0, c->stream, kin, vin, kout, vout, c->d_M, n, rebase, shift, fixed_bits,
                       c->d_hist, c->d_rowTot, nblk, 0, nullptr, nullptr);
    std::swap(kin, kout); std::swap(vin, vout);
  }
  HIPCHK(hipGetLastError());
  return MM_OK;
}

__device__ __forceinline__ u32 row_chunk_of(u64 mask, u32 n_chunks) { return n_chunks <= 1 ? 0u : (u32) ((mask * 0x9E3779B97F4A7C15ull) >> 40) % n_chunks; }
// rows (element set, bit pattern of its sum) of another context -> this one's table; empty entries (set 0) and, with n_chunks > 1, the rows
// of other chunks are skipped
__global__ __launch_bounds__(256) void k_seed_rows(const u64 *__restrict__ masks, const double *__restrict__ sums, u32 n, GlobalTable G, u32 chunk, u32 n_chunks) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || masks[i] == 0) return;
  if (n_chunks > 1 && row_chunk_of(masks[i], n_chunks) != chunk) return;
  const u32 s = gt_find_or_insert(G, masks[i]);
  if (s != NO_KEY) G.sum[s] = sums[i];
}

// segment bounds of every slot in one batch's slot-sorted pairs -> seg[0..GT_CAP) lows, seg[GT_CAP..2 GT_CAP) highs
static int ratio_bounds(mm_ctx *c, const u32 *keys, const u32 *info, size_t n, u32 *seg) {
  const u32 grid = (u32) std::min<size_t>((n + 255) / 256, 256 * 8);
  HIPCHK(hipMemsetAsync(seg, 0, 2 * GT_CAP * 4, c->stream));
  hipLaunchKernelGGL(k_seg_bounds, dim3(grid), dim3(256), 0, c->stream, keys, info, seg, seg + GT_CAP);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

// ordered accumulation of one batch's slot-sorted pairs (see k_ratio_rows); seg = the batch's bounds if already known; rows of one
// chunk only when nchunks > 1
static int ratio_accumulate(mm_ctx *c, const u32 *keys, const u32 *vals, const u32 *info, size_t n, const u32 *seg = nullptr, u32 chunk = 0,
                            u32 nchunks = 1) {
  if (!seg) { const int rc = ratio_bounds(c, keys, info, n, c->d_seg); if (rc) return rc; seg = c->d_seg; }
  hipLaunchKernelGGL(k_ratio_tiles, dim3((u32) (n / RATIO_CHUNK + 1)), dim3(256), 0, c->stream, keys, vals, info, c->G, c->d_tileTot, c->d_tileK, chunk, nchunks);
  hipLaunchKernelGGL(k_ratio_rows, dim3(GT_CAP / RATIO_SLOTS_PER_WG), dim3(RATIO_THREADS), 0, c->stream, vals, seg, seg + GT_CAP, c->G,
                     c->d_tileTot, c->d_tileK, chunk, nchunks);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

// a buffer for one parked batch of up to `cap` pairs: from the pool, or a new allocation (keys | values | segment bounds | batch scalars)
static int take_run_buffer(mm_ctx *c, size_t cap, mm_ctx::Run &r) {
  for (size_t k = 0; k < c->run_pool.size(); ++k) if (c->run_pool[k].cap >= cap) { r = c->run_pool[k]; c->run_pool.erase(c->run_pool.begin() + (long) k); r.n = 0; return MM_OK; }
  u32 *base = nullptr;
  HIPCHK(hipMalloc((void **) &base, cap * 8 + 2 * GT_CAP * 4 + 64));
  r.k = base; r.v = base + cap; r.seg = base + 2 * cap; r.info = r.seg + 2 * GT_CAP; r.cap = cap; r.n = 0;
  return MM_OK;
}
static void free_runs(mm_ctx *c) {       // parked batches -> their buffers back to the pool (the caller has synchronised the stream)
  for (auto &r : c->runs) c->run_pool.push_back(r);
  c->runs.clear();
}
static void free_run_pool(mm_ctx *c) {
  for (auto &r : c->run_pool) (void) hipFree(r.k);
  c->run_pool.clear();
}

// the batch's (index, NH) side list -> device, through one of two pinned staging sets (the caller's array is free on return)
static int stage_big(mm_ctx *c, const mm_nh_big *big, size_t n_big, size_t n, const uint2 **d_out) {
  *d_out = nullptr;
  if (n_big == 0) return MM_OK;
  if (!big) return fail(MM_ERR_ARG, "mm_annotate_batch: n_big > 0 without a list");
  for (size_t k = 0; k < n_big; ++k)
    if (big[k].index >= n || (k && big[k].index <= big[k - 1].index)) return fail(MM_ERR_ARG, "mm_annotate_batch: the NH side list must be ascending by tuple index, indices below n");
  const int b = c->big_cur; c->big_cur ^= 1;
  if (c->big_valid[b]) HIPCHK(hipEventSynchronize(c->ev_big[b]));   // the copy that last used this set (two batches ago)
  if (n_big > c->big_cap[b]) {
    (void) hipFree(c->d_big[b]); (void) hipHostFree(c->h_big[b]); c->d_big[b] = nullptr; c->h_big[b] = nullptr;
    const size_t cap = std::max<size_t>(2 * n_big, 1024);
    HIPCHK(hipMalloc((void **) &c->d_big[b], cap * sizeof(uint2))); HIPCHK(hipHostMalloc((void **) &c->h_big[b], cap * sizeof(uint2), hipHostMallocDefault));
    c->big_cap[b] = cap;
  }
  if (!c->ev_big[b]) HIPCHK(hipEventCreateWithFlags(&c->ev_big[b], hipEventDisableTiming));
  memcpy(c->h_big[b], big, n_big * sizeof(uint2));
  HIPCHK(hipMemcpyAsync(c->d_big[b], c->h_big[b], n_big * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(c->ev_big[b], c->stream)); c->big_valid[b] = true;
  *d_out = c->d_big[b];
  return MM_OK;
}

// one batch; fmt12: `aln` holds 12-byte tuples (mm_aln12) relative to qid_base, expanded on the device behind the copy
static int annotate_impl(mm_ctx *c, const void *aln, bool fmt12, uint64_t qid_base, size_t n, const mm_nh_big *big, size_t n_big, uint64_t qid_lo,
                         uint64_t qid_bound, uint32_t flags) {
  if (!c) return fail(MM_ERR_ARG, "mm_annotate_batch: null context");
  if (c->finalized) return fail(MM_ERR_STATE, "mm_annotate_batch: context already finalized (call mm_reset)");
  if (n == 0) return MM_OK;
  if (!aln) return fail(MM_ERR_ARG, "mm_annotate_batch: null tuples");
  if (n > 0xFFFFFFF0ull) return fail(MM_ERR_RANGE, "mm_annotate_batch: batch larger than 2^32 tuples");
  if (flags & ~(uint32_t) (MM_BATCH_ON_DEVICE | MM_BATCH_SINGLE_HIT)) return fail(MM_ERR_ARG, "mm_annotate_batch: unknown flag");
  const bool on_device = (flags & MM_BATCH_ON_DEVICE) != 0, single = (flags & MM_BATCH_SINGLE_HIT) != 0;
  if (single && n_big) return fail(MM_ERR_ARG, "mm_annotate_batch: MM_BATCH_SINGLE_HIT with an NH side list");
  if (qid_lo > qid_bound || qid_lo > 0xFFFFFFF0ull) qid_lo = 0;   // (a hint only)
  HIPCHK(hipSetDevice(c->ix->device));
  int rc = ensure_batch(c, n, !on_device || fmt12);
  if (rc) return rc;
  if (fmt12 && c->pk_cap < c->cap) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->copy_stream) HIPCHK(hipStreamSynchronize(c->copy_stream));
    for (int b = 0; b < 2; ++b) { (void) hipFree(c->d_pkBuf[b]); c->d_pkBuf[b] = nullptr; HIPCHK(hipMalloc((void **) &c->d_pkBuf[b], c->cap * 12)); }
    c->pk_cap = c->cap;
  }
  const bool dflt = c->opt.strategy == MM_STRAT_DEFAULT, ratio = c->opt.strategy == MM_STRAT_RATIO;
  if (dflt && !single) { rc = ensure_qids(c, qid_bound ? qid_bound : 1); if (rc) return rc; }
  const uint2 *d_big = nullptr;
  rc = stage_big(c, big, n_big, n, &d_big);
  if (rc) return rc;
  const int4 *d_aln = (const int4 *) aln;
  int staged = -1;
  const u32 egrid = (u32) std::min<size_t>((n + 255) / 256, 256 * 16);
  if (on_device && fmt12) {   // packed tuples already in HBM: expand into a staging set (no copy, but the set is in use until the batch is done)
    const int b = c->cur; c->cur ^= 1; staged = b;
    if (c->consumed_valid[b]) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_consumed[b], 0));
    hipLaunchKernelGGL(k_expand12, dim3(egrid), dim3(256), 0, c->stream, (const u32 *) aln, c->d_alnBuf[b], n, (u32) qid_base);
    c->d_aln = c->d_alnBuf[b];
    d_aln = c->d_aln;
  } else if (!on_device) {
    // copy stream: wait until the kernels that last read this staging set are done, copy, signal; the compute stream
    // waits for the signal.  With two sets the copy of this batch overlaps the kernels of the previous one.
    const int b = c->cur; c->cur ^= 1; staged = b;
    // (host side: the copy that used this set two batches ago has finished - that batch's host buffer is free again when this call
    // returns, which is the lifetime rule of include/mmannot_amd.h and keeps the host at most two batches ahead of the device)
    if (c->copied_valid[b]) HIPCHK(hipEventSynchronize(c->ev_copied[b]));
    if (c->consumed_valid[b]) HIPCHK(hipStreamWaitEvent(c->copy_stream, c->ev_consumed[b], 0));
    {
      PhaseTimer t(c, MM_PHASE_H2D);
      t.stream_override = c->copy_stream;
      t.restart();
      if (fmt12) HIPCHK(hipMemcpyAsync(c->d_pkBuf[b], aln, n * 12, hipMemcpyHostToDevice, c->copy_stream));   // 12 bytes per alignment cross the link
      else HIPCHK(hipMemcpyAsync(c->d_alnBuf[b], aln, n * sizeof(mm_aln), hipMemcpyHostToDevice, c->copy_stream));
    }
    HIPCHK(hipEventRecord(c->ev_copied[b], c->copy_stream)); c->copied_valid[b] = true;
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_copied[b], 0));
    if (fmt12) hipLaunchKernelGGL(k_expand12, dim3(egrid), dim3(256), 0, c->stream, c->d_pkBuf[b], c->d_alnBuf[b], n, (u32) qid_base);
    c->d_aln = c->d_alnBuf[b];
    d_aln = c->d_aln;
  }
  ProbeParams P;
  fill_params(c, P);
  const bool need_groups = (dflt && !single) || ratio;
  const u32 grid = (u32) std::min<size_t>((n + 255) / 256, 256 * 8);
  const u32 nblk = (u32) ((n + RS_TILE - 1) / RS_TILE);
  u64 *hm = c->d_hm;
  u32 *gkey = need_groups ? c->d_gkey : nullptr;
  // host-known key range of this batch's multi-hit records -> partition passes to launch (the device may need fewer).  One pass: that
  // layout is taken for granted (no order check, k_probe counts the buckets itself)
  const int span_bits = dflt ? bits_for((qid_bound ? qid_bound : 1) - qid_lo) : 16;
  const int n_passes = dflt ? leaf_layout(span_bits).passes : 2;
  const bool force_leaf = dflt && need_groups && n_passes == 1;
  // k_probe also counts the first sort pass's digits per tile: buckets of the leaf layout, or (-y ratio) the table slots of a one-pass sort
  const bool fuse_hist = (force_leaf || ratio) && !c->env_no_fused_hist;   // (the environment switch is for A/B measurements)
  if (fuse_hist) HIPCHK(hipMemsetAsync(c->d_hist, 0, (size_t) RS_BINS * nblk * sizeof(u32), c->stream));   // k_probe stores the non-zero counts only
  if (need_groups) hipLaunchKernelGGL(k_set_info, dim3(1), dim3(64), 0, c->stream, c->d_M, force_leaf ? ~(u32) qid_lo : 0u, force_leaf ? 1u : 0u);
  {
  PhaseTimer t(c, MM_PHASE_PROBE);
  // the instance that takes this launch's constants for granted, or the general one (side outputs, -y random, an index without slots,
  // more chromosomes than the LDS table holds; MMANNOT_AMD_GENERAL_PROBE=1 forces it for A/B measurements and tests)
  int var = PV_GENERAL;
  if (P.useSlots && P.nchr <= (u32) MM_LDS_CHR && !P.keep && P.strategy != MM_STRAT_RANDOM && !c->env_general_probe)
    var = ratio ? (fuse_hist ? PV_RATIO_HIST : PV_RATIO) : ((dflt && need_groups) ? (fuse_hist ? PV_GROUPS_HIST : PV_GROUPS) : PV_COUNT);
  // the pipelined walk over the fast slots (k_probe_fast) + the list of what it passes on (k_probe_list), where the index has them - or
  // k_probe's walk: same results, and which of the two is faster depends on annotation and options (measured on one box, same
  // batches: HS38-shaped, inclusion: 301 against 284 us per 2^24 tuples; Flybase6-shaped, -l 50: 183 against 229; profiles/r03c_*).
  // So a context measures: its first batches of at least 2^18 tuples take the two in turn, each bracketed by events, and once both
  // have a second timed batch (the first of each loads code) the faster one stays (mm_probe_path; MMANNOT_AMD_NO_FAST / MMANNOT_AMD_FAST in the environment decide without
  // measuring).
  bool fast = var != PV_GENERAL && P.fslots != nullptr && !c->env_no_fast;
  int tune_path = -1;
  if (fast && !c->env_fast && c->tune_choice == 0) {
    tune_collect(c);
    if (c->tune_choice == 0) {
      tune_path = (c->tune_turn++ & 1) ? 1 : 0;                  // 0 = pipelined, 1 = walk
      fast = tune_path == 0;
    }
  }
  if (fast && c->tune_choice == 2) fast = false;
  hipEvent_t tune_a = nullptr, tune_b = nullptr;
  if (tune_path >= 0 && n >= ((size_t) 1 << 18) && hipEventCreate(&tune_a) == hipSuccess && hipEventCreate(&tune_b) == hipSuccess)
    (void) hipEventRecord(tune_a, c->stream);
  if (fast) {
    u32 *const dlist2 = c->d_dlist + (size_t) nblk * RS_TILE;        // the compacted list (the first half: the waves' regions)
    // the list's length: two counters used in turn - k_probe_list zeroes the other one for the next batch (no memset launch per batch)
    u32 *const dcnt = c->d_dcnt + c->dcnt_cur, *const dcnt_next = c->d_dcnt + (c->dcnt_cur ^ 1);
    c->dcnt_cur ^= 1;
#define MM_LAUNCH_FAST(OVL, VAR)                                                                                                     \
  hipLaunchKernelGGL((k_probe_fast<OVL, VAR>), dim3(nblk), dim3(256), 0, c->stream, P, d_aln, single ? 1 : 0,                         \
                     (u32) qid_lo, n, gkey, c->d_gpay, c->d_M, c->G, c->d_stats, fuse_hist ? c->d_hist : (u32 *) nullptr, nblk, c->d_dlist, dlist2, dcnt)
#define MM_LAUNCH_FAST_VAR(OVL)                                       \
  switch (var) {                                                      \
    case PV_COUNT: MM_LAUNCH_FAST(OVL, PV_COUNT); break;              \
    case PV_GROUPS_HIST: MM_LAUNCH_FAST(OVL, PV_GROUPS_HIST); break;  \
    case PV_GROUPS: MM_LAUNCH_FAST(OVL, PV_GROUPS); break;            \
    case PV_RATIO: MM_LAUNCH_FAST(OVL, PV_RATIO); break;              \
    default: MM_LAUNCH_FAST(OVL, PV_RATIO_HIST);                      \
  }
#define MM_LAUNCH_LIST(OVL)                                                                                                          \
  hipLaunchKernelGGL((k_probe_list<OVL>), dim3(std::min<u32>(2048u, (u32) ((n + 255) / 256))), dim3(256), 0, c->stream, P, d_aln, d_big, (u32) n_big, \
                     single ? 1 : 0, (u32) qid_lo, gkey, c->d_gpay, c->d_M, c->G, c->d_stats, fuse_hist ? c->d_hist : (u32 *) nullptr, nblk, \
                     (const u32 *) dlist2, (const u32 *) dcnt, dcnt_next, c->d_walk)
    switch (c->opt.overlap_mode) {
      case MM_OVL_INCLUSION: MM_LAUNCH_FAST_VAR(MM_OVL_INCLUSION) MM_LAUNCH_LIST(MM_OVL_INCLUSION); break;
      case MM_OVL_PC: MM_LAUNCH_FAST_VAR(MM_OVL_PC) MM_LAUNCH_LIST(MM_OVL_PC); break;
      default: MM_LAUNCH_FAST_VAR(MM_OVL_NT) MM_LAUNCH_LIST(MM_OVL_NT);
    }
#undef MM_LAUNCH_LIST
#undef MM_LAUNCH_FAST_VAR
#undef MM_LAUNCH_FAST
  } else {
#define MM_LAUNCH_PROBE(OVL, VAR)                                                                                                  \
  hipLaunchKernelGGL((k_probe<OVL, VAR>), dim3(fuse_hist ? nblk : grid), dim3(256), 0, c->stream, P, d_aln, d_big, (u32) n_big,    \
                     single ? 1 : 0, (u32) qid_lo, n, hm, gkey, c->d_gpay, c->d_M, c->G, c->d_stats, c->d_walk,                   \
                     fuse_hist ? c->d_hist : (u32 *) nullptr, nblk)
#define MM_LAUNCH_PROBE_VAR(OVL)                                        \
  switch (var) {                                                        \
    case PV_COUNT: MM_LAUNCH_PROBE(OVL, PV_COUNT); break;               \
    case PV_GROUPS_HIST: MM_LAUNCH_PROBE(OVL, PV_GROUPS_HIST); break;   \
    case PV_GROUPS: MM_LAUNCH_PROBE(OVL, PV_GROUPS); break;             \
    case PV_RATIO: MM_LAUNCH_PROBE(OVL, PV_RATIO); break;               \
    case PV_RATIO_HIST: MM_LAUNCH_PROBE(OVL, PV_RATIO_HIST); break;     \
    default: MM_LAUNCH_PROBE(OVL, PV_GENERAL);                          \
  }
  switch (c->opt.overlap_mode) {
    case MM_OVL_INCLUSION: MM_LAUNCH_PROBE_VAR(MM_OVL_INCLUSION) break;
    case MM_OVL_PC: MM_LAUNCH_PROBE_VAR(MM_OVL_PC) break;
    default: MM_LAUNCH_PROBE_VAR(MM_OVL_NT)
  }
  }
#undef MM_LAUNCH_PROBE_VAR
#undef MM_LAUNCH_PROBE
  if (tune_a && tune_b) {
    // (while it measures - four batches - the context waits for each timed batch: the host would otherwise be many batches ahead of the
    // device before the first event completes, and the whole input would be spent taking turns)
    (void) hipEventRecord(tune_b, c->stream); c->tune_pending.push_back({tune_path, n, tune_a, tune_b});
    (void) hipEventSynchronize(tune_b);
    tune_collect(c);
  }
  }
  bool huge_nh = false;                                  // an NH >= 65535 can only come through the side list
  for (size_t k = 0; k < n_big && !huge_nh; ++k) huge_nh = big[k].nh >= 0xFFFFu;
  if (dflt && need_groups && !c->opt.keep_hit_masks && huge_nh) {
    switch (c->opt.overlap_mode) {
      case MM_OVL_INCLUSION:
        hipLaunchKernelGGL(k_probe_sets<MM_OVL_INCLUSION>, dim3(grid), dim3(256), 0, c->stream, P, d_aln, d_big, (u32) n_big, n, hm, c->d_M, c->G.err);
        break;
      case MM_OVL_PC:
        hipLaunchKernelGGL(k_probe_sets<MM_OVL_PC>, dim3(grid), dim3(256), 0, c->stream, P, d_aln, d_big, (u32) n_big, n, hm, c->d_M, c->G.err);
        break;
      default:
        hipLaunchKernelGGL(k_probe_sets<MM_OVL_NT>, dim3(grid), dim3(256), 0, c->stream, P, d_aln, d_big, (u32) n_big, n, hm, c->d_M, c->G.err);
    }
  }
  HIPCHK(hipGetLastError());
  c->last_n = n;
  c->last_aln = d_aln;
  if (need_groups) {
    {
      PhaseTimer t(c, MM_PHASE_SORT);
      rc = sort_pairs(c, n, n_passes, dflt ? 0 : 16, dflt ? 1 : 0, fuse_hist);
      if (!rc && dflt && n_passes > 1) {   // more than one pass: where each bucket lies is found by a pass of its own
        hipLaunchKernelGGL(k_bucket_bounds, dim3(grid), dim3(256), 0, c->stream, (const u64 *) c->d_kA, (const u64 *) c->d_kB, c->d_M, c->d_bLo, c->d_bHi,
                           c->d_bList, c->nbcap, c->G.err);
        HIPCHK(hipGetLastError());
      }
    }
    if (rc) return rc;
    PhaseTimer t(c, MM_PHASE_MERGE);
    if (dflt) {
      // exactly one of the two does the work (device-side decision, batch_is_leaf): k_leaf behind the partition passes,
      // k_chunk_default behind the plain compaction of name-grouped input
      hipLaunchKernelGGL(k_leaf, dim3(RS_BINS), dim3(LEAF_THREADS), 0, c->stream, (const u64 *) c->d_kA, (const u64 *) c->d_kB, c->d_M, c->d_rowTot, c->d_bLo, c->d_bHi,
                         c->d_bList, d_aln, d_big, (u32) n_big, hm, c->d_grpRem, c->d_grpU, c->qcap, c->G, c->d_stats, force_leaf ? 1 : 0);
      if (!force_leaf)
      hipLaunchKernelGGL(k_chunk_default, dim3(grid), dim3(256), 0, c->stream, c->d_kA, c->d_vA, c->d_kB, c->d_vB, c->d_M, d_aln, d_big, (u32) n_big, hm,
                         c->d_grpRem, c->d_grpU, c->qcap, c->G, c->d_stats, c->ix->E);
    } else if (c->opt.defer_ratio) {  // the sorted pairs are in A: park them - the run keeps A, the context goes on with a buffer from the pool
      mm_ctx::Run fresh;
      rc = take_run_buffer(c, c->cap, fresh);
      if (rc) return rc;
      mm_ctx::Run r;
      r.k = c->d_kA; r.v = c->d_vA; r.seg = c->d_kA + 2 * c->cap; r.info = r.seg + 2 * GT_CAP; r.cap = c->cap; r.n = n;
      c->d_kA = fresh.k; c->d_vA = fresh.v;
      HIPCHK(hipMemcpyAsync(r.info, c->d_M, INFO_WORDS * 4, hipMemcpyDeviceToDevice, c->stream));
      c->runs.push_back(r);
      rc = ratio_bounds(c, r.k, r.info, n, r.seg);        // every slot's segment of this batch: known now, off the chain's critical path
      if (rc) return rc;
    } else {
      rc = ratio_accumulate(c, c->d_kA, c->d_vA, c->d_M, n);
      if (rc) return rc;
    }
    HIPCHK(hipGetLastError());
  }
  if (staged >= 0) { HIPCHK(hipEventRecord(c->ev_consumed[staged], c->stream)); c->consumed_valid[staged] = true; }
  return MM_OK;
}

extern "C" int mm_annotate_batch(mm_ctx *c, const mm_aln *aln, size_t n, const mm_nh_big *big, size_t n_big, uint64_t qid_lo, uint64_t qid_bound,
                                 uint32_t flags) {
  return annotate_impl(c, aln, false, 0, n, big, n_big, qid_lo, qid_bound, flags);
}

extern "C" int mm_annotate_batch12(mm_ctx *c, const mm_aln12 *aln, size_t n, uint64_t qid_base, const mm_nh_big *big, size_t n_big, uint64_t qid_lo,
                                   uint64_t qid_bound, uint32_t flags) {
  if (qid_base > 0xFFFFFFF0ull) return fail(MM_ERR_RANGE, "mm_annotate_batch12: qid_base exceeds 2^32");
  return annotate_impl(c, aln, true, qid_base, n, big, n_big, qid_lo, qid_bound, flags);
}

static int check_table(mm_ctx *c) {
  u32 err = 0;
  HIPCHK(hipMemcpyAsync(&err, c->G.err, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (err == ERR_QID_BOUND) return fail(MM_ERR_ARG, "a read-name id (qid) of the batch is not below the qid_bound given to mm_annotate_batch; the batch's multi-hit records were not merged");
  if (err == ERR_SINGLE_HIT) return fail(MM_ERR_ARG, "a batch submitted with MM_BATCH_SINGLE_HIT holds a tuple with NH > 1");
  if (err == ERR_QID_LO) return fail(MM_ERR_ARG, "a tuple with NH > 1 has a read-name id below the qid_lo hint given to mm_annotate_batch");
  if (err == ERR_NH_LIST) return fail(MM_ERR_ARG, "a tuple's NH field is MM_ALN_NH_ESCAPE but the batch's NH side list has no entry for it");
  if (err) return fail(MM_ERR_TABLE_FULL, "more than 65535 distinct element sets");
  return MM_OK;
}

extern "C" int mm_sync(mm_ctx *c) {
  if (!c) return fail(MM_ERR_ARG, "mm_sync: null context");
  HIPCHK(hipSetDevice(c->ix->device));   // (one host thread may drive contexts on several devices)
  HIPCHK(hipStreamSynchronize(c->stream));
  return check_table(c);
}

extern "C" int mm_finalize(mm_ctx *c) {
  if (!c) return fail(MM_ERR_ARG, "mm_finalize: null context");
  HIPCHK(hipSetDevice(c->ix->device));   // (one host thread may drive contexts on several devices)
  if (!c->finalized && c->opt.strategy == MM_STRAT_DEFAULT && c->qcap) {
    PhaseTimer t(c, MM_PHASE_FLUSH);
    hipLaunchKernelGGL(k_flush_pending, dim3(256 * 4), dim3(256), 0, c->stream, c->d_grpRem, c->d_grpU, c->qcap, c->G, c->d_stats);
    HIPCHK(hipGetLastError());
  }
  c->finalized = true;
  return mm_sync(c);
}

// Multi-GPU -y ratio: the double sums are order dependent (mmannot.cpp:1730), so ranks that each hold a contiguous block of the
// file chain them: rank r seeds its rows with rank r-1's sums (mm_set_row_sums), replays its parked batches in order
// (mm_ratio_resolve) and passes its rows on.
extern "C" int mm_set_row_sums(mm_ctx *c, const uint64_t *masks, const double *sums, size_t n) {
  if (!c || (n && (!masks || !sums))) return fail(MM_ERR_ARG, "mm_set_row_sums: null argument");
  if (c->opt.strategy != MM_STRAT_RATIO) return fail(MM_ERR_STATE, "mm_set_row_sums: only meaningful under -y ratio");
  if (n == 0) return MM_OK;
  HIPCHK(hipSetDevice(c->ix->device));
  u64 *dm = nullptr; double *ds = nullptr;
  HIPCHK(hipMalloc((void **) &dm, n * 8)); HIPCHK(hipMalloc((void **) &ds, n * 8));
  HIPCHK(hipMemcpyAsync(dm, masks, n * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(ds, sums, n * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_seed_rows, dim3((u32) ((n + 255) / 256)), dim3(256), 0, c->stream, dm, ds, (u32) n, c->G, 0u, 1u);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void) hipFree(dm); (void) hipFree(ds);
  if (e != hipSuccess) return fail(MM_ERR_HIP, std::string("mm_set_row_sums: ") + hipGetErrorString(e));
  return check_table(c);
}

extern "C" size_t mm_ratio_pending(const mm_ctx *c) { return c ? c->runs.size() : 0; }

extern "C" int mm_ratio_resolve_n(mm_ctx *c, size_t max_runs) {
  if (!c) return fail(MM_ERR_ARG, "mm_ratio_resolve: null context");
  HIPCHK(hipSetDevice(c->ix->device));
  size_t done = 0;
  for (auto &r : c->runs) {
    if (done == max_runs) break;
    PhaseTimer t(c, MM_PHASE_MERGE);
    int rc = ratio_accumulate(c, r.k, r.v, r.info, r.n, r.seg);
    if (rc) return rc;
    ++done;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  for (size_t k = 0; k < done; ++k) c->run_pool.push_back(c->runs[k]);
  c->runs.erase(c->runs.begin(), c->runs.begin() + (long) done);
  return check_table(c);
}

extern "C" int mm_ratio_resolve(mm_ctx *c) { return mm_ratio_resolve_n(c, (size_t) -1); }

// Replays every parked batch for the rows of ONE chunk (of n_chunks, by the hash of the row's element set); the parked batches are
// released with the last chunk.  A rank of a chain can so pass chunk c's sums on while it works on chunk c + 1.
extern "C" int mm_ratio_resolve_chunk(mm_ctx *c, uint32_t chunk, uint32_t n_chunks) {
  if (!c || n_chunks == 0 || chunk >= n_chunks) return fail(MM_ERR_ARG, "mm_ratio_resolve_chunk: bad argument");
  HIPCHK(hipSetDevice(c->ix->device));
  for (auto &r : c->runs) {
    PhaseTimer t(c, MM_PHASE_MERGE);
    const int rc = ratio_accumulate(c, r.k, r.v, r.info, r.n, r.seg, chunk, n_chunks);
    if (rc) return rc;
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (chunk + 1 == n_chunks) free_runs(c);
  return check_table(c);
}
// ---- the -y ratio chain without host hops (round 3).  The whole table travels: MM_TABLE_SLOTS element sets and the bit patterns of their
// sums, empty slots with set 0 - 1 MB, fixed size, so neither side has to know how many rows there are; everything is asynchronous on
// the context's stream.
extern "C" int mm_table_export_device(mm_ctx *c, uint64_t *d_keys, double *d_sums) {
  if (!c || !d_keys || !d_sums) return fail(MM_ERR_ARG, "mm_table_export_device: null argument");
  HIPCHK(hipSetDevice(c->ix->device));
  HIPCHK(hipMemcpyAsync(d_keys, c->G.key, GT_CAP * 8, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_sums, c->G.sum, GT_CAP * 8, hipMemcpyDeviceToDevice, c->stream));
  return MM_OK;
}
extern "C" int mm_table_import_device(mm_ctx *c, const uint64_t *d_keys, const double *d_sums, size_t n, uint32_t chunk, uint32_t n_chunks) {
  if (!c || (n && (!d_keys || !d_sums)) || n_chunks == 0 || chunk >= n_chunks) return fail(MM_ERR_ARG, "mm_table_import_device: bad argument");
  if (c->opt.strategy != MM_STRAT_RATIO) return fail(MM_ERR_STATE, "mm_table_import_device: only meaningful under -y ratio");
  if (n == 0) return MM_OK;
  HIPCHK(hipSetDevice(c->ix->device));
  hipLaunchKernelGGL(k_seed_rows, dim3((u32) ((n + 255) / 256)), dim3(256), 0, c->stream, (const u64 *) d_keys, d_sums, (u32) n, c->G, chunk, n_chunks);
  HIPCHK(hipGetLastError());
  return MM_OK;
}
// replays parked batches [first, first + count) for the rows of one chunk, asynchronously; nothing is released (mm_ratio_release)
extern "C" int mm_ratio_replay(mm_ctx *c, size_t first, size_t count, uint32_t chunk, uint32_t n_chunks) {
  if (!c || n_chunks == 0 || chunk >= n_chunks) return fail(MM_ERR_ARG, "mm_ratio_replay: bad argument");
  HIPCHK(hipSetDevice(c->ix->device));
  for (size_t k = first; k < c->runs.size() && k - first < count; ++k) {
    PhaseTimer t(c, MM_PHASE_MERGE);
    const mm_ctx::Run &r = c->runs[k];
    const int rc = ratio_accumulate(c, r.k, r.v, r.info, r.n, r.seg, chunk, n_chunks);
    if (rc) return rc;
  }
  return MM_OK;
}
extern "C" int mm_ratio_release(mm_ctx *c) {
  if (!c) return fail(MM_ERR_ARG, "mm_ratio_release: null context");
  HIPCHK(hipSetDevice(c->ix->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  free_runs(c);
  return check_table(c);
}
// buffers for n_runs parked batches of up to batch_capacity tuples, allocated now instead of in the middle of a stream
extern "C" int mm_ratio_reserve(mm_ctx *c, size_t n_runs, size_t batch_capacity) {
  if (!c) return fail(MM_ERR_ARG, "mm_ratio_reserve: null context");
  if (c->opt.strategy != MM_STRAT_RATIO || !c->opt.defer_ratio) return MM_OK;
  HIPCHK(hipSetDevice(c->ix->device));
  int rc = ensure_batch(c, std::max<size_t>(batch_capacity, 1), false);
  if (rc) return rc;
  std::vector<mm_ctx::Run> got;
  for (size_t k = 0; k < n_runs; ++k) { mm_ctx::Run r; rc = take_run_buffer(c, c->cap, r); if (rc) break; got.push_back(r); }
  for (auto &r : got) c->run_pool.push_back(r);
  return rc;
}
// the table of src (any device) -> dst's rows of one chunk, ordered behind everything src's stream holds so far and ahead of what dst's
// stream gets next; asynchronous
extern "C" int mm_ratio_handoff(mm_ctx *src, mm_ctx *dst, uint32_t chunk, uint32_t n_chunks) {
  if (!src || !dst || src == dst) return fail(MM_ERR_ARG, "mm_ratio_handoff: bad argument");
  HIPCHK(hipSetDevice(src->ix->device));
  if (!src->ev_handoff) HIPCHK(hipEventCreateWithFlags(&src->ev_handoff, hipEventDisableTiming));
  HIPCHK(hipEventRecord(src->ev_handoff, src->stream));
  HIPCHK(hipSetDevice(dst->ix->device));
  if (!dst->d_xkeys) { HIPCHK(hipMalloc((void **) &dst->d_xkeys, GT_CAP * 8)); HIPCHK(hipMalloc((void **) &dst->d_xsums, GT_CAP * 8)); }
  HIPCHK(hipStreamWaitEvent(dst->stream, src->ev_handoff, 0));
  if (src->ix->device == dst->ix->device) {
    HIPCHK(hipMemcpyAsync(dst->d_xkeys, src->G.key, GT_CAP * 8, hipMemcpyDeviceToDevice, dst->stream));
    HIPCHK(hipMemcpyAsync(dst->d_xsums, src->G.sum, GT_CAP * 8, hipMemcpyDeviceToDevice, dst->stream));
  } else {
    HIPCHK(hipMemcpyPeerAsync(dst->d_xkeys, dst->ix->device, src->G.key, src->ix->device, GT_CAP * 8, dst->stream));
    HIPCHK(hipMemcpyPeerAsync(dst->d_xsums, dst->ix->device, src->G.sum, src->ix->device, GT_CAP * 8, dst->stream));
  }
  return mm_table_import_device(dst, (const uint64_t *) dst->d_xkeys, dst->d_xsums, GT_CAP, chunk, n_chunks);
}
/* chunk of a row, for callers that split mm_get_rows by chunk */
extern "C" uint32_t mm_row_chunk(uint64_t mask, uint32_t n_chunks) { return n_chunks <= 1 ? 0u : (uint32_t) ((mask * 0x9E3779B97F4A7C15ull) >> 40) % n_chunks; }

extern "C" int mm_set_profiling(mm_ctx *c, int on) {
  if (!c) return fail(MM_ERR_ARG, "mm_set_profiling: null context");
  HIPCHK(hipSetDevice(c->ix->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (auto &s : c->spans) { (void) hipEventDestroy(s.a); (void) hipEventDestroy(s.b); }
  c->spans.clear();
  c->profiling = (on == 2) ? 2 : (on != 0 ? 1 : 0);
  return MM_OK;
}

extern "C" int mm_get_phase_ms(mm_ctx *c, double ms[MM_N_PHASES], uint64_t launches[MM_N_PHASES]) {
  if (!c || !ms || !launches) return fail(MM_ERR_ARG, "mm_get_phase_ms: null argument");
  HIPCHK(hipSetDevice(c->ix->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int p = 0; p < MM_N_PHASES; ++p) { ms[p] = 0; launches[p] = 0; }
  for (auto &s : c->spans) {
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, s.a, s.b));
    ms[s.phase] += (double) t; ++launches[s.phase];
  }
  return MM_OK;
}

extern "C" int mm_get_stats(mm_ctx *c, uint64_t stats[MM_N_STATS]) {
  if (!c || !stats) return fail(MM_ERR_ARG, "mm_get_stats: null argument");
  HIPCHK(hipSetDevice(c->ix->device));   // (one host thread may drive contexts on several devices)
  u64 h[8];
  HIPCHK(hipMemcpyAsync(h, c->d_stats, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int i = 0; i < MM_N_STATS; ++i) stats[i] = h[i];
  return MM_OK;
}

#define ROWS_FIRST 8192u   // rows that travel with the count in the first copy (128 KB)
extern "C" int mm_get_rows(mm_ctx *c, uint64_t *masks, double *sums, size_t cap, size_t *n_rows) {
  if (!c || !n_rows) return fail(MM_ERR_ARG, "mm_get_rows: null argument");
  HIPCHK(hipSetDevice(c->ix->device));   // (one host thread may drive contexts on several devices)
  const bool ratio = c->opt.strategy == MM_STRAT_RATIO;
  u32 *const wgCount = (u32 *) (c->d_rows + 1 + 2 * (size_t) GT_CAP);
  hipLaunchKernelGGL(k_rows_count, dim3(ROWS_WG), dim3(1024), 0, c->stream, c->G, ratio ? 1 : 0, wgCount);
  hipLaunchKernelGGL(k_rows_write, dim3(ROWS_WG), dim3(1024), 0, c->stream, c->G, ratio ? 1 : 0, (const u32 *) wgCount, c->d_rows);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->h_rows, c->d_rows, (1 + 2 * (size_t) ROWS_FIRST) * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const size_t nr = (size_t) c->h_rows[0];
  *n_rows = nr;
  if (!masks) return MM_OK;
  if (nr > cap) return fail(MM_ERR_ARG, "mm_get_rows: capacity too small");
  if (nr > ROWS_FIRST) {
    HIPCHK(hipMemcpyAsync(c->h_rows + 1 + 2 * (size_t) ROWS_FIRST, c->d_rows + 1 + 2 * (size_t) ROWS_FIRST, 2 * (nr - ROWS_FIRST) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  for (size_t i = 0; i < nr; ++i) {
    masks[i] = c->h_rows[1 + 2 * i];
    if (sums) memcpy(&sums[i], &c->h_rows[2 + 2 * i], 8);
  }
  return MM_OK;
}

// The table for a merge on the device (ranks of a multi-process job, -y default / unique: tables add up): the 65 536 sets, their
// integer counts and the seven counters (d_stats[MM_N_STATS]), device to device on the context's stream - nothing waits for the host.
extern "C" int mm_table_export_counts_device(mm_ctx *c, uint64_t *d_keys, uint64_t *d_counts, uint64_t *d_stats) {
  if (!c || !d_keys || !d_counts || !d_stats) return fail(MM_ERR_ARG, "mm_table_export_counts_device: null argument");
  if (c->opt.strategy == MM_STRAT_RATIO) return fail(MM_ERR_STATE, "mm_table_export_counts_device: -y ratio sums are not mergeable by addition (mm_table_export_device)");
  HIPCHK(hipSetDevice(c->ix->device));
  HIPCHK(hipMemcpyAsync(d_keys, c->G.key, GT_CAP * 8, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_counts, c->G.cnt, GT_CAP * 8, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_stats, c->d_stats, MM_N_STATS * 8, hipMemcpyDeviceToDevice, c->stream));
  return MM_OK;
}
// ... and the other direction: n_tables tables in that layout, `stride` 64-bit words apart (>= 2 * MM_TABLE_SLOTS + MM_N_STATS; as an
// all_gather of every rank's export leaves them), are added to this context's table and counters - all but table `own`, which is this
// context's.  Asynchronous on the context's stream; the tables must be complete when it is called.  mm_get_rows / mm_get_stats then
// give the whole job's result, the same on every rank.
extern "C" int mm_table_merge_counts_device(mm_ctx *c, const uint64_t *d_tables, size_t n_tables, size_t stride, size_t own) {
  if (!c || !d_tables || n_tables == 0 || n_tables > 32 || stride < 2 * (size_t) GT_CAP + MM_N_STATS) return fail(MM_ERR_ARG, "mm_table_merge_counts_device: bad argument (1..32 tables)");
  if (c->opt.strategy == MM_STRAT_RATIO) return fail(MM_ERR_STATE, "mm_table_merge_counts_device: -y ratio sums are not mergeable by addition");
  HIPCHK(hipSetDevice(c->ix->device));
  hipLaunchKernelGGL(k_table_merge, dim3(256), dim3(256), 0, c->stream, (const u64 *) d_tables, (u32) n_tables, (u32) stride, (u32) std::min<size_t>(own, 0xFFFFFFFFu), c->G, c->d_stats);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_get_hit_masks(mm_ctx *c, uint64_t *out, size_t n) {
  if (!c || !out) return fail(MM_ERR_ARG, "mm_get_hit_masks: null argument");
  HIPCHK(hipSetDevice(c->ix->device));
  if (!c->opt.keep_hit_masks) return fail(MM_ERR_STATE, "mm_get_hit_masks: context created without keep_hit_masks");
  if (n > c->last_n) return fail(MM_ERR_ARG, "mm_get_hit_masks: n exceeds the last batch");
  HIPCHK(hipMemcpyAsync(out, c->d_hm, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return MM_OK;
}

extern "C" int mm_get_hit_intervals(mm_ctx *c, uint32_t *counts, uint32_t *idx, size_t cap, size_t *total) {
  if (!c || !counts || !total) return fail(MM_ERR_ARG, "mm_get_hit_intervals: null argument");
  if (!c->opt.keep_hit_masks) return fail(MM_ERR_STATE, "mm_get_hit_intervals: context created without keep_hit_masks");
  const size_t n = c->last_n;
  *total = 0;
  if (n == 0) return MM_OK;
  HIPCHK(hipSetDevice(c->ix->device));
  ProbeParams P;
  fill_params(c, P);
  const u32 grid = (u32) std::min<size_t>((n + 255) / 256, 256 * 8);
  u32 *d_cnt = c->d_gkey, *d_off = c->d_gpay;   // free between batches
#define MM_LAUNCH_HI(FILLV, OUTP)                                                                                              \
  switch (c->opt.overlap_mode) {                                                                                              \
    case MM_OVL_INCLUSION: hipLaunchKernelGGL((k_hit_intervals<MM_OVL_INCLUSION, FILLV>), dim3(grid), dim3(256), 0, c->stream, P, c->ix->d_actIdx, c->last_aln, n, c->d_hm, d_cnt, d_off, OUTP); break; \
    case MM_OVL_PC: hipLaunchKernelGGL((k_hit_intervals<MM_OVL_PC, FILLV>), dim3(grid), dim3(256), 0, c->stream, P, c->ix->d_actIdx, c->last_aln, n, c->d_hm, d_cnt, d_off, OUTP); break; \
    default: hipLaunchKernelGGL((k_hit_intervals<MM_OVL_NT, FILLV>), dim3(grid), dim3(256), 0, c->stream, P, c->ix->d_actIdx, c->last_aln, n, c->d_hm, d_cnt, d_off, OUTP); \
  }
  MM_LAUNCH_HI(false, (u32 *) nullptr)
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(counts, d_cnt, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  std::vector<u32> off(n);
  u64 tot = 0;
  for (size_t i = 0; i < n; ++i) { off[i] = (u32) tot; tot += counts[i]; }
  *total = (size_t) tot;
  if (!idx || tot == 0) return MM_OK;
  if (tot > cap) return fail(MM_ERR_ARG, "mm_get_hit_intervals: capacity too small");
  if (tot >= 0xFFFFFFF0ull) return fail(MM_ERR_RANGE, "mm_get_hit_intervals: too many intervals in one batch");
  u32 *d_out = nullptr;
  HIPCHK(hipMalloc((void **) &d_out, tot * 4));
  HIPCHK(hipMemcpyAsync(d_off, off.data(), n * 4, hipMemcpyHostToDevice, c->stream));
  MM_LAUNCH_HI(true, d_out)
#undef MM_LAUNCH_HI
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(idx, d_out, tot * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void) hipFree(d_out);
  if (e != hipSuccess) return fail(MM_ERR_HIP, std::string("mm_get_hit_intervals: ") + hipGetErrorString(e));
  return MM_OK;
}

extern "C" int mm_get_walk_stats(mm_ctx *c, uint64_t out[2]) {
  if (!c || !out) return fail(MM_ERR_ARG, "mm_get_walk_stats: null argument");
  HIPCHK(hipSetDevice(c->ix->device));
  HIPCHK(hipMemcpyAsync(out, c->d_walk, 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return MM_OK;
}

extern "C" int mm_probe_path(mm_ctx *c) {
  if (!c) return -1;
  if (c->env_no_fast || !c->ix->d_fslots) return 2;
  // launches the pipelined probe never takes (annotate_impl: the general instance of k_probe serves them)
  if (c->ix->nchr > MM_LDS_CHR || c->opt.keep_hit_masks || c->opt.strategy == MM_STRAT_RANDOM || c->env_general_probe || c->env_full_index_only) return 2;
  if (c->env_fast) return 1;
  (void) hipSetDevice(c->ix->device);
  tune_collect(c);
  return c->tune_choice;
}

extern "C" int mm_get_probe_stats(mm_ctx *c, uint64_t out[4]) {
  if (!c || !out) return fail(MM_ERR_ARG, "mm_get_probe_stats: null argument");
  HIPCHK(hipSetDevice(c->ix->device));
  HIPCHK(hipMemcpyAsync(out, c->d_walk, 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  out[3] = c->ix->d_fslots ? (uint64_t) c->ix->n_fslots * 32ull : 0;
  return MM_OK;
}

extern "C" int mm_synth_fill(void *d_aln, size_t n, const mm_synth_params *p, void *stream) {
  if (!d_aln || !p || !p->chr_len) return fail(MM_ERR_ARG, "mm_synth_fill: null argument");
  if (p->n_chromosomes < 1 || p->n_chromosomes > 64) return fail(MM_ERR_ARG, "mm_synth_fill: 1..64 chromosomes");
  if (p->nh_max < 1 || p->nh_max >= (int) MM_ALN_NH_ESCAPE || p->read_len < 1) return fail(MM_ERR_ARG, "mm_synth_fill: bad nh_max/read_len");
  SynthDev S;
  S.seed = p->seed; S.first = p->first; S.read_len = p->read_len; S.nh_max = p->nh_max; S.nchr = p->n_chromosomes;
  S.scatter_log2 = p->scatter_log2; S.pair_insert = p->pair_insert;
  if (p->pair_insert && p->nh_max != 1) return fail(MM_ERR_ARG, "mm_synth_fill: pairs are generated with NH = 1 only");
  S.cum[0] = 0;
  for (int c = 0; c < p->n_chromosomes; ++c) {
    if (p->chr_len[c] < 1 || p->chr_len[c] > 0x7FFFFF00) return fail(MM_ERR_RANGE, "mm_synth_fill: chromosome length");
    S.cum[c + 1] = S.cum[c] + (u64) p->chr_len[c];
  }
  if (n == 0) return MM_OK;
  const u32 grid = (u32) std::min<size_t>((n + 255) / 256, 256 * 8);
  hipLaunchKernelGGL(k_synth, dim3(grid), dim3(256), 0, (hipStream_t) stream, S, (int4 *) d_aln, n);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

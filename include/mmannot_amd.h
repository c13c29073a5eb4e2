/*
 * mmannot_amd.h — C ABI of the MI355X-native overlap-and-count engine.
 *
 * Drop-in boundary for the hot path of mzytnicki/mmannot v1.1 (all citations are into the
 * reference's single translation unit, mmannot.cpp).  The reference has no plugin / FFI
 * surface; the seam this library sits behind is the pair of member calls made once per
 * alignment by Counter::read (mmannot.cpp:1767-1782):
 *
 *     IntervalList::scan(Read&, vector<size_t>& regions, ...)      mmannot.cpp:1291-1332
 *     Counter::addCount(name, regions, intervals, nHits)           mmannot.cpp:1665-1739
 *
 * plus the end-of-file flush (mmannot.cpp:1783-1792), Counter::getCounts()/dump()
 * (mmannot.cpp:1803-1818) and TableCount (mmannot.cpp:1854-1901).  INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Plain pointers and sizes only; no torch / HIP types in any signature (a hipStream_t is
 * passed as void*).  All functions return MM_OK (0) or an MM_ERR_* code; mm_last_error()
 * gives the message of the last failure on the calling thread.
 *
 * The device path is the only path: every compute entry point fails with MM_ERR_HIP when
 * no gfx950 device/runtime is usable.  There is no CPU fallback.
 */
#ifndef MMANNOT_AMD_H
#define MMANNOT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_VERSION "mmannot-amd 0.1 (drop-in for mmannot v1.1)"
#define MM_MAX_ELEMENTS 64 /* an element set travels as a 64-bit mask */

enum {
  MM_OK = 0,
  MM_ERR_ARG = 1,        /* bad argument                                             */
  MM_ERR_HIP = 2,        /* HIP runtime / device failure (incl. no device)           */
  MM_ERR_RANGE = 3,      /* value outside the supported domain (E > 64, coord >= 2^31) */
  MM_ERR_TABLE_FULL = 4, /* more distinct element sets than the device table holds   */
  MM_ERR_NOMEM = 5,
  MM_ERR_STATE = 6,      /* call out of order (e.g. annotate after finalize)         */
  MM_ERR_IO = 7,
  MM_ERR_PARSE = 8
};

/* Strand of an Order element / of an annotation interval: enum class Strand, mmannot.cpp:49 */
enum { MM_STRAND_ALL = 0, MM_STRAND_F = 1, MM_STRAND_R = 2 };
/* Which intervalOverlapFunction -l selects, mmannot.cpp:992-1002, 1972-1977 */
enum { MM_OVL_INCLUSION = 0, MM_OVL_PC = 1, MM_OVL_NT = 2 };
/* -s U/F/R -> strandU/strandF/strandR, mmannot.cpp:836-844, 1981-1983 */
enum { MM_STRANDED_U = 0, MM_STRANDED_F = 1, MM_STRANDED_R = 2 };
/* -y, enum class Strategy mmannot.cpp:50, 2036-2039 */
enum { MM_STRAT_DEFAULT = 0, MM_STRAT_UNIQUE = 1, MM_STRAT_RANDOM = 2, MM_STRAT_RATIO = 3 };
/* Counter's seven counters in declaration order, mmannot.cpp:1663 */
enum { MM_STAT_HITS = 0, MM_STAT_READS, MM_STAT_UNIQUE, MM_STAT_AMBIGUOUS, MM_STAT_MULTIPLE, MM_STAT_UNASSIGNED, MM_STAT_RESCUED, MM_N_STATS };

/* The flattened Order section: what scan/getFirst/checkStrand need from Config
 * (mmannot.cpp:222-226, 370-374, 438-443, 463-470). */
typedef struct mm_elements {
  int32_t n_elements;         /* Config::getNElements()                         */
  const int32_t *line_of;     /* [E] Order line of element e (RegionType.id)    */
  const uint8_t *elem_strand; /* [E] MM_STRAND_* of the Order entry             */
  const uint8_t *is_up;       /* [E] entry's type is "upstream"                 */
  const uint8_t *is_down;     /* [E] entry's type is "downstream"               */
} mm_elements;

/* namespace Globals, mmannot.cpp:69-90 — the part the hot path reads */
typedef struct mm_options {
  int32_t overlap_mode;   /* MM_OVL_*                                  */
  float overlap_value;    /* Globals::overlap (float)                  */
  int32_t strandedness;   /* MM_STRANDED_*                             */
  int32_t strategy;       /* MM_STRAT_* (RANDOM is host-side only: MM_ERR_ARG here) */
  int32_t keep_hit_masks; /* debug/parity: keep the per-hit region sets of the last batch */
  int32_t defer_ratio;    /* -y ratio only: park each batch's ordered accumulation until mm_ratio_resolve (multi-GPU chaining) */
} mm_options;

/* One Read as handed to scan() (class Read, mmannot.cpp:846-903): 16 bytes, the whole per-alignment input stream
 * ("stream in: 16 B").
 *   start = record.start = pos0+1 (mmannot.cpp:1537), end after parseCigar (mmannot.cpp:852-875,
 *   may be start-1 for a record without CIGAR operations).
 *   tsn  bit 0       = record.strand (1 when flag 0x10 is clear, mmannot.cpp:608)
 *        bits 11..1  = record.nHits (NH; mmannot.cpp:1560, 1637-1641) when it is 0..2046; MM_ALN_NH_ESCAPE (2047) = the value
 *                      is in the batch's mm_nh_big side list (an aligner reporting more than 2046 hits of one read is rare)
 *        bits 31..12 = index into the *annotation's* chromosome table + 1; 0 = the read's chromosome is unknown to the
 *                      annotation (mmannot.cpp:1293-1302).  At most MM_ALN_MAX_CHROMOSOMES chromosomes.
 *   qid = dense id of the read name (the reference keys readCounts on the string, mmannot.cpp:1671). */
typedef struct mm_aln {
  int32_t start;
  int32_t end;
  uint32_t tsn;
  uint32_t qid;
} mm_aln;
#define MM_ALN_NH_BITS 11
#define MM_ALN_NH_ESCAPE ((1u << MM_ALN_NH_BITS) - 1u)
#define MM_ALN_MAX_CHROMOSOMES ((1 << (31 - MM_ALN_NH_BITS)) - 2)
#define MM_ALN_PACK(tid, strand_plus, nh) \
  ((((uint32_t) ((tid) + 1)) << (MM_ALN_NH_BITS + 1)) | ((((uint32_t) (nh)) < MM_ALN_NH_ESCAPE ? (uint32_t) (nh) : MM_ALN_NH_ESCAPE) << 1) | ((strand_plus) ? 1u : 0u))
#define MM_ALN_TID(tsn) ((int32_t) ((tsn) >> (MM_ALN_NH_BITS + 1)) - 1)
#define MM_ALN_STRAND(tsn) ((tsn) & 1u)
#define MM_ALN_NH_FIELD(tsn) (((tsn) >> 1) & MM_ALN_NH_ESCAPE)
/* side list of a batch: tuple `index` (its NH field holds MM_ALN_NH_ESCAPE) has nHits = nh; ascending by index */
typedef struct mm_nh_big {
  uint32_t index;
  uint32_t nh;
} mm_nh_big;

typedef struct mm_index mm_index; /* IntervalList's index members, mmannot.cpp:1086-1090, resident in HBM */
typedef struct mm_ctx mm_ctx;     /* one Counter (mmannot.cpp:1653-1664) on one GPU                      */

const char *mm_last_error(void);
const char *mm_version(void);

/* ---- index: replaces the sorted TypedInterval vector + bins of IntervalList (mmannot.cpp:1267-1284).
 * Input = the interval vector *after* the reference's std::sort (order is significant:
 * EvaluationStructure::set is last-writer-wins, mmannot.cpp:1023-1028), chr ascending.
 * bin_shift: log2 of the device bin width, 0 = choose automatically. */
/* number of HIP devices visible to the process (0 when there is none) */
int mm_device_count(void);
int mm_index_create(mm_index **out, const mm_elements *elems, int32_t n_chromosomes, size_t n_intervals,
                    const int32_t *chr, const int32_t *start, const int32_t *end, const uint8_t *type,
                    const uint8_t *strand, int device, int bin_shift);
void mm_index_destroy(mm_index *ix);
/* sizes of the device structures (bytes): full index [0]=interval records [1]=bin table [2]=active lists [3]=its bin shift;
 * compact index [4]=slots (64 B per bin) [5]=flat record lists [6]=its bin shift [7]=bitmap of non-empty slots (0 = not used) */
int mm_index_info(const mm_index *ix, uint64_t info[8]);

/* ---- context: replaces Counter (clear/read/getCounts/dump). stream: hipStream_t or NULL (a private stream). */
int mm_ctx_create(mm_ctx **out, const mm_index *ix, const mm_options *opt, void *stream);
void mm_ctx_destroy(mm_ctx *ctx);
/* Counter::clear(), mmannot.cpp:1742-1747 */
int mm_reset(mm_ctx *ctx);
/* optional: size the per-read-name state of -y default for qid_bound names ahead of time (a caller that knows how many reads are to
 * come), so that mm_annotate_batch never has to grow it in the middle of a stream */
int mm_reserve_qids(mm_ctx *ctx, uint64_t qid_bound);

/* One batch of the per-alignment loop of Counter::read (mmannot.cpp:1772-1781): for every tuple,
 * scan() then addCount().  Asynchronous on the context's stream.
 *   aln: n tuples; host memory (copied with hipMemcpyAsync on a copy stream that overlaps the previous batch's kernels;
 *        pinned memory - mm_host_alloc - for full PCIe rate; must stay valid until mm_sync() or until two further batches
 *        have been submitted) or, with MM_BATCH_ON_DEVICE, device memory (inputs already resident in HBM).
 *   big/n_big: host array, the NH values of the tuples whose NH field is MM_ALN_NH_ESCAPE, ascending by index; copied
 *        before the call returns.
 *   qid_bound: exclusive upper bound of every qid seen so far including this batch (sizes the
 *        per-read-name state of -y default; ignored by the other strategies).
 *   qid_lo: hint, 0 = none: no tuple of this batch with NH > 1 has a qid below it (the decoder knows; MM_ERR_ARG at the next
 *        mm_sync when one does).  The span qid_bound - qid_lo tells the host how many grouping passes the batch can need, so
 *        none is launched in vain.
 *   flags: MM_BATCH_ON_DEVICE; MM_BATCH_SINGLE_HIT = the caller promises that every tuple has NH <= 1 (checked on the
 *        device, MM_ERR_ARG at the next mm_sync otherwise): the multi-hit pipeline is not launched at all.
 * Batches must be submitted in file order (the multi-hit countdown is order dependent, mmannot.cpp:1669-1701). */
enum { MM_BATCH_ON_DEVICE = 1, MM_BATCH_SINGLE_HIT = 2 };
int mm_annotate_batch(mm_ctx *ctx, const mm_aln *aln, size_t n, const mm_nh_big *big, size_t n_big, uint64_t qid_lo, uint64_t qid_bound,
                      uint32_t flags);
/* Optional 12-byte link format.  From pinned host memory the path runs at the speed of the link (16 B per alignment at ~56 GB/s of
 * PCIe Gen5 x16 = 3.5 G alignments/s; the kernels take 27 G/s), so what crosses the link is worth packing: w[0] = start:31 | strand:1;
 * w[1] = length:14 (end - start + 1; 0 for a record without CIGAR operations) | NH:8 (255 = see the side list) | (chromosome+1) bits 9..0;
 * w[2] = (qid - qid_base):24 | (chromosome+1) bits 17..10.  mm_pack12 (host, where the tuples are produced) refuses a batch with a
 * tuple that does not fit (MM_ERR_RANGE: it then travels as mm_aln); out_big receives the NH values >= 255 (capacity big_cap).
 * mm_annotate_batch12 copies 12 bytes per tuple and expands them on the device (k_expand12) into the 16-byte tuples every kernel reads:
 * same results by construction.  Everything else as mm_annotate_batch. */
typedef struct mm_aln12 {
  uint32_t w[3];
} mm_aln12;
int mm_pack12(const mm_aln *aln, size_t n, const mm_nh_big *big, size_t n_big, uint64_t qid_base, mm_aln12 *out, mm_nh_big *out_big, size_t big_cap,
              size_t *out_n_big);
int mm_annotate_batch12(mm_ctx *ctx, const mm_aln12 *aln, size_t n, uint64_t qid_base, const mm_nh_big *big, size_t n_big, uint64_t qid_lo,
                        uint64_t qid_bound, uint32_t flags);
/* pinned (page-locked) host memory for tuple batches: hipHostMalloc / hipHostFree */
void *mm_host_alloc(size_t bytes);
void mm_host_free(void *p);
int mm_sync(mm_ctx *ctx);
/* End of input: the flush of still-pending multi-hit reads, mmannot.cpp:1783-1792. Synchronises. */
int mm_finalize(mm_ctx *ctx);

/* Counter's counters (valid after mm_sync / mm_finalize). */
int mm_get_stats(mm_ctx *ctx, uint64_t stats[MM_N_STATS]);
/* Counter::getCounts(), mmannot.cpp:1803: one row per distinct element set, unsorted.
 * sums[] is the reference's double accumulator (exactly integral except under -y ratio). */
int mm_get_rows(mm_ctx *ctx, uint64_t *masks, double *sums, size_t cap, size_t *n_rows);
/* Parity/debug: region set of every tuple of the last batch (requires keep_hit_masks);
 * ~0 marks a tuple skipped by -y unique (mmannot.cpp:1773). */
int mm_get_hit_masks(mm_ctx *ctx, uint64_t *out, size_t n);
/* -M support, EvaluationStructure::getIds (mmannot.cpp:1077-1081): for every tuple of the last batch, the indices (in the
 * sorted interval vector given to mm_index_create) of the intervals that passed scan() for the elements of its final set.
 * counts[n] always; idx (capacity cap) may be NULL to learn *total first.  Requires keep_hit_masks; the batch's tuples must
 * still be valid (device batches: not yet overwritten by the caller). */
int mm_get_hit_intervals(mm_ctx *ctx, uint32_t *counts, uint32_t *idx, size_t cap, size_t *total);
/* Profiling aids since the last mm_reset.  out[0]: candidate records visited by the probe (requires keep_hit_masks);
 * out[1]: batches whose multi-hit records arrived already grouped by read name (non-decreasing qids, as in an aligner's
 * unsorted output) and therefore skipped the name sort. */
int mm_get_walk_stats(mm_ctx *ctx, uint64_t out[2]);
/* The same two, and out[2]: tuples the pipelined probe (k_probe_fast: 32-byte slots of 4-byte records, three tuples in flight per lane)
 * handed on to the general walk since the last mm_reset - bins with more than 7 records, reads longer than the slots' saturating fields
 * allow or without CIGAR operations, NH values from the side list, hits that need the max-overlap / min-distance selection
 * (mmannot.cpp:1049-1075); out[3]: bytes of the fast slot table (0: the index has none). */
int mm_get_probe_stats(mm_ctx *ctx, uint64_t out[4]);
/* Which probe the context runs where both apply (no side outputs, at most 256 chromosomes): 1 = the pipelined one (k_probe_fast +
 * k_probe_list), 2 = the walk (k_probe), 0 = still measuring - the first batches of at least 2^18 tuples take the two in turn, timed
 * with events on the context's stream, and the faster one stays (the results are the same either way).  MMANNOT_AMD_NO_FAST=1 /
 * MMANNOT_AMD_FAST=1 in the environment (read at mm_ctx_create) decide without measuring. */
int mm_probe_path(mm_ctx *ctx);

/* Multi-GPU -y ratio.  regionCounts[set] += 1.0/nHits is a file-order double sum (mmannot.cpp:1730): ranks holding consecutive
 * blocks of the file chain it.  With mm_options.defer_ratio the batches are probed and sorted as usual but their accumulation is
 * parked; rank r then calls mm_set_row_sums with the rows of rank r-1 (none for rank 0), mm_ratio_resolve (replays its parked
 * batches in order), and hands its mm_get_rows on to rank r+1. */
int mm_set_row_sums(mm_ctx *ctx, const uint64_t *masks, const double *sums, size_t n);
int mm_ratio_resolve(mm_ctx *ctx);
int mm_ratio_resolve_n(mm_ctx *ctx, size_t max_runs);   /* replays only the first max_runs parked batches */
/* Rows are independent of each other, so a chain of ranks can be pipelined: replay all parked batches for the rows of chunk `chunk`
 * only (mm_row_chunk(mask, n_chunks) == chunk), hand those rows on, go on with the next chunk.  The parked batches are released with
 * the last chunk. */
int mm_ratio_resolve_chunk(mm_ctx *ctx, uint32_t chunk, uint32_t n_chunks);
uint32_t mm_row_chunk(uint64_t mask, uint32_t n_chunks);
size_t mm_ratio_pending(const mm_ctx *ctx);            /* parked batches not yet replayed */
/* The same chain without a host hop (round 3): the whole table travels device to device - MM_TABLE_SLOTS element sets and the bit
 * patterns of their sums, empty slots with set 0 - and every call is asynchronous on the context's stream.
 *   mm_table_export_device: the context's table -> d_keys[MM_TABLE_SLOTS], d_sums[MM_TABLE_SLOTS] (device memory);
 *   mm_table_import_device: n entries (device memory; set 0 = none) seed the context's rows - with n_chunks > 1 only the rows of `chunk`;
 *   mm_ratio_replay: parked batches [first, first + count) for the rows of one chunk, nothing released, no synchronisation;
 *   mm_ratio_release: synchronises, gives the parked batches' buffers back to the context's pool, reports errors;
 *   mm_ratio_reserve: buffers for n_runs parked batches of up to batch_capacity tuples now (else they are allocated when first needed);
 *   mm_ratio_handoff: src's table -> dst's rows of one chunk, across devices of one process (peer copy), stream-ordered on both sides. */
#define MM_TABLE_SLOTS 65536
int mm_table_export_device(mm_ctx *ctx, uint64_t *d_keys, double *d_sums);
int mm_table_import_device(mm_ctx *ctx, const uint64_t *d_keys, const double *d_sums, size_t n, uint32_t chunk, uint32_t n_chunks);
int mm_ratio_replay(mm_ctx *ctx, size_t first, size_t count, uint32_t chunk, uint32_t n_chunks);
int mm_ratio_release(mm_ctx *ctx);
int mm_ratio_reserve(mm_ctx *ctx, size_t n_runs, size_t batch_capacity);
int mm_ratio_handoff(mm_ctx *src, mm_ctx *dst, uint32_t chunk, uint32_t n_chunks);
/* -y default / unique, ranks of a multi-process job: tables add up (Counter::getCounts of every rank, mmannot.cpp:1803, summed).  For a
 * merge that stays on the devices: d_keys[MM_TABLE_SLOTS] (0 = empty slot), d_counts[MM_TABLE_SLOTS], d_stats[MM_N_STATS], device to
 * device on the context's stream (mm_sync before another stream reads them).  MM_ERR_STATE under -y ratio (sums do not add up). */
int mm_table_export_counts_device(mm_ctx *ctx, uint64_t *d_keys, uint64_t *d_counts, uint64_t *d_stats);
/* ... and back: n_tables (1..32) exported tables, each d_keys | d_counts | d_stats contiguous and `stride` 64-bit words apart (what an
 * all_gather of every rank's export leaves in device memory), are added to this context's table and counters - all but table `own`,
 * this context's.  Asynchronous on the context's stream.  mm_get_stats / mm_get_rows then return the whole job's result. */
int mm_table_merge_counts_device(mm_ctx *ctx, const uint64_t *d_tables, size_t n_tables, size_t stride, size_t own);

/* ---- one input on several GPUs (replaces the reference's thread-per-file pool, mmannot.cpp:2117-2143, which can only
 * spread whole files).  One context per device, each on its own copy of the index.
 * mm_partition (host): deals the tuples of one decoded batch out by qid mod n_parts, file order kept inside each part, read names
 * renumbered qid / n_parts (dense per part; the caller passes ceil(qid_bound / n_parts) on).  out[p] needs room for n tuples,
 * out_big[p] for big_cap entries; out_max_nh[p] / out_qid_lo[p] (may be NULL) = the part's largest NH / smallest renumbered qid with NH > 1.  Every hit of a name lands in one part,
 * which is what the multi-hit countdown of -y default needs (mmannot.cpp:1669-1701); -y unique may use it too.
 * -y ratio instead sends whole batches round-robin (batch j to context j mod n, contexts created with defer_ratio): its per-row
 * double sums are file-order dependent (mmannot.cpp:1730).
 * mm_merge (host, after mm_finalize on every context): the input's table and counters.  default / unique: rows and counters add
 * up.  ratio: the parked accumulations are replayed in file order, each context starting from the bit patterns its predecessor
 * ended with.  masks NULL: only *n_rows and stats.  (Ranks that are separate processes merge with an all_gather of the keys and
 * an all_reduce of the dense count vector over RCCL instead: mmannot_amd/dist.py.) */
int mm_partition(const mm_aln *aln, size_t n, const mm_nh_big *big, size_t n_big, int n_parts, mm_aln *const *out, size_t *out_n,
                 mm_nh_big *const *out_big, size_t big_cap, size_t *out_n_big, uint32_t *out_max_nh, uint64_t *out_qid_lo);
/* host threads mm_partition uses (default: the cores the process may run on, at most 16; MM_PARTITION_THREADS in the environment):
 * n >= 1 sets the number - taken as given, also for small batches - n = 0 only asks; returns the number in use. */
int mm_partition_threads(int n);
int mm_merge(mm_ctx *const *ctxs, int n_ctx, int strategy, uint64_t *masks, double *sums, size_t cap, size_t *n_rows,
             uint64_t stats[MM_N_STATS]);

/* Live timing with HIP events recorded on the context's stream around each phase of
 * mm_annotate_batch / mm_finalize (bench.py's roofline leg).  on = 0 off (default), 1 every phase, 2 the probe phase and the
 * host-to-device copies only (every recorded event costs a few microseconds of stream time; the copies have a stream of their own). */
enum { MM_PHASE_H2D = 0, MM_PHASE_PROBE, MM_PHASE_COMPACT, MM_PHASE_SORT, MM_PHASE_MERGE, MM_PHASE_FLUSH, MM_N_PHASES };
int mm_set_profiling(mm_ctx *ctx, int on);
/* synchronises; ms[p] = summed duration of phase p since mm_set_profiling(ctx,1), launches[p] = how many spans */
int mm_get_phase_ms(mm_ctx *ctx, double ms[MM_N_PHASES], uint64_t launches[MM_N_PHASES]);

/* ---- host finalisation (pure host code, no device): TableCount, mmannot.cpp:1854-1901 */
/* `unsigned int v = round(double)` as the reference's x86-64 build evaluates it, mmannot.cpp:1868 */
uint32_t mm_round_to_uint(double v);
/* std::sort over the vector<size_t> keys, mmannot.cpp:1883: sorts rows in place */
void mm_sort_rows(uint64_t *masks, double *sums, size_t n_rows);

/* ---- cold path (host, once per run): configuration file + GTF -> the sorted typed interval vector.
 * Replaces Config::parse (mmannot.cpp:237-382) and the IntervalList constructor (mmannot.cpp:1094-1290),
 * including Synonyms / Introns / Vicinity expansion and the reference's std::sort order. */
typedef struct mm_annotation mm_annotation;
int mm_annotation_load(mm_annotation **out, const char *config_path, const char *gtf_path, uint32_t upstream, uint32_t downstream);
void mm_annotation_free(mm_annotation *a);
const char *mm_annotation_error(void);                 /* message of the last failed mm_annotation_load (what the reference prints before exit(1)) */
const char *mm_annotation_log(const mm_annotation *a); /* what the reference writes to stderr while loading */
int32_t mm_annotation_n_elements(const mm_annotation *a);
int32_t mm_annotation_n_chromosomes(const mm_annotation *a);
size_t mm_annotation_n_intervals(const mm_annotation *a);
const char *mm_annotation_element_name(const mm_annotation *a, int32_t e);  /* Config::getName, mmannot.cpp:445-462 */
const char *mm_annotation_chromosome(const mm_annotation *a, int32_t c);
int mm_annotation_chromosome_has_intervals(const mm_annotation *a, int32_t c); /* is the name a key of `bins` (mmannot.cpp:1278, 1293) */
const char *mm_annotation_interval_id(const mm_annotation *a, size_t i);    /* TypedInterval::getId (for -M) */
int mm_annotation_elements(const mm_annotation *a, mm_elements *out);       /* pointers stay owned by the annotation */
int mm_annotation_intervals(const mm_annotation *a, const int32_t **chr, const int32_t **start, const int32_t **end,
                            const uint8_t **type, const uint8_t **strand);
int mm_index_from_annotation(mm_index **out, const mm_annotation *a, int device, int bin_shift);

/* ---- reads decoder (host): BAM / SAM -> tuples in file order.  Replaces Reader / SamReader / BamReader and
 * Read::parseCigar (mmannot.cpp:846-903, 1339-1650), including XA alternative hits (mmannot.cpp:1360-1399, 1412-1422). */
typedef struct mm_reader mm_reader;
enum { MM_FMT_AUTO = 0, MM_FMT_SAM = 1, MM_FMT_BAM = 2 };
/* XamRecord::setFlags ignores its argument in v1.1 (mmannot.cpp:606-609), so a verbatim run treats every record as '+'
 * or every record as '-' depending on stack garbage.  NONE = intended meaning (flag 0x10). */
enum { MM_STRAND_COMPAT_NONE = 0, MM_STRAND_COMPAT_PLUS = 1, MM_STRAND_COMPAT_MINUS = 2 };
int mm_reader_open(mm_reader **out, const char *path, int format, const mm_annotation *an, int strand_compat,
                   int unique_only /* -y unique: reads with NH != 1 never reach scan(), so they raise no chromosome warning */,
                   int keep_names /* keep qid -> name (needed by -m) */);
const char *mm_reader_error(const mm_reader *r);
/* Reader::gotoNextRead x cap: fills up to cap tuples; *n = 0 at end of file.  NH values that do not fit the tuple's NH field go
 * to big[] (capacity big_cap >= 1; the batch ends early when it is full), *n_big of them.  *max_nh (may be NULL) receives the
 * largest NH of the batch: <= 1 allows MM_BATCH_SINGLE_HIT. */
int mm_reader_next(mm_reader *r, mm_aln *aln, size_t cap, size_t *n, mm_nh_big *big, size_t big_cap, size_t *n_big, uint32_t *max_nh);
/* smallest qid among the tuples with NH > 1 of the batch mm_reader_next delivered last (mm_annotate_batch's qid_lo; the qid bound when there is none) */
uint64_t mm_reader_qid_lo(const mm_reader *r);

uint64_t mm_reader_qid_bound(const mm_reader *r);
uint64_t mm_reader_n_reads(const mm_reader *r);       /* the "lines read" counter of Counter::read, mmannot.cpp:1782 */
const char *mm_reader_name(const mm_reader *r, uint32_t qid);
const char *mm_reader_take_log(mm_reader *r);         /* stderr text produced since the last call */
void mm_reader_close(mm_reader *r);

/* Raw DEFLATE (RFC 1951) of one BGZF block's payload by the reader's own block decoder -> out[0..out_len).  0 = decoded and the
 * stream ended exactly at out_len; non-zero = this decoder does not take the stream (the reader then hands the block to zlib, as it
 * does for every block when MM_READER_ZLIB=1).  Replaces the zlib calls behind the reference's gzread (mmannot.cpp:903-960 read its
 * BAM through zlib); exported for the differential tests (tests/test_decoder.py). */
int mm_inflate_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);

/* ---- synthetic workload generators for bench.py / tests (device kernels, counter-based RNG;
 * tests/synth_ref.py holds the numpy twin).  See DESIGN.md "Synthetic inputs". */
typedef struct mm_synth_params {
  uint64_t seed;
  uint64_t first;          /* index of the first alignment of this batch in the stream        */
  int32_t read_len;        /* every read is <read_len>M                                        */
  int32_t nh_max;          /* 1: every read NH=1; k>1 (<= 2046): groups with NH = 1..k, exactly NH hits */
  int32_t n_chromosomes;
  const int64_t *chr_len;  /* host array [n_chromosomes]                                       */
  uint32_t scatter_log2;   /* hits of one read are spread over windows of 2^scatter_log2 tuples */
  uint32_t pair_insert;    /* > 0 (needs nh_max 1): tuples 2p, 2p+1 are mates - one name, opposite strands,
                            * starts pair_insert + 0..63 apart (BASELINE.json configs[3])          */
} mm_synth_params;
int mm_synth_fill(void *d_aln, size_t n, const mm_synth_params *p, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MMANNOT_AMD_H */

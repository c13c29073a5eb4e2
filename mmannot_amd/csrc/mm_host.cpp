// mm_host.cpp — host finalisation of the device results: TableCount (mmannot.cpp:1854-1901).
// Pure host code: sorting a few hundred rows and rounding them is what the reference does
// once per input file after the hot loop; nothing here touches alignments.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "../../include/mmannot_amd.h"

// TableCount::addCounter stores `unsigned int v = round(count.second)` (mmannot.cpp:1868, 1872).
// The reference is built for x86-64 with g++ -O3, where double -> unsigned int is a 64-bit
// cvttsd2si followed by truncation to 32 bits; infinities / NaN / |v| >= 2^63 give the
// "integer indefinite" 0x8000000000000000, i.e. 0.  -y ratio with NH = 0 (an NH tag stored
// in a signed BAM type, mmannot.cpp:1596-1599) reaches this with 1.0/0 = inf.
extern "C" uint32_t mm_round_to_uint(double v) {
  const double r = std::round(v);  // half away from zero
  if (!(r > -9223372036854775808.0 && r < 9223372036854775808.0)) return 0u;
  return (uint32_t) (uint64_t) (int64_t) r;
}

// std::sort over vector<vector<size_t>> (mmannot.cpp:1883): lexicographic order of the
// ascending element-id lists; a proper prefix sorts first.
static bool list_less(uint64_t a, uint64_t b) {
  while (a && b) {
    const int ea = __builtin_ctzll(a), eb = __builtin_ctzll(b);
    if (ea != eb) return ea < eb;
    a &= a - 1; b &= b - 1;
  }
  return b != 0;
}

extern "C" void mm_sort_rows(uint64_t *masks, double *sums, size_t n) {
  if (!masks || n < 2) return;
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), (size_t) 0);
  std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return list_less(masks[x], masks[y]); });
  std::vector<uint64_t> m(n);
  std::vector<double> s(n);
  for (size_t i = 0; i < n; ++i) { m[i] = masks[idx[i]]; if (sums) s[i] = sums[idx[i]]; }
  for (size_t i = 0; i < n; ++i) { masks[i] = m[i]; if (sums) sums[i] = s[i]; }
}

// 16-byte tuples -> the 12-byte link format (mm_aln12): start:31 | strand:1, length:14 | NH:8 | (chromosome+1) low 10 bits,
// (qid - qid_base):24 | (chromosome+1) high 8 bits.  Host code run where the tuples are produced; MM_ERR_RANGE (nothing usable written)
// when a tuple does not fit - then the batch travels as 16-byte tuples.
extern "C" int mm_pack12(const mm_aln *aln, size_t n, const mm_nh_big *big, size_t n_big, uint64_t qid_base, mm_aln12 *out, mm_nh_big *out_big,
                         size_t big_cap, size_t *out_n_big) {
  if ((n && (!aln || !out)) || !out_n_big || (n_big && !big)) return MM_ERR_ARG;
  size_t bi = 0, nb = 0;
  for (size_t i = 0; i < n; ++i) {
    const mm_aln &a = aln[i];
    const int64_t len = (int64_t) a.end - (int64_t) a.start + 1;          // 0 for a record without CIGAR operations
    const uint32_t tid1 = a.tsn >> (MM_ALN_NH_BITS + 1);
    if (a.start < 0 || len < 0 || len > 0x3FFF || tid1 >= (1u << 18) || (uint64_t) a.qid < qid_base || (uint64_t) a.qid - qid_base >= (1u << 24)) return MM_ERR_RANGE;
    uint32_t nh = MM_ALN_NH_FIELD(a.tsn);
    if (nh == MM_ALN_NH_ESCAPE) {
      while (bi < n_big && big[bi].index < i) ++bi;
      if (bi >= n_big || big[bi].index != i) return MM_ERR_ARG;            // escaped tuple without an entry
      nh = big[bi].nh;
    }
    uint32_t nh8 = nh;
    if (nh >= 255u) {
      if (!out_big || nb >= big_cap) return MM_ERR_RANGE;
      out_big[nb].index = (uint32_t) i; out_big[nb].nh = nh; ++nb;
      nh8 = 255u;
    }
    const uint32_t rel = (uint32_t) ((uint64_t) a.qid - qid_base);
    out[i].w[0] = (uint32_t) a.start | ((a.tsn & 1u) << 31);
    out[i].w[1] = (uint32_t) len | (nh8 << 14) | ((tid1 & 0x3FFu) << 22);
    out[i].w[2] = rel | ((tid1 >> 10) << 24);
  }
  *out_n_big = nb;
  return MM_OK;
}

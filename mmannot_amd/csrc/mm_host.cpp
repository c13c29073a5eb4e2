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

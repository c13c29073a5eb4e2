// One thread, hot caches: the reader's own DEFLATE decoder (mmannot_amd/csrc/mm_inflate.h) against zlib's inflate() on a 64 KB block
// of BAM-shaped records with sequence and qualities, compressed at levels 1 and 6.
//   g++ -O2 -std=c++17 -o /tmp/bench_inflate tools/bench_inflate.cpp -lz && /tmp/bench_inflate
#include "../mmannot_amd/csrc/mm_inflate.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <chrono>
int main() {
  std::mt19937_64 rng(7);
  // BAM-like: 36 fixed bytes (mostly small ints), read name, cigar, 50 bytes 4-bit seq (random), 100 qualities (skewed), NH tag
  std::vector<uint8_t> data;
  while (data.size() < 65280 - 300) {
    uint32_t pos = rng() % 100000000;
    uint8_t fixed[36] = {220,0,0,0, 1,0,0,0, 0,0,0,0, 12,255, 73,18, 1,0,0,0, 100,0,0,0, 255,255,255,255, 255,255,255,255, 0,0,0,0};
    memcpy(fixed + 8, &pos, 4);
    data.insert(data.end(), fixed, fixed + 36);
    char name[16]; int nl = snprintf(name, sizeof name, "r%09llu", (unsigned long long) (rng() % 1000000000)); data.insert(data.end(), name, name + nl + 1);
    uint8_t cig[4] = {0x40, 6, 0, 0}; data.insert(data.end(), cig, cig + 4);
    for (int i = 0; i < 50; ++i) data.push_back((uint8_t) ((1u << (rng() % 4)) << 4 | (1u << (rng() % 4))));
    for (int i = 0; i < 100; ++i) { int q = 40 - (int) (rng() % 100 < 80 ? rng() % 4 : rng() % 30); data.push_back((uint8_t) q); }
    uint8_t tag[4] = {'N', 'H', 'C', (uint8_t) (1 + rng() % 5)}; data.insert(data.end(), tag, tag + 4);
  }
  mminf::Tables *T = new mminf::Tables;
  for (int level : {1, 6}) {
    uLongf cl = compressBound(data.size()) + 64;
    std::vector<uint8_t> comp(cl);
    z_stream zs{}; deflateInit2(&zs, level, Z_DEFLATED, -15, 8, 0);
    zs.next_in = data.data(); zs.avail_in = data.size(); zs.next_out = comp.data(); zs.avail_out = cl;
    deflate(&zs, Z_FINISH); size_t clen = cl - zs.avail_out; deflateEnd(&zs);
    std::vector<uint8_t> out(data.size());
    const int reps = 3000;
    auto t0 = std::chrono::steady_clock::now();
    int bad = 0;
    for (int r = 0; r < reps; ++r) bad += mminf::inflate_block(comp.data(), clen, out.data(), out.size(), *T);
    auto t1 = std::chrono::steady_clock::now();
    z_stream zi{}; inflateInit2(&zi, -15);
    for (int r = 0; r < reps; ++r) { inflateReset(&zi); zi.next_in = comp.data(); zi.avail_in = clen; zi.next_out = out.data(); zi.avail_out = out.size(); bad += inflate(&zi, Z_FINISH) != Z_STREAM_END; }
    auto t2 = std::chrono::steady_clock::now();
    double a = std::chrono::duration<double>(t1 - t0).count(), b = std::chrono::duration<double>(t2 - t1).count();
    printf("level %d: ratio %.2f  mine %.0f MB/s  zlib %.0f MB/s  (bad %d)\n", level, (double) data.size() / clen, reps * data.size() / a / 1e6, reps * data.size() / b / 1e6, bad);
  }
}

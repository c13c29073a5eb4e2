// Raw DEFLATE (RFC 1951) decoder for one BGZF block - the reads decoder's inner loop on files whose records carry sequence and
// qualities, where zlib's inflate() was three quarters of the decoding time.  A BGZF block is at most 64 KB, complete in memory,
// and its inflated size is known from the gzip trailer, which is all this decoder handles: input and output are flat buffers,
// there is no streaming state, no window beyond the output itself, and every check that could fail ends the call with a non-zero
// code - the caller then hands the block to zlib, so an unusual stream costs time, never correctness.
//
// The method is the usual table-driven one: canonical Huffman codes expanded into a primary table indexed by the next
// TABLE_BITS input bits (longer codes continue in sub-tables), a 64-bit bit buffer refilled with one unaligned load, and match
// copies in 8-byte steps where the distance allows.  tests/test_decoder.py holds it against zlib on every block type, level
// and a few thousand random streams; MM_READER_ZLIB=1 switches it off.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace mminf {

enum { LITLEN_BITS = 11, DIST_BITS = 8, PRE_BITS = 7, MAX_CODE_LEN = 15 };
// table entry: bits 0-4 = number of input bits this entry consumes (the code's length, or for a sub-table pointer the primary
// index width), bits 5-7 = kind, bits 8-12 = number of extra bits that follow the code, bits 16-31 = value (literal, base
// length, base distance, or offset of the sub-table)
// (K_LIT2: two literals whose codes fit the primary index together, value = first | second << 8; the two literal kinds are the ones
// with the low two kind bits clear)
enum { K_LITERAL = 0, K_BASE = 1, K_END = 2, K_SUB = 3, K_LIT2 = 4, K_INVALID = 5 };
static inline bool is_lit(uint32_t e) { return (e & (3u << 5)) == 0; }
static inline uint32_t entry(uint32_t nbits, uint32_t kind, uint32_t extra, uint32_t value) { return nbits | (kind << 5) | (extra << 8) | (value << 16); }

struct Tables {
  uint32_t litlen[(1u << LITLEN_BITS) + 2400];   // primary + sub-tables (a code that needs more is left to zlib)
  uint32_t dist[(1u << DIST_BITS) + 2048];
  uint32_t pre[1u << PRE_BITS];
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t reverse_bits(uint32_t v, int n) {
  uint32_t r = 0;
  for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; }
  return r;
}

// which: 0 = literal/length alphabet, 1 = distances, 2 = code-length alphabet.  Returns false for a code that is over-subscribed
// or incomplete (but for the distances of a block without matches, or with a single distance), or that needs more sub-table room.
static inline bool build(const uint8_t *lens, int n, uint32_t *table, int table_bits, size_t table_cap, int which) {
  uint16_t count[MAX_CODE_LEN + 1] = {0};
  for (int i = 0; i < n; ++i) ++count[lens[i]];
  const uint32_t psize0 = 1u << table_bits;
  if (count[0] == n) {                                // no code at all: legal for the distances of a block made of literals only
    if (which != 1) return false;
    for (uint32_t i = 0; i < psize0; ++i) table[i] = entry(1, K_INVALID, 0, 0);
    return true;
  }
  uint32_t left = 1;
  for (int l = 1; l <= MAX_CODE_LEN; ++l) {
    left <<= 1;
    if (count[l] > left) return false;
    left -= count[l];
  }
  // incomplete: only the one case the format allows - a single distance code, of one bit (its other half stays invalid)
  if (left != 0 && !(which == 1 && count[1] == 1 && count[0] == n - 1)) return false;
  uint16_t next[MAX_CODE_LEN + 2];
  uint32_t code = 0;
  for (int l = 1; l <= MAX_CODE_LEN; ++l) { code = (code + count[l - 1] * (l > 1 ? 1u : 0u)) << 1; next[l] = (uint16_t) code; }
  // (count[0] must not shift into the first code: the line above skips it for l == 1)
  const uint32_t psize = 1u << table_bits;
  for (uint32_t i = 0; i < psize; ++i) table[i] = entry(0, K_INVALID, 0, 0);
  size_t used = psize;
  // sub-tables: for every primary prefix that long codes share, the width of its sub-table is the longest code under it
  uint8_t sub_bits[1u << LITLEN_BITS];
  if (table_bits < MAX_CODE_LEN) memset(sub_bits, 0, psize);
  auto make = [&](int sym, int len) -> uint32_t {
    if (which == 0) {
      if (sym < 256) return entry((uint32_t) len, K_LITERAL, 0, (uint32_t) sym);
      if (sym == 256) return entry((uint32_t) len, K_END, 0, 0);
      if (sym > 285) return entry((uint32_t) len, K_INVALID, 0, 0);
      return entry((uint32_t) len, K_BASE, LEN_EXTRA[sym - 257], LEN_BASE[sym - 257]);
    }
    if (which == 1) {
      if (sym > 29) return entry((uint32_t) len, K_INVALID, 0, 0);
      return entry((uint32_t) len, K_BASE, DIST_EXTRA[sym], DIST_BASE[sym]);
    }
    return entry((uint32_t) len, K_LITERAL, 0, (uint32_t) sym);
  };
  // first pass over the long codes: sub-table widths
  {
    uint16_t nx[MAX_CODE_LEN + 2];
    memcpy(nx, next, sizeof nx);
    for (int sym = 0; sym < n; ++sym) {
      const int len = lens[sym];
      if (len <= table_bits) { if (len) ++nx[len]; continue; }
      const uint32_t rev = reverse_bits(nx[len]++, len);
      const uint32_t pfx = rev & (psize - 1);
      if ((uint8_t) (len - table_bits) > sub_bits[pfx]) sub_bits[pfx] = (uint8_t) (len - table_bits);
    }
    for (uint32_t p = 0; p < psize; ++p)
      if (table_bits < MAX_CODE_LEN && sub_bits[p]) {
        const size_t size = (size_t) 1 << sub_bits[p];
        if (used + size > table_cap) return false;
        table[p] = entry((uint32_t) table_bits, K_SUB, sub_bits[p], (uint32_t) used);
        for (size_t i = 0; i < size; ++i) table[used + i] = entry(0, K_INVALID, 0, 0);
        used += size;
      }
  }
  for (int sym = 0; sym < n; ++sym) {
    const int len = lens[sym];
    if (!len) continue;
    const uint32_t rev = reverse_bits(next[len]++, len);
    if (len <= table_bits) {
      const uint32_t e = make(sym, len);
      for (uint32_t i = rev; i < psize; i += 1u << len) table[i] = e;
    } else {
      const uint32_t pfx = rev & (psize - 1);
      const uint32_t sb = sub_bits[pfx], off = table[pfx] >> 16;
      const uint32_t e = make(sym, len - table_bits);            // consumes the rest of the code, after the primary bits
      for (uint32_t i = rev >> table_bits; i < (1u << sb); i += 1u << (len - table_bits)) table[off + i] = e;
    }
  }
  // Literal pairs.  A stream of mostly literals (packed bases, qualities) is decoded one table look-up after the other, each waiting
  // for the previous code's length; where two literal codes fit into the primary index together, one look-up yields both.
  if (which == 0) {
    uint32_t prim[1u << LITLEN_BITS];
    memcpy(prim, table, sizeof prim);
    for (uint32_t i = 0; i < psize; ++i) {
      const uint32_t e = prim[i], l1 = e & 31u;
      if (((e >> 5) & 7u) != K_LITERAL || l1 >= (uint32_t) table_bits) continue;
      const uint32_t e2 = prim[i >> l1], l2 = e2 & 31u;
      if (((e2 >> 5) & 7u) == K_LITERAL && l1 + l2 <= (uint32_t) table_bits) table[i] = entry(l1 + l2, K_LIT2, 0, (e >> 16) | ((e2 >> 16) << 8));
    }
  }
  return true;
}

static inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

// 0 = out[0..out_len) holds the block's data and the stream ended exactly there; anything else: not decoded (use zlib)
static inline int inflate_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, Tables &T) {
  const uint8_t *ip = in, *const iend = in + in_len;
  uint8_t *op = out, *const oend = out + out_len;
  uint64_t bb = 0;
  uint32_t bc = 0;
  // bits beyond the end of the input read as zero; `over` counts the bytes made up this way and is checked at the end of each block
  size_t over = 0;
#define MMI_REFILL()                                                                                   \
  do {                                                                                                 \
    if (iend - ip >= 8) { bb |= load64(ip) << bc; ip += (63 - bc) >> 3; bc |= 56; }                    \
    else while (bc <= 56) { if (ip < iend) bb |= (uint64_t) *ip++ << bc; else ++over; bc += 8; }      \
  } while (0)
#define MMI_DROP(N) do { bb >>= (N); bc -= (N); } while (0)
  for (;;) {
    MMI_REFILL();
    const uint32_t last = (uint32_t) bb & 1u, type = ((uint32_t) bb >> 1) & 3u;
    MMI_DROP(3);
    if (type == 0) {                                   // stored
      MMI_DROP(bc & 7u);
      // give the whole bytes still in the bit buffer back to the input
      while (bc >= 8) { if (over) --over; else --ip; bc -= 8; }
      bb = 0; bc = 0;
      if (over || iend - ip < 4) return 1;
      const uint32_t len = ip[0] | ((uint32_t) ip[1] << 8), nlen = ip[2] | ((uint32_t) ip[3] << 8);
      ip += 4;
      if ((len ^ nlen) != 0xFFFFu || (size_t) (iend - ip) < len || (size_t) (oend - op) < len) return 2;
      memcpy(op, ip, len);
      ip += len; op += len;
    } else if (type == 3) return 3;
    else {
      if (type == 1) {                                 // fixed codes
        uint8_t l[288 + 32];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        for (int i = 0; i < 32; ++i) l[288 + i] = 5;
        if (!build(l, 288, T.litlen, LITLEN_BITS, sizeof T.litlen / 4, 0) || !build(l + 288, 32, T.dist, DIST_BITS, sizeof T.dist / 4, 1)) return 4;
      } else {                                         // dynamic codes
        const uint32_t hlit = ((uint32_t) bb & 31u) + 257, hdist = (((uint32_t) bb >> 5) & 31u) + 1, hclen = (((uint32_t) bb >> 10) & 15u) + 4;
        MMI_DROP(14);
        if (hlit > 286 || hdist > 30) return 5;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t pl[19] = {0};
        for (uint32_t i = 0; i < hclen; ++i) {
          if (bc < 3) MMI_REFILL();
          pl[order[i]] = (uint8_t) (bb & 7u);
          MMI_DROP(3);
        }
        if (!build(pl, 19, T.pre, PRE_BITS, sizeof T.pre / 4, 2)) return 6;
        uint8_t l[286 + 30 + 138];
        uint32_t i = 0;
        while (i < hlit + hdist) {
          MMI_REFILL();
          const uint32_t e = T.pre[bb & ((1u << PRE_BITS) - 1)];
          if (((e >> 5) & 7u) != K_LITERAL) return 7;
          MMI_DROP(e & 31u);
          const uint32_t sym = e >> 16;
          if (sym < 16) { l[i++] = (uint8_t) sym; continue; }
          uint32_t rep, val = 0;
          if (sym == 16) { if (!i) return 8; val = l[i - 1]; rep = 3 + ((uint32_t) bb & 3u); MMI_DROP(2); }
          else if (sym == 17) { rep = 3 + ((uint32_t) bb & 7u); MMI_DROP(3); }
          else { rep = 11 + ((uint32_t) bb & 127u); MMI_DROP(7); }
          if (i + rep > hlit + hdist) return 9;
          memset(l + i, (int) val, rep);
          i += rep;
        }
        if (l[256] == 0) return 10;                    // no end-of-block code
        if (!build(l, (int) hlit, T.litlen, LITLEN_BITS, sizeof T.litlen / 4, 0) || !build(l + hlit, (int) hdist, T.dist, DIST_BITS, sizeof T.dist / 4, 1)) return 11;
      }
      // the block's symbols.  Fast loop while both buffers have room to spare: no bounds checks per symbol, up to four literals per
      // refill (4 x 11 bits of the 56), matches copied 8 bytes at a time (the spare room covers the overshoot); whatever is left - the
      // last few hundred bytes of a block, or a short input - goes through the careful loop below, which also ends the block.
      bool ended = false;
      while ((size_t) (oend - op) >= 8 + 258 + 8 && (size_t) (iend - ip) >= 16) {      // (16: two 8-byte loads, the second up to 7 bytes on)
        bb |= load64(ip) << bc; ip += (63 - bc) >> 3; bc |= 56;
        uint32_t e = T.litlen[bb & ((1u << LITLEN_BITS) - 1)];
        if (is_lit(e)) {                                // one or two literals out of the primary table
          int lits = 0;
          do {
            MMI_DROP(e & 31u);
            op[0] = (uint8_t) (e >> 16); op[1] = (uint8_t) (e >> 24);      // (the second byte is only kept for a pair)
            op += 1 + ((e >> 7) & 1u);
            e = T.litlen[bb & ((1u << LITLEN_BITS) - 1)];
          } while (is_lit(e) && ++lits < 4);
          if (is_lit(e)) continue;
          bb |= load64(ip) << bc; ip += (63 - bc) >> 3; bc |= 56;
        }
        if (((e >> 5) & 7u) == K_SUB) { MMI_DROP(LITLEN_BITS); e = T.litlen[(e >> 16) + (bb & ((1u << ((e >> 8) & 31u)) - 1))]; }
        const uint32_t kind = (e >> 5) & 7u;
        MMI_DROP(e & 31u);
        if (kind == K_END) { ended = true; break; }
        if (kind == K_LITERAL) { *op++ = (uint8_t) (e >> 16); continue; }
        if (kind != K_BASE) return 13;
        const uint32_t xl = (e >> 8) & 31u;
        const uint32_t len = (e >> 16) + ((uint32_t) bb & ((1u << xl) - 1));
        MMI_DROP(xl);
        uint32_t d = T.dist[bb & ((1u << DIST_BITS) - 1)];
        if (((d >> 5) & 7u) == K_SUB) { MMI_DROP(DIST_BITS); d = T.dist[(d >> 16) + (bb & ((1u << ((d >> 8) & 31u)) - 1))]; }
        if (((d >> 5) & 7u) != K_BASE) return 14;
        MMI_DROP(d & 31u);
        const uint32_t xd = (d >> 8) & 31u;
        const uint32_t dist = (d >> 16) + ((uint32_t) bb & ((1u << xd) - 1));
        MMI_DROP(xd);
        if (dist > (size_t) (op - out)) return 15;
        const uint8_t *src = op - dist;
        if (dist >= 8) {
          uint8_t *dst = op;
          const uint8_t *const stop = op + len;
          do { memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < stop);
        } else if (dist == 1) memset(op, *src, len);
        else for (uint32_t i = 0; i < len; ++i) op[i] = src[i];
        op += len;
      }
      while (!ended) {
        MMI_REFILL();                                   // >= 56 bits: a length code, its extra bits, a distance code and its extra bits need at most 48
        uint32_t e = T.litlen[bb & ((1u << LITLEN_BITS) - 1)];
        if (((e >> 5) & 7u) == K_SUB) { MMI_DROP(LITLEN_BITS); e = T.litlen[(e >> 16) + (bb & ((1u << ((e >> 8) & 31u)) - 1))]; }
        const uint32_t kind = (e >> 5) & 7u;
        MMI_DROP(e & 31u);
        if (kind == K_LITERAL) {
          if (op == oend) return 12;
          *op++ = (uint8_t) (e >> 16);
          continue;
        }
        if (kind == K_LIT2) {
          if (oend - op < 2) return 12;
          *op++ = (uint8_t) (e >> 16); *op++ = (uint8_t) (e >> 24);
          continue;
        }
        if (kind == K_END) break;
        if (kind != K_BASE) return 13;
        const uint32_t xl = (e >> 8) & 31u;
        const uint32_t len = (e >> 16) + ((uint32_t) bb & ((1u << xl) - 1));
        MMI_DROP(xl);
        uint32_t d = T.dist[bb & ((1u << DIST_BITS) - 1)];
        if (((d >> 5) & 7u) == K_SUB) { MMI_DROP(DIST_BITS); d = T.dist[(d >> 16) + (bb & ((1u << ((d >> 8) & 31u)) - 1))]; }
        if (((d >> 5) & 7u) != K_BASE) return 14;
        MMI_DROP(d & 31u);
        const uint32_t xd = (d >> 8) & 31u;
        const uint32_t dist = (d >> 16) + ((uint32_t) bb & ((1u << xd) - 1));
        MMI_DROP(xd);
        if (dist > (size_t) (op - out) || len > (size_t) (oend - op)) return 15;
        const uint8_t *src = op - dist;
        if (dist >= 8 && (size_t) (oend - op) >= len + 8) {        // 8 bytes at a time; may write up to 7 bytes past the match, inside this block's output
          uint8_t *dst = op;
          const uint8_t *const stop = op + len;
          do { memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < stop);
        } else if (dist == 1) memset(op, *src, len);
        else for (uint32_t i = 0; i < len; ++i) op[i] = src[i];
        op += len;
      }
      if (over > (bc >> 3)) return 16;                 // the block used bits the input does not have
    }
    if (last) break;
  }
#undef MMI_REFILL
#undef MMI_DROP
  return op == oend ? 0 : 17;
}

}  // namespace mminf

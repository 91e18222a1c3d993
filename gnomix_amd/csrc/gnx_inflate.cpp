// gnx_inflate.cpp — raw DEFLATE (RFC 1951) decoder for BGZF blocks, written for throughput.
//
// The reference reads its query through gzip.open + scikit-allel (src/utils.py:55-81); its demo query is a .vcf.gz.  A BGZF file
// is a sequence of independent <= 64 KiB gzip members, so the reader inflates them on every core — but with zlib's inflate()
// (byte-wise bit reader, 9-bit tables, one symbol per loop trip) the 16 host cores of a GPU box produce ~5 GB/s of text and the
// GPU waits for the parser 98 % of the time (round 4: 6.3 k haplotypes/s from .vcf.gz against 54.8 k from plain text).
// This decoder is the well-known word-at-a-time design:
//   * a 64-bit bit buffer refilled with ONE unaligned 8-byte load (>= 56 valid bits after every refill: a whole
//     length/distance pair — 15 + 5 + 15 + 13 bits — needs no second refill),
//   * an 11-bit primary table for literal/length codes (most symbols of genotype text: one lookup), 8-bit for distances,
//     canonical-Huffman subtables behind both for longer codes,
//   * literals decoded up to three per trip, matches copied in 8-byte words (byte-wise only when the distance is < 8),
//   * a margin-free fast loop while both buffers have slack, a bounds-checked loop for the last bytes of a block.
// Output goes to a caller-owned buffer of known size (BGZF stores ISIZE): a block that does not decode to exactly that size
// is an error and the caller falls back to zlib for it.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/gnomix_io.h"

namespace {

constexpr int LL_BITS = 11, D_BITS = 8;
constexpr int LL_ENOUGH = 2048 + 1024, D_ENOUGH = 256 + 512;  // primary + subtables (zlib's ENOUGH bounds for these roots are lower)

// table entry: bits 0-4 code length to consume (subtable pointer: the primary bits), bits 8-12 extra bits (subtable pointer: its index
// bits), bit 13 end of block, bit 14 subtable pointer, bit 15 literal, bits 16-31 literal / length base / distance base / subtable start
constexpr uint32_t F_EOB = 1u << 13, F_SUB = 1u << 14, F_LIT = 1u << 15, F_BAD = 1u << 7;

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t ll_entry(int sym, int len) {
  if (sym < 256) return F_LIT | ((uint32_t)sym << 16) | (uint32_t)len;
  if (sym == 256) return F_EOB | (uint32_t)len;
  if (sym > 285) return F_BAD | (uint32_t)len;
  return ((uint32_t)kLenBase[sym - 257] << 16) | ((uint32_t)kLenExtra[sym - 257] << 8) | (uint32_t)len;
}
inline uint32_t d_entry(int sym, int len) {
  if (sym > 29) return F_BAD | (uint32_t)len;
  return ((uint32_t)kDistBase[sym] << 16) | ((uint32_t)kDistExtra[sym] << 8) | (uint32_t)len;
}

// canonical Huffman decoding table with `root` primary bits (zlib's inflate_table construction: codes in order of length, the
// bit-reversed code incremented backwards, a subtable opened whenever a long code's root prefix changes).  Returns false for an
// over-subscribed set, or an incomplete one other than the single-code case RFC 1951 allows for distances.
//
// This function restates the table construction of zlib's inftrees.c (inflate_table) closely — the same counting / offset / fill
// loop with its variables — on this decoder's own entry format.  zlib is third-party code (not part of the Gnomix reference), used
// under its licence, whose notice follows:
//
//   zlib.h -- interface of the 'zlib' general purpose compression library
//   Copyright (C) 1995-2024 Jean-loup Gailly and Mark Adler
//
//   This software is provided 'as-is', without any express or implied warranty.  In no event will the authors be held liable for
//   any damages arising from the use of this software.
//
//   Permission is granted to anyone to use this software for any purpose, including commercial applications, and to alter it and
//   redistribute it freely, subject to the following restrictions:
//
//   1. The origin of this software must not be misrepresented; you must not claim that you wrote the original software. If you use
//      this software in a product, an acknowledgment in the product documentation would be appreciated but is not required.
//   2. Altered source versions must be plainly marked as such, and must not be misrepresented as being the original software.
//   3. This notice may not be removed or altered from any source distribution.
//
//   Jean-loup Gailly        Mark Adler
//   jloup@gzip.org          madler@alumni.caltech.edu
//
// (Altered: this is NOT zlib's source; it is a re-implementation of one of its algorithms inside a different decoder.)
template <typename MakeEntry>
bool build_table(const uint8_t* lens, int n, int root, uint32_t* table, int enough, MakeEntry make, bool allow_incomplete) {
  uint16_t count[16] = {0}, offs[16], work[320];
  for (int i = 0; i < n; ++i) count[lens[i]]++;
  int max = 15;
  while (max >= 1 && count[max] == 0) --max;
  if (max == 0) {  // no codes at all: every lookup is invalid
    for (int i = 0; i < (1 << root); ++i) table[i] = F_BAD | 1u;
    return allow_incomplete;
  }
  int min = 1;
  while (min < max && count[min] == 0) ++min;
  int left = 1;
  for (int len = 1; len <= 15; ++len) {
    left <<= 1;
    left -= count[len];
    if (left < 0) return false;  // over-subscribed
  }
  if (left > 0 && !(allow_incomplete && max == 1)) return false;  // incomplete set: only zlib's single one-bit code (distances)
  offs[1] = 0;
  for (int len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
  for (int i = 0; i < n; ++i)
    if (lens[i]) work[offs[lens[i]]++] = (uint16_t)i;

  if (root > max) root = max;  // (the caller indexes with its own root: replicate below)
  const int croot = root;
  int len = min, sym = 0, curr = croot, drop = 0;
  uint32_t huff = 0, low = (uint32_t)-1;
  const uint32_t mask = (1u << croot) - 1;
  uint32_t* next = table;
  int used = 1 << croot;
  if (used > enough) return false;
  uint16_t cnt[16];
  memcpy(cnt, count, sizeof(cnt));
  for (;;) {
    const uint32_t here = make((int)work[sym], len - drop);
    const uint32_t incr = 1u << (len - drop);
    uint32_t fill = 1u << curr;
    const uint32_t tmin = fill;
    do {
      fill -= incr;
      next[(huff >> drop) + fill] = here;
    } while (fill != 0);
    uint32_t inc = 1u << (len - 1);
    while (huff & inc) inc >>= 1;
    if (inc != 0) {
      huff &= inc - 1;
      huff += inc;
    } else
      huff = 0;
    ++sym;
    if (--cnt[len] == 0) {
      if (len == max) break;
      len = lens[work[sym]];
    }
    if (len > croot && (huff & mask) != low) {
      if (drop == 0) drop = croot;
      next += tmin;
      curr = len - drop;
      int lft = 1 << curr;
      while (curr + drop < max) {
        lft -= cnt[curr + drop];
        if (lft <= 0) break;
        ++curr;
        lft <<= 1;
      }
      used += 1 << curr;
      if (used > enough) return false;
      low = huff & mask;
      table[low] = F_SUB | ((uint32_t)(next - table) << 16) | ((uint32_t)curr << 8) | (uint32_t)croot;
    }
  }
  if (huff != 0) {  // incomplete code (single distance code): the unused slot is invalid
    next[huff >> drop] = F_BAD | (uint32_t)(len - drop);
  }
  return true;
}

struct Tables {
  uint32_t ll[LL_ENOUGH];
  uint32_t d[D_ENOUGH];
  int ll_root, d_root;
};

// build_table shrinks the root to the longest code; the decoder always indexes with LL_BITS / D_BITS: replicate a smaller table
void widen(uint32_t* t, int have, int want) {
  for (int b = have; b < want; ++b) memcpy(t + (1 << b), t, sizeof(uint32_t) << b);
}

bool build_pair(Tables& T, const uint8_t* ll_lens, int nll, const uint8_t* d_lens, int nd) {
  int mx = 0;
  for (int i = 0; i < nll; ++i) mx = ll_lens[i] > mx ? ll_lens[i] : mx;
  if (mx == 0 || ll_lens[256] == 0) return false;  // no end-of-block code
  if (!build_table(ll_lens, nll, LL_BITS, T.ll, LL_ENOUGH, ll_entry, false)) return false;
  if (mx < LL_BITS) widen(T.ll, mx, LL_BITS);
  int md = 0;
  for (int i = 0; i < nd; ++i) md = d_lens[i] > md ? d_lens[i] : md;
  if (!build_table(d_lens, nd, D_BITS, T.d, D_ENOUGH, d_entry, true)) return false;
  if (md == 0) {
    for (int i = 0; i < (1 << D_BITS); ++i) T.d[i] = F_BAD | 1u;
  } else if (md < D_BITS)
    widen(T.d, md, D_BITS);
  return true;
}

const Tables& fixed_tables() {
  static const Tables* T = [] {
    Tables* t = new Tables();
    uint8_t ll[288], d[32];
    for (int i = 0; i < 144; ++i) ll[i] = 8;
    for (int i = 144; i < 256; ++i) ll[i] = 9;
    for (int i = 256; i < 280; ++i) ll[i] = 7;
    for (int i = 280; i < 288; ++i) ll[i] = 8;
    for (int i = 0; i < 32; ++i) d[i] = 5;
    build_pair(*t, ll, 288, d, 32);
    return t;
  }();
  return *T;
}

struct Bits {
  const uint8_t* in;
  const uint8_t* in_end;
  uint64_t buf = 0;
  int n = 0;  // valid bits
  int over = 0;  // bytes of zero padding consumed past the end (a valid stream never needs more than a few)
  inline void refill_fast() {  // needs 8 readable bytes at `in`
    uint64_t w;
    memcpy(&w, in, 8);
    buf |= w << n;
    in += (63 - n) >> 3;
    n |= 56;
  }
  inline void refill_safe() {
    while (n <= 56) {
      if (in < in_end) buf |= (uint64_t)*in++ << n;
      else ++over;
      n += 8;
    }
  }
  inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1)); }
  inline void drop(int k) { buf >>= k; n -= k; }
};

// one compressed block's symbols.  FAST: no bounds checks — the caller guarantees >= 8 input bytes and >= 274 output bytes of slack
// at the top of every trip.  Returns 0 at end of block, 1 when the slack ran out (continue in the other mode), -1 on corrupt data.
template <bool FAST>
__attribute__((always_inline)) inline int run_block(Bits& bits, const Tables& T, uint8_t* out_begin, uint8_t*& outp, uint8_t* out_end) {
  // the bit reader lives in locals for the whole block: `out` is a byte pointer, which may alias anything — through the Bits
  // reference every stored byte would force the buffer, the count and the input pointer back to memory and in again
  Bits b = bits;
  uint8_t* out = outp;
  const uint32_t* const ll = T.ll;
  const uint32_t* const dt = T.d;
  int rc;
  for (;;) {
    if (FAST) {
      if (b.in_end - b.in < 8 || out_end - out < 274) { rc = 1; break; }
      b.refill_fast();
    } else {
      b.refill_safe();
      if (b.over > 8) { rc = -1; break; }
    }
    uint32_t e = ll[b.peek(LL_BITS)];
    if (e & F_SUB) {
      b.drop((int)(e & 31));
      e = ll[(e >> 16) + b.peek((int)((e >> 8) & 31))];
    }
    b.drop((int)(e & 31));
    if (e & F_LIT) {
      if (!FAST && out >= out_end) { rc = -1; break; }
      *out++ = (uint8_t)(e >> 16);
      if (FAST) {  // up to two more literals from the bits already in the buffer (>= 56 - 15 bits are left)
        e = ll[b.peek(LL_BITS)];
        if ((e & (F_LIT | F_SUB)) == F_LIT) {
          b.drop((int)(e & 31));
          *out++ = (uint8_t)(e >> 16);
          e = ll[b.peek(LL_BITS)];
          if ((e & (F_LIT | F_SUB)) == F_LIT) {
            b.drop((int)(e & 31));
            *out++ = (uint8_t)(e >> 16);
          }
        }
      }
      continue;
    }
    if (e & (F_EOB | F_BAD)) {
      rc = (e & F_EOB) ? 0 : -1;
      break;
    }
    const int xl = (int)((e >> 8) & 31);
    const uint32_t length = (e >> 16) + b.peek(xl);
    b.drop(xl);
    if (!FAST) b.refill_safe();
    uint32_t f = dt[b.peek(D_BITS)];
    if (f & F_SUB) {
      b.drop((int)(f & 31));
      f = dt[(f >> 16) + b.peek((int)((f >> 8) & 31))];
    }
    if (f & F_BAD) { rc = -1; break; }
    b.drop((int)(f & 31));
    const int xd = (int)((f >> 8) & 31);
    if (FAST && b.n < xd) b.refill_fast();  // (15 + 5 + 15 bits are gone at worst: 21 are left, 13 may be needed — never taken)
    const uint32_t dist = (f >> 16) + b.peek(xd);
    b.drop(xd);
    if (dist > (size_t)(out - out_begin)) { rc = -1; break; }
    const uint8_t* src = out - dist;
    if (FAST) {
      uint8_t* const end = out + length;
      if (dist >= 8) {
        uint64_t w;
        memcpy(&w, src, 8);  // most matches of genotype text are a field or two long: one word, no loop
        memcpy(out, &w, 8);
        if (length > 8) {
          src += 8;
          out += 8;
          do {
            memcpy(&w, src, 8);
            memcpy(out, &w, 8);
            src += 8;
            out += 8;
          } while (out < end);
        }
      } else if (dist == 1) {
        memset(out, *src, length);
      } else {
        // a short period ("0|0\t" repeats at distance 4): the first 8 bytes one by one — each may read what the one before wrote —
        // then the copy continues in words from the multiple of the period that is >= 8 (its bytes exist by now)
        for (int i = 0; i < 8; ++i) out[i] = src[i];
        static const uint8_t kWide[8] = {0, 8, 8, 9, 8, 10, 12, 14};
        out += 8;
        src = out - kWide[dist];
        while (out < end) {
          uint64_t w;
          memcpy(&w, src, 8);
          memcpy(out, &w, 8);
          src += 8;
          out += 8;
        }
      }
      out = end;
    } else {
      if ((size_t)(out_end - out) < length) { rc = -1; break; }
      for (uint32_t i = 0; i < length; ++i) out[i] = src[i];
      out += length;
    }
  }
  bits = b;
  outp = out;
  return rc;
}

}  // namespace

// raw DEFLATE stream `in` -> exactly out_n bytes at `out`.  0: ok; -1: corrupt, truncated, or a size other than out_n.
// (two clones, chosen once by the loader: with BMI2 the variable shifts of the bit reader are shrx / bzhi, without a detour through cl)
extern "C" __attribute__((target_clones("default", "bmi2"))) int gnx_io_inflate_raw(const uint8_t* in, size_t in_n, uint8_t* out, size_t out_n) {
  if ((!in && in_n) || (!out && out_n)) return -1;
  Bits b;
  b.in = in;
  b.in_end = in + in_n;
  uint8_t* o = out;
  uint8_t* const o_end = out + out_n;
  Tables dyn;
  for (;;) {
    b.refill_safe();
    const uint32_t last = b.peek(1), type = (b.peek(3) >> 1);
    b.drop(3);
    if (type == 0) {  // stored: skip to the byte boundary, LEN / NLEN, bytes
      b.drop(b.n & 7);
      b.refill_safe();
      const uint32_t len = b.peek(16);
      b.drop(16);
      const uint32_t nlen = b.peek(16);
      b.drop(16);
      if ((len ^ nlen) != 0xFFFFu) return -1;
      // give the whole bytes still in the bit buffer back to the input (zero padding appended past the end is not input)
      const int real = (b.n >> 3) - b.over;
      if (real < 0) return -1;  // LEN / NLEN themselves came out of the padding
      const uint8_t* p = b.in - real;
      b.buf = 0;
      b.n = 0;
      b.over = 0;
      if ((size_t)(b.in_end - p) < len || (size_t)(o_end - o) < len) return -1;
      memcpy(o, p, len);
      o += len;
      b.in = p + len;
    } else if (type == 1 || type == 2) {
      const Tables* T = &fixed_tables();
      if (type == 2) {
        b.refill_safe();
        const int hlit = (int)b.peek(5) + 257;
        b.drop(5);
        const int hdist = (int)b.peek(5) + 1;
        b.drop(5);
        const int hclen = (int)b.peek(4) + 4;
        b.drop(4);
        if (hlit > 286 || hdist > 30) return -1;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; ++i) {
          b.refill_safe();
          cl[order[i]] = (uint8_t)b.peek(3);
          b.drop(3);
        }
        uint32_t clt[128 + 64];
        auto cl_entry = [](int sym, int len) { return ((uint32_t)sym << 16) | (uint32_t)len; };
        int mc = 0;
        for (int i = 0; i < 19; ++i) mc = cl[i] > mc ? cl[i] : mc;
        if (mc == 0 || !build_table(cl, 19, 7, clt, 128 + 64, cl_entry, false)) return -1;
        if (mc < 7) widen(clt, mc, 7);
        uint8_t lens[286 + 30 + 138];
        int i = 0;
        const int total = hlit + hdist;
        while (i < total) {
          b.refill_safe();
          const uint32_t e = clt[b.peek(7)];
          b.drop((int)(e & 31));
          const int sym = (int)(e >> 16);
          if (sym < 16) lens[i++] = (uint8_t)sym;
          else {
            int rep;
            uint8_t v = 0;
            if (sym == 16) {
              if (i == 0) return -1;
              v = lens[i - 1];
              rep = 3 + (int)b.peek(2);
              b.drop(2);
            } else if (sym == 17) {
              rep = 3 + (int)b.peek(3);
              b.drop(3);
            } else {
              rep = 11 + (int)b.peek(7);
              b.drop(7);
            }
            if (i + rep > total) return -1;
            memset(lens + i, v, (size_t)rep);
            i += rep;
          }
        }
        if (b.over > 8) return -1;
        if (!build_pair(dyn, lens, hlit, lens + hlit, hdist)) return -1;
        T = &dyn;
      }
      for (;;) {
        int rc = run_block<true>(b, *T, out, o, o_end);
        if (rc == 1) rc = run_block<false>(b, *T, out, o, o_end);
        if (rc == 0) break;
        if (rc < 0) return -1;
      }
    } else
      return -1;
    if (last) break;
  }
  return (o == o_end && b.over <= 8) ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------------------------------------
// CRC-32 (the gzip / BGZF trailer's: reflected polynomial 0xEDB88320) by carry-less multiplication.
// A BGZF block carries the CRC of its text; the fast decoder above checks sizes only, so a block whose payload was damaged into
// ANOTHER valid deflate stream of the same length would pass silently.  zlib's crc32() at ~1.3 GB/s per thread would
// cost half of what the decoder gained; folding 64 bytes per iteration with PCLMULQDQ runs at > 10 GB/s per thread, so the check is
// always on.  Method: "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction" (Gopal et al., Intel, 2009): four
// 128-bit lanes folded by x^(512+-32) mod P, reduced to one by x^(128+-32), then 128 -> 64 -> 32 bits by Barrett reduction.
// The bytes before the first and after the last whole 16-byte piece, and hosts without the instruction, go through zlib's crc32().
// ------------------------------------------------------------------------------------------------------------------------------
#include <immintrin.h>
#include <zlib.h>

namespace {
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(const uint8_t* p, size_t n, uint32_t crc) {  // n: multiple of 16, >= 64; crc: raw register (not inverted back)
  const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);   // x^(4*128-32), x^(4*128+32) mod P (bit-reflected)
  const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);   // x^(128-32), x^(128+32)
  const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124);                // x^64
  const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);   // mu, P
  const __m128i lo32 = _mm_set_epi32(0, ~0, 0, ~0);
  __m128i a = _mm_loadu_si128((const __m128i*)p), b = _mm_loadu_si128((const __m128i*)(p + 16));
  __m128i c = _mm_loadu_si128((const __m128i*)(p + 32)), d = _mm_loadu_si128((const __m128i*)(p + 48));
  a = _mm_xor_si128(a, _mm_cvtsi32_si128((int)crc));
  p += 64; n -= 64;
  auto fold = [](__m128i x, __m128i k, __m128i next) __attribute__((target("pclmul,sse4.1"))) {
    return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), next);
  };
  while (n >= 64) {
    a = fold(a, k1k2, _mm_loadu_si128((const __m128i*)p));
    b = fold(b, k1k2, _mm_loadu_si128((const __m128i*)(p + 16)));
    c = fold(c, k1k2, _mm_loadu_si128((const __m128i*)(p + 32)));
    d = fold(d, k1k2, _mm_loadu_si128((const __m128i*)(p + 48)));
    p += 64; n -= 64;
  }
  a = fold(a, k3k4, b);
  a = fold(a, k3k4, c);
  a = fold(a, k3k4, d);
  while (n >= 16) {
    a = fold(a, k3k4, _mm_loadu_si128((const __m128i*)p));
    p += 16; n -= 16;
  }
  // 128 -> 64 bits
  __m128i t = _mm_clmulepi64_si128(a, k3k4, 0x10);
  a = _mm_xor_si128(_mm_srli_si128(a, 8), t);
  t = _mm_srli_si128(a, 4);
  a = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(a, lo32), k5, 0x00), t);
  // Barrett: 64 -> 32 bits
  t = _mm_clmulepi64_si128(_mm_and_si128(a, lo32), poly, 0x10);
  t = _mm_clmulepi64_si128(_mm_and_si128(t, lo32), poly, 0x00);
  a = _mm_xor_si128(a, t);
  return (uint32_t)_mm_extract_epi32(a, 1);
}
}  // namespace

extern "C" uint32_t gnx_io_crc32(const uint8_t* data, size_t n) {
  static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  if (!have || n < 64) return (uint32_t)crc32(0L, data, (uInt)n);
  const size_t body = n & ~(size_t)15;
  uint32_t c = ~crc32_clmul(data, body, 0xFFFFFFFFu);          // the CRC of the first `body` bytes, finalised
  return body < n ? (uint32_t)crc32(c, data + body, (uInt)(n - body)) : c;
}

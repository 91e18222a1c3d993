// gnx_vcf.cpp — phased VCF text -> variant-major 2-bit genotypes on every core of the box (host code, no HIP).
// Contract: include/gnomix_io.h (gnx_vcf_read and the accessors).  Reference: src/utils.py:55-81 (read_vcf = scikit-allel's
// C parser behind gzip.open), the step the reference's own notebook names as the largest cost of a run.
//
// ONE pass.  The record area is cut into chunks of a few MB; a worker claims a chunk, brings its bytes into a buffer of its
// own with pread (measured on the 2 x 64-core host of an MI355X box, scripts/dev/io_probe.cpp: 50 GB/s with 32-64 readers,
// against 22 GB/s through a shared mapping — whose page-table setup and 0.1 s/4 GB teardown it also avoids) and parses the
// records that START in the chunk into chunk-local columns.  When all chunks are done a prefix sum places every chunk and the
// columns are copied into the final arrays in parallel (the 2-bit matrix is an eighth of the text).  gzip input is inflated
// into memory first (BGZF: every 64 KB block on its own worker; a plain gzip stream is serial by construction) and the
// chunks point into that buffer.
//
// A record whose FORMAT is exactly "GT" and whose sample area is 4 * n_samples - 1 bytes long is tried on the fixed-width
// path: 32 bytes of text ("a|b\t" x 8) are validated and squeezed to 16 two-bit fields with a handful of AVX2 operations
// (allele byte ^ '0' is 0, 1 or 0x1E for '.': its low two bits ARE the code); anything else (other FORMAT keys, multi-digit
// alleles, haploid calls) takes the per-sample path.
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>

#include "gnx_io.h"

namespace {
// ------------------------------------------------------------------------------------------------------------------------
// text source: a file descriptor (plain text, pread) or memory (inflated gzip / BGZF)
// ------------------------------------------------------------------------------------------------------------------------
struct BgzfBlock {
  size_t poff, psize, uoff, usize;  // payload (raw deflate) in the file, place in the text
  uint32_t crc;                     // CRC-32 of the block's text (the member's trailer)
};

struct Source {
  int fd = -1;
  // BGZF: the file stays mapped and compressed; fetch() inflates the blocks that cover a window into the calling thread's buffer,
  // so the text of a whole-chromosome query (gigabytes) is never written to DRAM and read back — it is parsed out of the cache
  // it was inflated into (inflating everything first cost 0.75 s of the 0.88 s a chr22 x 5 000-sample .vcf.gz took)
  const uint8_t* zmap = nullptr;
  size_t zmap_n = 0;
  std::vector<BgzfBlock> blocks;
  const char* mem = nullptr;
  char* owned = nullptr;
  size_t n = 0;
  int compression = 0;
  int64_t file_bytes = 0;
  size_t owned_map = 0;  // > 0: `owned` is an anonymous mapping of this many bytes (transparent huge pages), else malloc
  ~Source() {
    if (fd >= 0) close(fd);
    if (owned_map) munmap(owned, owned_map);
    else free(owned);
    if (zmap) munmap(const_cast<uint8_t*>(zmap), zmap_n);
  }
  bool use_zlib = false;  // GNX_VCF_ZLIB (read per open: an A/B switch): zlib's inflate() instead of gnx_io_inflate_raw
  bool inflate_block(const uint8_t* in, size_t in_n, uint8_t* out, size_t out_n) const {
    if (out_n == 0) return true;
    if (!use_zlib && gnx_io_inflate_raw(in, in_n, out, out_n) == 0) return true;
    z_stream s;
    memset(&s, 0, sizeof(s));
    if (inflateInit2(&s, -15) != Z_OK) return false;
    s.next_in = const_cast<Bytef*>(in);
    s.avail_in = (uInt)in_n;
    s.next_out = (Bytef*)out;
    s.avail_out = (uInt)out_n;
    const bool ok = inflate(&s, Z_FINISH) == Z_STREAM_END && s.avail_out == 0;
    inflateEnd(&s);
    return ok;
  }
  // bytes [off, off + len) -> pointer; file sources copy into buf (grown as needed)
  const char* fetch(size_t off, size_t len, std::vector<char>& buf, std::string* err) const {
    if (mem) return mem + off;
    if (!blocks.empty()) {
      if (len == 0) return "";
      // the block that holds byte `off`: the last one with uoff <= off
      size_t lo = 0, hi = blocks.size();
      while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (blocks[mid].uoff <= off) lo = mid;
        else hi = mid;
      }
      const size_t base = blocks[lo].uoff, need_end = off + len;
      size_t j = lo, end = base;
      while (end < need_end && j < blocks.size()) {
        end = blocks[j].uoff + blocks[j].usize;
        ++j;
      }
      if (end < need_end) {
        *err = "read past the end of the BGZF text";
        return nullptr;
      }
      try {
        if (buf.size() < end - base) buf.resize(end - base);
      } catch (const std::bad_alloc&) {  // (this runs inside a parsing thread: nothing may be thrown through it)
        *err = "out of memory inflating a BGZF window";
        return nullptr;
      }
      for (size_t k = lo; k < j; ++k)
        if (!inflate_block(zmap + blocks[k].poff, blocks[k].psize, (uint8_t*)buf.data() + (blocks[k].uoff - base), blocks[k].usize) ||
            gnx_io_crc32((const uint8_t*)buf.data() + (blocks[k].uoff - base), blocks[k].usize) != blocks[k].crc) {
          *err = "corrupt BGZF block (does not inflate to its stored size and CRC-32)";
          return nullptr;
        }
      return buf.data() + (off - base);
    }
    if (buf.size() < len) buf.resize(len);
    size_t got = 0;
    while (got < len) {
      const ssize_t r = pread(fd, buf.data() + got, len - got, (off_t)(off + got));
      if (r < 0) {
        if (errno == EINTR) continue;
        *err = std::string("read error: ") + strerror(errno);
        return nullptr;
      }
      if (r == 0) {
        *err = "file shrank while it was read";
        return nullptr;
      }
      got += (size_t)r;
    }
    return buf.data();
  }
};

inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd32(const uint8_t* p) { return rd16(p) | (rd16(p + 2) << 16); }

bool bgzf_table(const uint8_t* z, size_t zn, std::vector<BgzfBlock>& blocks, size_t& total) {
  size_t o = 0;
  total = 0;
  while (o < zn) {
    if (zn - o < 18) return false;
    const uint8_t* h = z + o;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) return false;  // FEXTRA only: what bgzip writes
    const size_t xend = 12 + (size_t)rd16(h + 10);
    if (zn - o < xend + 8) return false;
    int64_t bsize = -1;
    for (size_t x = 12; x + 4 <= xend;) {
      const uint32_t slen = rd16(h + x + 2);
      if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= xend) bsize = (int64_t)rd16(h + x + 4) + 1;
      x += 4 + slen;
    }
    if (bsize < (int64_t)xend + 8 || (size_t)bsize > zn - o) return false;
    BgzfBlock b;
    b.poff = o + xend;
    b.psize = (size_t)bsize - xend - 8;
    b.usize = rd32(h + bsize - 4);
    b.crc = rd32(h + bsize - 8);
    if (b.usize > 65536) return false;  // not BGZF (the format bounds a block's text at 64 KiB): the serial gzip path decides
    b.uoff = total;
    total += b.usize;
    blocks.push_back(b);
    o += (size_t)bsize;
  }
  return !blocks.empty();
}

int inflate_serial(const uint8_t* z, size_t zn, char** out, size_t* out_n) {
  size_t cap = std::max<size_t>((size_t)64 << 20, zn * 6);
  char* buf = (char*)malloc(cap);
  if (!buf) return gnx_io_fail(GNX_ENOMEM, "vcf: out of memory inflating");
  z_stream s;
  memset(&s, 0, sizeof(s));
  if (inflateInit2(&s, 15 + 32) != Z_OK) {
    free(buf);
    return gnx_io_fail(GNX_EINVAL, "vcf: inflateInit2 failed");
  }
  size_t in_pos = 0, produced = 0;
  int rc_out = GNX_OK;
  for (;;) {
    if (s.avail_in == 0 && in_pos < zn) {
      const size_t take = std::min<size_t>(zn - in_pos, (size_t)1 << 30);
      s.next_in = const_cast<Bytef*>(z + in_pos);
      s.avail_in = (uInt)take;
      in_pos += take;
    }
    if (produced == cap) {
      cap *= 2;
      char* nb = (char*)realloc(buf, cap);
      if (!nb) {
        rc_out = gnx_io_fail(GNX_ENOMEM, "vcf: out of memory inflating");
        break;
      }
      buf = nb;
    }
    const size_t room = std::min<size_t>(cap - produced, (size_t)1 << 30);
    s.next_out = (Bytef*)buf + produced;
    s.avail_out = (uInt)room;
    const int rc = inflate(&s, Z_NO_FLUSH);
    produced += room - s.avail_out;
    if (rc == Z_STREAM_END) {
      if (s.avail_in == 0 && in_pos >= zn) break;
      if (inflateReset(&s) != Z_OK) {  // next member of a multi-member file
        rc_out = gnx_io_fail(GNX_EINVAL, "vcf: inflateReset failed");
        break;
      }
      continue;
    }
    if (rc != Z_OK && rc != Z_BUF_ERROR) {
      rc_out = gnx_io_fail(GNX_EINVAL, std::string("vcf: corrupt gzip stream (") + (s.msg ? s.msg : "inflate error") + ")");
      break;
    }
    if (rc == Z_BUF_ERROR && s.avail_in == 0 && in_pos >= zn) {
      rc_out = gnx_io_fail(GNX_EINVAL, "vcf: truncated gzip stream");
      break;
    }
  }
  inflateEnd(&s);
  if (rc_out != GNX_OK) {
    free(buf);
    return rc_out;
  }
  *out = buf;
  *out_n = produced;
  return GNX_OK;
}

int open_source(const char* path, int n_threads, Source& t) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return gnx_io_fail(GNX_EINVAL, std::string("vcf: cannot open ") + path + ": " + strerror(errno));
  struct stat st;
  if (fstat(fd, &st) != 0) {
    close(fd);
    return gnx_io_fail(GNX_EINVAL, std::string("vcf: cannot stat ") + path);
  }
  t.file_bytes = (int64_t)st.st_size;
  const size_t zn = (size_t)st.st_size;
  uint8_t magic[2] = {0, 0};
  if (zn >= 2 && pread(fd, magic, 2, 0) != 2) {
    close(fd);
    return gnx_io_fail(GNX_EINVAL, std::string("vcf: cannot read ") + path);
  }
  if (zn < 2 || magic[0] != 0x1f || magic[1] != 0x8b) {
    t.fd = fd;
    t.n = zn;
    return GNX_OK;
  }
  void* m = mmap(nullptr, zn, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return gnx_io_fail(GNX_ENOMEM, std::string("vcf: mmap failed for ") + path);
  const uint8_t* z = (const uint8_t*)m;
  std::vector<BgzfBlock> blocks;
  size_t total = 0;
  if (bgzf_table(z, zn, blocks, total)) {
    (void)n_threads;  // the blocks are inflated by the parsing threads, window by window (Source::fetch)
    t.zmap = z;
    t.zmap_n = zn;
    t.blocks = std::move(blocks);
    t.use_zlib = getenv("GNX_VCF_ZLIB") != nullptr;
    t.n = total;
    t.compression = 2;
    return GNX_OK;
  }
  char* buf = nullptr;
  size_t bn = 0;
  const int rc = inflate_serial(z, zn, &buf, &bn);
  munmap(m, zn);
  if (rc != GNX_OK) return rc;
  t.owned = buf;
  t.mem = buf;
  t.n = bn;
  t.compression = 1;
  return GNX_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// records
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kVarCols = 6;  // CHROM ID REF ALT0 ALT1 ALT2

struct ChunkOut {
  std::vector<int64_t> pos;
  std::vector<float> qual;
  std::vector<uint8_t> match;  // CHROM == region
  std::vector<uint8_t> gt;     // rows x ldg
  std::string blob[kVarCols];
  std::vector<uint32_t> len[kVarCols];
  std::vector<gnx_vcf_ovf> ovf;  // row = chunk-local
  int64_t n_match = 0, row0 = 0, fast = 0, general = 0;
  std::string err;
};

inline bool is_record(const char* s, const char* e) {
  if (e > s && e[-1] == '\r') --e;
  return e > s && *s != '#';
}

// --- fixed-width genotype area: "a|b\t" per sample, a, b in {'0','1','.'}, separator '|' or '/' ---------------------------
__attribute__((target("avx2"))) inline void gt_step_avx2(const char* q, uint8_t* o, __m256i& bad) {
  const __m256i c30 = _mm256_set1_epi8(0x30), cfe = _mm256_set1_epi8((char)0xFE), c1e = _mm256_set1_epi8(0x1E);
  const __m256i cbar = _mm256_set1_epi8('|'), cslash = _mm256_set1_epi8('/'), ctab = _mm256_set1_epi8('\t');
  const __m256i m_even = _mm256_set1_epi16(0x00FF);           // allele bytes (offsets 0, 2 of each sample)
  const __m256i m_sep = _mm256_set1_epi32(0x0000FF00);        // offset 1
  const __m256i m_tab = _mm256_set1_epi32((int)0xFF000000u);  // offset 3
  const __m256i v = _mm256_loadu_si256((const __m256i*)q);
  const __m256i y = _mm256_xor_si256(v, c30);
  const __m256i ok_allele = _mm256_or_si256(_mm256_cmpeq_epi8(_mm256_and_si256(y, cfe), _mm256_setzero_si256()), _mm256_cmpeq_epi8(y, c1e));
  const __m256i ok_sep = _mm256_or_si256(_mm256_cmpeq_epi8(v, cbar), _mm256_cmpeq_epi8(v, cslash));
  const __m256i ok_tab = _mm256_cmpeq_epi8(v, ctab);
  const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_and_si256(ok_allele, m_even), _mm256_and_si256(ok_sep, m_sep)),
                                     _mm256_and_si256(ok_tab, m_tab));
  bad = _mm256_or_si256(bad, _mm256_xor_si256(ok, _mm256_set1_epi8((char)0xFF)));
  const uint32_t p0 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(y, 7));  // bit 0 of every byte
  const uint32_t p1 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(y, 6));  // bit 1 of every byte
  const uint32_t r = (p0 & 0x55555555u) | ((p1 & 0x55555555u) << 1);            // allele bytes sit at even positions
  memcpy(o, &r, 4);
}

__attribute__((target("avx2"))) bool gt_fast_avx2(const char* p, int64_t ns, uint8_t* out) {
  const int64_t len = 4 * ns - 1;
  const int64_t full = len / 32;
  __m256i bad = _mm256_setzero_si256();
  for (int64_t g = 0; g < full; ++g) gt_step_avx2(p + 32 * g, out + 4 * g, bad);
  const int64_t rem = len - 32 * full;
  if (rem > 0) {
    alignas(32) char tail[32];
    for (int i = 0; i < 32; i += 4) memcpy(tail + i, "0|0\t", 4);
    memcpy(tail, p + 32 * full, (size_t)rem);
    tail[rem] = '\t';
    uint8_t o4[4];
    gt_step_avx2(tail, o4, bad);
    const int64_t nbytes = (2 * (ns - 8 * full) + 3) / 4;
    memcpy(out + 4 * full, o4, (size_t)nbytes);
  }
  return _mm256_testz_si256(bad, bad) != 0;
}

// the same test and the same bit gather on 64 bytes = 16 samples (AVX-512BW: byte compares and vpmovb2m deliver 64-bit masks)
__attribute__((target("avx512f,avx512bw"))) inline void gt_step_avx512(const char* q, uint8_t* o, uint64_t& bad) {
  const __m512i v = _mm512_loadu_si512((const void*)q);
  const __m512i y = _mm512_xor_si512(v, _mm512_set1_epi8(0x30));
  const uint64_t ok_allele = _mm512_cmpeq_epi8_mask(_mm512_and_si512(y, _mm512_set1_epi8((char)0xFE)), _mm512_setzero_si512()) |
                             _mm512_cmpeq_epi8_mask(y, _mm512_set1_epi8(0x1E));
  const uint64_t ok_sep = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('|')) | _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('/'));
  const uint64_t ok_tab = _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8('\t'));
  const uint64_t ok = (ok_allele & 0x5555555555555555ull) | (ok_sep & 0x2222222222222222ull) | (ok_tab & 0x8888888888888888ull);
  bad |= ~ok;
  const uint64_t p0 = _mm512_movepi8_mask(_mm512_slli_epi16(y, 7));  // bit 0 of every byte
  const uint64_t p1 = _mm512_movepi8_mask(_mm512_slli_epi16(y, 6));  // bit 1 of every byte
  const uint64_t r = (p0 & 0x5555555555555555ull) | ((p1 & 0x5555555555555555ull) << 1);
  memcpy(o, &r, 8);
}

__attribute__((target("avx512f,avx512bw"))) bool gt_fast_avx512(const char* p, int64_t ns, uint8_t* out) {
  const int64_t len = 4 * ns - 1;
  const int64_t full = len / 64;
  uint64_t bad = 0;
  for (int64_t g = 0; g < full; ++g) gt_step_avx512(p + 64 * g, out + 8 * g, bad);
  const int64_t rem = len - 64 * full;
  if (rem > 0) {
    alignas(64) char tail[64];
    for (int i = 0; i < 64; i += 4) memcpy(tail + i, "0|0\t", 4);
    memcpy(tail, p + 64 * full, (size_t)rem);
    tail[rem] = '\t';
    uint8_t o8[8];
    gt_step_avx512(tail, o8, bad);
    const int64_t nbytes = (2 * (ns - 16 * full) + 3) / 4;
    memcpy(out + 8 * full, o8, (size_t)nbytes);
  }
  return bad == 0;
}

bool gt_fast_scalar(const char* p, int64_t ns, uint8_t* out) {
  uint32_t acc = 0;
  int nf = 0;
  int64_t ob = 0;
  for (int64_t s = 0; s < ns; ++s) {
    const char* q = p + 4 * s;
    const unsigned a = (unsigned char)q[0] ^ 0x30u, b = (unsigned char)q[2] ^ 0x30u;
    if (!((a <= 1 || a == 0x1E) && (b <= 1 || b == 0x1E) && (q[1] == '|' || q[1] == '/'))) return false;
    if (s + 1 < ns && q[3] != '\t') return false;
    acc |= ((a & 3u) | ((b & 3u) << 2)) << (4 * nf);
    if (++nf == 2) {
      out[ob++] = (uint8_t)acc;
      acc = 0;
      nf = 0;
    }
  }
  if (nf) out[ob++] = (uint8_t)acc;
  return true;
}

struct ParseCfg {
  int64_t ns, ldg;
  int simd;  // fixed-width genotype path: 0 scalar, 1 AVX2 (8 samples per step), 2 AVX-512BW (16 samples per step)
  const std::string* region;
};

inline const char* find_tab(const char* s, const char* e) { return (const char*)memchr(s, '\t', (size_t)(e - s)); }

// one record, appended to the chunk's columns.  Returns nullptr or the error.
const char* parse_record(const char* s, const char* e, const ParseCfg& cfg, ChunkOut& co) {
  if (e > s && e[-1] == '\r') --e;
  const char* f[10];
  f[0] = s;
  for (int i = 1; i <= 9; ++i) {
    const char* t = find_tab(f[i - 1], e);
    if (!t) return "record with fewer than 10 columns";
    f[i] = t + 1;
  }
  auto fld = [&](int i, const char*& b, const char*& en) {
    b = f[i];
    en = f[i + 1] - 1;
  };
  const char *b, *en;
  const int64_t r = (int64_t)co.pos.size();
  // CHROM
  fld(0, b, en);
  co.blob[0].append(b, en);
  co.len[0].push_back((uint32_t)(en - b));
  const std::string& reg = *cfg.region;
  const bool m = !reg.empty() && (size_t)(en - b) == reg.size() && memcmp(b, reg.data(), reg.size()) == 0;
  co.match.push_back(m ? 1 : 0);
  co.n_match += m;
  // POS
  fld(1, b, en);
  {
    int64_t p = 0;
    auto rr = std::from_chars(b, en, p);
    if (rr.ec != std::errc() || rr.ptr != en) return "POS is not an integer";
    co.pos.push_back(p);
  }
  // ID, REF
  fld(2, b, en);
  co.blob[1].append(b, en);
  co.len[1].push_back((uint32_t)(en - b));
  fld(3, b, en);
  co.blob[2].append(b, en);
  co.len[2].push_back((uint32_t)(en - b));
  // ALT: first three alternates, the rest dropped (scikit-allel's default alt_number = 3)
  fld(4, b, en);
  {
    const char* a = b;
    for (int k = 0; k < 3; ++k) {
      if (a > en) {
        co.len[3 + k].push_back(0);
        continue;
      }
      const char* c = (const char*)memchr(a, ',', (size_t)(en - a));
      const char* ae = c ? c : en;
      co.blob[3 + k].append(a, ae);
      co.len[3 + k].push_back((uint32_t)(ae - a));
      a = ae + 1;
    }
  }
  // QUAL
  fld(5, b, en);
  {
    float q = NAN;
    if (!(en - b == 1 && *b == '.') && en > b) {
      auto rr = std::from_chars(b, en, q);
      if (rr.ec != std::errc()) q = NAN;
    }
    co.qual.push_back(q);
  }
  // genotypes
  fld(8, b, en);
  const size_t need = (size_t)(r + 1) * (size_t)cfg.ldg;
  if (co.gt.size() < need) co.gt.resize(std::max(need, co.gt.size() * 2));
  uint8_t* row = co.gt.data() + (size_t)r * cfg.ldg;
  const char* g = f[9];
  const int64_t ns = cfg.ns;
  if (en - b == 2 && b[0] == 'G' && b[1] == 'T' && e - g == 4 * ns - 1) {
    const bool ok = cfg.simd == 2 ? gt_fast_avx512(g, ns, row) : cfg.simd == 1 ? gt_fast_avx2(g, ns, row) : gt_fast_scalar(g, ns, row);
    if (ok) {
      const int64_t used = (2 * ns + 3) / 4;
      if (cfg.ldg > used) memset(row + used, 0, (size_t)(cfg.ldg - used));
      ++co.fast;
      return nullptr;
    }
  }
  ++co.general;
  memset(row, 0, (size_t)cfg.ldg);
  // index of the GT key in FORMAT (-1: absent -> every call missing)
  int gi = -1;
  {
    int k = 0;
    const char* a = b;
    while (a <= en) {
      const char* c = (const char*)memchr(a, ':', (size_t)(en - a));
      const char* ae = c ? c : en;
      if (ae - a == 2 && a[0] == 'G' && a[1] == 'T') {
        gi = k;
        break;
      }
      if (!c) break;
      a = c + 1;
      ++k;
    }
  }
  const char* q = g;
  for (int64_t sidx = 0; sidx < ns; ++sidx) {
    if (q > e) return "record with fewer sample columns than the header";
    const char* t = find_tab(q, e);
    const char* fe = t ? t : e;
    int al[2] = {-1, -1};
    if (gi >= 0) {
      const char* a = q;
      bool have = true;
      for (int k = 0; k < gi; ++k) {
        const char* c = (const char*)memchr(a, ':', (size_t)(fe - a));
        if (!c) {
          have = false;
          break;
        }
        a = c + 1;
      }
      if (have) {
        for (int h = 0; h < 2; ++h) {
          if (a < fe && *a >= '0' && *a <= '9') {
            int v = 0;
            while (a < fe && *a >= '0' && *a <= '9') {
              v = std::min(v * 10 + (*a - '0'), 127);
              ++a;
            }
            al[h] = v;
          } else if (a < fe && *a == '.') {
            ++a;
          } else {
            break;
          }
          if (h == 0) {
            if (a < fe && (*a == '|' || *a == '/')) ++a;
            else break;
          }
        }
      }
    }
    for (int h = 0; h < 2; ++h) {
      const int64_t hap = 2 * sidx + h;
      const int a = al[h];
      const unsigned code = a < 0 ? 2u : a <= 1 ? (unsigned)a : 3u;
      row[hap >> 2] |= (uint8_t)(code << (2 * (hap & 3)));
      if (a >= 2) co.ovf.push_back({r, (int32_t)hap, a});
    }
    q = fe + 1;
  }
  if (q <= e) return "record with more sample columns than the header";
  return nullptr;
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
// the reader
// ------------------------------------------------------------------------------------------------------------------------
int gnx_io_vcf_read(const char* path, const char* region, int n_threads, gnx_io_alloc_fn alloc, gnx_io_free_fn release, void* user,
                    int pinned, gnx_vcf** out) {
  if (!path || !out || !alloc || !release) return gnx_io_fail(GNX_EINVAL, "vcf_read: bad arguments");
  *out = nullptr;
  const int nt = gnx_io_stream_threads(n_threads);
  const double t0 = gnx_io_now();
  Source src;
  int rc = open_source(path, gnx_io_threads(n_threads), src);  // inflating BGZF blocks is compute: every core
  if (rc != GNX_OK) return rc;
  std::unique_ptr<gnx_vcf> V(new gnx_vcf());
  V->info.n_threads = nt;
  V->info.compression = src.compression;
  V->info.file_bytes = src.file_bytes;
  V->info.text_bytes = (int64_t)src.n;
  V->info.gt2_pinned = pinned;
  // ---- header: '##' lines, then the column line; the record area starts behind it ------------------------------------------
  size_t d0 = 0;
  bool have_cols = false;
  {
    std::vector<char> hb;
    std::string err;
    size_t want = std::min<size_t>(src.n, (size_t)1 << 20);
    for (;;) {
      const char* p = want ? src.fetch(0, want, hb, &err) : "";
      if (!p) return gnx_io_fail(GNX_EINVAL, std::string("vcf: ") + err + " in " + path);
      const char* end = p + want;
      const char* q = p;
      bool complete = true;
      gnx_strcol meta, samples;
      have_cols = false;
      while (q < end && *q == '#') {
        const char* nl = (const char*)memchr(q, '\n', (size_t)(end - q));
        if (!nl && want < src.n) {
          complete = false;  // header line cut by the window
          break;
        }
        const char* le = nl ? nl : end;
        if (le - q >= 2 && q[1] == '#') {
          meta.blob.append(q, nl ? nl + 1 : le);
        } else {
          const char* ce = le;
          if (ce > q && ce[-1] == '\r') --ce;
          const char* c = q;
          int k = 0;
          samples = gnx_strcol();
          while (c <= ce) {
            const char* t = find_tab(c, ce);
            const char* fe = t ? t : ce;
            if (k >= 9) {
              samples.blob.append(c, fe);
              samples.off.push_back((int64_t)samples.blob.size());
            }
            ++k;
            if (!t) break;
            c = t + 1;
          }
          have_cols = true;
        }
        q = nl ? nl + 1 : end;
      }
      if (complete && (q < end || want >= src.n)) {
        d0 = (size_t)(q - p);
        meta.off.push_back((int64_t)meta.blob.size());
        V->col[GNX_VCF_META] = std::move(meta);
        V->col[GNX_VCF_SAMPLES] = std::move(samples);
        break;
      }
      want = std::min(src.n, want * 4);  // the header is longer than the window
    }
  }
  const int64_t ns = (int64_t)V->col[GNX_VCF_SAMPLES].off.size() - 1;
  V->info.n_samples = ns;
  if (!have_cols) return gnx_io_fail(GNX_EINVAL, std::string("vcf: no #CHROM header line in ") + path);
  if (ns <= 0) return gnx_io_fail(GNX_EINVAL, std::string("vcf: no sample columns in ") + path);
  const int64_t ldg = ((2 * ns + 15) / 16) * 4;
  V->info.ldg = ldg;
  const double t1 = gnx_io_now();
  // ---- chunks --------------------------------------------------------------------------------------------------------------------
  const size_t dn = src.n - d0;
  size_t cs = dn / ((size_t)nt * 6) + 1;
  cs = std::min<size_t>(std::max<size_t>(cs, (size_t)1 << 18), (size_t)8 << 20);
  if (const char* e = getenv("GNX_IO_CHUNK")) cs = std::max<size_t>(64, (size_t)atoll(e));  // tests: many chunks on small files
  const int64_t n_chunks = (int64_t)((dn + cs - 1) / cs);
  std::vector<ChunkOut> chunks((size_t)n_chunks);
  const std::string reg = region ? region : "";
  // AVX2 where the CPU has it.  The 64-byte AVX-512BW step is built and tested too (GNX_IO_SIMD=avx512) but measured SLOWER on the
  // MI355X box's EPYC 9575F: read_vcf 0.142 s against 0.118 s for chr22 x 5 000 samples — the pass is bound by bringing the text in
  // (pread), not by the 32-byte steps.  GNX_IO_SIMD=scalar|avx2|avx512 selects (never wider than the CPU offers; tests run all three).
  const int widest = __builtin_cpu_supports("avx512bw") ? 2 : __builtin_cpu_supports("avx2") ? 1 : 0;
  int simd = std::min(widest, 1);
  if (const char* e = getenv("GNX_IO_SIMD")) simd = std::min(widest, !strcmp(e, "scalar") ? 0 : !strcmp(e, "avx2") ? 1 : 2);
  if (getenv("GNX_IO_NO_AVX2")) simd = 0;
  ParseCfg cfg{ns, ldg, simd, &reg};
  {
    std::atomic<int64_t> next{0};
    const size_t est_line = (size_t)(4 * ns + 48);
    gnx_io_parallel((int)std::max<int64_t>(1, std::min<int64_t>(n_chunks, nt)), [&](int) {
      std::vector<char> buf;
      for (;;) {
        const int64_t k = next.fetch_add(1, std::memory_order_relaxed);
        if (k >= n_chunks) break;
        ChunkOut& co = chunks[(size_t)k];
        const size_t a = d0 + (size_t)k * cs, b = std::min(src.n, a + cs);
        const size_t lo = k > 0 ? a - 1 : a;  // one byte back: is `a` the start of a line?
        size_t hi = std::min(src.n, b + est_line + 4096);
        const char* w = src.fetch(lo, hi - lo, buf, &co.err);
        if (!w) break;
        // first record that starts at or behind a
        size_t s = a;
        if (k > 0) {
          const char* nl = (const char*)memchr(w, '\n', b - lo);
          if (!nl) continue;  // no line starts in this chunk
          s = lo + (size_t)(nl - w) + 1;
        }
        const size_t rows_est = cs / est_line + 8;
        co.pos.reserve(rows_est);
        co.qual.reserve(rows_est);
        co.match.reserve(rows_est);
        co.gt.resize(rows_est * (size_t)ldg);
        while (s < b && co.err.empty()) {
          const char* nl = (const char*)memchr(w + (s - lo), '\n', hi - s);
          while (!nl && hi < src.n) {  // the record runs past the window: fetch more
            hi = std::min(src.n, hi + std::max<size_t>(est_line, (hi - lo)));
            w = src.fetch(lo, hi - lo, buf, &co.err);
            if (!w) break;
            nl = (const char*)memchr(w + (s - lo), '\n', hi - s);
          }
          if (!w) break;
          const size_t e = nl ? lo + (size_t)(nl - w) : src.n;
          const char* ls = w + (s - lo);
          const char* le = w + (e - lo);
          if (is_record(ls, le)) {
            const char* err = parse_record(ls, le, cfg, co);
            if (err) co.err = std::string(err) + " (record at byte " + std::to_string((int64_t)s) + ")";
          }
          s = e + 1;
        }
      }
    });
  }
  for (auto& c : chunks)
    if (!c.err.empty()) return gnx_io_fail(GNX_EINVAL, std::string("vcf: ") + c.err + " in " + path);
  const double t2 = gnx_io_now();
  // ---- placement ----------------------------------------------------------------------------------------------------------------
  int64_t total = 0, matched = 0;
  for (auto& c : chunks) {
    total += (int64_t)c.pos.size();
    matched += c.n_match;
  }
  const bool filter = !reg.empty() && matched > 0;
  V->info.region_fallback = (!reg.empty() && matched == 0 && total > 0) ? 1 : 0;
  const int64_t nv = filter ? matched : total;
  {
    int64_t r = 0;
    for (auto& c : chunks) {
      c.row0 = r;
      r += filter ? c.n_match : (int64_t)c.pos.size();
    }
  }
  V->info.n_variants = nv;
  V->pos.resize((size_t)nv);
  V->qual.resize((size_t)nv);
  const double t3 = gnx_io_now();
  V->gt2 = (uint8_t*)alloc(user, std::max<size_t>((size_t)nv * (size_t)ldg, 64));
  if (!V->gt2) return gnx_io_fail(GNX_ENOMEM, "vcf: cannot allocate the genotype matrix");
  V->release = release;
  V->user = user;
  const double t4 = gnx_io_now();
  {
    std::atomic<int64_t> next{0};
    gnx_vcf* Vp = V.get();
    gnx_io_parallel((int)std::max<int64_t>(1, std::min<int64_t>(n_chunks, nt)), [&](int) {
      for (;;) {
        const int64_t k = next.fetch_add(1, std::memory_order_relaxed);
        if (k >= n_chunks) break;
        ChunkOut& co = chunks[(size_t)k];
        const int64_t n = (int64_t)co.pos.size();
        if (!filter) {
          if (n) {
            memcpy(Vp->pos.data() + co.row0, co.pos.data(), (size_t)n * 8);
            memcpy(Vp->qual.data() + co.row0, co.qual.data(), (size_t)n * 4);
            memcpy(Vp->gt2 + (size_t)co.row0 * ldg, co.gt.data(), (size_t)n * ldg);
          }
        } else {
          int64_t r = co.row0;
          for (int64_t i = 0; i < n; ++i)
            if (co.match[(size_t)i]) {
              Vp->pos[(size_t)r] = co.pos[(size_t)i];
              Vp->qual[(size_t)r] = co.qual[(size_t)i];
              memcpy(Vp->gt2 + (size_t)r * ldg, co.gt.data() + (size_t)i * ldg, (size_t)ldg);
              ++r;
            }
        }
        std::vector<uint8_t>().swap(co.gt);
      }
    });
  }
  // the small columns: strings and the side list of large alleles
  for (int f = 0; f < kVarCols; ++f) {
    gnx_strcol& col = V->col[f];
    size_t bytes = 0;
    for (auto& c : chunks) bytes += c.blob[f].size();
    col.blob.reserve(bytes);
    col.off.reserve((size_t)nv + 1);
    for (auto& c : chunks) {
      if (!filter) {
        col.blob.append(c.blob[f]);
        int64_t o = col.off.back();
        for (uint32_t l : c.len[f]) {
          o += l;
          col.off.push_back(o);
        }
      } else {
        size_t o = 0;
        for (size_t i = 0; i < c.len[f].size(); ++i) {
          const uint32_t l = c.len[f][i];
          if (c.match[i]) {
            col.blob.append(c.blob[f], o, l);
            col.off.push_back((int64_t)col.blob.size());
          }
          o += l;
        }
      }
    }
  }
  for (auto& c : chunks) {
    if (!filter) {
      for (auto x : c.ovf) {
        x.row += c.row0;
        V->ovf.push_back(x);
      }
    } else if (!c.ovf.empty()) {
      std::vector<int64_t> dest(c.match.size());
      int64_t r = c.row0;
      for (size_t i = 0; i < c.match.size(); ++i) dest[i] = c.match[i] ? r++ : -1;
      for (auto x : c.ovf)
        if (dest[(size_t)x.row] >= 0) {
          x.row = dest[(size_t)x.row];
          V->ovf.push_back(x);
        }
    }
    V->info.n_fast_lines += c.fast;
    V->info.n_general_lines += c.general;
  }
  V->info.n_overflow = (int64_t)V->ovf.size();
  const double t5 = gnx_io_now();
  V->info.seconds_load = t1 - t0;
  V->info.seconds_parse = t2 - t1;
  V->info.seconds_alloc = t4 - t3;
  V->info.seconds_merge = (t3 - t2) + (t5 - t4);
  *out = V.release();
  return GNX_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// accessors
// ------------------------------------------------------------------------------------------------------------------------
extern "C" void gnx_vcf_free(gnx_vcf* v) {
  if (!v) return;
  if (v->gt2 && v->release) v->release(v->user, v->gt2);
  delete v;
}
extern "C" int gnx_vcf_get_info(const gnx_vcf* v, gnx_vcf_info* out) {
  if (!v || !out) return GNX_EINVAL;
  *out = v->info;
  return GNX_OK;
}
extern "C" const uint8_t* gnx_vcf_gt2(const gnx_vcf* v) { return v ? v->gt2 : nullptr; }
extern "C" const int64_t* gnx_vcf_pos(const gnx_vcf* v) { return v ? v->pos.data() : nullptr; }
extern "C" const float* gnx_vcf_qual(const gnx_vcf* v) { return v ? v->qual.data() : nullptr; }
extern "C" int gnx_vcf_strings(const gnx_vcf* v, int field, const char** blob, const int64_t** offsets, int64_t* n) {
  if (!v || field < 0 || field > 7 || !blob || !offsets || !n) return gnx_io_fail(GNX_EINVAL, "vcf_strings: bad arguments");
  const gnx_strcol& c = v->col[field];
  *blob = c.blob.data();
  *offsets = c.off.data();
  *n = (int64_t)c.off.size() - 1;
  return GNX_OK;
}

extern "C" int gnx_vcf_gt_int8(const gnx_vcf* v, int8_t* out, int n_threads) {
  if (!v || !out) return gnx_io_fail(GNX_EINVAL, "vcf_gt_int8: bad arguments");
  const int64_t nv = v->info.n_variants, nh = 2 * v->info.n_samples, ldg = v->info.ldg;
  static const int8_t kCode[4] = {0, 1, -1, 2};
  const int64_t nblk = (nv + 255) / 256;
  std::atomic<int64_t> next{0};
  gnx_io_parallel((int)std::max<int64_t>(1, std::min<int64_t>(nblk, gnx_io_stream_threads(n_threads))), [&](int) {
    for (;;) {
      const int64_t blk = next.fetch_add(1, std::memory_order_relaxed);
      if (blk >= nblk) break;
      for (int64_t r = blk * 256; r < std::min(nv, blk * 256 + 256); ++r) {
        const uint8_t* row = v->gt2 + (size_t)r * ldg;
        int8_t* o = out + (size_t)r * nh;
        for (int64_t h = 0; h < nh; ++h) o[h] = kCode[(row[h >> 2] >> (2 * (h & 3))) & 3];
      }
    }
  });
  for (const gnx_vcf_ovf& x : v->ovf) out[(size_t)x.row * nh + x.hap] = (int8_t)x.allele;
  return GNX_OK;
}

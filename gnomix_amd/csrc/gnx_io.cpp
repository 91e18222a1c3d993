// gnx_io.cpp — the file side of the hot path, host code only (no HIP): phased VCF text -> variant-major 2-bit genotypes on
// every core of the box, and the .msp / .fb / phased-VCF writers.  Contract: include/gnomix_io.h.
//
// Reference: src/utils.py:55-81 (read_vcf = scikit-allel's C parser behind gzip.open), src/postprocess.py:84-126 (write_msp /
// write_fb: numpy / pandas string conversion), src/utils.py:247-329 (npy_to_vcf: one pandas column of "a|b" strings per sample).
//
// Reader.  The text is mapped (plain), or inflated into memory (BGZF: every 64 KB block on its own thread; plain gzip: one
// stream, serial).  Pass 1 cuts the record area into chunks at line starts and counts records (and records whose CHROM is the
// requested region) per chunk; a prefix sum gives every chunk its first output row, so pass 2 parses straight into the final
// arrays with no merge step for the genotype matrix.  A record whose FORMAT is exactly "GT" and whose sample area is
// 4*n_samples-1 bytes long is tried on the fixed-width path: 32 bytes of text ("a|b\t" x 8) are validated and squeezed to
// 16 two-bit fields with a handful of AVX2 operations (allele byte ^ '0' is 0, 1 or 0x1E for '.': its low two bits ARE the
// code); anything else (extra FORMAT keys, multi-digit alleles, haploid calls) takes the per-sample path.
//
// Writers.  Rows are formatted in parallel, each by one worker into its own buffer, and written with pwrite at offsets that
// are chained from row to row (row r's offset is published by whoever formatted row r-1 as soon as its length is known):
// no barrier, one row buffer per worker, rows land in order.
#include "gnx_io.h"

#include <fcntl.h>
#include <immintrin.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------------------------------------
// errors, clock, pool
// ------------------------------------------------------------------------------------------------------------------------
static thread_local std::string g_io_err;
void gnx_io_set_error(const std::string& msg) { g_io_err = msg; }
extern "C" const char* gnx_io_last_error(void) { return g_io_err.c_str(); }
static int io_fail(int code, const std::string& msg) {
  g_io_err = msg;
  return code;
}
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int gnx_io_threads(int requested) {
  if (requested > 0) return std::min(requested, 1024);
  if (const char* e = getenv("GNX_IO_THREADS")) {
    const int v = atoi(e);
    if (v > 0) return std::min(v, 1024);
  }
  cpu_set_t set;
  int n = 0;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)std::thread::hardware_concurrency();
  return std::max(1, std::min(n, 512));
}

namespace {
// Workers sleep on a condition variable between jobs; the pool is created on first use and never destroyed (its threads
// end with the process: a static destructor joining them at exit could deadlock inside a host interpreter's shutdown).
class Pool {
 public:
  static Pool& get() {
    static Pool* p = new Pool();
    return *p;
  }
  void run(int n, const std::function<void(int)>& fn) {
    if (n <= 1) {
      fn(0);
      return;
    }
    std::lock_guard<std::mutex> job_lock(job_mu_);
    {
      std::unique_lock<std::mutex> lk(mu_);
      while ((int)threads_.size() < n - 1) {
        const int tid = (int)threads_.size() + 1;
        threads_.emplace_back([this, tid] { loop(tid); });
        threads_.back().detach();
      }
      fn_ = &fn;
      n_active_ = n;
      pending_ = n - 1;
      ++gen_;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void loop(int tid) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int)>* fn = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (tid < n_active_) fn = fn_;
      }
      if (fn) {
        (*fn)(tid);
        std::unique_lock<std::mutex> lk(mu_);
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  std::mutex job_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> threads_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_active_ = 0, pending_ = 0;
  uint64_t gen_ = 0;
};
}  // namespace

void gnx_io_parallel(int n_workers, const std::function<void(int)>& fn) { Pool::get().run(n_workers, fn); }

// dynamic loop over [0, n) in grains: body(i, tid)
template <class F>
static void par_for(int64_t n, int n_threads, F&& body) {
  if (n <= 0) return;
  const int nt = (int)std::min<int64_t>(n, n_threads);
  std::atomic<int64_t> next{0};
  gnx_io_parallel(nt, [&](int tid) {
    for (;;) {
      const int64_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n) break;
      body(i, tid);
    }
  });
}

// ------------------------------------------------------------------------------------------------------------------------
// numbers -> text
// ------------------------------------------------------------------------------------------------------------------------
static inline char* put_int(char* o, int64_t v) {
  auto r = std::to_chars(o, o + 24, v);
  return r.ptr;
}

// numpy's str() of a float scalar (scalartypes.c.src: *_either): shortest round-trip digits (Dragon4, unique mode = what
// Ryu's shortest gives), positional when 1e-4 <= |x| < 1e16 (at least one digit after the point), else scientific with a
// two-digit exponent.  Writes at most 32 characters, returns the end.
template <typename T>
static inline char* put_float(char* o, T v) {
  if (v != v) {
    memcpy(o, "nan", 3);
    return o + 3;
  }
  if (std::isinf(v)) {
    if (v < 0) *o++ = '-';
    memcpy(o, "inf", 3);
    return o + 3;
  }
  if (std::signbit(v)) {
    *o++ = '-';
    v = -v;
  }
  if (v == 0) {
    memcpy(o, "0.0", 3);
    return o + 3;
  }
  char sc[40];
  auto r = std::to_chars(sc, sc + 40, v, std::chars_format::scientific);  // d[.ddd]e[+-]XX
  // digits
  char dig[24];
  int k = 0;
  const char* p = sc;
  while (p < r.ptr && *p != 'e') {
    if (*p != '.') dig[k++] = *p;
    ++p;
  }
  ++p;  // 'e'
  const bool eneg = (*p == '-');
  ++p;
  int e = 0;
  while (p < r.ptr) e = e * 10 + (*p++ - '0');
  if (eneg) e = -e;
  const double av = (double)v;
  if (av >= 1e-4 && av < 1e16) {
    if (e >= 0) {
      const int ni = e + 1;  // digits before the point
      if (k <= ni) {
        memcpy(o, dig, k);
        o += k;
        for (int i = k; i < ni; ++i) *o++ = '0';
        *o++ = '.';
        *o++ = '0';
      } else {
        memcpy(o, dig, ni);
        o += ni;
        *o++ = '.';
        memcpy(o, dig + ni, k - ni);
        o += k - ni;
      }
    } else {
      *o++ = '0';
      *o++ = '.';
      for (int i = 0; i < -e - 1; ++i) *o++ = '0';
      memcpy(o, dig, k);
      o += k;
    }
    return o;
  }
  *o++ = dig[0];
  if (k > 1) {
    *o++ = '.';
    memcpy(o, dig + 1, k - 1);
    o += k - 1;
  }
  *o++ = 'e';
  *o++ = e < 0 ? '-' : '+';
  int ae = e < 0 ? -e : e;
  if (ae >= 100) {
    *o++ = (char)('0' + ae / 100);
    ae %= 100;
  }
  *o++ = (char)('0' + ae / 10);
  *o++ = (char)('0' + ae % 10);
  return o;
}

extern "C" int gnx_format_floats(const void* values, int is_f64, int64_t n, char* out, int64_t* off) {
  if (n < 0 || (n > 0 && (!values || !out || !off))) return io_fail(GNX_EINVAL, "format_floats: bad arguments");
  char* o = out;
  if (off) off[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    o = is_f64 ? put_float<double>(o, ((const double*)values)[i]) : put_float<float>(o, ((const float*)values)[i]);
    off[i + 1] = o - out;
  }
  return GNX_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// text source: mmap, gzip, BGZF
// ------------------------------------------------------------------------------------------------------------------------
namespace {
struct Text {
  const char* p = nullptr;
  size_t n = 0;
  void* map = nullptr;
  size_t map_len = 0;
  char* owned = nullptr;
  int compression = 0;
  int64_t file_bytes = 0;
  ~Text() { release(); }
  void release() {
    if (map) munmap(map, map_len);
    map = nullptr;
    free(owned);
    owned = nullptr;
    p = nullptr;
    n = 0;
  }
};

inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd32(const uint8_t* p) { return rd16(p) | (rd16(p + 2) << 16); }

// BGZF block table: (compressed offset, compressed size, payload offset, payload size, uncompressed offset, uncompressed size)
struct BgzfBlock {
  size_t coff, csize, poff, psize, uoff, usize;
};

bool bgzf_table(const uint8_t* z, size_t zn, std::vector<BgzfBlock>& blocks, size_t& total) {
  size_t o = 0;
  total = 0;
  while (o < zn) {
    if (zn - o < 18) return false;
    const uint8_t* h = z + o;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
    const uint32_t xlen = rd16(h + 10);
    if (zn - o < 12 + (size_t)xlen + 8) return false;
    int64_t bsize = -1;
    size_t x = 12;
    const size_t xend = 12 + (size_t)xlen;
    while (x + 4 <= xend) {
      const uint32_t slen = rd16(h + x + 2);
      if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2 && x + 6 <= xend) bsize = (int64_t)rd16(h + x + 4) + 1;
      x += 4 + slen;
    }
    if (bsize < (int64_t)xend + 8 || (size_t)bsize > zn - o) return false;
    if (h[3] & ~4) return false;  // name / comment / header crc: not what bgzip writes
    BgzfBlock b;
    b.coff = o;
    b.csize = (size_t)bsize;
    b.poff = o + xend;
    b.psize = (size_t)bsize - xend - 8;
    b.usize = rd32(h + bsize - 4);
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
  if (!buf) return io_fail(GNX_ENOMEM, "vcf: out of memory inflating");
  z_stream s;
  memset(&s, 0, sizeof(s));
  if (inflateInit2(&s, 15 + 32) != Z_OK) {
    free(buf);
    return io_fail(GNX_EINVAL, "vcf: inflateInit2 failed");
  }
  size_t in_pos = 0, produced = 0;
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
        inflateEnd(&s);
        free(buf);
        return io_fail(GNX_ENOMEM, "vcf: out of memory inflating");
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
        inflateEnd(&s);
        free(buf);
        return io_fail(GNX_EINVAL, "vcf: inflateReset failed");
      }
      continue;
    }
    if (rc != Z_OK && rc != Z_BUF_ERROR) {
      inflateEnd(&s);
      free(buf);
      return io_fail(GNX_EINVAL, std::string("vcf: corrupt gzip stream (") + (s.msg ? s.msg : "inflate error") + ")");
    }
    if (rc == Z_BUF_ERROR && s.avail_in == 0 && in_pos >= zn) {
      inflateEnd(&s);
      free(buf);
      return io_fail(GNX_EINVAL, "vcf: truncated gzip stream");
    }
  }
  inflateEnd(&s);
  *out = buf;
  *out_n = produced;
  return GNX_OK;
}

int load_text(const char* path, int n_threads, Text& t) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return io_fail(GNX_EINVAL, std::string("vcf: cannot open ") + path + ": " + strerror(errno));
  struct stat st;
  if (fstat(fd, &st) != 0) {
    close(fd);
    return io_fail(GNX_EINVAL, std::string("vcf: cannot stat ") + path);
  }
  t.file_bytes = (int64_t)st.st_size;
  if (st.st_size == 0) {
    close(fd);
    return GNX_OK;
  }
  void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return io_fail(GNX_ENOMEM, std::string("vcf: mmap failed for ") + path);
  madvise(m, (size_t)st.st_size, MADV_WILLNEED);
  const uint8_t* z = (const uint8_t*)m;
  const size_t zn = (size_t)st.st_size;
  if (zn < 2 || z[0] != 0x1f || z[1] != 0x8b) {
    t.map = m;
    t.map_len = zn;
    t.p = (const char*)m;
    t.n = zn;
    t.compression = 0;
    return GNX_OK;
  }
  std::vector<BgzfBlock> blocks;
  size_t total = 0;
  if (bgzf_table(z, zn, blocks, total)) {
    char* buf = (char*)malloc(std::max<size_t>(total, 1));
    if (!buf) {
      munmap(m, zn);
      return io_fail(GNX_ENOMEM, "vcf: out of memory inflating");
    }
    std::atomic<int> bad{0};
    std::atomic<int64_t> next{0};
    const int64_t nb = (int64_t)blocks.size();
    gnx_io_parallel((int)std::min<int64_t>(nb, n_threads), [&](int) {
      z_stream s;
      memset(&s, 0, sizeof(s));
      if (inflateInit2(&s, -15) != Z_OK) {
        bad = 1;
        return;
      }
      for (;;) {
        const int64_t i0 = next.fetch_add(16, std::memory_order_relaxed);
        if (i0 >= nb) break;
        for (int64_t i = i0; i < std::min(nb, i0 + 16); ++i) {
          const BgzfBlock& b = blocks[(size_t)i];
          if (b.usize == 0) continue;
          inflateReset(&s);
          s.next_in = const_cast<Bytef*>(z + b.poff);
          s.avail_in = (uInt)b.psize;
          s.next_out = (Bytef*)buf + b.uoff;
          s.avail_out = (uInt)b.usize;
          const int rc = inflate(&s, Z_FINISH);
          if (rc != Z_STREAM_END || s.avail_out != 0) bad = 1;
        }
      }
      inflateEnd(&s);
    });
    munmap(m, zn);
    if (bad) {
      free(buf);
      return io_fail(GNX_EINVAL, "vcf: corrupt BGZF block");
    }
    t.owned = buf;
    t.p = buf;
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
  t.p = buf;
  t.n = bn;
  t.compression = 1;
  return GNX_OK;
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
// the parsed file
// ------------------------------------------------------------------------------------------------------------------------
struct StrCol {
  std::string blob;
  std::vector<int64_t> off{0};
};
struct Ovf {
  int64_t row;
  int32_t hap, allele;
};
struct gnx_vcf {
  gnx_vcf_info info{};
  std::vector<int64_t> pos;
  std::vector<float> qual;
  StrCol col[8];
  uint8_t* gt2 = nullptr;
  gnx_io_free_fn release = nullptr;
  void* user = nullptr;
  std::vector<Ovf> ovf;
};

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
  if (!v || field < 0 || field > 7 || !blob || !offsets || !n) return io_fail(GNX_EINVAL, "vcf_strings: bad arguments");
  const StrCol& c = v->col[field];
  *blob = c.blob.data();
  *offsets = c.off.data();
  *n = (int64_t)c.off.size() - 1;
  return GNX_OK;
}

namespace {
constexpr int kVarCols = 6;  // CHROM ID REF ALT0 ALT1 ALT2

struct ChunkOut {
  std::string blob[kVarCols];
  std::vector<uint32_t> len[kVarCols];
  std::vector<Ovf> ovf;
  int64_t n_lines = 0, n_match = 0, row0 = 0, fast = 0, general = 0;
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
  bool avx2;
};

inline const char* find_tab(const char* s, const char* e) { return (const char*)memchr(s, '\t', (size_t)(e - s)); }

// one record -> row r.  Returns an empty string or the error.
const char* parse_record(const char* s, const char* e, const ParseCfg& cfg, int64_t r, gnx_vcf* V, ChunkOut& co) {
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
  // CHROM
  fld(0, b, en);
  co.blob[0].append(b, en);
  co.len[0].push_back((uint32_t)(en - b));
  // POS
  fld(1, b, en);
  {
    int64_t p = 0;
    auto rr = std::from_chars(b, en, p);
    if (rr.ec != std::errc() || rr.ptr != en) return "POS is not an integer";
    V->pos[(size_t)r] = p;
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
    V->qual[(size_t)r] = q;
  }
  // genotypes
  fld(8, b, en);
  uint8_t* row = V->gt2 + (size_t)r * cfg.ldg;
  const char* g = f[9];
  const int64_t ns = cfg.ns;
  if (en - b == 2 && b[0] == 'G' && b[1] == 'T' && e - g == 4 * ns - 1) {
    const bool ok = cfg.avx2 ? gt_fast_avx2(g, ns, row) : gt_fast_scalar(g, ns, row);
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

int gnx_io_vcf_read(const char* path, const char* region, int n_threads, gnx_io_alloc_fn alloc, gnx_io_free_fn release, void* user,
                    int pinned, gnx_vcf** out) {
  if (!path || !out || !alloc || !release) return io_fail(GNX_EINVAL, "vcf_read: bad arguments");
  *out = nullptr;
  const int nt = gnx_io_threads(n_threads);
  const double t0 = now_s();
  Text text;
  int rc = load_text(path, nt, text);
  if (rc != GNX_OK) return rc;
  const double t1 = now_s();
  std::unique_ptr<gnx_vcf> V(new gnx_vcf());
  V->info.n_threads = nt;
  V->info.compression = text.compression;
  V->info.file_bytes = text.file_bytes;
  V->info.text_bytes = (int64_t)text.n;
  V->info.gt2_pinned = pinned;
  // ---- header -------------------------------------------------------------------------------------------------------
  const char* p = text.p;
  const char* end = text.p + text.n;
  bool have_cols = false;
  while (p < end && *p == '#') {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* le = nl ? nl : end;
    if (le - p >= 2 && p[1] == '#') {
      V->col[GNX_VCF_META].blob.append(p, nl ? nl + 1 : le);
    } else {
      const char* ce = le;
      if (ce > p && ce[-1] == '\r') --ce;
      const char* q = p;
      int k = 0;
      while (q <= ce) {
        const char* t = find_tab(q, ce);
        const char* fe = t ? t : ce;
        if (k >= 9) {
          V->col[GNX_VCF_SAMPLES].blob.append(q, fe);
          V->col[GNX_VCF_SAMPLES].off.push_back((int64_t)V->col[GNX_VCF_SAMPLES].blob.size());
        }
        ++k;
        if (!t) break;
        q = t + 1;
      }
      have_cols = true;
    }
    p = nl ? nl + 1 : end;
  }
  V->col[GNX_VCF_META].off.push_back((int64_t)V->col[GNX_VCF_META].blob.size());
  const int64_t ns = (int64_t)V->col[GNX_VCF_SAMPLES].off.size() - 1;
  V->info.n_samples = ns;
  if (!have_cols) return io_fail(GNX_EINVAL, std::string("vcf: no #CHROM header line in ") + path);
  if (ns <= 0) return io_fail(GNX_EINVAL, std::string("vcf: no sample columns in ") + path);
  const int64_t ldg = ((2 * ns + 15) / 16) * 4;
  V->info.ldg = ldg;
  // ---- pass 1: chunks, record counts, region matches --------------------------------------------------------------------
  const char* d0 = p;
  const size_t dn = (size_t)(end - d0);
  const int64_t n_chunks = std::max<int64_t>(1, std::min<int64_t>((int64_t)(dn >> 16) + 1, (int64_t)nt * 8));
  std::vector<const char*> bound((size_t)n_chunks + 1);
  bound[0] = d0;
  bound[(size_t)n_chunks] = end;
  for (int64_t k = 1; k < n_chunks; ++k) {
    const char* q = d0 + dn / (size_t)n_chunks * (size_t)k;
    if (q <= d0) {
      bound[(size_t)k] = d0;
      continue;
    }
    const char* nl = (const char*)memchr(q - 1, '\n', (size_t)(end - (q - 1)));
    bound[(size_t)k] = nl ? nl + 1 : end;
  }
  for (int64_t k = 1; k <= n_chunks; ++k) bound[(size_t)k] = std::max(bound[(size_t)k], bound[(size_t)k - 1]);
  std::vector<ChunkOut> chunks((size_t)n_chunks);
  const std::string reg = region ? region : "";
  const bool use_region = !reg.empty();
  par_for(n_chunks, nt, [&](int64_t k, int) {
    ChunkOut& co = chunks[(size_t)k];
    const char* q = bound[(size_t)k];
    const char* qe = bound[(size_t)k + 1];
    while (q < qe) {
      const char* nl = (const char*)memchr(q, '\n', (size_t)(qe - q));
      const char* le = nl ? nl : qe;
      if (is_record(q, le)) {
        ++co.n_lines;
        if (use_region && (size_t)(le - q) > reg.size() && q[reg.size()] == '\t' && memcmp(q, reg.data(), reg.size()) == 0) ++co.n_match;
      }
      q = nl ? nl + 1 : qe;
    }
  });
  int64_t total = 0, matched = 0;
  for (auto& c : chunks) {
    total += c.n_lines;
    matched += c.n_match;
  }
  const bool filter = use_region && matched > 0;
  V->info.region_fallback = (use_region && matched == 0 && total > 0) ? 1 : 0;
  const int64_t nv = filter ? matched : total;
  {
    int64_t r = 0;
    for (auto& c : chunks) {
      c.row0 = r;
      r += filter ? c.n_match : c.n_lines;
    }
  }
  V->info.n_variants = nv;
  const double t2 = now_s();
  // ---- pass 2 ---------------------------------------------------------------------------------------------------------------
  V->pos.resize((size_t)nv);
  V->qual.resize((size_t)nv);
  V->gt2 = (uint8_t*)alloc(user, std::max<size_t>((size_t)nv * (size_t)ldg, 64));
  if (!V->gt2) return io_fail(GNX_ENOMEM, "vcf: cannot allocate the genotype matrix");
  V->release = release;
  V->user = user;
  ParseCfg cfg{ns, ldg, __builtin_cpu_supports("avx2") != 0 && !getenv("GNX_IO_NO_AVX2")};
  gnx_vcf* Vp = V.get();
  par_for(n_chunks, nt, [&](int64_t k, int) {
    ChunkOut& co = chunks[(size_t)k];
    const char* q = bound[(size_t)k];
    const char* qe = bound[(size_t)k + 1];
    int64_t r = co.row0;
    while (q < qe && co.err.empty()) {
      const char* nl = (const char*)memchr(q, '\n', (size_t)(qe - q));
      const char* le = nl ? nl : qe;
      if (is_record(q, le)) {
        const bool take = !filter || ((size_t)(le - q) > reg.size() && q[reg.size()] == '\t' && memcmp(q, reg.data(), reg.size()) == 0);
        if (take) {
          const char* err = parse_record(q, le, cfg, r, Vp, co);
          if (err) co.err = std::string(err) + " (record " + std::to_string(r + 1) + ", byte " + std::to_string((int64_t)(q - text.p)) + ")";
          ++r;
        }
      }
      q = nl ? nl + 1 : qe;
    }
  });
  for (auto& c : chunks)
    if (!c.err.empty()) return io_fail(GNX_EINVAL, std::string("vcf: ") + c.err + " in " + path);
  // ---- merge the small per-chunk columns -----------------------------------------------------------------------------------
  for (int f = 0; f < kVarCols; ++f) {
    StrCol& col = V->col[f];
    size_t bytes = 0;
    for (auto& c : chunks) bytes += c.blob[f].size();
    col.blob.reserve(bytes);
    col.off.reserve((size_t)nv + 1);
    for (auto& c : chunks) {
      col.blob.append(c.blob[f]);
      int64_t o = col.off.back();
      for (uint32_t l : c.len[f]) {
        o += l;
        col.off.push_back(o);
      }
    }
  }
  for (auto& c : chunks) {
    V->ovf.insert(V->ovf.end(), c.ovf.begin(), c.ovf.end());
    V->info.n_fast_lines += c.fast;
    V->info.n_general_lines += c.general;
  }
  V->info.n_overflow = (int64_t)V->ovf.size();
  const double t3 = now_s();
  V->info.seconds_load = t1 - t0;
  V->info.seconds_index = t2 - t1;
  V->info.seconds_parse = t3 - t2;
  *out = V.release();
  return GNX_OK;
}

extern "C" int gnx_vcf_gt_int8(const gnx_vcf* v, int8_t* out, int n_threads) {
  if (!v || !out) return io_fail(GNX_EINVAL, "vcf_gt_int8: bad arguments");
  const int64_t nv = v->info.n_variants, nh = 2 * v->info.n_samples, ldg = v->info.ldg;
  static const int8_t kCode[4] = {0, 1, -1, 2};
  par_for((nv + 255) / 256, gnx_io_threads(n_threads), [&](int64_t blk, int) {
    for (int64_t r = blk * 256; r < std::min(nv, blk * 256 + 256); ++r) {
      const uint8_t* row = v->gt2 + (size_t)r * ldg;
      int8_t* o = out + (size_t)r * nh;
      for (int64_t h = 0; h < nh; ++h) o[h] = kCode[(row[h >> 2] >> (2 * (h & 3))) & 3];
    }
  });
  for (const Ovf& x : v->ovf) out[(size_t)x.row * nh + x.hap] = (int8_t)x.allele;
  return GNX_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// writers
// ------------------------------------------------------------------------------------------------------------------------
namespace {
// n_blocks text blocks, produced by format(block, buffer) on any worker, land in block order behind `head`.
// Buffered write()s to ONE file are serialised by the kernel (the inode lock is held across the copy into the page cache:
// ~1.5 GB/s whatever the thread count), so the file is sized to an upper bound, mapped, filled by all workers at their
// chained offsets and cut to its true length at the end; where the mapping is refused the blocks go out with pwrite.
template <class F>
int write_blocks(const char* path, const char* head, int64_t head_len, int64_t n_blocks, int n_threads, int64_t bound, F&& format) {
  if (!path || head_len < 0 || (head_len > 0 && !head)) return io_fail(GNX_EINVAL, "write: bad arguments");
  const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) return io_fail(GNX_EINVAL, std::string("write: cannot open ") + path + ": " + strerror(errno));
  char* map = nullptr;
  const int64_t map_len = head_len + bound;
  if (map_len > 0 && !getenv("GNX_IO_NO_MMAP") && ftruncate(fd, (off_t)map_len) == 0) {
    bool reserved = true;
    struct statfs sf;
    const bool tmpfs = fstatfs(fd, &sf) == 0 && sf.f_type == 0x01021994;  // TMPFS_MAGIC: pages come from RAM either way
    if (!tmpfs) {
      const int e = posix_fallocate(fd, 0, (off_t)map_len);  // no SIGBUS on a full disk later
      if (e == ENOSPC || e == EFBIG || e == EDQUOT) reserved = false;
    }
    if (reserved) {
      void* m = mmap(nullptr, (size_t)map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      if (m != MAP_FAILED) map = (char*)m;
    }
    if (!map && ftruncate(fd, 0) != 0) {
      close(fd);
      return io_fail(GNX_EINVAL, std::string("write: cannot reset ") + path);
    }
  }
  auto put = [&](const char* b, size_t n, int64_t off) -> bool {
    if (map) {
      memcpy(map + off, b, n);
      return true;
    }
    while (n > 0) {
      const ssize_t w = pwrite(fd, b, n, (off_t)off);
      if (w < 0) {
        if (errno == EINTR) continue;
        return false;
      }
      b += w;
      n -= (size_t)w;
      off += w;
    }
    return true;
  };
  bool ok = put(head, (size_t)head_len, 0);
  std::unique_ptr<std::atomic<int64_t>[]> off(new std::atomic<int64_t>[(size_t)n_blocks + 1]);
  for (int64_t i = 0; i <= n_blocks; ++i) off[(size_t)i].store(-1, std::memory_order_relaxed);
  off[0].store(head_len, std::memory_order_release);
  std::atomic<int64_t> next{0};
  std::atomic<int> failed{ok ? 0 : 1};
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_blocks, gnx_io_threads(n_threads)));
  gnx_io_parallel(nt, [&](int) {
    std::vector<char> buf;
    for (;;) {
      const int64_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n_blocks) break;
      buf.clear();
      format(i, buf);
      int64_t o;
      while ((o = off[(size_t)i].load(std::memory_order_acquire)) < 0) std::this_thread::yield();
      off[(size_t)i + 1].store(o + (int64_t)buf.size(), std::memory_order_release);
      if (map && o + (int64_t)buf.size() > map_len) failed = 2;  // a formatter exceeded its own bound: never write outside the map
      else if (!failed.load(std::memory_order_relaxed) && !put(buf.data(), buf.size(), o)) failed = 1;
    }
  });
  const int64_t final_len = off[(size_t)n_blocks].load(std::memory_order_acquire);
  int rc_trunc = 0;
  if (map) {
    munmap(map, (size_t)map_len);
    rc_trunc = ftruncate(fd, (off_t)std::max<int64_t>(final_len, 0));
  }
  const int cr = close(fd);
  if (failed == 2) return io_fail(GNX_EINVAL, std::string("write: internal size bound exceeded for ") + path);
  if (failed || cr != 0 || rc_trunc != 0) return io_fail(GNX_EINVAL, std::string("write: I/O error on ") + path + ": " + strerror(errno));
  return GNX_OK;
}

inline void grow(std::vector<char>& buf, size_t used, size_t need) {
  if (buf.size() < used + need) buf.resize(std::max(buf.size() * 2, used + need));
}
}  // namespace

extern "C" int gnx_write_msp(const char* path, const char* head, int64_t head_len, const char* pb, const int64_t* po, const int32_t* labels,
                             int64_t N, int64_t ldl, int64_t W, int n_threads) {
  if (N < 0 || W < 0 || ldl < W || (W > 0 && (!pb || !po)) || (N > 0 && W > 0 && !labels)) return io_fail(GNX_EINVAL, "write_msp: bad arguments");
  const int nt = gnx_io_threads(n_threads);
  // labels (N, ldl) -> (W, N): a row of text reads one contiguous run
  std::vector<int32_t> T((size_t)W * (size_t)N);
  const int64_t nbw = (W + 15) / 16, nbn = (N + 255) / 256;
  par_for(nbw * nbn, nt, [&](int64_t t, int) {
    const int64_t w0 = (t % nbw) * 16, n0 = (t / nbw) * 256;
    const int64_t w1 = std::min(W, w0 + 16), n1 = std::min(N, n0 + 256);
    for (int64_t n = n0; n < n1; ++n)
      for (int64_t w = w0; w < w1; ++w) T[(size_t)w * N + n] = labels[(size_t)n * ldl + w];
  });
  return write_blocks(path, head, head_len, W, nt, (W > 0 ? po[W] - po[0] : 0) + W * (N * 12 + 2), [&](int64_t w, std::vector<char>& buf) {
    const size_t plen = (size_t)(po[w + 1] - po[w]);
    buf.resize(plen + (size_t)N * 12 + 2);
    char* o = buf.data();
    memcpy(o, pb + po[w], plen);
    o += plen;
    const int32_t* row = T.data() + (size_t)w * N;
    for (int64_t n = 0; n < N; ++n) {
      *o++ = '\t';
      const int32_t v = row[n];
      if (v >= 0 && v < 10) *o++ = (char)('0' + v);
      else o = put_int(o, v);
    }
    *o++ = '\n';
    buf.resize((size_t)(o - buf.data()));
  });
}

extern "C" int gnx_write_fb(const char* path, const char* head, int64_t head_len, const char* pb, const int64_t* po, const void* proba,
                            int is_f64, int64_t N, int64_t W, int64_t A, int n_threads) {
  if (N < 0 || W < 0 || A < 0 || (W > 0 && (!pb || !po)) || (N > 0 && W > 0 && A > 0 && !proba)) return io_fail(GNX_EINVAL, "write_fb: bad arguments");
  const size_t per = is_f64 ? 26 : 17;
  return write_blocks(path, head, head_len, W, gnx_io_threads(n_threads), (W > 0 ? po[W] - po[0] : 0) + W * (N * A * (int64_t)per + 2),
                      [&](int64_t w, std::vector<char>& buf) {
    const size_t plen = (size_t)(po[w + 1] - po[w]);
    buf.resize(plen + (size_t)N * (size_t)A * per + 2);
    char* o = buf.data();
    memcpy(o, pb + po[w], plen);
    o += plen;
    for (int64_t n = 0; n < N; ++n) {
      const size_t base = ((size_t)n * W + w) * A;
      if (is_f64) {
        const double* v = (const double*)proba + base;
        for (int64_t a = 0; a < A; ++a) {
          *o++ = '\t';
          if (v[a] == v[a]) o = put_float<double>(o, v[a]);  // NaN -> empty (pandas na_rep)
        }
      } else {
        const float* v = (const float*)proba + base;
        for (int64_t a = 0; a < A; ++a) {
          *o++ = '\t';
          if (v[a] == v[a]) o = put_float<float>(o, v[a]);
        }
      }
    }
    *o++ = '\n';
    buf.resize((size_t)(o - buf.data()));
  });
}

namespace {
struct GtLut {
  char t[256][8];
  explicit GtLut(bool dot) {
    const char* sym = dot ? "01.3" : "0123";
    for (int b = 0; b < 256; ++b) {
      char* o = t[b];
      o[0] = sym[b & 3];
      o[1] = '|';
      o[2] = sym[(b >> 2) & 3];
      o[3] = '\t';
      o[4] = sym[(b >> 4) & 3];
      o[5] = '|';
      o[6] = sym[(b >> 6) & 3];
      o[7] = '\t';
    }
  }
};
const GtLut kGtLut(false), kGtLutDot(true);

// "\t a|b \t c|d ..." of one gt2 row; returns the end (no newline)
inline char* put_gt_row(char* o, const uint8_t* row, int64_t ns, const GtLut& lut = kGtLut) {
  *o++ = '\t';
  const int64_t nb = ns / 2;
  for (int64_t b = 0; b < nb; ++b) {
    memcpy(o, lut.t[row[b]], 8);
    o += 8;
  }
  if (ns & 1) {
    memcpy(o, lut.t[row[nb]], 4);
    o += 4;
  }
  return o - 1;  // drop the trailing tab
}

constexpr int64_t kVcfBlock = 64;  // variants per text block
}  // namespace

extern "C" int gnx_write_vcf_gt2(const char* path, const char* head, int64_t head_len, const char* pb, const int64_t* po, const uint8_t* G,
                                 int64_t V, int64_t ldg, int64_t ns, int missing_as_dot, int n_threads) {
  if (V < 0 || ns <= 0 || ldg < (2 * ns + 3) / 4 || (V > 0 && (!pb || !po || !G))) return io_fail(GNX_EINVAL, "write_vcf_gt2: bad arguments");
  const GtLut& lut = missing_as_dot ? kGtLutDot : kGtLut;
  return write_blocks(path, head, head_len, (V + kVcfBlock - 1) / kVcfBlock, gnx_io_threads(n_threads), (V > 0 ? po[V] - po[0] : 0) + V * (ns * 4 + 10),
                      [&](int64_t blk, std::vector<char>& buf) {
    const int64_t v0 = blk * kVcfBlock, v1 = std::min(V, v0 + kVcfBlock);
    buf.resize((size_t)(po[v1] - po[v0]) + (size_t)(v1 - v0) * ((size_t)ns * 4 + 10));
    char* o = buf.data();
    for (int64_t v = v0; v < v1; ++v) {
      const size_t plen = (size_t)(po[v + 1] - po[v]);
      memcpy(o, pb + po[v], plen);
      o += plen;
      o = put_gt_row(o, G + (size_t)v * ldg, ns, lut);
      *o++ = '\n';
    }
    buf.resize((size_t)(o - buf.data()));
  });
}

extern "C" int gnx_write_phased_vcf(const char* path, const char* head, int64_t head_len, const gnx_vcf* src, const int64_t* rows, int64_t V,
                                    const char* ref_blob, const int64_t* ref_off, const char* alt_blob, const int64_t* alt_off, const uint8_t* G,
                                    int64_t ldg, int64_t ns, int n_threads) {
  if (!src || V < 0 || ns <= 0 || ldg < (2 * ns + 3) / 4 || (V > 0 && (!rows || !G)) || (ref_blob && !ref_off) || (alt_blob && !alt_off))
    return io_fail(GNX_EINVAL, "write_phased_vcf: bad arguments");
  const int64_t nsrc = src->info.n_variants;
  for (int64_t v = 0; v < V; ++v)
    if (rows[v] < 0 || rows[v] >= nsrc) return io_fail(GNX_EINVAL, "write_phased_vcf: row index outside the source VCF");
  auto col = [&](int f, int64_t r, const char*& b, size_t& n) {
    const StrCol& c = src->col[f];
    b = c.blob.data() + c.off[(size_t)r];
    n = (size_t)(c.off[(size_t)r + 1] - c.off[(size_t)r]);
  };
  int64_t bound = 0;
  for (int64_t v = 0; v < V; ++v) {
    const int64_t r = rows[v];
    auto len = [&](int f) { return src->col[f].off[(size_t)r + 1] - src->col[f].off[(size_t)r]; };
    bound += len(GNX_VCF_CHROM) + len(GNX_VCF_ID) + (ref_blob ? ref_off[v + 1] - ref_off[v] : len(GNX_VCF_REF)) +
             (alt_blob ? alt_off[v + 1] - alt_off[v] : len(GNX_VCF_ALT0)) + 96 + ns * 4;
  }
  return write_blocks(path, head, head_len, (V + kVcfBlock - 1) / kVcfBlock, gnx_io_threads(n_threads), bound, [&](int64_t blk, std::vector<char>& buf) {
    const int64_t v0 = blk * kVcfBlock, v1 = std::min(V, v0 + kVcfBlock);
    size_t used = 0;
    for (int64_t v = v0; v < v1; ++v) {
      const int64_t r = rows[v];
      const char *s0, *s1, *s2, *s3;
      size_t n0, n1, n2, n3;
      col(GNX_VCF_CHROM, r, s0, n0);
      col(GNX_VCF_ID, r, s1, n1);
      if (ref_blob) {
        s2 = ref_blob + ref_off[v];
        n2 = (size_t)(ref_off[v + 1] - ref_off[v]);
      } else col(GNX_VCF_REF, r, s2, n2);
      if (alt_blob) {
        s3 = alt_blob + alt_off[v];
        n3 = (size_t)(alt_off[v + 1] - alt_off[v]);
      } else col(GNX_VCF_ALT0, r, s3, n3);
      grow(buf, used, n0 + n1 + n2 + n3 + 96 + (size_t)ns * 4);
      char* o = buf.data() + used;
      memcpy(o, s0, n0);
      o += n0;
      *o++ = '\t';
      o = put_int(o, src->pos[(size_t)r]);
      *o++ = '\t';
      memcpy(o, s1, n1);
      o += n1;
      *o++ = '\t';
      memcpy(o, s2, n2);
      o += n2;
      *o++ = '\t';
      memcpy(o, s3, n3);
      o += n3;
      *o++ = '\t';
      const float q = src->qual[(size_t)r];
      if (q == q) o = put_float<float>(o, q);
      memcpy(o, "\tPASS\t.\tGT", 10);
      o += 10;
      o = put_gt_row(o, G + (size_t)v * ldg, ns);
      *o++ = '\n';
      used = (size_t)(o - buf.data());
    }
    buf.resize(used);
  });
}

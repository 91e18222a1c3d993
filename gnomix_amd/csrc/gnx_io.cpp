// gnx_io.cpp — the file side of the hot path, host code only (no HIP): worker pool, numpy's number text, and the .msp / .fb /
// phased-VCF writers (the VCF reader lives in gnx_vcf.cpp).  Contract: include/gnomix_io.h.
//
// Reference: src/utils.py:55-81 (read_vcf = scikit-allel's C parser behind gzip.open), src/postprocess.py:84-126 (write_msp /
// write_fb: numpy / pandas string conversion), src/utils.py:247-329 (npy_to_vcf: one pandas column of "a|b" strings per sample).
//
// Writers.  Text blocks (a window's row of the .msp / .fb, 64 records of a VCF) are formatted in parallel into a ring of
// buffers and written by ONE thread, in order, with plain write(): on the 2 x 64-core host of an MI355X box one writer moves
// 8.8 GB/s into tmpfs and 14 GB/s into the page cache of a disk file, 16 or more concurrent pwrite()s 2-3.7 GB/s, and
// stores through a shared mapping collapse to 0.6 GB/s at 256 threads (scripts/dev/io_probe.cpp).
#include "gnx_io.h"

#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

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
extern "C" const char* gnx_io_last_error(void) { return g_io_err.c_str(); }
int gnx_io_fail(int code, const std::string& msg) {
  g_io_err = msg;
  return code;
}
static inline int io_fail(int code, const std::string& msg) { return gnx_io_fail(code, msg); }
double gnx_io_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
  const int q = gnx_io_cpu_quota();
  if (q > 0) n = std::min(n, q);
  return std::max(1, std::min(n, 512));
}

// CPUs' worth of time the container may use per period (cgroup v2 cpu.max, v1 cfs quota), rounded up; 0: no limit.  The GPU
// boxes of this project show 256 logical CPUs and a quota of 16: threads beyond the quota are throttled, not run — the same
// sha256 loop finishes 26 thread-equivalents with 32 threads and 13 with 256.
int gnx_io_cpu_quota() {
  static const int cached = [] {
    long long quota = -1, period = 100000;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      if (fscanf(f, "%31s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
      fclose(f);
    } else {
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
      }
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(g, "%lld", &period) != 1) period = 100000;
        fclose(g);
      }
    }
    if (quota <= 0 || period <= 0) return 0;
    return (int)((quota + period - 1) / period);
  }();
  return cached;
}

// threads for work that streams memory (pread + parse, format + write).  Measured on an MI355X box (2 x 64 cores shown, CPU
// quota 16): the reader peaks at 32 workers — 79 GB/s of text, 60 GB/s with 16, 18 GB/s with 256 — and the writers at 16-64
// (scripts/dev/vcf_io_probe.py): these workers spend part of their time in page-cache copies and faults, so twice the quota
// keeps the allowed CPUs busy; without a quota 32 is where one file stops scaling.
int gnx_io_stream_threads(int requested) {
  if (requested > 0 || getenv("GNX_IO_THREADS")) return gnx_io_threads(requested);
  const int q = gnx_io_cpu_quota();
  cpu_set_t set;
  int n = 0;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)std::thread::hardware_concurrency();
  return std::max(1, std::min(std::min(n, 32), q > 0 ? 2 * q : 32));
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

// ---- float32 -> the shortest decimal that reads back as the same float (then the closest one, then the even one): Schubfach
// (R. Giulietti, "The Schubfach way to render doubles", 2020), binary32 instance.  v = c * 2^q -> f * 10^e in a few 64-bit
// multiplications and no loop; the same digits numpy's Dragon4 "unique" mode and Ryu print (tests: millions of values against numpy).
// kG1[k + 31] = floor(10^k * 2^-r) >> 63 with r = floor(log2 10^k) - 125, k = -31 .. 45 (126-bit constants, upper 63 bits).
static const uint64_t kG1[77] = {
    0x40e7599625a1fe7aull, 0x51212ffbaf0a7e18ull, 0x65697bfa9acd1d9full, 0x7ec3daf941806506ull,
    0x4f3a68dbc8f03f24ull, 0x63090312bb2c4eedull, 0x7bcb43d769f762a8ull, 0x4d5f0a66a23a9da9ull,
    0x60b6cd004ac94513ull, 0x78e480405d7b9658ull, 0x4b8ed0283a6d3df7ull, 0x5e72843249088d75ull,
    0x760f253edb4ab0d2ull, 0x49c97747490eae83ull, 0x5c3bd5191b525a24ull, 0x734aca5f6226f0adull,
    0x480ebe7b9d58566cull, 0x5a126e1a84ae6c07ull, 0x709709a125da0709ull, 0x465e6604b7a84465ull,
    0x57f5ff85e592557full, 0x6df37f675ef6eadfull, 0x44b82fa09b5a52cbull, 0x55e63b88c230e77eull,
    0x6b5fca6af2bd215eull, 0x431bde82d7b634daull, 0x53e2d6238da3c211ull, 0x68db8bac710cb295ull,
    0x4189374bc6a7ef9dull, 0x51eb851eb851eb85ull, 0x6666666666666666ull, 0x4000000000000000ull,
    0x5000000000000000ull, 0x6400000000000000ull, 0x7d00000000000000ull, 0x4e20000000000000ull,
    0x61a8000000000000ull, 0x7a12000000000000ull, 0x4c4b400000000000ull, 0x5f5e100000000000ull,
    0x7735940000000000ull, 0x4a817c8000000000ull, 0x5d21dba000000000ull, 0x746a528800000000ull,
    0x48c2739500000000ull, 0x5af3107a40000000ull, 0x71afd498d0000000ull, 0x470de4df82000000ull,
    0x58d15e1762800000ull, 0x6f05b59d3b200000ull, 0x4563918244f40000ull, 0x56bc75e2d6310000ull,
    0x6c6b935b8bbd4000ull, 0x43c33c1937564800ull, 0x54b40b1f852bda00ull, 0x69e10de76676d080ull,
    0x422ca8b0a00a4250ull, 0x52b7d2dcc80cd2e4ull, 0x6765c793fa10079dull, 0x409f9cbc7c4a04c2ull,
    0x50c783eb9b5c85f2ull, 0x64f964e68233a76full, 0x7e37be2022c0914bull, 0x4ee2d6d415b85aceull,
    0x629b8c891b267182ull, 0x7b426fab61f00de3ull, 0x4d0985cb1d3608aeull, 0x604be73de4838ad9ull,
    0x785ee10d5da46d90ull, 0x4b3b4ca85a86c47aull, 0x5e0a1fd271287598ull, 0x758ca7c70d7292feull,
    0x4977e8dc68679bdfull, 0x5bd5e313828182d6ull, 0x72cb5bd86321e38cull, 0x47bf19673df52e37ull,
    0x59aedfc10d7279c5ull,
};
static inline int64_t sf_flog10pow2(int64_t e) { return (e * 661971961083LL) >> 41; }
static inline int64_t sf_flog10_three_quarters_pow2(int64_t e) { return (e * 661971961083LL - 274743187321LL) >> 41; }
static inline int64_t sf_flog2pow10(int64_t e) { return (e * 913124641741LL) >> 38; }
static inline uint32_t sf_rop(uint64_t g, uint64_t cp) {
  const uint64_t x1 = (uint64_t)(((unsigned __int128)g * cp) >> 64);
  return (uint32_t)((x1 >> 31) | (((x1 & 0xffffffffull) + 0xffffffffull) >> 32));
}
// bits of a positive, finite, non-zero float -> f (no trailing zeros), e: the value prints as the digits of f times 10^e
static inline void f32_shortest(uint32_t bits, uint32_t& f_out, int& e_out) {
  const uint32_t t = bits & 0x7fffffu, bq = bits >> 23;
  const int64_t q = bq ? (int64_t)bq - 150 : -149;
  const uint64_t c = bq ? (0x800000u | t) : t;
  const uint32_t out = (uint32_t)(c & 1);
  const uint64_t cb = c << 2, cbr = cb + 2;
  uint64_t cbl;
  int64_t k;
  if (c != 0x800000u || q == -149) {
    cbl = cb - 2;
    k = sf_flog10pow2(q);
  } else {  // a power of two: the gap below is half the gap above
    cbl = cb - 1;
    k = sf_flog10_three_quarters_pow2(q);
  }
  const int h = (int)(q + sf_flog2pow10(-k) + 33);
  const uint64_t g = kG1[-k + 31] + 1;
  const uint32_t vb = sf_rop(g, cb << h), vbl = sf_rop(g, cbl << h), vbr = sf_rop(g, cbr << h);
  const uint32_t s = vb >> 2;
  uint32_t f;
  bool done = false;
  if (s >= 100) {
    const uint32_t sp10 = 10 * (s / 10), tp10 = sp10 + 10;
    const bool upin = vbl + out <= (sp10 << 2), wpin = (tp10 << 2) + out <= vbr;
    if (upin != wpin) {
      f = upin ? sp10 : tp10;
      done = true;
    }
  }
  if (!done) {
    const uint32_t t1 = s + 1;
    const bool uin = vbl + out <= (s << 2), win = (t1 << 2) + out <= vbr;
    if (uin != win) f = uin ? s : t1;
    else {
      const int32_t cmp = (int32_t)vb - (int32_t)((s + t1) << 1);
      f = (cmp < 0 || (cmp == 0 && (s & 1) == 0)) ? s : t1;
    }
  }
  int e = (int)k;
  while (f % 10 == 0) {  // f > 0
    f /= 10;
    ++e;
  }
  f_out = f;
  e_out = e;
}

// decimal digits of f (1 <= f < 10^9) into d (16 bytes are the caller's), two at a time from the end; returns their number
static const char kD2[] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263"
    "646566676869707172737475767778798081828384858687888990919293949596979899";
static inline int f32_digits(uint32_t f, char* d) {
  const int nd = f >= 100000000u ? 9 : f >= 10000000u ? 8 : f >= 1000000u ? 7 : f >= 100000u ? 6 : f >= 10000u ? 5 : f >= 1000u ? 4 : f >= 100u ? 3 : f >= 10u ? 2 : 1;
  char* p = d + nd;
  while (f >= 100) {
    const uint32_t r = f % 100;
    f /= 100;
    p -= 2;
    memcpy(p, kD2 + 2 * r, 2);
  }
  if (f >= 10) memcpy(p - 2, kD2 + 2 * f, 2);
  else p[-1] = (char)('0' + f);
  return nd;
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
  // digits dig[0 .. k) and the exponent e of the first one (d.ddd x 10^e)
  char dig[24];
  int k = 0, e = 0;
  if constexpr (sizeof(T) == 4) {
    uint32_t bits, f;
    memcpy(&bits, &v, 4);
    int e10;
    f32_shortest(bits, f, e10);
    k = f32_digits(f, dig);
    e = e10 + k - 1;
    if (e < 0 && e >= -4 && v >= 1e-4f && (double)v >= 1e-4) {  // 1e-4 <= v < 1 (numpy's rule reads the VALUE, not the digits: float32(1e-4) prints 1e-04): "0." + (-e - 1) zeros + digits — nearly every probability; fixed-size stores only
      memcpy(o, "0.000000", 8);
      memcpy(o + 1 - e, dig, 16);  // (the caller's buffer has 32 characters per number)
      return o + 1 - e + k;
    }
  } else {
    char sc[40];
    auto r = std::to_chars(sc, sc + 40, v, std::chars_format::scientific);  // d[.ddd]e[+-]XX
    const char* p = sc;
    while (p < r.ptr && *p != 'e') {
      if (*p != '.') dig[k++] = *p;
      ++p;
    }
    ++p;  // 'e'
    const bool eneg = (*p == '-');
    ++p;
    while (p < r.ptr) e = e * 10 + (*p++ - '0');
    if (eneg) e = -e;
  }
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
// writers
// ------------------------------------------------------------------------------------------------------------------------
namespace {
// n_blocks text blocks, produced by format(block, buffer) on the workers, land in block order behind `head`: a ring of
// buffers between the formatters and the calling thread, which is the only writer.  `bound` (an upper bound of the text)
// is unused by this strategy and kept for the callers' documentation of their sizes.
template <class F>
int write_blocks(const char* path, const char* head, int64_t head_len, int64_t n_blocks, int n_threads, int64_t bound, F&& format) {
  (void)bound;
  if (!path || head_len < 0 || (head_len > 0 && !head)) return io_fail(GNX_EINVAL, "write: bad arguments");
  const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) return io_fail(GNX_EINVAL, std::string("write: cannot open ") + path + ": " + strerror(errno));
  auto put = [&](const char* b, size_t n) -> bool {
    while (n > 0) {
      const ssize_t w = write(fd, b, n);
      if (w < 0) {
        if (errno == EINTR) continue;
        return false;
      }
      b += w;
      n -= (size_t)w;
    }
    return true;
  };
  bool ok = put(head, (size_t)head_len);
  int err_no = ok ? 0 : errno;
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_blocks, gnx_io_threads(n_threads)));
  if (n_blocks > 0 && ok) {
    const int64_t K = std::max<int64_t>(2, std::min<int64_t>(n_blocks, (int64_t)2 * nt));  // ring slots
    std::vector<std::vector<char>> ring((size_t)K);
    std::unique_ptr<std::atomic<int64_t>[]> filled(new std::atomic<int64_t>[(size_t)K]);  // block held by the slot, -1: free
    for (int64_t i = 0; i < K; ++i) filled[(size_t)i].store(-1, std::memory_order_relaxed);
    std::atomic<int64_t> next{0}, written{0};
    std::atomic<int> failed{0};
    auto pause = [](int& spins) {
      if (++spins < 64) std::this_thread::yield();
      else std::this_thread::sleep_for(std::chrono::microseconds(50));
    };
    auto produce = [&]() {
      for (;;) {
        const int64_t i = next.fetch_add(1, std::memory_order_relaxed);
        if (i >= n_blocks) break;
        int spins = 0;
        while (i >= written.load(std::memory_order_acquire) + K && !failed.load(std::memory_order_relaxed)) pause(spins);
        if (failed.load(std::memory_order_relaxed)) break;
        std::vector<char>& buf = ring[(size_t)(i % K)];
        buf.clear();
        format(i, buf);
        filled[(size_t)(i % K)].store(i, std::memory_order_release);
      }
    };
    auto drain = [&]() {  // the writer: blocks in order
      for (int64_t i = 0; i < n_blocks; ++i) {
        int spins = 0;
        while (filled[(size_t)(i % K)].load(std::memory_order_acquire) != i) pause(spins);
        const std::vector<char>& buf = ring[(size_t)(i % K)];
        if (!put(buf.data(), buf.size())) {
          // a failed write (disk full, file size limit) ends the job: the producers see `failed` and leave without filling the
          // blocks they claimed, so waiting for those would never end (write_msp / write_fb hung on ENOSPC with > 1 thread)
          err_no = errno;
          failed.store(1, std::memory_order_release);
          return;
        }
        written.store(i + 1, std::memory_order_release);
      }
    };
    if (nt == 1) {  // one thread: format and write in turn
      std::vector<char> buf;
      for (int64_t i = 0; i < n_blocks && ok; ++i) {
        buf.clear();
        format(i, buf);
        if (!put(buf.data(), buf.size())) {
          err_no = errno;
          ok = false;
        }
      }
    } else {
      gnx_io_parallel(nt, [&](int tid) {
        if (tid == 0) drain();
        else produce();
      });
      ok = !failed.load();
    }
  }
  const int cr = close(fd);
  if (!ok || cr != 0) return io_fail(GNX_EINVAL, std::string("write: I/O error on ") + path + ": " + strerror(err_no ? err_no : errno));
  return GNX_OK;
}

inline void grow(std::vector<char>& buf, size_t used, size_t need) {
  if (buf.size() < used + need) buf.resize(std::max(buf.size() * 2, used + need));
}
}  // namespace

int gnx_io_write_file(const char* path, const char* head, size_t head_len, const char* body, size_t body_len) {
  const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) return io_fail(GNX_EINVAL, std::string("write: cannot open ") + path + ": " + strerror(errno));
  bool ok = true;
  int err_no = 0;
  for (int part = 0; part < 2 && ok; ++part) {
    const char* b = part ? body : head;
    size_t n = part ? body_len : head_len;
    while (n > 0) {
      const ssize_t w = write(fd, b, std::min<size_t>(n, (size_t)1 << 30));
      if (w < 0) {
        if (errno == EINTR) continue;
        ok = false;
        err_no = errno;
        break;
      }
      b += w;
      n -= (size_t)w;
    }
  }
  const int cr = close(fd);
  if (!ok || cr != 0) return io_fail(GNX_EINVAL, std::string("write: I/O error on ") + path + ": " + strerror(err_no ? err_no : errno));
  return GNX_OK;
}

extern "C" int gnx_write_msp(const char* path, const char* head, int64_t head_len, const char* pb, const int64_t* po, const int32_t* labels,
                             int64_t N, int64_t ldl, int64_t W, int n_threads) {
  if (N < 0 || W < 0 || ldl < W || (W > 0 && (!pb || !po)) || (N > 0 && W > 0 && !labels)) return io_fail(GNX_EINVAL, "write_msp: bad arguments");
  const int nt = gnx_io_stream_threads(n_threads);
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
  // (re-laying (N, W, A) window-major first, so that a line reads one contiguous run, was measured: 0.082-0.086 s against 0.074-0.080 s
  //  for chr22 x 10 000 haplotypes — the strided reads are not what bounds this writer)
  return write_blocks(path, head, head_len, W, gnx_io_stream_threads(n_threads), (W > 0 ? po[W] - po[0] : 0) + W * (N * A * (int64_t)per + 2),
                      [&](int64_t w, std::vector<char>& buf) {
    const size_t plen = (size_t)(po[w + 1] - po[w]);
    buf.resize(plen + (size_t)N * (size_t)A * per + 2 + 32);  // + 32: put_float<float> stores 16 digit bytes whatever their number
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
  return write_blocks(path, head, head_len, (V + kVcfBlock - 1) / kVcfBlock, gnx_io_stream_threads(n_threads), (V > 0 ? po[V] - po[0] : 0) + V * (ns * 4 + 10),
                      [&](int64_t blk, std::vector<char>& buf) {
    const int64_t v0 = blk * kVcfBlock, v1 = std::min(V, v0 + kVcfBlock);
    buf.resize((size_t)(po[v1] - po[v0]) + (size_t)(v1 - v0) * ((size_t)ns * 4 + 10) + 32);
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
    const gnx_strcol& c = src->col[f];
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
  return write_blocks(path, head, head_len, (V + kVcfBlock - 1) / kVcfBlock, gnx_io_stream_threads(n_threads), bound, [&](int64_t blk, std::vector<char>& buf) {
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

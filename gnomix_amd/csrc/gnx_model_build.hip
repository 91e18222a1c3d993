// gnx_model_build.hip — model preparation behind gnx_model_load (include/gnomix_hip.h): the host-side re-layout of a
// gnx_model_desc into what the kernels read — fixed-point digit planes and chunk tables of the logistic base, packed / rank-quantised /
// pointer-node trees of the tree smoother, per-window heaps of the forest bases, bit planes and dual coefficients of the CovRSK
// base, the CRF's tables — validated on the way (every index a kernel will follow is checked here, once).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <thread>
#include <utility>

#include "gnx_internal.h"

static inline int fail(gnx_ctx* ctx, int code, const std::string& msg) { return gnx_fail(ctx, code, msg); }

#define HIPCHK(ctx, expr)                                                                      \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess)                                                                     \
      return fail((ctx), GNX_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
  } while (0)

// [0, n) cut into contiguous ranges, one per host thread (at most 16, at least `grain` items each): fn(lo, hi, thread index).
// Model preparation is host arithmetic over tens of millions of weights (a whole-genome set of logistic models: 2.5e8): single-threaded
// it was most of gnx_model_load's time.  A thread that cannot be started (resource limits) is run inline: nothing is thrown.
template <typename F>
static unsigned parallel_ranges(size_t n, size_t grain, F&& fn) {
  unsigned nth = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  nth = (unsigned)std::min<size_t>(nth, std::max<size_t>(1, n / std::max<size_t>(grain, 1)));
  if (nth <= 1) {
    fn((size_t)0, n, 0u);
    return 1;
  }
  std::vector<std::thread> th;
  th.reserve(nth);
  for (unsigned t = 0; t < nth; ++t) {
    const size_t lo = n * t / nth, hi = n * (t + 1) / nth;
    try { th.emplace_back([&fn, lo, hi, t]() { fn(lo, hi, t); }); } catch (...) { fn(lo, hi, t); }
  }
  for (auto& t : th) t.join();
  return nth;
}
static unsigned parallel_max_threads() { return std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u); }

// ---- the logistic base's prepared planes as a relocatable blob (gnx_model_export_prepared -> gnx_model_desc.prepared) -------------
// | GnxPreparedHdr | wscale[W] float64 | V8 | pad to 16 | V2 or V2F |  — valid for exactly one (coefficients, geometry, ABI, plane
// settings); everything else of the model (chunk / run tables, intercepts) is cheap and always rebuilt from the description.
struct GnxPreparedHdr {
  char magic[8];                 // "GNXPLR1"
  uint32_t hdr_bytes, abi;
  int64_t C, M, ctx, W;
  int32_t A, NT, NT2, EPR, flat, reserved;
  int64_t n_chunks, n_runs;
  uint64_t key;                  // lr_key(): the coefficient rows the planes were made from
  int64_t v8_bytes, v2_bytes;
};
static const char kPreparedMagic[8] = {'G', 'N', 'X', 'P', 'L', 'R', '1', 0};

// 64-bit hash of the coefficient rows a model uses (row i, class a: the first width_i float64 of its lr_ldc), one hash per window
// folded in window order: different weights, window widths or row order give a different key (not cryptographic: a cache key)
static uint64_t lr_key(const gnx_model_desc* d, int64_t W, int64_t M_, int64_t rem) {
  std::vector<uint64_t> hw((size_t)W);
  parallel_ranges((size_t)W, 16, [&](size_t lo, size_t hi, unsigned) {
    for (size_t i = lo; i < hi; ++i) {
      const int64_t width = M_ + ((int64_t)i == W - 1 ? rem : 0);
      uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)i;
      for (int a = 0; a < d->A; ++a) {
        const double* row = d->lr_coef + ((size_t)i * d->A + a) * (size_t)d->lr_ldc;
        for (int64_t k = 0; k < width; ++k) {
          uint64_t x;
          std::memcpy(&x, row + k, 8);
          h = (h ^ x) * 0xD6E8FEB86659FD93ull;
          h ^= h >> 29;
        }
      }
      hw[i] = h;
    }
  });
  uint64_t h = 0xA0761D6478BD642Full;
  for (uint64_t x : hw) { h = (h ^ x) * 0xE7037ED1A0B428DBull; h ^= h >> 32; }
  return h;
}

template <typename T>
static int dev_upload_raw(gnx_model* m, const void* src, size_t bytes, const T** out, size_t pad_bytes) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes + pad_bytes + 16);
  if (e != hipSuccess) return gnx_fail(m->ctx, GNX_ENOMEM, std::string("hipMalloc model: ") + hipGetErrorString(e));
  m->dev_allocs.push_back(p);
  m->info.device_bytes += (int64_t)(bytes + pad_bytes + 16);
  if (bytes && (e = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice)) != hipSuccess) return gnx_fail(m->ctx, GNX_EHIP, std::string("hipMemcpy model: ") + hipGetErrorString(e));
  if ((e = hipMemset((char*)p + bytes, 0, pad_bytes + 16)) != hipSuccess) return gnx_fail(m->ctx, GNX_EHIP, std::string("hipMemset model: ") + hipGetErrorString(e));
  *out = (const T*)p;
  return GNX_OK;
}

// ------------------------------------------------------------------------------------------------
// model preparation: logistic base
// ------------------------------------------------------------------------------------------------
int gnx_build_lr(gnx_model* m, const gnx_model_desc* d) {
  gnx_ctx* ctx = m->ctx;
  const int64_t C = d->C, M = d->M, cx = d->ctx;
  const int A = d->A;
  const int64_t W = C / M, rem = C - M * W, M_ = M + 2 * cx, Cp = C + 2 * cx;
  if (!d->lr_coef || !d->lr_intercept) return fail(ctx, GNX_EINVAL, "logistic base: lr_coef / lr_intercept is NULL");
  if (d->lr_ldc < M_ + rem) return fail(ctx, GNX_EINVAL, "logistic base: lr_ldc < M + 2*ctx + rem");
  const int64_t R = (M_ + M - 1) / M;
  const int64_t NC = R * A;
  const int NT = (int)((NC + 15) / 16);
  if (NT > 4)
    return fail(ctx, GNX_EUNSUPPORTED, "logistic base: ceil((M+2ctx)/M)*A > 64 class columns per SNP (context ratio too large)");

  auto wstart = [&](int64_t i) { return i * M; };                                // padded coords
  auto wend = [&](int64_t i) { return (i < W - 1) ? i * M + M_ : Cp; };          // padded coords, exclusive
  std::vector<int64_t> fpos((size_t)W);                                          // flush position, real coords
  for (int64_t i = 0; i < W; ++i) fpos[(size_t)i] = std::min<int64_t>(wend(i) - cx, C);

  // pieces end at distinct flush positions
  std::vector<int64_t> bounds{0};
  for (int64_t i = 0; i < W; ++i)
    if (fpos[(size_t)i] > bounds.back()) bounds.push_back(fpos[(size_t)i]);
  if (bounds.back() != C) return fail(ctx, GNX_EINVAL, "logistic base: internal piece construction failed");
  const size_t n_pieces = bounds.size() - 1;

  std::vector<int32_t> chunk_j0, chunk_flush0, chunk_nflush, piece_chunk0(n_pieces + 1);
  std::vector<int64_t> chunk_end;  // real end (exclusive) of the piece the chunk belongs to
  {
    int64_t wi = 0;
    for (size_t k = 0; k < n_pieces; ++k) {
      piece_chunk0[k] = (int32_t)chunk_j0.size();
      const int64_t b0 = bounds[k], b1 = bounds[k + 1];
      const int64_t nreal = (b1 - b0 + 63) / 64;
      // every piece holds an EVEN number of chunks (the kernels step two chunks at a time and flush between steps):
      // an odd piece gets one all-zero chunk that re-reads the bytes of its last chunk
      const int64_t nch = nreal + (nreal & 1);
      int64_t f0 = wi, nf = 0;
      while (wi < W && fpos[(size_t)wi] == b1) { ++wi; ++nf; }
      for (int64_t c = 0; c < nch; ++c) {
        const bool dummy = c >= nreal;
        chunk_j0.push_back((int32_t)(b0 + 64 * std::min(c, nreal - 1)));
        chunk_end.push_back(dummy ? b0 : b1);  // j >= chunk_end skips every weight of a dummy chunk
        const bool last = (c == nch - 1);
        chunk_flush0.push_back(last && nf ? (int32_t)f0 : -1);
        chunk_nflush.push_back(last ? (int32_t)nf : 0);
      }
    }
    piece_chunk0[n_pieces] = (int32_t)chunk_j0.size();
  }
  const size_t n_chunks = chunk_j0.size();

  std::vector<int32_t> win_chunk0((size_t)W), win_chunk1((size_t)W);
  for (int64_t i = 0; i < W; ++i) {
    const int64_t s = std::max<int64_t>(wstart(i) - cx, 0);
    size_t k = (size_t)(std::upper_bound(bounds.begin(), bounds.end(), s) - bounds.begin()) - 1;
    if (k >= n_pieces) k = n_pieces - 1;
    win_chunk0[(size_t)i] = piece_chunk0[k];
    size_t kf = (size_t)(std::lower_bound(bounds.begin(), bounds.end(), fpos[(size_t)i]) - bounds.begin());
    win_chunk1[(size_t)i] = piece_chunk0[kf];  // piece kf-1 ends at fpos -> one past its last chunk
  }

  // fragment-ordered, reflect-folded weights
  const char* impl = std::getenv("GNX_BASE_LR_IMPL");  // "i8" (default, exact fixed point) or "f64" (f64 MFMA)
  m->lr_i8 = !(impl && std::string(impl) == "f64");
  // prepared planes handed in with the description (gnx_model_desc.prepared): validated here and below, used instead of the two
  // passes over the coefficients; ANY mismatch is GNX_ESTALE and loads nothing
  const GnxPreparedHdr* ph = nullptr;
  auto stale = [&](const char* what) { return fail(ctx, GNX_ESTALE, std::string("prepared planes: ") + what + " (load without them and export them again)"); };
  if (m->lr_i8) m->lr_key = lr_key(d, W, M_, rem);
  if (d->prepared) {
    if (!m->lr_i8) return stale("made for the int8 kernels, GNX_BASE_LR_IMPL=f64 is set");
    if (d->prepared_bytes < (int64_t)sizeof(GnxPreparedHdr)) return stale("truncated (no header)");
    ph = static_cast<const GnxPreparedHdr*>(d->prepared);
    if (std::memcmp(ph->magic, kPreparedMagic, 8) != 0 || ph->hdr_bytes != sizeof(GnxPreparedHdr)) return stale("not a prepared-planes blob of this library");
    if (ph->abi != (uint32_t)GNX_ABI_VERSION) return stale("written by another ABI version");
    if (ph->C != C || ph->M != M || ph->ctx != cx || ph->W != W || ph->A != A || ph->NT != NT || ph->n_chunks != (int64_t)n_chunks)
      return stale("another model geometry");
    if (ph->key != m->lr_key) return stale("other coefficients");
    if (ph->v8_bytes != (int64_t)(n_chunks * (size_t)NT * 7 * 64 * 16) || ph->v2_bytes < 0 ||
        d->prepared_bytes != (int64_t)sizeof(GnxPreparedHdr) + W * 8 + ((ph->v8_bytes + 15) & ~(int64_t)15) + ph->v2_bytes)
      return stale("truncated or of an unexpected size");
  }
  std::vector<double> V(ph ? 0 : n_chunks * 16 * (size_t)NT * 64, 0.0);
  std::vector<int32_t> Vwin(m->lr_i8 && !ph ? V.size() : 0, -1);
  std::vector<double> maxabs((size_t)W, 0.0);
  const double* coef = d->lr_coef;
  const int64_t ldc = d->lr_ldc;
  std::vector<std::vector<double>> maxabs_t(parallel_max_threads());  // per-thread window maxima, merged below
  if (!ph) parallel_ranges(n_chunks, 64, [&](size_t c_lo, size_t c_hi, unsigned tid) {
  std::vector<double>& maxabs = maxabs_t[tid];
  maxabs.assign((size_t)W, 0.0);
  for (size_t c = c_lo; c < c_hi; ++c)
    for (int t = 0; t < 16; ++t)
      for (int kq = 0; kq < 4; ++kq) {
        const int64_t j = (int64_t)chunk_j0[c] + 16 * kq + t;
        if (j >= chunk_end[c]) continue;  // zero rows pad the piece to a multiple of 64 SNPs
        const int64_t p = j + cx;
        const int64_t i0 = std::min<int64_t>(p / M, W - 1);
        for (int64_t slot = 0; slot < R; ++slot) {
          int64_t i = i0 - (((i0 - slot) % R + R) % R);
          if (i < 0 || p >= wend(i)) continue;
          const int64_t ws_ = wstart(i), we_ = wend(i);
          int64_t pp[3];
          int np = 0;
          if (j < cx) pp[np++] = cx - 1 - j;             // left reflection (base.py:42)
          pp[np++] = p;                                   // direct
          if (j >= C - cx) pp[np++] = 2 * C + cx - 1 - j; // right reflection (base.py:43)
          for (int a = 0; a < A; ++a) {
            double wsum = 0.0;
            bool any = false;
            for (int q = 0; q < np; ++q)
              if (pp[q] >= ws_ && pp[q] < we_) {
                wsum += coef[((size_t)i * A + a) * (size_t)ldc + (size_t)(pp[q] - ws_)];
                any = true;
              }
            if (!any) continue;
            const int64_t col = slot * A + a;
            const int nt = (int)(col / 16), c16 = (int)(col % 16);
            const size_t vi = ((c * 16 + (size_t)t) * NT + (size_t)nt) * 64 + (size_t)(kq * 16 + c16);
            V[vi] = wsum;
            if (m->lr_i8) {
              Vwin[vi] = (int32_t)i;
              maxabs[(size_t)i] = std::max(maxabs[(size_t)i], std::fabs(wsum));
            }
          }
        }
      }
  });
  for (const auto& mt : maxabs_t)
    for (size_t i = 0; i < mt.size(); ++i) maxabs[i] = std::max(maxabs[i], mt[i]);

  std::vector<double> icpt(d->lr_intercept, d->lr_intercept + (size_t)W * A);
  int rc;
  if (m->lr_i8) {
    // exact fixed point: q = round(c * 2^f_w), |q| < 2^54, seven balanced base-256 digits per weight
    std::vector<int> fexp((size_t)W, 0);
    std::vector<double> wscale((size_t)W, 1.0);
    const uint8_t* blob = static_cast<const uint8_t*>(d->prepared);
    const uint8_t* blob_v8 = ph ? blob + sizeof(GnxPreparedHdr) + (size_t)W * 8 : nullptr;
    const uint8_t* blob_v2 = ph ? blob_v8 + ((ph->v8_bytes + 15) & ~(int64_t)15) : nullptr;
    if (ph) std::memcpy(wscale.data(), blob + sizeof(GnxPreparedHdr), (size_t)W * 8);
    else for (int64_t i = 0; i < W; ++i) {
      if (!(maxabs[(size_t)i] < 1e300)) return fail(ctx, GNX_EINVAL, "logistic base: non-finite coefficient");
      if (maxabs[(size_t)i] > 0.0) fexp[(size_t)i] = 53 - std::ilogb(maxabs[(size_t)i]);
      wscale[(size_t)i] = std::ldexp(1.0, -fexp[(size_t)i]);
    }
    m->lr_v8_bytes = (int64_t)(n_chunks * (size_t)NT * 7 * 64 * 16);
    std::vector<int8_t> V8(ph ? 0 : (size_t)m->lr_v8_bytes, 0);
    if (!ph) parallel_ranges(n_chunks, 64, [&](size_t c_lo, size_t c_hi, unsigned) {
    for (size_t c = c_lo; c < c_hi; ++c)
      for (int t = 0; t < 16; ++t)
        for (int nt = 0; nt < NT; ++nt)
          for (int ln = 0; ln < 64; ++ln) {
            const size_t vi = ((c * 16 + (size_t)t) * NT + (size_t)nt) * 64 + (size_t)ln;
            const int32_t wi = Vwin[vi];
            if (wi < 0 || V[vi] == 0.0) continue;
            long long q = std::llrint(std::ldexp(V[vi], fexp[(size_t)wi]));
            for (int l = 0; l < 7; ++l) {
              long long dg = (l < 6) ? ((((q + 128) % 256) + 256) % 256) - 128 : q;
              V8[(((c * NT + (size_t)nt) * 7 + (size_t)l) * 64 + (size_t)ln) * 16 + (size_t)t] = (int8_t)dg;
              q = (q - dg) / 256;
            }
          }
    });
    if (ph) rc = dev_upload_raw(m, blob_v8, (size_t)ph->v8_bytes, &m->lr.V8, 64);
    else rc = gnx_dev_upload(m, V8, &m->lr.V8, 64);
    if (rc != GNX_OK) return rc;
#ifdef GNX_EXPERIMENTS
    // flat column tiles for k_base_logistic_i8_fl: column q = slot * 7 + limb, ceil(NC * 7 / 16) tiles instead of NT * 7
    // (A = 12 at the default context: 11 instead of 14); an experiment that measured slower (see the kernel's header): built only
    // when GNX_LR_FLAT=1 asks for it at model load
    const int NF = (int)((NC * 7 + 15) / 16);
    if (!ph && NF < NT * 7 && NF >= 8 && NF <= 11 && A <= 16 && std::getenv("GNX_LR_FLAT") && std::atoi(std::getenv("GNX_LR_FLAT")) > 0) {
      std::vector<int8_t> V8F(n_chunks * (size_t)NF * 64 * 16, 0);
      for (size_t c = 0; c < n_chunks; ++c)
        for (int ft = 0; ft < NF; ++ft)
          for (int ln = 0; ln < 64; ++ln) {
            const int q = ft * 16 + (ln & 15), slot = q / 7, l = q - 7 * slot;
            if (slot >= NC) continue;
            const int8_t* src = &V8[(((c * NT + (size_t)(slot / 16)) * 7 + (size_t)l) * 64 + (size_t)((ln & ~15) + slot % 16)) * 16];
            std::copy(src, src + 16, &V8F[((c * NF + (size_t)ft) * 64 + (size_t)ln) * 16]);
          }
      if ((rc = gnx_dev_upload(m, V8F, &m->lr.V8F, 64)) != GNX_OK) return rc;
      m->lr.NF = NF;
    }
#endif
    if ((rc = gnx_dev_upload(m, wscale, &m->lr.wscale)) != GNX_OK) return rc;
    // ---- the same weights for the 2-bit-native pass (k_base_logistic_p2.hip): the pieces walked in dword-aligned runs of 256 SNPs
    // (a piece [b0, b1) starts at SNP b0 & ~15: the up to 15 SNPs before b0 and everything from b1 on meet zero weights).
    // Lane kq of a row's four lanes holds SNPs [64 kq, 64 kq + 64) of the run as four 32-bit words; entry k multiplies word k of every
    // lane, whose in-register unpack puts SNP field f = 4 (t & 3) + (t >> 2) at k position t: k position (kq, t) of entry k of a run
    // starting at SNP s is SNP  s + 16 EPR kq + 16 k + 4 (t & 3) + (t >> 2)  (EPR = entries per run, 4 or 8).  Same folded weights, same f_w, same digits as V8.
    {
      // column tiles of the 2-bit pass: one tile (column = slot * A + class, as in V8) when the R * A class columns of a SNP fit 16;
      // otherwise one tile PER SLOT (column = class) and one pass per tile (k_base_logistic_p2.hip)
      const int NT2 = NC <= 16 ? 1 : (A <= 16 ? (int)R : 0);
      const int64_t cs2 = NT2 == 1 ? A : 16;
      const char* p2env = std::getenv("GNX_LR_P2");  // 0: never build the planes; 2: build them whatever the padding costs (tests)
      const int p2mode = p2env ? std::atoi(p2env) : 1;
      // a run is RS = 256 SNPs (64 packed bytes per row visit, 4 MFMA entries).  Runs of 512 (128 bytes = whole cache lines per row
      // visit, 8 entries; GNX_LR_P2_RUN=512) were built as the "next lever" and measured the same: 0.595 / 0.608 ms at config 2,
      // 9.52 / 9.59 ms at config 5a (scripts/dev/p2_runlen.sh) — the pass is bound on the matrix / epilogue side by then, and the
      // longer run costs 16 KB more LDS.  Short pieces (small windows) would multiply mostly padding: such models keep the int8 kernels.
      auto padded_with = [&](int64_t rs) {
        int64_t p = 0;
        for (size_t k = 0; k < n_pieces; ++k) p += ((bounds[k + 1] - (bounds[k] & ~(int64_t)15) + rs - 1) / rs) * rs;
        return p;
      };
      int64_t RS = 256;
      if (const char* e = std::getenv("GNX_LR_P2_RUN")) { const int v = std::atoi(e); if (v == 256 || v == 512) RS = v; }
      const int EPR = (int)(RS / 64);
      const bool want_p2 = NT2 > 0 && p2mode != 0 && (p2mode == 2 || padded_with(RS) * 4 <= C * 5);
      if (ph && ((ph->v2_bytes > 0) != want_p2)) return stale("made with other GNX_LR_P2 settings");
      if (want_p2) {
        std::vector<int32_t> run_byte, run_flush0, run_nflush, piece_run0(n_pieces + 1);
        std::vector<int64_t> run_s, run_b0, run_b1;
        {
          int64_t wi = 0;
          for (size_t k = 0; k < n_pieces; ++k) {
            piece_run0[k] = (int32_t)run_byte.size();
            const int64_t b0 = bounds[k], b1 = bounds[k + 1];
            const int64_t s0 = b0 & ~(int64_t)15;  // 32-bit aligned in the packed row: a load that is not dword-aligned is split by the
                                                    // texture addresser (measured: ~4.5 L1 tag accesses per lane instead of ~0.5)
            const int64_t nr = (b1 - s0 + RS - 1) / RS;
            int64_t f0 = wi, nf = 0;
            while (wi < W && fpos[(size_t)wi] == b1) { ++wi; ++nf; }
            for (int64_t r = 0; r < nr; ++r) {
              run_byte.push_back((int32_t)((s0 + RS * r) / 4));
              run_s.push_back(s0 + RS * r); run_b0.push_back(b0); run_b1.push_back(b1);
              const bool last = (r == nr - 1);
              run_flush0.push_back(last && nf ? (int32_t)f0 : -1);
              run_nflush.push_back(last ? (int32_t)nf : 0);
            }
          }
          piece_run0[n_pieces] = (int32_t)run_byte.size();
        }
        const size_t n_runs = run_byte.size();
        std::vector<int32_t> win_run0((size_t)W), win_run1((size_t)W);
        for (int64_t i = 0; i < W; ++i) {
          const int64_t s = std::max<int64_t>(wstart(i) - cx, 0);
          size_t k = (size_t)(std::upper_bound(bounds.begin(), bounds.end(), s) - bounds.begin()) - 1;
          if (k >= n_pieces) k = n_pieces - 1;
          win_run0[(size_t)i] = piece_run0[k];
          const size_t kf = (size_t)(std::lower_bound(bounds.begin(), bounds.end(), fpos[(size_t)i]) - bounds.begin());
          win_run1[(size_t)i] = piece_run0[kf];
        }
        // R * A == 24 class columns: flat column tiles (BaseLRDev::V2F) instead of one tile per slot; GNX_LR_P2_FLAT=0 keeps the slot tiles
        const char* flat_env = std::getenv("GNX_LR_P2_FLAT");
        const bool flat = NC == GNX_LR_FLAT_COLS && A <= 16 && EPR == 4 && !(flat_env && std::atoi(flat_env) == 0);
        const int NTB = flat ? GNX_LR_FLAT_TILES : NT2;  // 1 KB digit blocks per entry: flat tiles, or column tiles x 7 limbs below
        const size_t entry_bytes = flat ? (size_t)NTB * 64 * 16 : (size_t)NT2 * 7 * 64 * 16;
        m->lr_v2_bytes = (int64_t)(n_runs * (size_t)EPR * entry_bytes);
        if (ph && (ph->v2_bytes != m->lr_v2_bytes || ph->EPR != EPR || ph->NT2 != NT2 || ph->flat != (flat ? 1 : 0) || ph->n_runs != (int64_t)n_runs))
          return stale("made with other GNX_LR_P2 / GNX_LR_P2_RUN / GNX_LR_P2_FLAT settings");
        std::vector<int8_t> V2(ph ? 0 : (size_t)m->lr_v2_bytes, 0);
        auto fill_runs = [&](size_t r_lo, size_t r_hi) {
          std::vector<double> wsum((size_t)A);
          std::vector<uint8_t> any((size_t)A);
          for (size_t r = r_lo; r < r_hi; ++r)
            for (int k = 0; k < EPR; ++k)
              for (int kq = 0; kq < 4; ++kq)
                for (int t = 0; t < 16; ++t) {
                  const int64_t j = run_s[r] + 16 * EPR * kq + 16 * k + 4 * (t & 3) + (t >> 2);  // lane kq holds 16 EPR SNPs of the run, word k of them
                  if (j < run_b0[r] || j >= run_b1[r]) continue;
                  const int64_t p = j + cx;
                  const int64_t i0 = std::min<int64_t>(p / M, W - 1);
                  for (int64_t slot = 0; slot < R; ++slot) {
                    const int64_t i = i0 - (((i0 - slot) % R + R) % R);
                    if (i < 0 || p >= wend(i)) continue;
                    const int64_t ws_ = wstart(i), we_ = wend(i);
                    int64_t pp[3];
                    int np = 0;
                    if (j < cx) pp[np++] = cx - 1 - j;
                    pp[np++] = p;
                    if (j >= C - cx) pp[np++] = 2 * C + cx - 1 - j;
                    for (int a = 0; a < A; ++a) {
                      double acc = 0.0;
                      bool got = false;
                      for (int q = 0; q < np; ++q)
                        if (pp[q] >= ws_ && pp[q] < we_) {
                          acc += coef[((size_t)i * A + a) * (size_t)ldc + (size_t)(pp[q] - ws_)];
                          got = true;
                        }
                      if (!got || acc == 0.0) continue;
                      const int64_t col = flat ? slot * A + a : slot * cs2 + a;
                      const int nt = (int)(col / 16), c16 = (int)(col % 16);
                      long long q = std::llrint(std::ldexp(acc, fexp[(size_t)i]));
                      for (int l = 0; l < 7; ++l) {
                        const long long dg = (l < 6) ? ((((q + 128) % 256) + 256) % 256) - 128 : q;
                        if (flat) {  // flat column 24 l + col: tile and lane column of it
                          const int fq = GNX_LR_FLAT_COLS * l + (int)col;
                          V2[(((r * (size_t)EPR + (size_t)k) * NTB + (size_t)(fq >> 4)) * 64 + (size_t)(kq * 16 + (fq & 15))) * 16 + (size_t)t] = (int8_t)dg;
                        } else
                        V2[((((r * (size_t)EPR + (size_t)k) * NT2 + (size_t)nt) * 7 + (size_t)l) * 64 + (size_t)(kq * 16 + c16)) * 16 + (size_t)t] = (int8_t)dg;
                        q = (q - dg) / 256;
                      }
                    }
                  }
                }
        };
        if (!ph) parallel_ranges(n_runs, 64, [&](size_t lo, size_t hi, unsigned) { fill_runs(lo, hi); });
        if (ph) rc = dev_upload_raw(m, blob_v2, (size_t)ph->v2_bytes, flat ? &m->lr.V2F : &m->lr.V2, 64);
        else rc = gnx_dev_upload(m, V2, flat ? &m->lr.V2F : &m->lr.V2, 64);
        if (rc != GNX_OK) return rc;
        if ((rc = gnx_dev_upload(m, run_byte, &m->lr.run_byte)) != GNX_OK) return rc;
        if ((rc = gnx_dev_upload(m, run_flush0, &m->lr.run_flush0)) != GNX_OK) return rc;
        if ((rc = gnx_dev_upload(m, run_nflush, &m->lr.run_nflush)) != GNX_OK) return rc;
        if ((rc = gnx_dev_upload(m, win_run0, &m->lr.win_run0)) != GNX_OK) return rc;
        if ((rc = gnx_dev_upload(m, win_run1, &m->lr.win_run1)) != GNX_OK) return rc;
        m->lr_h_win_run0 = win_run0;
        m->lr_h_win_run1 = win_run1;
        m->lr.n_runs = (int32_t)n_runs;
        m->lr.NT2 = NT2;
        m->lr.EPR = EPR;
      }
    }
  } else if ((rc = gnx_dev_upload(m, V, &m->lr.V)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, icpt, &m->lr.icpt)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, chunk_j0, &m->lr.chunk_j0)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, chunk_flush0, &m->lr.chunk_flush0)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, chunk_nflush, &m->lr.chunk_nflush)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, win_chunk0, &m->lr.win_chunk0)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, win_chunk1, &m->lr.win_chunk1)) != GNX_OK) return rc;
  m->lr_h_win_chunk0 = win_chunk0;
  m->lr_h_win_chunk1 = win_chunk1;
  m->lr.n_chunks = (int32_t)n_chunks;
  {
    int32_t mx = 1;
    for (size_t k = 0; k < n_pieces; ++k) mx = std::max(mx, piece_chunk0[k + 1] - piece_chunk0[k]);
    m->lr.max_piece_chunks = mx;
  }
  m->lr.R = (int32_t)R;
  m->lr.NT = NT;
  return GNX_OK;
}

// ------------------------------------------------------------------------------------------------
// model preparation: xgboost-schema trees -> class-major complete heaps
// ------------------------------------------------------------------------------------------------
// Depth of one tree given as child arrays (-1 at leaves), or -1 when it is not a tree: child index out of range, a node
// reachable twice (cycle / shared subtree: the visit count is bounded by the node count, so a malformed input costs
// O(n_nodes), not 2^depth) or deeper than 64.  Iterative: nothing here recurses on caller-supplied data.
static int checked_tree_depth(const int32_t* left, const int32_t* right, int32_t n_nodes) {
  if (n_nodes <= 0) return -1;
  std::vector<uint8_t> seen((size_t)n_nodes, 0);
  std::vector<std::pair<int32_t, int32_t>> stack{{0, 0}};
  int depth = 0;
  while (!stack.empty()) {
    const auto [nid, dep] = stack.back();
    stack.pop_back();
    if (nid < 0 || nid >= n_nodes || seen[(size_t)nid] || dep > 64) return -1;
    seen[(size_t)nid] = 1;
    depth = std::max(depth, dep);
    const int32_t l = left[nid], r = right[nid];
    if (l == -1) continue;  // leaf (xgboost / sklearn mark leaves by left == -1)
    stack.push_back({l, dep + 1});
    stack.push_back({r, dep + 1});
  }
  return depth;
}

// node offsets of concatenated trees: start at 0, strictly increasing, end at the caller's node count when given
static bool checked_tree_offsets(const int32_t* off, int32_t n_trees, int32_t n_nodes) {
  if (!off || off[0] != 0) return false;
  for (int32_t t = 0; t < n_trees; ++t)
    if (off[t + 1] <= off[t]) return false;
  return n_nodes <= 0 || off[n_trees] == n_nodes;
}

// subtree rooted at xgboost node `nid` (or a replicated early leaf) -> heap slot j of the packed layout
static void tree_fill(const gnx_model_desc* d, int32_t o, int32_t nid, uint32_t j, int depth, int D, uint8_t* out) {
  const uint32_t half = 1u << (D - 1);
  const bool leaf = d->left[o + nid] == -1;
  const float inf = std::numeric_limits<float>::infinity();
  if (depth == D - 1) {  // last split level: 16-byte node carrying both leaves
    uint32_t foff = 0;
    float thr = inf, ll, lr;
    if (leaf) { ll = lr = d->cond[o + nid]; }  // early leaf: dummy split, both sides the leaf value
    else {
      foff = (uint32_t)d->feat[o + nid] * 4u;
      thr = d->cond[o + nid];
      ll = d->cond[o + d->left[o + nid]];
      lr = d->cond[o + d->right[o + nid]];
    }
    uint8_t* p = out + (size_t)(j - half) * 16;
    std::memcpy(p, &foff, 4); std::memcpy(p + 4, &thr, 4); std::memcpy(p + 8, &ll, 4); std::memcpy(p + 12, &lr, 4);
    return;
  }
  uint32_t foff = 0;
  float thr = inf;  // early leaf: always go left (f < +inf), both subtrees replicate the leaf
  if (!leaf) { foff = (uint32_t)d->feat[o + nid] * 4u; thr = d->cond[o + nid]; }
  uint8_t* p = out + (size_t)half * 16 + (size_t)(j - 1) * 8;
  std::memcpy(p, &foff, 4); std::memcpy(p + 4, &thr, 4);
  tree_fill(d, o, leaf ? nid : d->left[o + nid], 2 * j, depth + 1, D, out);
  tree_fill(d, o, leaf ? nid : d->right[o + nid], 2 * j + 1, depth + 1, D, out);
}

// rank-quantised tree (layout in gnx_internal.h: SmoothXGBDev::rk_packed)
static void tree_fill_rk(const gnx_model_desc* d, int32_t o, int32_t nid, uint32_t j, int depth, int D,
                         const std::vector<float>& U, int stride, uint32_t* nodes, float* leaves) {
  const bool leaf = d->left[o + nid] == -1;
  if (depth == D) {
    leaves[j - (1u << D)] = d->cond[o + nid];
    return;
  }
  uint32_t word = 0xFFFFu << 16;  // early leaf: every rank is < 0xFFFF -> left; both subtrees replicate the leaf
  if (!leaf) {
    const float thr = d->cond[o + nid];
    uint32_t field;
    if (thr != thr || thr == -std::numeric_limits<float>::infinity()) field = 0;           // p < thr never holds
    else if (thr == std::numeric_limits<float>::infinity()) field = 0xFFFFu;                // always holds
    else field = (uint32_t)(std::lower_bound(U.begin(), U.end(), thr) - U.begin()) + 1u;    // p < U[k] <=> rank(p) < k+1
    const int f = d->feat[o + nid], A = d->A;
    const uint32_t off = (uint32_t)(((f % A) * stride + f / A) * 2);
    word = (field << 16) | off;
  }
  nodes[j] = word;
  tree_fill_rk(d, o, leaf ? nid : d->left[o + nid], 2 * j, depth + 1, D, U, stride, nodes, leaves);
  tree_fill_rk(d, o, leaf ? nid : d->right[o + nid], 2 * j + 1, depth + 1, D, U, stride, nodes, leaves);
}

static int build_xgb_rk(gnx_model* m, const gnx_model_desc* d, const std::vector<int32_t>& order, int D) {
  const char* impl = std::getenv("GNX_SMOOTH_IMPL");  // default: 16-bit ranks with pointer nodes; "rk" heap-index nodes, "h64" lane = haplotype, "f32" float features
  if (impl && std::string(impl) == "f32") return GNX_OK;
  const int A = d->A, S = d->S;
  std::vector<float> U;
  for (int t = 0; t < d->n_trees; ++t)
    for (int32_t k = d->tree_off[t]; k < d->tree_off[t + 1]; ++k)
      if (d->left[k] != -1 && std::isfinite(d->cond[k])) U.push_back(d->cond[k]);
  std::sort(U.begin(), U.end());
  U.erase(std::unique(U.begin(), U.end()), U.end());
  // segments per strip: 2-4 independent walks per lane are enough to cover the LDS latency and keep the strips small
  // (measured on chr22: 3 -> 1.73 ms, 2 -> 1.76, 6 -> 1.85); among those the split of the chromosome that wastes the
  // fewest 64-window segments
  const int nseg = (int)((d->C / d->M + 63) / 64);
  int rpl = 1;
  if (nseg >= 2) {
    int best_waste = 1 << 30;
    for (int r : {3, 2, 4}) {
      const int waste = (nseg + r - 1) / r * r - nseg;
      if (waste < best_waste) { best_waste = waste; rpl = r; }
    }
  }
  if (const char* e = std::getenv("GNX_RK_RPL")) rpl = std::max(1, std::min(GNX_RK_RPL_MAX, std::atoi(e)));
  int stride = rpl * 64 + S - 1;
  stride += stride & 1;
  if (U.size() > 65000 || (size_t)A * stride * 2 > 65535) return GNX_OK;  // does not fit 16 bits: float kernel only
  if (U.empty()) U.push_back(0.5f);
  const int K = (int)U.size();
  // bucket b covers [b/1024, (b+1)/1024) (bucket 0 also everything below, bucket 1023 everything above):
  // rank(p) lies in [#{U < lower edge}, #{U < upper edge}]
  std::vector<uint32_t> lut(1024);
  int steps = 0;
  for (int b = 0; b < 1024; ++b) {
    const int lo = b == 0 ? 0 : (int)(std::lower_bound(U.begin(), U.end(), (float)b / 1024.0f) - U.begin());
    const int hi = b == 1023 ? K : (int)(std::lower_bound(U.begin(), U.end(), (float)(b + 1) / 1024.0f) - U.begin());
    lut[(size_t)b] = (uint32_t)lo | ((uint32_t)hi << 16);
    int st = 0;
    while ((1 << st) < hi - lo + 1) ++st;
    steps = std::max(steps, st);
  }
  const int tree_bytes = 8 << D;
  int group_bytes = 4096;  // per staging buffer; two of them per block
  if (const char* e = std::getenv("GNX_RK_GROUP_BYTES")) group_bytes = std::max(tree_bytes, std::min(8192, std::atoi(e)));
  const int G = std::max(1, group_bytes / tree_bytes);
  std::vector<int32_t> group_tree0, group_class;
  {
    int in_group = 0, cur = -1;
    for (size_t k = 0; k < order.size(); ++k) {
      const int c = d->tree_class[order[k]];
      if (c != cur || in_group == G) { group_tree0.push_back((int32_t)k); group_class.push_back(c); in_group = 0; cur = c; }
      ++in_group;
    }
    group_tree0.push_back((int32_t)order.size());
  }
  std::vector<uint8_t> packed(order.size() * (size_t)tree_bytes, 0);
  for (size_t k = 0; k < order.size(); ++k) {
    uint8_t* tb = packed.data() + k * tree_bytes;
    tree_fill_rk(d, d->tree_off[order[k]], 0, 1, 0, D, U, stride, reinterpret_cast<uint32_t*>(tb),
                 reinterpret_cast<float*>(tb + ((size_t)4 << D)));
  }
  int rc;
  if ((rc = gnx_dev_upload(m, packed, &m->xgb.rk_packed, 64)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, U, &m->xgb.rk_thr, 64)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, lut, &m->xgb.rk_lut)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, group_tree0, &m->xgb.rk_group_tree0)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, group_class, &m->xgb.rk_group_class)) != GNX_OK) return rc;
  m->xgb.rk_K = K; m->xgb.rk_steps = steps; m->xgb.rk_stride = stride; m->xgb.rk_tree_bytes = tree_bytes;
  m->xgb.rk_n_groups = (int32_t)group_class.size(); m->xgb.rk_max_group = G; m->xgb.rk_rpl = rpl;
  // measured (chr22, 10 000 haplotypes, MI355X): rk 1.83 ms; h64 2.22 ms + 0.23 ms of rank pre-pass — conflict-free, and slower
  // (DESIGN.md 4.2b): the rank kernel stays the default, h64 is what GNX_SMOOTH_IMPL=h64 selects
  // (round 3, same inputs: rk 1.83 ms, pointer nodes 1.78 ms, bit-identical: default where the shape allows, "rk" = heap-index nodes)
#ifdef GNX_EXPERIMENTS
  m->xgb.impl = (impl && std::string(impl) == "h64") ? 2 : (impl && std::string(impl) == "rk") ? 1 : 3;
#else
  m->xgb.impl = (impl && std::string(impl) == "rk") ? 1 : 3;
#endif
  // ---- the same trees for k_gnofix: heap-index nodes whose offsets address a [class][pitch] tile of 2S+2 window positions ----
  if ((size_t)A * (2 * S + 2) * 2 <= 65535) {
    const int pitch = 2 * S + 2;
    const size_t tw = (size_t)gnx_gf_tree_words(D);
    std::vector<uint32_t> gf((order.size() + GNX_GF_PAD_TREES) * tw, 0u);
    for (size_t k = 0; k < order.size(); ++k) {
      uint32_t* tb = gf.data() + k * tw;
      tree_fill_rk(d, d->tree_off[order[k]], 0, 1, 0, D, U, pitch, tb, reinterpret_cast<float*>(tb + ((size_t)1 << D)));
    }
    if ((rc = gnx_dev_upload(m, gf, &m->xgb.gf_packed, 64)) != GNX_OK) return rc;
    std::vector<int> cnt(A, 0);
    int mc = 0;
    for (size_t k = 0; k < order.size(); ++k) mc = std::max(mc, ++cnt[d->tree_class[order[k]]]);
    m->xgb.gf_pitch = pitch; m->xgb.gf_max_class = mc;
  }
  // ---- the same trees with pointer nodes (k_smooth_xgb_rk<.., PTR>): the walk keeps the ADDRESS of its node, no heap index ---
  if (D >= 2 && D <= 6) {
    const int tbp = 12 << D;
    const int Gp = std::max(2, (4096 / tbp) & ~1);  // even: trees are walked in pairs
    std::vector<int32_t> gp_tree0, gp_class;
    {
      int in_group = 0, cur = -1;
      for (size_t k = 0; k < order.size(); ++k) {
        const int c = d->tree_class[order[k]];
        if (c != cur || in_group == Gp) { gp_tree0.push_back((int32_t)k); gp_class.push_back(c); in_group = 0; cur = c; }
        ++in_group;
      }
      gp_tree0.push_back((int32_t)order.size());
    }
    std::vector<uint8_t> pp(order.size() * (size_t)tbp, 0);
    std::vector<uint32_t> nodes((size_t)1 << D);
    std::vector<float> leaves((size_t)1 << D);
    size_t g = 0;
    for (size_t k = 0; k < order.size(); ++k) {
      while ((size_t)gp_tree0[g + 1] <= k) ++g;
      const uint32_t base = (uint32_t)(k - (size_t)gp_tree0[g]) * (uint32_t)tbp;  // the tree's first byte inside its group
      std::fill(nodes.begin(), nodes.end(), 0u);
      tree_fill_rk(d, d->tree_off[order[k]], 0, 1, 0, D, U, stride, nodes.data(), leaves.data());
      uint32_t* o = reinterpret_cast<uint32_t*>(pp.data() + k * tbp);
      auto kids = [&](uint32_t j) {
        const uint32_t l = 2 * j < (1u << D) ? base + 8u * (2 * j) : base + (8u << D) + 4u * (2 * j - (1u << D));
        const uint32_t step = 2 * j < (1u << D) ? 8u : 4u;
        return l | ((l + step) << 16);
      };
      for (uint32_t j = 2; j < (1u << D); ++j) { o[2 * j] = nodes[j]; o[2 * j + 1] = kids(j); }
      o[0] = nodes[2]; o[1] = nodes[3]; o[2] = nodes[1]; o[3] = kids(2);
      memcpy(pp.data() + k * tbp + ((size_t)8 << D), leaves.data(), sizeof(float) << D);
    }
    if ((rc = gnx_dev_upload(m, pp, &m->xgb.rp_packed, 64)) != GNX_OK) return rc;
    if ((rc = gnx_dev_upload(m, gp_tree0, &m->xgb.rp_group_tree0)) != GNX_OK) return rc;
    if ((rc = gnx_dev_upload(m, gp_class, &m->xgb.rp_group_class)) != GNX_OK) return rc;
    m->xgb.rp_tree_bytes = tbp; m->xgb.rp_n_groups = (int32_t)gp_class.size(); m->xgb.rp_max_group = Gp;
  }
#ifdef GNX_EXPERIMENTS
  // ---- the same trees for k_smooth_xgb_h64 (lane = haplotype): pointer nodes whose w0 carries the feature's SLOT s * A + a --------
  if (D >= 2 && D <= 6 && (size_t)S * A < 65536) {
    const int tb8 = 12 << D;
    // a staging group = as many trees of one class as the LDS holds beside the 16-wave strip (two buffers): one block per CU means
    // nothing covers a block barrier, so there should be few of them (chr22 / A = 7: a class = 100 trees = one group, 7 barriers
    // instead of 35); a multiple of the trees walked side by side, at most 32 KB (staging registers)
    int G8 = std::max(4, (4096 / tb8) & ~3);
    {
      const size_t strip16 = (size_t)(16 * 3 + S - 1) * A * 128;
      if (strip16 + 2 * (size_t)G8 * tb8 <= (size_t)160 * 1024) {
        int per_class = 0;
        std::vector<int> cnt(A, 0);
        for (size_t k = 0; k < order.size(); ++k) per_class = std::max(per_class, ++cnt[d->tree_class[order[k]]]);
        const int fit = (int)((((size_t)160 * 1024 - strip16) / 2) / tb8) & ~3;
        G8 = std::max(G8, std::min({fit, (32768 / tb8) & ~3, (per_class + 3) & ~3}));
      }
    }
    std::vector<int32_t> g8_tree0, g8_class;
    {
      int in_group = 0, cur = -1;
      for (size_t k = 0; k < order.size(); ++k) {
        const int c = d->tree_class[order[k]];
        if (c != cur || in_group == G8) { g8_tree0.push_back((int32_t)k); g8_class.push_back(c); in_group = 0; cur = c; }
        ++in_group;
      }
      g8_tree0.push_back((int32_t)order.size());
    }
    std::vector<uint8_t> p8(order.size() * (size_t)tb8, 0);
    std::vector<uint32_t> nodes((size_t)1 << D);
    std::vector<float> leaves((size_t)1 << D);
    size_t g = 0;
    for (size_t k = 0; k < order.size(); ++k) {
      while ((size_t)g8_tree0[g + 1] <= k) ++g;
      const uint32_t base = (uint32_t)(k - (size_t)g8_tree0[g]) * (uint32_t)tb8;
      std::fill(nodes.begin(), nodes.end(), 0u);
      tree_fill_rk(d, d->tree_off[order[k]], 0, 1, 0, D, U, stride, nodes.data(), leaves.data());  // (field << 16) | strip offset
      for (uint32_t j = 1; j < (1u << D); ++j) {
        // the rank layout's byte offset (a * stride + s) * 2 back to (s, a); an early leaf's word (offset 0) reads slot 0
        const uint32_t h = (nodes[j] & 0xffffu) / 2, a = h / (uint32_t)stride, sidx = h - a * (uint32_t)stride;
        nodes[j] = (nodes[j] & 0xffff0000u) | (sidx * (uint32_t)A + a);
      }
      uint32_t* o = reinterpret_cast<uint32_t*>(p8.data() + k * tb8);
      auto kids = [&](uint32_t j) {
        const uint32_t l = 2 * j < (1u << D) ? base + 8u * (2 * j) : base + (8u << D) + 4u * (2 * j - (1u << D));
        const uint32_t step = 2 * j < (1u << D) ? 8u : 4u;
        return l | ((l + step) << 16);
      };
      for (uint32_t j = 2; j < (1u << D); ++j) { o[2 * j] = nodes[j]; o[2 * j + 1] = kids(j); }
      o[0] = nodes[2]; o[1] = nodes[3]; o[2] = nodes[1]; o[3] = kids(2);
      memcpy(p8.data() + k * tb8 + ((size_t)8 << D), leaves.data(), sizeof(float) << D);
    }
    if ((rc = gnx_dev_upload(m, p8, &m->xgb.h8_packed, 64)) != GNX_OK) return rc;
    if ((rc = gnx_dev_upload(m, g8_tree0, &m->xgb.h8_group_tree0)) != GNX_OK) return rc;
    if ((rc = gnx_dev_upload(m, g8_class, &m->xgb.h8_group_class)) != GNX_OK) return rc;
    m->xgb.h8_tree_bytes = tb8; m->xgb.h8_n_groups = (int32_t)g8_class.size(); m->xgb.h8_max_group = G8;
  }
#endif
  return GNX_OK;
}

#ifdef GNX_EXPERIMENTS  // the bit-sliced smoother is parked (scripts/dev/rejected/k_smooth_xgb_bs.hip): its tables are built in that build only
// bit-sliced trees (layout in gnx_internal.h: SmoothXGBDev::bs_nodes); Uc = per-class sorted thresholds, Y = the kernel's LDS map
static void tree_fill_bs(const gnx_model_desc* d, int32_t o, int32_t nid, uint32_t j, int depth, const std::vector<std::vector<float>>& Uc,
                         const std::vector<int32_t>& binoff, const GnxBsLayout& Y, uint32_t* words, float* leaves) {
  const bool leaf = d->left[o + nid] == -1;
  if (depth == 4) {
    leaves[j - 16u] = d->cond[o + nid];
    return;
  }
  // early leaf: both subtrees replicate the leaf, any node will do (class 0, rank field 0: row 0 = "every window goes right")
  uint32_t a = 0, kf = 0, sw = 0;
  if (!leaf) {
    const float thr = d->cond[o + nid];
    const int f = d->feat[o + nid], A = d->A;
    a = (uint32_t)(f % A);
    sw = (uint32_t)(f / A);
    const std::vector<float>& U = Uc[a];
    if (thr != thr || thr == -std::numeric_limits<float>::infinity()) kf = 0;                       // p < thr never holds
    else if (thr == std::numeric_limits<float>::infinity()) kf = (uint32_t)U.size() + 1u;          // holds unless p is NaN
    else kf = (uint32_t)(std::lower_bound(U.begin(), U.end(), thr) - U.begin()) + 1u;              // p < U[k] <=> rank(p) < k+1
  }
  const uint32_t cnt_addr = (uint32_t)Y.off_cnt + (uint32_t)binoff[a] + kf;
  words[2 * j] = (sw & 31u) | (cnt_addr << 16);
  words[2 * j + 1] = (uint32_t)Y.off_P + a * (uint32_t)(Y.nr * Y.rb) + (sw >> 5) * 4u;
  tree_fill_bs(d, o, leaf ? nid : d->left[o + nid], 2 * j, depth + 1, Uc, binoff, Y, words, leaves);
  tree_fill_bs(d, o, leaf ? nid : d->right[o + nid], 2 * j + 1, depth + 1, Uc, binoff, Y, words, leaves);
}

// k_smooth_xgb_bs: ensembles of depth <= 4 whose padded chunk fits byte counters and whose tables fit the LDS
static int build_xgb_bs(gnx_model* m, const gnx_model_desc* d, const std::vector<int32_t>& order, int D) {
  const char* impl = std::getenv("GNX_SMOOTH_IMPL");
  if (impl && std::string(impl) == "f32") return GNX_OK;
  const int A = d->A, S = d->S;
  if (D > 4 || A > 16) return GNX_OK;
  std::vector<std::vector<float>> Uc((size_t)A);
  for (int t = 0; t < d->n_trees; ++t)
    for (int32_t k = d->tree_off[t]; k < d->tree_off[t + 1]; ++k)
      if (d->left[k] != -1 && std::isfinite(d->cond[k])) Uc[(size_t)(d->feat[k] % A)].push_back(d->cond[k]);
  std::vector<int32_t> uoff((size_t)A + 1, 0), binoff((size_t)A + 1, 0);
  std::vector<float> U;
  for (int c = 0; c < A; ++c) {
    std::vector<float>& u = Uc[(size_t)c];
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    if (u.empty()) u.push_back(0.5f);
    if (u.size() > 60000) return GNX_OK;
    U.insert(U.end(), u.begin(), u.end());
    uoff[(size_t)c + 1] = (int32_t)U.size();
    // ranks 0..K_c and the NaN bin (the class's LAST counter), padded to 16 bytes: a wave scans its class's counters as words
    binoff[(size_t)c + 1] = binoff[(size_t)c] + (((int32_t)u.size() + 2 + 15) & ~15);
  }
  const int nbins = binoff[(size_t)A];
  if (nbins > 65000) return GNX_OK;
  int wc = 128;  // windows per chunk; counters are bytes: the padded chunk must stay below 256 windows
  if (wc + S - 1 > 255 || gnx_bs_layout(A, S, wc, nbins).total > 160 * 1024) return GNX_OK;
  const GnxBsLayout Y = gnx_bs_layout(A, S, wc, nbins);
  std::vector<uint32_t> lut((size_t)A * 1024);
  int steps = 0;
  for (int c = 0; c < A; ++c) {
    const std::vector<float>& u = Uc[(size_t)c];
    const int K = (int)u.size();
    for (int b = 0; b < 1024; ++b) {
      const int lo = b == 0 ? 0 : (int)(std::lower_bound(u.begin(), u.end(), (float)b / 1024.0f) - u.begin());
      const int hi = b == 1023 ? K : (int)(std::lower_bound(u.begin(), u.end(), (float)(b + 1) / 1024.0f) - u.begin());
      lut[(size_t)c * 1024 + (size_t)b] = (uint32_t)lo | ((uint32_t)hi << 16);
      int st = 0;
      while ((1 << st) < hi - lo + 1) ++st;
      steps = std::max(steps, st);
    }
  }
  std::vector<uint32_t> nodes(order.size() * 32, 0u);
  std::vector<float> leaves(order.size() * 16, 0.f);
  for (size_t k = 0; k < order.size(); ++k)
    tree_fill_bs(d, d->tree_off[order[k]], 0, 1, 0, Uc, binoff, Y, nodes.data() + k * 32, leaves.data() + k * 16);
  std::vector<int32_t> ct0((size_t)A + 1, 0);
  for (int t = 0; t < d->n_trees; ++t) ct0[(size_t)d->tree_class[t] + 1] += 1;
  for (int c = 0; c < A; ++c) ct0[(size_t)c + 1] += ct0[(size_t)c];
  int rc;
  if ((rc = gnx_dev_upload(m, nodes, &m->xgb.bs_nodes, 128)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, leaves, &m->xgb.bs_leaves, 64)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, U, &m->xgb.bs_thr, 64)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, lut, &m->xgb.bs_lut)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, uoff, &m->xgb.bs_uoff)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, binoff, &m->xgb.bs_binoff)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, ct0, &m->xgb.bs_class_tree0)) != GNX_OK) return rc;
  m->xgb.bs_steps = steps; m->xgb.bs_nbins = nbins; m->xgb.bs_wc = wc; m->xgb.bs_nthr = (int32_t)U.size();
  for (int c = 0; c < A; ++c) m->xgb.bs_maxbins = std::max(m->xgb.bs_maxbins, binoff[(size_t)c + 1] - binoff[(size_t)c]);
  if (impl && std::string(impl) == "bs") m->xgb.impl = 4;
  return GNX_OK;
}
#endif  // GNX_EXPERIMENTS

int gnx_build_xgb(gnx_model* m, const gnx_model_desc* d) {
  gnx_ctx* ctx = m->ctx;
  const int A = d->A, S = d->S, F = S * A;
  if (d->n_trees <= 0 || !d->tree_off || !d->left || !d->right || !d->feat || !d->cond || !d->tree_class)
    return fail(ctx, GNX_EINVAL, "xgb smoother: tree arrays missing");
  const int64_t W = d->C / d->M;
  if (W < 2 * (int64_t)S)  // src/Smooth/models.py:13
    return fail(ctx, GNX_EINVAL, "Smoother size to large for given window size. ");
  if (!checked_tree_offsets(d->tree_off, d->n_trees, d->n_nodes))
    return fail(ctx, GNX_EINVAL, "xgb smoother: tree_off must start at 0, increase strictly and end at n_nodes");
  int D = 1;
  for (int t = 0; t < d->n_trees; ++t) {
    const int32_t o = d->tree_off[t], nn = d->tree_off[t + 1] - o;
    const int dep = checked_tree_depth(d->left + o, d->right + o, nn);
    if (dep < 0) return fail(ctx, GNX_EINVAL, "xgb smoother: malformed tree (child index out of range, node reachable twice or depth > 64)");
    D = std::max(D, dep);
    if (d->tree_class[t] < 0 || d->tree_class[t] >= A) return fail(ctx, GNX_EINVAL, "xgb smoother: tree_class out of range");
    for (int32_t k = 0; k < nn; ++k)
      if (d->left[o + k] != -1 && (d->feat[o + k] < 0 || d->feat[o + k] >= F))
        return fail(ctx, GNX_EINVAL, "xgb smoother: split feature outside the S*A sliding window");
  }
  if (D > 8) return fail(ctx, GNX_EUNSUPPORTED, "xgb smoother: tree depth > 8");
  const int tree_bytes = gnx_tree_bytes(D);
  const int G = std::max(1, std::min(24, 16384 / tree_bytes));

  std::vector<int32_t> order;
  order.reserve((size_t)d->n_trees);
  std::vector<int32_t> group_tree0, group_class;
  for (int c = 0; c < A; ++c) {
    int in_group = 0;
    for (int t = 0; t < d->n_trees; ++t) {
      if (d->tree_class[t] != c) continue;
      if (in_group == 0) { group_tree0.push_back((int32_t)order.size()); group_class.push_back(c); }
      order.push_back(t);
      if (++in_group == G) in_group = 0;
    }
  }
  group_tree0.push_back((int32_t)order.size());
  std::vector<int32_t> class_tree0((size_t)A + 1, 0);
  for (int t = 0; t < d->n_trees; ++t) class_tree0[(size_t)d->tree_class[t] + 1] += 1;
  for (int c = 0; c < A; ++c) class_tree0[(size_t)c + 1] += class_tree0[(size_t)c];
  std::vector<uint8_t> packed((size_t)d->n_trees * tree_bytes, 0);
  for (size_t k = 0; k < order.size(); ++k)
    tree_fill(d, d->tree_off[order[k]], 0, 1, 0, D, packed.data() + k * tree_bytes);

  int rc;
  if ((rc = gnx_dev_upload(m, packed, &m->xgb.packed, 64)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, group_tree0, &m->xgb.group_tree0)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, group_class, &m->xgb.group_class)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, class_tree0, &m->class_tree0)) != GNX_OK) return rc;
  m->xgb.n_groups = (int32_t)group_class.size();
  m->xgb.n_trees = d->n_trees;
  m->xgb.D = D;
  m->xgb.tree_bytes = tree_bytes;
  m->xgb.max_group = G;
  m->xgb.base_score = d->base_score;
  m->info.n_trees = d->n_trees;
  m->info.tree_depth = D;
  if ((rc = build_xgb_rk(m, d, order, D)) != GNX_OK) return rc;
#ifdef GNX_EXPERIMENTS
  if ((rc = build_xgb_bs(m, d, order, D)) != GNX_OK) return rc;
#endif
  return GNX_OK;
}

// ------------------------------------------------------------------------------------------------
// model preparation: forest base — per-window xgboost-schema trees -> per-window class-major complete heaps
// ------------------------------------------------------------------------------------------------
// Forest trees use their own compact heap: 2^D node words (slot 0 unused) followed by 2^D float leaves.  A node word is
// (SNP index within the window << 4) | left-mask, bit v of the mask = "a SNP of value v goes left": SNPs only take
// the values 0..3, so `float(v) < threshold` and the missing code's default direction fold into 4 bits at load time.
static void forest_fill(const gnx_model_desc* d, int32_t o, int32_t nid, uint32_t j, int depth, int D, uint32_t off, uint32_t* nodes,
                        float* leaves) {
  const bool leaf = d->fb_left[o + nid] == -1;
  if (depth == D) {  // D is the ensemble's maximum depth: nid is a leaf here
    leaves[j - (1u << D)] = d->fb_cond[o + nid];
    return;
  }
  uint32_t word = 0xFu;  // early leaf: every value goes left, both subtrees replicate the leaf
  if (!leaf) {
    const float thr = d->fb_cond[o + nid];
    const bool dl = d->fb_default_left && d->fb_default_left[o + nid];
    uint32_t mask = 0;
    for (int v = 0; v < 4; ++v) {
      const bool left = (v == d->fb_missing) ? dl : ((float)v < thr);
      mask |= (left ? 1u : 0u) << v;
    }
    word = (((uint32_t)d->fb_feat[o + nid] + off) << 4) | mask;  // off = window start mod 16 (the tile's words are anchored globally)
  }
  nodes[j] = word;
  forest_fill(d, o, leaf ? nid : d->fb_left[o + nid], 2 * j, depth + 1, D, off, nodes, leaves);
  forest_fill(d, o, leaf ? nid : d->fb_right[o + nid], 2 * j + 1, depth + 1, D, off, nodes, leaves);
}

int gnx_build_forest(gnx_model* m, const gnx_model_desc* d) {
  gnx_ctx* ctx = m->ctx;
  const int A = d->A;
  const int64_t C = d->C, M = d->M, W = C / M, rem = C - M * W, M_ = M + 2 * d->ctx;
  if (d->fb_n_trees <= 0 || !d->fb_win_tree0 || !d->fb_tree_off || !d->fb_left || !d->fb_right || !d->fb_feat || !d->fb_cond)
    return fail(ctx, GNX_EINVAL, "forest base: tree arrays missing");
  if (A > 2 && !d->fb_tree_class) return fail(ctx, GNX_EINVAL, "forest base: fb_tree_class is NULL");
  if (d->fb_missing < 0 || d->fb_missing > 3) return fail(ctx, GNX_EINVAL, "forest base: missing code must be in [0, 3]");
  if (!(d->fb_base_score > 0.f && d->fb_base_score < 1.f) && A == 2)
    return fail(ctx, GNX_EINVAL, "forest base: binary:logistic needs base_score in (0, 1)");
  if (d->fb_win_tree0[0] != 0 || d->fb_win_tree0[W] != d->fb_n_trees)
    return fail(ctx, GNX_EINVAL, "forest base: fb_win_tree0 must run from 0 to fb_n_trees");
  for (int64_t w = 0; w < W; ++w)
    if (d->fb_win_tree0[w + 1] < d->fb_win_tree0[w]) return fail(ctx, GNX_EINVAL, "forest base: fb_win_tree0 not monotone");
  if (!checked_tree_offsets(d->fb_tree_off, d->fb_n_trees, d->fb_n_nodes))
    return fail(ctx, GNX_EINVAL, "forest base: fb_tree_off must start at 0, increase strictly and end at fb_n_nodes");
  int D = 1, max_trees = 0;
  for (int64_t w = 0; w < W; ++w) {
    const int32_t t0 = d->fb_win_tree0[w], t1 = d->fb_win_tree0[w + 1];
    max_trees = std::max(max_trees, t1 - t0);
    const int64_t width = (w == W - 1) ? M_ + rem : M_;
    for (int32_t t = t0; t < t1; ++t) {
      const int32_t o = d->fb_tree_off[t], nn = d->fb_tree_off[t + 1] - o;
      const int dep = checked_tree_depth(d->fb_left + o, d->fb_right + o, nn);
      if (dep < 0) return fail(ctx, GNX_EINVAL, "forest base: malformed tree (child index out of range, node reachable twice or depth > 64)");
      D = std::max(D, dep);
      if (A > 2 && (d->fb_tree_class[t] < 0 || d->fb_tree_class[t] >= A))
        return fail(ctx, GNX_EINVAL, "forest base: fb_tree_class out of range");
      for (int32_t k = 0; k < nn; ++k)
        if (d->fb_left[o + k] != -1 && (d->fb_feat[o + k] < 0 || d->fb_feat[o + k] >= width))
          return fail(ctx, GNX_EINVAL, "forest base: split feature outside the window's padded slice");
    }
  }
  if (D > 8) return fail(ctx, GNX_EUNSUPPORTED, "forest base: tree depth > 8");
  if (C < 16) return fail(ctx, GNX_EUNSUPPORTED, "forest base: fewer than 16 SNPs");
  const int tree_bytes = 8 << D;
  const int max_words = gnx_forest_ring_words(M_ + rem);
  if (gnx_forest_lds_bytes(A, max_words, max_trees, tree_bytes, 64) > (size_t)160 * 1024)
    return fail(ctx, GNX_EUNSUPPORTED, "forest base: one window's trees and SNPs exceed the 160 KB LDS");

  std::vector<uint8_t> packed((size_t)d->fb_n_trees * tree_bytes, 0);
  std::vector<int32_t> win_tree0(d->fb_win_tree0, d->fb_win_tree0 + W + 1);
  std::vector<int32_t> wct((size_t)W * (A + 1), 0);
  size_t k = 0;
  for (int64_t w = 0; w < W; ++w) {
    const int32_t t0 = win_tree0[(size_t)w], t1 = win_tree0[(size_t)w + 1];
    int32_t* ct = wct.data() + (size_t)w * (A + 1);
    for (int c = 0; c < (A == 2 ? 1 : A); ++c) {
      ct[c] = (int32_t)(k - (size_t)t0);
      for (int32_t t = t0; t < t1; ++t) {
        if (A > 2 && d->fb_tree_class[t] != c) continue;
        uint8_t* tb = packed.data() + k * tree_bytes;
        forest_fill(d, d->fb_tree_off[t], 0, 1, 0, D, (uint32_t)((w * M) & 15), reinterpret_cast<uint32_t*>(tb),
                    reinterpret_cast<float*>(tb + ((size_t)4 << D)));
        ++k;
      }
    }
    for (int c = (A == 2 ? 1 : A); c <= A; ++c) ct[c] = t1 - t0;
  }
  // k_base_forest2's node words: the loader words above, baked for their window (first word, ring size) at load time
  std::vector<uint32_t> nodes2((size_t)d->fb_n_trees << D, 0);
  for (int64_t w = 0; w < W; ++w) {
    const uint32_t ring = (uint32_t)gnx_forest_ring_words(w == W - 1 ? M_ + rem : M_), g0 = (uint32_t)((w * M) >> 4);
    for (int32_t t = win_tree0[(size_t)w]; t < win_tree0[(size_t)w + 1]; ++t) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(packed.data() + (size_t)t * tree_bytes);
      for (uint32_t j = 1; j < (1u << D); ++j) nodes2[((size_t)t << D) + j] = gnx_forest2_node(src[j], g0, ring);
    }
  }
  int rc;
  if ((rc = gnx_dev_upload(m, packed, &m->forest.packed, 64)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, nodes2, &m->forest.nodes2, 64)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, win_tree0, &m->forest.win_tree0)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, wct, &m->forest.win_class_tree0)) != GNX_OK) return rc;
  m->forest.D = D; m->forest.tree_bytes = tree_bytes; m->forest.max_trees = max_trees; m->forest.max_words = max_words;
  m->forest.missing = d->fb_missing; m->forest.base_score = d->fb_base_score;
  return GNX_OK;
}

// ------------------------------------------------------------------------------------------------
// model preparation: random-forest base — sklearn tree arrays -> mask-node heaps + expanded leaf rows
// ------------------------------------------------------------------------------------------------
static void rf_fill(const gnx_model_desc* d, int32_t o, int32_t nid, uint32_t j, int depth, int D, uint32_t off, uint32_t* nodes,
                    double* leafval) {
  const bool leaf = d->rf_left[o + nid] == -1;
  if (depth == D) {
    std::memcpy(leafval + (size_t)(j - (1u << D)) * d->A, d->rf_value + (size_t)(o + nid) * d->A, (size_t)d->A * sizeof(double));
    return;
  }
  uint32_t word = 0xFu;  // early leaf: every value goes left, both subtrees replicate the leaf
  if (!leaf) {
    uint32_t mask = 0;
    for (int v = 0; v < 4; ++v) mask |= (((double)(float)v <= d->rf_thr[o + nid]) ? 1u : 0u) << v;  // _tree.pyx: X[i, f] <= threshold
    word = (((uint32_t)d->rf_feat[o + nid] + off) << 4) | mask;
  }
  nodes[j] = word;
  rf_fill(d, o, leaf ? nid : d->rf_left[o + nid], 2 * j, depth + 1, D, off, nodes, leafval);
  rf_fill(d, o, leaf ? nid : d->rf_right[o + nid], 2 * j + 1, depth + 1, D, off, nodes, leafval);
}

int gnx_build_rforest(gnx_model* m, const gnx_model_desc* d) {
  gnx_ctx* ctx = m->ctx;
  const int A = d->A;
  const int64_t C = d->C, M = d->M, W = C / M, rem = C - M * W, M_ = M + 2 * d->ctx;
  if (d->rf_n_trees <= 0 || !d->rf_win_tree0 || !d->rf_tree_off || !d->rf_left || !d->rf_right || !d->rf_feat || !d->rf_thr || !d->rf_value)
    return fail(ctx, GNX_EINVAL, "rforest base: tree arrays missing");
  if (d->rf_win_tree0[0] != 0 || d->rf_win_tree0[W] != d->rf_n_trees)
    return fail(ctx, GNX_EINVAL, "rforest base: rf_win_tree0 must run from 0 to rf_n_trees");
  if (C < 16) return fail(ctx, GNX_EUNSUPPORTED, "rforest base: fewer than 16 SNPs");
  if (!checked_tree_offsets(d->rf_tree_off, d->rf_n_trees, d->rf_n_nodes))
    return fail(ctx, GNX_EINVAL, "rforest base: rf_tree_off must start at 0, increase strictly and end at rf_n_nodes");
  int D = 1, max_trees = 0;
  for (int64_t w = 0; w < W; ++w) {
    const int32_t t0 = d->rf_win_tree0[w], t1 = d->rf_win_tree0[w + 1];
    if (t1 <= t0) return fail(ctx, GNX_EINVAL, "rforest base: every window needs at least one tree");
    max_trees = std::max(max_trees, t1 - t0);
    const int64_t width = (w == W - 1) ? M_ + rem : M_;
    for (int32_t t = t0; t < t1; ++t) {
      const int32_t o = d->rf_tree_off[t], nn = d->rf_tree_off[t + 1] - o;
      const int dep = checked_tree_depth(d->rf_left + o, d->rf_right + o, nn);
      if (dep < 0) return fail(ctx, GNX_EINVAL, "rforest base: malformed tree (child index out of range, node reachable twice or depth > 64)");
      D = std::max(D, dep);
      for (int32_t k = 0; k < nn; ++k)
        if (d->rf_left[o + k] != -1 && (d->rf_feat[o + k] < 0 || d->rf_feat[o + k] >= width))
          return fail(ctx, GNX_EINVAL, "rforest base: split feature outside the window's padded slice");
    }
  }
  if (D > 8) return fail(ctx, GNX_EUNSUPPORTED, "rforest base: tree depth > 8");
  const int tree_bytes = std::max(16, 4 << D);
  const int max_words = gnx_forest_ring_words(M_ + rem);
  if (gnx_forest_lds_bytes(A, max_words, max_trees, tree_bytes, 64) > (size_t)160 * 1024)
    return fail(ctx, GNX_EUNSUPPORTED, "rforest base: one window's trees and SNPs exceed the 160 KB LDS");
  std::vector<uint8_t> packed((size_t)d->rf_n_trees * tree_bytes, 0);
  std::vector<double> leafval((size_t)d->rf_n_trees * ((size_t)1 << D) * A, 0.0);
  for (int64_t w = 0; w < W; ++w)
    for (int32_t t = d->rf_win_tree0[w]; t < d->rf_win_tree0[w + 1]; ++t)
      rf_fill(d, d->rf_tree_off[t], 0, 1, 0, D, (uint32_t)((w * M) & 15), reinterpret_cast<uint32_t*>(packed.data() + (size_t)t * tree_bytes),
              leafval.data() + (size_t)t * ((size_t)1 << D) * A);
  std::vector<int32_t> win_tree0(d->rf_win_tree0, d->rf_win_tree0 + W + 1);
  int rc;
  if ((rc = gnx_dev_upload(m, packed, &m->forest.packed, 64)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, leafval, &m->forest.rf_leafval)) != GNX_OK) return rc;
  {  // k_base_forest2's node words, baked per window as for the boosted-tree base (a record holds nothing but its 2^D node words)
    std::vector<uint32_t> nodes2((size_t)d->rf_n_trees << D, 0);
    for (int64_t w = 0; w < W; ++w) {
      const uint32_t ring = (uint32_t)gnx_forest_ring_words(w == W - 1 ? M_ + rem : M_), g0 = (uint32_t)((w * M) >> 4);
      for (int32_t t = d->rf_win_tree0[w]; t < d->rf_win_tree0[w + 1]; ++t) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(packed.data() + (size_t)t * tree_bytes);
        for (uint32_t j = 1; j < (1u << D); ++j) nodes2[((size_t)t << D) + j] = gnx_forest2_node(src[j], g0, ring);
      }
    }
    if ((rc = gnx_dev_upload(m, nodes2, &m->forest.nodes2, 64)) != GNX_OK) return rc;
  }
  if ((rc = gnx_dev_upload(m, win_tree0, &m->forest.win_tree0)) != GNX_OK) return rc;
  m->forest.D = D; m->forest.tree_bytes = tree_bytes; m->forest.max_trees = max_trees; m->forest.max_words = max_words;
  return GNX_OK;
}

// ------------------------------------------------------------------------------------------------
// model preparation: CovRSK / SVC base — support vectors as bit-planes, run-length table g
// ------------------------------------------------------------------------------------------------
int gnx_build_covrsk(gnx_model* m, const gnx_model_desc* d) {
  gnx_ctx* ctx = m->ctx;
  const int A = d->A, P = A * (A - 1) / 2;
  const int64_t C = d->C, M = d->M, W = C / M, rem = C - M * W, M_ = M + 2 * d->ctx;
  if (!d->svc) return fail(ctx, GNX_EINVAL, "covrsk base: svc array is NULL");
  if (A > 13) return fail(ctx, GNX_EUNSUPPORTED, "covrsk base: more than 13 ancestries");
  std::vector<SvcWinDev> wins((size_t)W);
  std::vector<uint32_t> svbits, gtab;
  std::vector<double> coef;
  std::vector<std::pair<std::vector<int32_t>, int32_t>> gkeys;  // (ms, width) -> offset
  std::vector<int32_t> goffs;
  int max_nw = 0, max_width = 0;
  for (int64_t i = 0; i < W; ++i) {
    const gnx_svc_window& sw = d->svc[i];
    const int64_t width = (i == W - 1) ? M_ + rem : M_;
    if (sw.width != width) return fail(ctx, GNX_EINVAL, "covrsk base: svc[i].width != window width (M+2ctx, +rem for the last)");
    const bool poly = sw.kernel_kind == GNX_SVC_KERNEL_POLY;
    if (!sw.xfit || !sw.support || !sw.dual_coef || !sw.intercept || !sw.prob_a || !sw.prob_b || !sw.n_support || sw.n_sv <= 0 ||
        (poly ? (!sw.run_value || !(sw.poly_p > 0.0)) : (!sw.ms || sw.n_ms <= 0)))
      return fail(ctx, GNX_EINVAL, "covrsk base: incomplete svc window");
    if (sw.kernel_kind != GNX_SVC_KERNEL_SUBSTRINGS && !poly) return fail(ctx, GNX_EINVAL, "covrsk base: unknown kernel_kind");
    SvcWinDev& wd = wins[(size_t)i];
    wd.width = (int32_t)width;
    wd.nw = (int32_t)((width + 31) / 32);
    wd.n_sv = sw.n_sv;
    max_nw = std::max(max_nw, wd.nw);
    max_width = std::max(max_width, wd.width);
    int acc = 0;
    for (int c = 0; c < A; ++c) { wd.cls_start[c] = acc; acc += sw.n_support[c]; }
    wd.cls_start[A] = acc;
    if (acc != sw.n_sv) return fail(ctx, GNX_EINVAL, "covrsk base: sum(n_support) != n_sv");
    // g(L) = sum_{m in Ms, m <= L} (L - m + 1): K adds g(run length) per maximal match run
    std::vector<int32_t> ms;
    if (!poly) ms.assign(sw.ms, sw.ms + sw.n_ms);
    int32_t goff = poly ? 0 : -1;
    for (size_t k = 0; k < gkeys.size(); ++k)
      if (gkeys[k].first == ms && gkeys[k].second == wd.width) goff = goffs[k];
    if (goff < 0) {
      goff = (int32_t)gtab.size();
      for (int64_t Lr = 0; Lr <= width; ++Lr) {
        uint64_t g = 0;
        for (int32_t mm : ms) if (mm >= 1 && mm <= Lr) g += (uint64_t)(Lr - mm + 1);
        gtab.push_back((uint32_t)g);
      }
      gkeys.push_back({ms, wd.width});
      goffs.push_back(goff);
    }
    wd.g_off = goff;
    wd.n_ms = poly ? 0 : sw.n_ms;
    wd.poly = poly ? 1 : 0;
    wd.rv_off = 0;
    wd.poly_p = poly ? sw.poly_p : 0.0;
    if (poly) m->svc.fast_nw.push_back(-1);  // its own kernel
    else {  // fast path: lengths are a prefix of what CovSample(seed=37) yields, and the window fits 16 words
      static const int32_t canon[] = {1, 4, 8, 39, 42, 117, 376};
      bool ok = sw.n_ms <= 7 && wd.nw <= 16;
      for (int k = 0; ok && k < sw.n_ms; ++k) ok = (sw.ms[k] == canon[k]);
      if (std::getenv("GNX_COVRSK_GENERIC")) ok = false;
      m->svc.fast_nw.push_back(ok ? wd.nw : 0);
    }
    wd.sv_off = (int64_t)svbits.size();
    for (int k = 0; k < sw.n_sv; ++k) {
      const int32_t r = sw.support[k];
      if (r < 0 || r >= sw.n_fit) return fail(ctx, GNX_EINVAL, "covrsk base: support index out of range");
      const int8_t* row = sw.xfit + (size_t)r * width;
      const size_t base = svbits.size();
      svbits.resize(base + 2 * (size_t)wd.nw, 0u);
      for (int64_t t = 0; t < width; ++t) {
        const uint32_t v = (uint32_t)(uint8_t)row[t];
        if (v > 3) return fail(ctx, GNX_EUNSUPPORTED, "covrsk base: training symbols outside {0,1,2,3}");
        svbits[base + (size_t)(t >> 5)] |= (v & 1u) << (t & 31);
        svbits[base + (size_t)wd.nw + (size_t)(t >> 5)] |= ((v >> 1) & 1u) << (t & 31);
      }
    }
    wd.coef_off = (int64_t)coef.size();
    coef.insert(coef.end(), sw.dual_coef, sw.dual_coef + (size_t)(A - 1) * sw.n_sv);
    coef.insert(coef.end(), sw.intercept, sw.intercept + P);
    coef.insert(coef.end(), sw.prob_a, sw.prob_a + P);
    coef.insert(coef.end(), sw.prob_b, sw.prob_b + P);
    if (poly) {
      wd.rv_off = (int64_t)coef.size();
      coef.insert(coef.end(), sw.run_value, sw.run_value + (size_t)width + 1);
    }
  }
  if (gtab.empty()) gtab.push_back(0u);
  if (gnx_covrsk_lds_bytes(A, max_nw, max_width) > 160 * 1024)
    return fail(ctx, GNX_EUNSUPPORTED, "covrsk base: window too wide for the LDS working set");
  int rc;
  if ((rc = gnx_dev_upload(m, wins, &m->svc.win)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, svbits, &m->svc.svbits, 64)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, coef, &m->svc.coef)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, gtab, &m->svc.gtab)) != GNX_OK) return rc;
  m->svc.max_nw = max_nw;
  m->svc.max_width = max_width;
  return GNX_OK;
}

int gnx_build_crf(gnx_model* m, const gnx_model_desc* d) {
  gnx_ctx* ctx = m->ctx;
  const int A = d->A;
  if (!d->crf_state || !d->crf_trans) return fail(ctx, GNX_EINVAL, "crf smoother: crf_state / crf_trans is NULL");
  std::vector<double> st(d->crf_state, d->crf_state + (size_t)A * A), et((size_t)A * A);
  for (int i = 0; i < A * A; ++i) et[(size_t)i] = std::exp(d->crf_trans[i]);
  // How many windows the forward recurrence may run between two rescalings (k_smooth_crf_ck): with B on the simplex one window
  // multiplies a normalised alpha by a factor within e^(+-r), r = max|theta| + max|tau|; float64 holds e^(+-708).  r is doubled
  // for slack (B rows that sum to more than 1), so eight windows need r <= 37.5, four r <= 75, two r <= 150 — a trained model's r is ~10.
  double smax = 0.0, tmax = 0.0;
  for (int i = 0; i < A * A; ++i) {
    smax = std::max(smax, std::fabs(d->crf_state[i]));
    tmax = std::max(tmax, std::fabs(d->crf_trans[i]));
  }
  const double r = smax + tmax;
  m->crf_norm_mask = !(r <= 150.0) ? 0 : (r <= 37.5 ? 7 : (r <= 75.0 ? 3 : 1));
  int rc;
  if ((rc = gnx_dev_upload(m, st, &m->crf_state)) != GNX_OK) return rc;
  if ((rc = gnx_dev_upload(m, et, &m->crf_etrans)) != GNX_OK) return rc;
  return GNX_OK;
}


// ------------------------------------------------------------------------------------------------
// gnx_model_export_prepared (include/gnomix_hip.h): the planes of a loaded logistic model, read back from the device
// ------------------------------------------------------------------------------------------------
extern "C" int gnx_model_export_prepared(gnx_model* m, void* buf, int64_t cap, int64_t* bytes) {
  if (!m || !bytes) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  *bytes = 0;
  if (m->info.base_kind != GNX_BASE_LOGISTIC || !m->lr_i8 || !m->lr.V8) return GNX_OK;   // nothing to prepare for other bases
  const int64_t W = m->info.W;
  const int64_t v8p = (m->lr_v8_bytes + 15) & ~(int64_t)15;
  const int64_t total = (int64_t)sizeof(GnxPreparedHdr) + W * 8 + v8p + m->lr_v2_bytes;
  *bytes = total;
  if (!buf) return GNX_OK;
  if (cap < total) return fail(ctx, GNX_EINVAL, "export_prepared: buffer smaller than *bytes");
  GNX_BIND_DEVICE(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  uint8_t* out = static_cast<uint8_t*>(buf);
  GnxPreparedHdr h{};
  std::memcpy(h.magic, kPreparedMagic, 8);
  h.hdr_bytes = (uint32_t)sizeof(GnxPreparedHdr);
  h.abi = (uint32_t)GNX_ABI_VERSION;
  h.C = m->info.C; h.M = m->info.M; h.ctx = m->info.ctx; h.W = W;
  h.A = m->info.A; h.NT = m->lr.NT; h.NT2 = m->lr.NT2; h.EPR = m->lr.EPR; h.flat = m->lr.V2F ? 1 : 0;
  h.n_chunks = m->lr.n_chunks; h.n_runs = m->lr.n_runs;
  h.key = m->lr_key;
  h.v8_bytes = m->lr_v8_bytes; h.v2_bytes = m->lr_v2_bytes;
  std::memcpy(out, &h, sizeof(h));
  out += sizeof(h);
  HIPCHK(ctx, hipMemcpy(out, m->lr.wscale, (size_t)W * 8, hipMemcpyDeviceToHost));
  out += W * 8;
  HIPCHK(ctx, hipMemcpy(out, m->lr.V8, (size_t)m->lr_v8_bytes, hipMemcpyDeviceToHost));
  std::memset(out + m->lr_v8_bytes, 0, (size_t)(v8p - m->lr_v8_bytes));
  out += v8p;
  if (m->lr_v2_bytes > 0) HIPCHK(ctx, hipMemcpy(out, m->lr.V2F ? m->lr.V2F : m->lr.V2, (size_t)m->lr_v2_bytes, hipMemcpyDeviceToHost));
  return GNX_OK;
}

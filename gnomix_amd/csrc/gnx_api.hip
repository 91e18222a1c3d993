// gnx_api.hip — the C ABI of libgnomix_hip.so (include/gnomix_hip.h): contexts, model preparation
// (weight re-layout, tree packing), host staging, profiling.  The kernels live in k_*.hip.
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

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
int gnx_fail(gnx_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}
static inline int fail(gnx_ctx* ctx, int code, const std::string& msg) { return gnx_fail(ctx, code, msg); }

#define HIPCHK(ctx, expr)                                                                      \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess)                                                                     \
      return fail((ctx), GNX_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
  } while (0)

static int ws_reserve(gnx_ctx* ctx, gnx_devbuf& b, size_t bytes);
int gnx_ws_reserve(gnx_ctx* ctx, gnx_devbuf& b, size_t bytes) { return ws_reserve(ctx, b, bytes); }
static int ws_reserve(gnx_ctx* ctx, gnx_devbuf& b, size_t bytes) {
  if (bytes <= b.cap) return GNX_OK;
  GNX_BIND_DEVICE(ctx);  // hipMalloc allocates on the calling thread's current device
  if (b.p) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  const size_t cap = bytes + (bytes >> 3) + 256;
  hipError_t e = hipMalloc(&b.p, cap);
  if (e != hipSuccess) {
    b.p = nullptr;
    return fail(ctx, GNX_ENOMEM, std::string("hipMalloc(") + std::to_string(cap) + "): " + hipGetErrorString(e));
  }
  b.cap = cap;
  return GNX_OK;
}

// ------------------------------------------------------------------------------------------------
// profiling: one hipEvent pair per launch on the context stream
// ------------------------------------------------------------------------------------------------
struct ProfScope {
  gnx_ctx* ctx;
  int kid;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(gnx_ctx* c, int k) : ctx(c), kid(k) {
    if (!ctx->prof) return;
    auto get = [&]() {
      hipEvent_t e = nullptr;
      if (!ctx->prof_pool.empty()) { e = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); }
      else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
      return e;
    };
    a = get();
    b = get();
    if (a && b) (void)hipEventRecord(a, ctx->stream);
  }
  ~ProfScope() {
    if (!ctx->prof || !a || !b) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->prof_pending.push_back({a, b, kid});
  }
};

static void prof_drain(gnx_ctx* ctx) {
  for (auto& pr : ctx->prof_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(pr.b) == hipSuccess && hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
      ctx->prof_ms[pr.kid] += ms;
      ctx->prof_n[pr.kid] += 1;
    }
    ctx->prof_pool.push_back(pr.a);
    ctx->prof_pool.push_back(pr.b);
  }
  ctx->prof_pending.clear();
}

// every grow-only device workspace of a context (gnx_ctx_free releases them, gnx_debug_ws_devices reports where they live)
static std::vector<gnx_devbuf*> ctx_workspaces(gnx_ctx* ctx) {
  return {&ctx->ws_x, &ctx->ws_b32, &ctx->ws_b64, &ctx->ws_p32, &ctx->ws_p64, &ctx->ws_lab, &ctx->ws_misc, &ctx->ws_scale, &ctx->ws_bits,
          &ctx->ws_lastrow, &ctx->ws_rpair, &ctx->ws_y0, &ctx->ws_cal, &ctx->ws_marg, &ctx->ws_pk, &ctx->ws_xu, &ctx->ws_xu2, &ctx->ws_psi,
          &ctx->ws_gt2, &ctx->ws_src, &ctx->ws_gt2o, &ctx->ws_rank, &ctx->ws_fb, &ctx->ws_fb_body};
}

// ------------------------------------------------------------------------------------------------
// ABI
// ------------------------------------------------------------------------------------------------
// the development / test knobs, read once per context (gnx_init)
static void read_tune(gnx_tune& t) {
  auto geti = [](const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; };
  t.lr_bpc = geti("GNX_LR_BPC", 0);
  t.lr_want = geti("GNX_LR_WANT", 0);
  if (const char* e = std::getenv("GNX_LR_TUNE")) std::sscanf(e, "%d,%d", &t.lr_mt, &t.lr_waves);
  t.lr_flags = geti("GNX_LR_FLAGS", 0);
  t.lr_dl = geti("GNX_LR_DL", -1);
  t.lr_flat = geti("GNX_LR_FLAT", -1);
  t.lr_ws = geti("GNX_LR_WS", 0);
  t.lr_w512 = geti("GNX_LR_W512", 0);
  t.lr_ws_pw = geti("GNX_LR_WS_PW", 2);
  t.lr_nbuf = geti("GNX_LR_NBUF", 0);
  t.lr_p2 = geti("GNX_LR_P2", 1);
  t.p2_decline = geti("GNX_P2_DECLINE", 0);
  if (const char* e = std::getenv("GNX_P2_TUNE")) std::sscanf(e, "%d,%d,%d,%d,%d", &t.p2_mt, &t.p2_cw, &t.p2_ew, &t.p2_xsn, &t.p2_nbuf);
  t.sm_nw = geti("GNX_SM_NW", 0);
  t.sm_pair = geti("GNX_SM_PAIR", 1);
  if (const char* e = std::getenv("GNX_SM_TUNE")) std::sscanf(e, "%d,%d", &t.smf_rpl, &t.smf_nw);
  if (const char* e = std::getenv("GNX_CRF_FLAGS")) t.crf_flags = atoi(e);
  if (const char* e = std::getenv("GNX_LDS_PAD")) std::sscanf(e, "%d,%d", &t.lr_lds_pad, &t.sm_lds_pad);
  if (const char* e = std::getenv("GNX_CRF_IMPL")) t.crf_impl = !std::strcmp(e, "scan") ? 1 : !std::strcmp(e, "row") ? 2 : !std::strcmp(e, "lanes") ? 3 : !std::strcmp(e, "quad") ? 4 : !std::strcmp(e, "mm") ? 5 : 0;
  t.forest_threads = geti("GNX_FOREST_T", 0);
  t.forest_wrun = geti("GNX_FOREST_WRUN", 0);
  t.forest_halves = geti("GNX_FOREST_H", 0);
  t.forest_flags = geti("GNX_FOREST_FLAGS", 0);
  t.forest_impl = geti("GNX_FOREST_IMPL", 0);
  if (const char* e = std::getenv("GNX_HOST_BATCH")) t.host_batch = std::atoll(e);
  t.h2d_overlap = geti("GNX_H2D_OVERLAP", 1);
  t.debug = std::getenv("GNX_DEBUG") ? atoi(std::getenv("GNX_DEBUG")) : 0;
  if (const char* e = std::getenv("GNX_GNOFIX_IMPL")) t.gnofix_impl = std::string(e) == "f32" ? 1 : 0;
  t.gnofix_threads = geti("GNX_GNOFIX_T", 0);
  t.gnofix_aux = geti("GNX_GNOFIX_AUX", 1);
}

extern "C" {

int gnx_abi_version(void) { return GNX_ABI_VERSION; }

int gnx_build_flags(void) {
#ifdef GNX_EXPERIMENTS
  return GNX_BUILD_EXPERIMENTS;
#else
  return 0;
#endif
}

// HIP maps streams onto 4 hardware queues by default and two streams on one queue serialise; a context uses up to five (see
// gnomix_amd/_lib.py: load).  Effective when the runtime has not been initialised yet; an embedding application that has already
// used HIP sets GPU_MAX_HW_QUEUES itself.
static void want_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

int gnx_device_count(void) {
  want_hw_queues();
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int gnx_init(int device, gnx_ctx** out) {
  if (!out) return GNX_EINVAL;
  *out = nullptr;
  gnx_ctx* ctx = new (std::nothrow) gnx_ctx();
  if (!ctx) return GNX_ENOMEM;
  ctx->device = device;
  want_hw_queues();
  gnx_device_scope bind(ctx);  // the caller's current device is put back on return
  hipError_t e = bind.err;
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    // the context is returned so the caller can read the message, but it is unusable
    ctx->err = std::string("gnx_init: ") + hipGetErrorString(e);
    ctx->stream = nullptr;
    *out = ctx;
    return GNX_EHIP;
  }
  ctx->own_stream = true;
  ctx->usable = true;
  read_tune(ctx->tune);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
  *out = ctx;
  return GNX_OK;
}

void gnx_ctx_free(gnx_ctx* ctx) {
  if (!ctx) return;
  gnx_device_scope bind(ctx);
  if (ctx->usable) (void)hipStreamSynchronize(ctx->stream);
  prof_drain(ctx);
  for (auto e : ctx->prof_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : {ctx->ev_in[0], ctx->ev_in[1], ctx->ev_done[0], ctx->ev_done[1], ctx->ev_out[0], ctx->ev_out[1]})
    if (e) (void)hipEventDestroy(e);
  if (ctx->s_aux) (void)hipStreamDestroy(ctx->s_aux);
  for (auto e : ctx->ev_aux)
    if (e) (void)hipEventDestroy(e);
  if (ctx->s_in) (void)hipStreamDestroy(ctx->s_in);
  if (ctx->s_out) (void)hipStreamDestroy(ctx->s_out);
  for (gnx_devbuf* b : ctx_workspaces(ctx))
    if (b->p) (void)hipFree(b->p);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* gnx_last_error(const gnx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int gnx_set_stream(gnx_ctx* ctx, void* hip_stream) {
  if (!ctx) return GNX_EINVAL;
  if (!ctx->own_stream && ctx->stream == (hipStream_t)hip_stream) return GNX_OK;
  GNX_BIND_DEVICE(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream) {
    (void)hipStreamDestroy(ctx->stream);
    ctx->own_stream = false;
  }
  ctx->stream = (hipStream_t)hip_stream;  // NULL = the default stream (what torch uses unless told otherwise)
  return GNX_OK;
}

int gnx_reset_stream(gnx_ctx* ctx) {
  if (!ctx) return GNX_EINVAL;
  if (ctx->own_stream) return GNX_OK;
  GNX_BIND_DEVICE(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  ctx->own_stream = true;
  return GNX_OK;
}

}  // extern "C"

// ---- page-locked host buffers are expensive to make and to release (hipHostMalloc / hipHostFree pin and unpin every page: ~0.15 ms
// per MB on the MI355X box, i.e. 35 ms for the 2-bit genotype matrix of chr22 x 5 000 samples and 20 ms for its outputs, per call).
// A long-lived process (a server, bench.py's repeated passes) asks for the same sizes again and again: released buffers of at least
// 1 MB wait in a small process-wide list (one process drives one GPU) and are handed out again to requests they fit without more
// than 2x slack.  GNX_PIN_CACHE_MB bounds what the list may hold (default 1024, 0 = off).
#include <mutex>
#include <unordered_map>
namespace {
struct PinCache {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;            // every buffer this module allocated -> its capacity
  std::vector<std::pair<void*, size_t>> idle;        // released, reusable
  size_t idle_bytes = 0, limit = (size_t)1024 << 20;
  PinCache() {
    if (const char* e = std::getenv("GNX_PIN_CACHE_MB")) limit = (size_t)std::max(0LL, std::atoll(e)) << 20;
  }
};
PinCache& pin_cache() {
  static PinCache* c = new PinCache();  // never destroyed: buffers may be released during interpreter shutdown
  return *c;
}
}  // namespace
void* gnx_pin_alloc(size_t bytes) {
  PinCache& c = pin_cache();
  bytes = bytes ? bytes : 1;
  {
    std::lock_guard<std::mutex> g(c.mu);
    int best = -1;
    for (int i = 0; i < (int)c.idle.size(); ++i)
      if (c.idle[(size_t)i].second >= bytes && c.idle[(size_t)i].second <= 2 * bytes + ((size_t)1 << 20) &&
          (best < 0 || c.idle[(size_t)i].second < c.idle[(size_t)best].second))
        best = i;
    if (best >= 0) {
      void* p = c.idle[(size_t)best].first;
      c.idle_bytes -= c.idle[(size_t)best].second;
      c.idle.erase(c.idle.begin() + best);
      return p;
    }
  }
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
    // the idle list may be what exhausts the pinnable memory: release it and try once more
    std::vector<std::pair<void*, size_t>> drop;
    {
      std::lock_guard<std::mutex> g(c.mu);
      drop.swap(c.idle);
      c.idle_bytes = 0;
      for (auto& d : drop) c.live.erase(d.first);
    }
    for (auto& d : drop) (void)hipHostFree(d.first);
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
  }
  std::lock_guard<std::mutex> g(c.mu);
  c.live[p] = bytes;
  return p;
}
void gnx_pin_free(void* p) {
  if (!p) return;
  PinCache& c = pin_cache();
  {
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.live.find(p);
    if (it != c.live.end() && it->second >= ((size_t)1 << 20) && c.idle_bytes + it->second <= c.limit && c.idle.size() < 8) {
      c.idle.emplace_back(p, it->second);
      c.idle_bytes += it->second;
      return;
    }
    if (it != c.live.end()) c.live.erase(it);
  }
  (void)hipHostFree(p);
}

extern "C" {

int gnx_host_alloc(gnx_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return GNX_EINVAL;
  *out = nullptr;
  if (!ctx->usable) return fail(ctx, GNX_ESTATE, "context has no device (gnx_init failed)");
  GNX_BIND_DEVICE(ctx);
  *out = gnx_pin_alloc(bytes);
  if (!*out) return fail(ctx, GNX_ENOMEM, "hipHostMalloc(" + std::to_string(bytes) + ") failed");
  return GNX_OK;
}

int gnx_host_flags(const void* p, unsigned* flags) {
  if (!p || !flags) return GNX_EINVAL;
  return hipHostGetFlags(flags, const_cast<void*>(p)) == hipSuccess ? GNX_OK : GNX_EHIP;
}

int gnx_host_free(gnx_ctx* ctx, void* p) {
  if (!ctx) return GNX_EINVAL;
  gnx_pin_free(p);
  return GNX_OK;
}

int gnx_debug_ws_devices(gnx_ctx* ctx, int32_t* out, int32_t n) {
  if (!ctx || n < 0 || (n > 0 && !out)) return GNX_EINVAL;
  int32_t live = 0;
  for (gnx_devbuf* b : ctx_workspaces(ctx)) {
    if (!b->p) continue;
    hipPointerAttribute_t at;
    HIPCHK(ctx, hipPointerGetAttributes(&at, b->p));
    if (live < n) out[live] = (int32_t)at.device;
    ++live;
  }
  return live;
}

int gnx_synchronize(gnx_ctx* ctx) {
  if (!ctx) return GNX_EINVAL;
  GNX_BIND_DEVICE(ctx);
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GNX_OK;
}

int gnx_model_load(gnx_ctx* ctx, const gnx_model_desc* d, gnx_model** out) {
  if (!ctx || !out) return GNX_EINVAL;
  *out = nullptr;
  if (!ctx->usable) return fail(ctx, GNX_ESTATE, "context has no device (gnx_init failed)");
  if (!d) return fail(ctx, GNX_EINVAL, "model description is NULL");
  if (d->abi_version != GNX_ABI_VERSION) return fail(ctx, GNX_EINVAL, "gnx_model_desc.abi_version mismatch");
  if (d->A < 2 || d->A > 32) return fail(ctx, GNX_EINVAL, "A (ancestries) must be in [2, 32]");
  if (d->M <= 0 || d->C < d->M || d->ctx < 0 || d->ctx > d->C) return fail(ctx, GNX_EINVAL, "bad C / M / ctx");
  if (d->C % d->M == 0)  // src/Base/base.py:158 + gnomix.py:124-125
    return fail(ctx, GNX_EINVAL, "C % M == 0: the reference's window slicing (base.py:158) requires a remainder");
  if (d->C > (int64_t)1 << 30) return fail(ctx, GNX_EINVAL, "C too large");
  GNX_BIND_DEVICE(ctx);
  gnx_model* m = new (std::nothrow) gnx_model();
  if (!m) return fail(ctx, GNX_ENOMEM, "host allocation failed");
  m->ctx = ctx;
  m->info.C = d->C; m->info.M = d->M; m->info.ctx = d->ctx; m->info.W = d->C / d->M;
  m->info.A = d->A; m->info.S = d->S; m->info.base_kind = d->base_kind; m->info.smooth_kind = d->smooth_kind;
  int rc = GNX_OK;
  switch (d->base_kind) {
    case GNX_BASE_NONE: break;
    case GNX_BASE_LOGISTIC: rc = gnx_build_lr(m, d); break;
    case GNX_BASE_COVRSK_SVC: rc = gnx_build_covrsk(m, d); break;
    case GNX_BASE_FOREST: rc = gnx_build_forest(m, d); break;
    case GNX_BASE_RFOREST: rc = gnx_build_rforest(m, d); break;
    default: rc = fail(ctx, GNX_EINVAL, "unknown base_kind");
  }
  if (rc == GNX_OK) switch (d->smooth_kind) {
    case GNX_SMOOTH_NONE: break;
    case GNX_SMOOTH_XGB:
      if (d->S <= 0 || d->S % 2 == 0) rc = fail(ctx, GNX_EINVAL, "S must be odd and positive (smooth.py:14)");
      else rc = gnx_build_xgb(m, d);
      break;
    case GNX_SMOOTH_CRF: rc = gnx_build_crf(m, d); break;
    case GNX_SMOOTH_CNN:
      if (d->S <= 0 || d->S % 2 == 0) rc = fail(ctx, GNX_EINVAL, "S must be odd and positive (smooth.py:14)");
      else if (!d->cnn_weight || !d->cnn_bias) rc = fail(ctx, GNX_EINVAL, "cnn smoother: cnn_weight / cnn_bias is NULL");
      else {
        // torch's (out, in, k) -> [in][k][out padded to AP]: one tap's output weights are a contiguous, wave-uniform run that the
        // kernel fetches with scalar loads; bias padded likewise
        const int A = d->A, S = d->S, AP = gnx_cnn_ap(A);
        std::vector<float> wv((size_t)A * S * AP, 0.f), bv((size_t)AP, 0.f);
        for (int ao = 0; ao < A; ++ao)
          for (int ai = 0; ai < A; ++ai)
            for (int k = 0; k < S; ++k) wv[((size_t)ai * S + k) * AP + ao] = d->cnn_weight[((size_t)ao * A + ai) * S + k];
        for (int ao = 0; ao < A; ++ao) bv[(size_t)ao] = d->cnn_bias[ao];
        if ((rc = gnx_dev_upload(m, wv, &m->cnn_weight)) == GNX_OK) rc = gnx_dev_upload(m, bv, &m->cnn_bias);
      }
      break;
    default: rc = fail(ctx, GNX_EINVAL, "unknown smooth_kind");
  }
  if (rc == GNX_OK && d->calib_off) {
    if (!d->calib_x || !d->calib_y) rc = fail(ctx, GNX_EINVAL, "calibrator: calib_x / calib_y is NULL");
    else {
      std::vector<int32_t> off(d->calib_off, d->calib_off + d->A + 1);
      bool ok = off[0] == 0;
      for (int c = 0; c < d->A && ok; ++c) ok = off[(size_t)c + 1] > off[(size_t)c];
      if (!ok) rc = fail(ctx, GNX_EINVAL, "calibrator: calib_off must start at 0 and give every class >= 1 threshold");
      else {
        std::vector<double> cx(d->calib_x, d->calib_x + off[(size_t)d->A]), cy(d->calib_y, d->calib_y + off[(size_t)d->A]);
        if ((rc = gnx_dev_upload(m, off, &m->calib_off)) == GNX_OK && (rc = gnx_dev_upload(m, cx, &m->calib_x)) == GNX_OK)
          rc = gnx_dev_upload(m, cy, &m->calib_y);
        m->calib_f32 = d->calib_is_f32 != 0;
      }
    }
  }
  if (rc != GNX_OK) {
    gnx_model_free(m);
    return rc;
  }
  *out = m;
  return GNX_OK;
}

int gnx_model_set_calibrate(gnx_model* m, int on) {
  if (!m) return GNX_EINVAL;
  m->calibrate_on = on != 0;
  return GNX_OK;
}

void gnx_model_free(gnx_model* m) {
  if (!m) return;
  gnx_ctx unbound;  // (a model without a context frees on the caller's device)
  if (!m->ctx) (void)hipGetDevice(&unbound.device);
  gnx_device_scope bind(m->ctx ? m->ctx : &unbound);
  if (m->ctx && m->ctx->usable) (void)hipStreamSynchronize(m->ctx->stream);
  for (void* p : m->dev_allocs) (void)hipFree(p);
  delete m;
}

int gnx_model_get_info(const gnx_model* m, gnx_model_info* out) {
  if (!m || !out) return GNX_EINVAL;
  *out = m->info;
  return GNX_OK;
}

// ---- base ------------------------------------------------------------------------------------------
int gnx_base_predict_dev(gnx_model* m, const int8_t* dX, int64_t N, int64_t ldx, float* d_b32, double* d_b64) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (N < 0 || ldx < m->info.C || (N > 0 && !dX)) return fail(ctx, GNX_EINVAL, "base_predict: bad X / N / ldx");
  if (N == 0 || (!d_b32 && !d_b64)) return GNX_OK;
  GNX_BIND_DEVICE(ctx);
  if (m->info.base_kind == GNX_BASE_COVRSK_SVC) {
    const int64_t Cp = m->info.C + 2 * m->info.ctx, nwp = (Cp + 31) / 32 + 2;
    int rc = ws_reserve(ctx, ctx->ws_bits, (size_t)N * 2 * nwp * 4);
    if (rc != GNX_OK) return rc;
    ProfScope ps(ctx, GNX_K_BASE_COVRSK);
    HIPCHK(ctx, gnx_launch_pack_bits(dX, N, ldx, m->info.C, m->info.ctx, nwp, (uint32_t*)ctx->ws_bits.p, ctx->stream));
    CovRSKLaunch L{};
    L.planes = (const uint32_t*)ctx->ws_bits.p; L.N = N; L.nwp = nwp; L.M = m->info.M;
    L.W = (int32_t)m->info.W; L.A = m->info.A;
    L.win = m->svc.win; L.svbits = m->svc.svbits; L.coef = m->svc.coef; L.gtab = m->svc.gtab;
    L.max_nw = m->svc.max_nw; L.max_width = m->svc.max_width;
    L.host_fast_nw = m->svc.fast_nw.data();
    L.b32 = d_b32; L.b64 = d_b64;
    {
      const size_t per_hap = (size_t)m->info.W * (m->info.A * (m->info.A - 1) / 2) * sizeof(double);
      int64_t haps = std::max<int64_t>(64, (((int64_t)256 << 20) / (int64_t)per_hap) / 64 * 64);
      haps = std::min<int64_t>(haps, (N + 63) / 64 * 64);
      if ((rc = ws_reserve(ctx, ctx->ws_rpair, (size_t)haps * per_hap)) != GNX_OK) return rc;
      L.rpair = (double*)ctx->ws_rpair.p;
      L.rpair_haps = haps;
    }
    if (!ctx->s_aux) {
      HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_aux, hipStreamNonBlocking));
      for (int b = 0; b < 2; ++b) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_aux[b], hipEventDisableTiming));
    }
    L.aux = ctx->s_aux; L.ev_fork = ctx->ev_aux[0]; L.ev_join = ctx->ev_aux[1]; L.n_cu = ctx->n_cu;
    HIPCHK(ctx, gnx_launch_covrsk(L, ctx->stream));
    return GNX_OK;
  }
  if (m->info.base_kind == GNX_BASE_FOREST || m->info.base_kind == GNX_BASE_RFOREST) {
    ProfScope ps(ctx, GNX_K_BASE_FOREST);
    ForestLaunch L{};
    L.X = dX; L.N = N; L.ldx = ldx; L.C = m->info.C; L.ctx = m->info.ctx; L.M = m->info.M;
    L.width = m->info.M + 2 * m->info.ctx;
    L.width_last = L.width + (m->info.C - m->info.M * m->info.W);
    L.W = (int32_t)m->info.W; L.A = m->info.A; L.D = m->forest.D; L.tree_bytes = m->forest.tree_bytes;
    L.max_trees = m->forest.max_trees; L.missing = m->forest.missing;
    L.base_score = m->forest.base_score;
    L.packed = m->forest.packed; L.win_tree0 = m->forest.win_tree0; L.win_class_tree0 = m->forest.win_class_tree0;
    L.rf_leafval = m->forest.rf_leafval;
    L.b32 = d_b32; L.b64 = d_b64;
    // the two-blocks-per-CU kernel wherever its 128-haplotype tile fits the LDS (both tree bases); GNX_FOREST_IMPL=1: the 256-haplotype kernel
    const bool v2 = m->forest.nodes2 && ctx->tune.forest_impl != 1 && (!L.rf_leafval || L.A <= 32) &&
                    gnx_forest2_lds_bytes(L.A, gnx_forest_ring_words(L.width_last), L.max_trees, L.D, L.rf_leafval != nullptr) <= (size_t)160 * 1024;
    if (v2) {
      if (!ctx->s_aux) {  // the one wider last window (a grid of N / 128 blocks) runs beside the main grid on a side stream
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_aux, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_aux[b], hipEventDisableTiming));
      }
      HIPCHK(ctx, gnx_launch_base_forest2(L, m->forest.nodes2, ctx->n_cu, ctx->tune, ctx->stream, ctx->s_aux, ctx->ev_aux[0], ctx->ev_aux[1]));
    }
    else HIPCHK(ctx, gnx_launch_base_forest(L, ctx->n_cu, ctx->tune, ctx->stream));
    return GNX_OK;
  }
  if (m->info.base_kind != GNX_BASE_LOGISTIC) return fail(ctx, GNX_ESTATE, "model has no base classifier");
  // the kernels fetch 16-byte pieces of 64-SNP chunks unconditionally: only the last row could read past X
  {
    const size_t need = (size_t)m->info.C + 128;
    if (ctx->ws_lastrow.cap < need) {
      int rc = ws_reserve(ctx, ctx->ws_lastrow, need);
      if (rc != GNX_OK) return rc;
      HIPCHK(ctx, hipMemsetAsync(ctx->ws_lastrow.p, 0, ctx->ws_lastrow.cap, ctx->stream));
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_lastrow.p, dX + (N - 1) * ldx, (size_t)m->info.C, hipMemcpyDeviceToDevice, ctx->stream));
  }
  BaseLRLaunch L{};
  L.X = dX;
  L.last_row = (const int8_t*)ctx->ws_lastrow.p;
  L.N = N; L.ldx = ldx; L.d = m->lr;
  L.h_win_chunk0 = m->lr_h_win_chunk0.data(); L.h_win_chunk1 = m->lr_h_win_chunk1.data();
  L.W = (int32_t)m->info.W; L.A = m->info.A;
  L.b32 = d_b32; L.b64 = d_b64;
  ProfScope ps(ctx, GNX_K_BASE_LOGISTIC);
  if (m->lr_i8) {
    const bool dl = ctx->tune.lr_dl < 0 ? m->lr.NT >= 2 : ctx->tune.lr_dl != 0;
    hipError_t e = hipErrorNotSupported;
#ifdef GNX_EXPERIMENTS  // make EXPERIMENTS=1: the measured-slower structures of DESIGN.md 5.2 (scripts/dev/rejected/)
    if (ctx->tune.lr_ws) e = gnx_launch_base_logistic_i8_ws(L, ctx->n_cu, ctx->tune, ctx->stream);
    if (e == hipErrorNotSupported && ctx->tune.lr_w512) e = gnx_launch_base_logistic_i8_w512(L, ctx->n_cu, ctx->tune, ctx->stream);
    if (e == hipErrorNotSupported && ctx->tune.lr_flat > 0 && m->lr.V8F) e = gnx_launch_base_logistic_i8_fl(L, ctx->n_cu, ctx->tune, ctx->stream);
#endif
    if (e == hipErrorNotSupported && dl) e = gnx_launch_base_logistic_i8_dl(L, ctx->n_cu, ctx->tune, ctx->stream);
    if (e == hipErrorNotSupported) e = gnx_launch_base_logistic_i8(L, ctx->n_cu, ctx->tune, ctx->stream);  // > 2 column tiles
    HIPCHK(ctx, e);
  }
  else HIPCHK(ctx, gnx_launch_base_logistic(L, ctx->n_cu, ctx->stream));
  return GNX_OK;
}

// ---- smoother ------------------------------------------------------------------------------------
static int smooth_raw_dev(gnx_model* m, const void* dB, int b_is_f64, int64_t N, float* d_p32, double* d_p64, int32_t* d_lab);

int gnx_smooth_predict_dev(gnx_model* m, const void* dB, int b_is_f64, int64_t N, float* d_p32, double* d_p64,
                           int32_t* d_lab) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  GNX_BIND_DEVICE(ctx);
  if (!(m->calibrate_on && m->calib_off)) return smooth_raw_dev(m, dB, b_is_f64, N, d_p32, d_p64, d_lab);
  if (N < 0 || (N > 0 && !dB)) return fail(ctx, GNX_EINVAL, "smooth_predict: bad B / N");
  if (N == 0) return GNX_OK;
  // Smoother.predict_proba with a calibrator (smooth.py:46-52): raw smoother probabilities -> Calibrator.transform
  const size_t n = (size_t)N * m->info.W * m->info.A;
  const bool nat64 = m->info.smooth_kind == GNX_SMOOTH_CRF;
  int rc = ws_reserve(ctx, ctx->ws_cal, n * (nat64 ? 8 : 4));
  if (rc != GNX_OK) return rc;
  rc = smooth_raw_dev(m, dB, b_is_f64, N, nat64 ? nullptr : (float*)ctx->ws_cal.p, nat64 ? (double*)ctx->ws_cal.p : nullptr, nullptr);
  if (rc != GNX_OK) return rc;
  CalibLaunch L{};
  L.in = ctx->ws_cal.p; L.in_is_f64 = nat64; L.R = N * m->info.W; L.A = m->info.A;
  L.off = m->calib_off; L.x = m->calib_x; L.y = m->calib_y; L.thr_f32 = m->calib_f32;
  L.out64 = d_p64; L.out32 = d_p32; L.labels = d_lab;
  ProfScope ps(ctx, GNX_K_CALIBRATE);
  HIPCHK(ctx, gnx_launch_calibrate(L, ctx->stream));
  return GNX_OK;
}

static int smooth_raw_dev(gnx_model* m, const void* dB, int b_is_f64, int64_t N, float* d_p32, double* d_p64, int32_t* d_lab) {
  gnx_ctx* ctx = m->ctx;
  if (N < 0 || (N > 0 && !dB)) return fail(ctx, GNX_EINVAL, "smooth_predict: bad B / N");
  if (N == 0) return GNX_OK;
  if (m->info.smooth_kind == GNX_SMOOTH_XGB) {
    const size_t n = (size_t)N * m->info.W * m->info.A;
    if (!d_p32) {  // margins are parked in the f32 output: borrow a workspace
      int rc = ws_reserve(ctx, ctx->ws_misc, n * sizeof(float));
      if (rc != GNX_OK) return rc;
      d_p32 = (float*)ctx->ws_misc.p;
    }
    SmoothXGBLaunch L{};
    L.B = dB; L.b_is_f64 = b_is_f64; L.N = N;
    L.W = (int32_t)m->info.W; L.A = m->info.A; L.S = m->info.S;
    L.d = m->xgb; L.proba = d_p32; L.proba64 = d_p64; L.labels = d_lab;
#ifdef GNX_EXPERIMENTS
    const bool h64 = m->xgb.rk_packed && m->xgb.impl == 2 && gnx_smooth_h64_waves(m->xgb, m->info.A, m->info.S) > 0;
    if (m->xgb.impl == 2 && !h64) return fail(ctx, GNX_EUNSUPPORTED, "GNX_SMOOTH_IMPL=h64: the model's strip does not fit the LDS");
#else
    [[maybe_unused]] constexpr bool h64 = false;
#endif
    if (m->xgb.rk_packed) {
      const size_t n_pad = (size_t)((N + 63) / 64 * 64) * m->info.W * m->info.A;   // h64 parks whole 64-haplotype lines
      int rc = ws_reserve(ctx, ctx->ws_marg, n_pad * sizeof(float));
      if (rc != GNX_OK) return rc;
      L.marg = (float*)ctx->ws_marg.p;
#ifdef GNX_EXPERIMENTS
      if (h64 && (rc = ws_reserve(ctx, ctx->ws_rank, gnx_smooth_h64_rank_bytes(N, L.W, L.A, L.S))) != GNX_OK) return rc;
#endif
    }
#ifdef GNX_EXPERIMENTS
    const bool bs = m->xgb.impl == 4 && gnx_smooth_bs_fits(m->xgb, m->info.A, m->info.S);
    if (bs) {  // the rank pre-pass parks one 16-bit counter index per probability
      int rc = ws_reserve(ctx, ctx->ws_marg, gnx_smooth_bs_scratch_bytes(N, L.W, L.A));
      if (rc != GNX_OK) return rc;
    }
#endif
    ProfScope ps(ctx, GNX_K_SMOOTH_XGB);
#ifdef GNX_EXPERIMENTS
    if (bs) {
      HIPCHK(ctx, gnx_launch_smooth_xgb_bs(L, (uint16_t*)ctx->ws_marg.p, ctx->n_cu, ctx->tune, ctx->stream));
      return GNX_OK;
    }
#endif
#ifdef GNX_EXPERIMENTS
    if (h64) HIPCHK(ctx, gnx_launch_smooth_xgb_h64(L, (uint16_t*)ctx->ws_rank.p, ctx->tune, ctx->stream));
    else
#endif
    if (m->xgb.rk_packed) HIPCHK(ctx, gnx_launch_smooth_xgb_rk(L, ctx->tune, ctx->stream));
    else HIPCHK(ctx, gnx_launch_smooth_xgb(L, ctx->tune, ctx->stream));
    return GNX_OK;
  }
  if (m->info.smooth_kind == GNX_SMOOTH_CRF) {
    const size_t n = (size_t)N * m->info.W * m->info.A;
    int rc;
    // psi and the parked alphas live in context scratch with a padded tail (the scan's chunked loads may overrun the last row)
    if ((rc = ws_reserve(ctx, ctx->ws_misc, n * sizeof(double) + 4096)) != GNX_OK) return rc;
    double* alpha = (double*)ctx->ws_misc.p;
    if ((rc = ws_reserve(ctx, ctx->ws_scale, (size_t)2 * N * m->info.W * sizeof(double))) != GNX_OK) return rc;  // (c_t, 1/c_t) pairs
    if ((rc = ws_reserve(ctx, ctx->ws_psi, n * sizeof(double) + 4096)) != GNX_OK) return rc;
    SmoothCRFLaunch L{};
    L.psi = (double*)ctx->ws_psi.p;
    L.B = dB; L.b_is_f64 = b_is_f64; L.N = N; L.W = (int32_t)m->info.W; L.A = m->info.A;
    L.state = m->crf_state; L.etrans = m->crf_etrans; L.norm_mask = m->crf_norm_mask;
    L.alpha = alpha; L.scale = (double*)ctx->ws_scale.p;
    L.proba64 = d_p64; L.proba32 = d_p32; L.labels = d_lab;
    ProfScope ps(ctx, GNX_K_SMOOTH_CRF);
    HIPCHK(ctx, gnx_launch_smooth_crf(L, ctx->tune, ctx->stream));
    return GNX_OK;
  }
  if (m->info.smooth_kind == GNX_SMOOTH_CNN) {
    SmoothCNNLaunch L{};
    L.B = dB; L.b_is_f64 = b_is_f64; L.N = N; L.W = (int32_t)m->info.W; L.A = m->info.A; L.S = m->info.S;
    L.weight = m->cnn_weight; L.bias = m->cnn_bias;
    L.proba32 = d_p32; L.proba64 = d_p64; L.labels = d_lab;
    ProfScope ps(ctx, GNX_K_SMOOTH_CNN);
    HIPCHK(ctx, gnx_launch_smooth_cnn(L, ctx->stream));
    return GNX_OK;
  }
  return fail(ctx, GNX_ESTATE, "model has no smoother");
}

// ---- packed (2-bit) input: the 2-bit-native logistic pass where the model has its planes, otherwise widen + the int8 kernels ----
static bool lr_p2_usable(const gnx_model* m) {
  return m->info.base_kind == GNX_BASE_LOGISTIC && m->lr_i8 && (m->lr.V2 || m->lr.V2F) && m->ctx->tune.lr_p2 != 0;
}

int gnx_base_predict_packed_dev(gnx_model* m, const uint8_t* dP, int64_t N, int64_t ldp, float* d_b32, double* d_b64) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  const int64_t C = m->info.C, rowb = (C + 3) / 4;
  if (N < 0 || ldp < rowb || (N > 0 && !dP)) return fail(ctx, GNX_EINVAL, "base_predict_packed: bad P / N / ldp");
  if (N == 0 || (!d_b32 && !d_b64)) return GNX_OK;
  GNX_BIND_DEVICE(ctx);
  if (lr_p2_usable(m)) {
    // the kernel fetches 64-byte runs unconditionally: only the last row could read past the matrix
    const size_t need = (size_t)C + 128;
    if (ctx->ws_lastrow.cap < need) {
      int rc = ws_reserve(ctx, ctx->ws_lastrow, need);
      if (rc != GNX_OK) return rc;
      HIPCHK(ctx, hipMemsetAsync(ctx->ws_lastrow.p, 0, ctx->ws_lastrow.cap, ctx->stream));  // the int8 pass counts on a zero tail
    }
    HIPCHK(ctx, hipMemsetAsync(ctx->ws_lastrow.p, 0, (size_t)rowb + 128, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_lastrow.p, dP + (N - 1) * ldp, (size_t)rowb, hipMemcpyDeviceToDevice, ctx->stream));
    BaseLRLaunch L{};
    L.X = reinterpret_cast<const int8_t*>(dP);
    L.last_row = (const int8_t*)ctx->ws_lastrow.p;
    L.N = N; L.ldx = ldp; L.d = m->lr;
    L.h_win_chunk0 = m->lr_h_win_run0.data(); L.h_win_chunk1 = m->lr_h_win_run1.data();
    L.W = (int32_t)m->info.W; L.A = m->info.A;
    L.b32 = d_b32; L.b64 = d_b64;
    ProfScope ps(ctx, GNX_K_BASE_LOGISTIC);
    // GNX_P2_DECLINE=1 (tests): behave as if no instantiation fitted, so that the widening fallback below runs
    const hipError_t e = ctx->tune.p2_decline ? hipErrorNotSupported : gnx_launch_base_logistic_p2(L, ctx->n_cu, ctx->tune, ctx->stream);
    if (e != hipErrorNotSupported) {
      HIPCHK(ctx, e);
      return GNX_OK;
    }
  }
  // widen to int8 and run the int8 kernels.  The file routes (gnx_infer_gt2_range / gnx_phase_gt2_range) build their packed rows IN
  // ws_xu: widening those in place would read and write one buffer at overlapping strides (and growing ws_xu would free the input),
  // so rows that live in ws_xu are widened into a workspace of their own.
  const int64_t ldx = ((C + 15) / 16) * 16;
  const char* in0 = reinterpret_cast<const char*>(dP);
  const bool in_xu = ctx->ws_xu.p && in0 >= (const char*)ctx->ws_xu.p && in0 < (const char*)ctx->ws_xu.p + ctx->ws_xu.cap;
  gnx_devbuf& wide = in_xu ? ctx->ws_xu2 : ctx->ws_xu;
  int rc = ws_reserve(ctx, wide, (size_t)N * ldx + 256);
  if (rc != GNX_OK) return rc;
  HIPCHK(ctx, gnx_launch_unpack2(dP, N, ldp, C, (int8_t*)wide.p, ldx, ctx->stream));
  return gnx_base_predict_dev(m, (const int8_t*)wide.p, N, ldx, d_b32, d_b64);
}

// base + smoother with B kept in context scratch; X int8 (packed == false) or 2-bit rows
static int infer_any_dev(gnx_model* m, const void* dIn, bool packed, int64_t N, int64_t ld, float* d_p32, double* d_p64, int32_t* d_lab) {
  gnx_ctx* ctx = m->ctx;
  if (N <= 0) return N == 0 ? GNX_OK : fail(ctx, GNX_EINVAL, "infer: N < 0");
  GNX_BIND_DEVICE(ctx);
  const size_t n = (size_t)N * m->info.W * m->info.A;
  const bool f64 = (m->info.smooth_kind == GNX_SMOOTH_CRF);  // CRF consumes float64 base probabilities
  int rc = ws_reserve(ctx, f64 ? ctx->ws_b64 : ctx->ws_b32, n * (f64 ? 8 : 4));
  if (rc != GNX_OK) return rc;
  float* b32 = f64 ? nullptr : (float*)ctx->ws_b32.p;
  double* b64 = f64 ? (double*)ctx->ws_b64.p : nullptr;
  rc = packed ? gnx_base_predict_packed_dev(m, (const uint8_t*)dIn, N, ld, b32, b64) : gnx_base_predict_dev(m, (const int8_t*)dIn, N, ld, b32, b64);
  if (rc != GNX_OK) return rc;
  return gnx_smooth_predict_dev(m, f64 ? ctx->ws_b64.p : ctx->ws_b32.p, f64, N, d_p32, d_p64, d_lab);
}

int gnx_infer_dev(gnx_model* m, const int8_t* dX, int64_t N, int64_t ldx, float* d_p32, double* d_p64, int32_t* d_lab) {
  if (!m) return GNX_EINVAL;
  return infer_any_dev(m, dX, false, N, ldx, d_p32, d_p64, d_lab);
}

// ---- host-pointer entry points: stage, run, copy back, synchronise -------------------------------------
static int64_t hap_batch(const gnx_model* m, int64_t N, int64_t ldx) {
  // bound the staging workspaces (~1 GiB of X per batch); whole individuals per batch
  int64_t nb = ((int64_t)1 << 30) / std::max<int64_t>(ldx, 1);
  if (m->ctx->tune.host_batch > 0) nb = m->ctx->tune.host_batch;  // tests: force several batches on small inputs
  nb = std::max<int64_t>(2, nb & ~(int64_t)1);
  return std::min(N, nb);
}

int gnx_base_predict(gnx_model* m, const int8_t* X, int64_t N, int64_t ldx, float* b32, double* b64) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (N < 0 || ldx < m->info.C || (N > 0 && !X)) return fail(ctx, GNX_EINVAL, "base_predict: bad X / N / ldx");
  if (N == 0) return GNX_OK;
  GNX_BIND_DEVICE(ctx);
  const size_t WA = (size_t)m->info.W * m->info.A;
  const int64_t nb = hap_batch(m, N, ldx);
  int rc;
  if ((rc = ws_reserve(ctx, ctx->ws_x, (size_t)nb * ldx + 64)) != GNX_OK) return rc;
  if (b32 && (rc = ws_reserve(ctx, ctx->ws_b32, nb * WA * 4)) != GNX_OK) return rc;
  if (b64 && (rc = ws_reserve(ctx, ctx->ws_b64, nb * WA * 8)) != GNX_OK) return rc;
  for (int64_t n0 = 0; n0 < N; n0 += nb) {
    const int64_t n = std::min(nb, N - n0);
    const size_t xbytes = (size_t)(n - 1) * ldx + m->info.C;
    HIPCHK(ctx, hipMemcpyAsync(ctx->ws_x.p, X + n0 * ldx, xbytes, hipMemcpyHostToDevice, ctx->stream));
    rc = gnx_base_predict_dev(m, (const int8_t*)ctx->ws_x.p, n, ldx, b32 ? (float*)ctx->ws_b32.p : nullptr,
                              b64 ? (double*)ctx->ws_b64.p : nullptr);
    if (rc != GNX_OK) return rc;
    if (b32) HIPCHK(ctx, hipMemcpyAsync(b32 + n0 * WA, ctx->ws_b32.p, n * WA * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (b64) HIPCHK(ctx, hipMemcpyAsync(b64 + n0 * WA, ctx->ws_b64.p, n * WA * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return GNX_OK;
}

int gnx_smooth_predict(gnx_model* m, const void* B, int b_is_f64, int64_t N, float* p32, double* p64, int32_t* lab) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (N < 0 || (N > 0 && !B)) return fail(ctx, GNX_EINVAL, "smooth_predict: bad B / N");
  if (N == 0) return GNX_OK;
  GNX_BIND_DEVICE(ctx);
  const size_t WA = (size_t)m->info.W * m->info.A, Wn = (size_t)m->info.W;
  const size_t esz = b_is_f64 ? 8 : 4;
  int rc;
  gnx_devbuf& wb = b_is_f64 ? ctx->ws_b64 : ctx->ws_b32;
  if ((rc = ws_reserve(ctx, wb, N * WA * esz)) != GNX_OK) return rc;
  const bool xgb = m->info.smooth_kind == GNX_SMOOTH_XGB;
  if ((p32 || xgb) && (rc = ws_reserve(ctx, ctx->ws_p32, N * WA * 4)) != GNX_OK) return rc;
  if (p64 && (rc = ws_reserve(ctx, ctx->ws_p64, N * WA * 8)) != GNX_OK) return rc;
  if (lab && (rc = ws_reserve(ctx, ctx->ws_lab, N * Wn * 4)) != GNX_OK) return rc;
  HIPCHK(ctx, hipMemcpyAsync(wb.p, B, N * WA * esz, hipMemcpyHostToDevice, ctx->stream));
  rc = gnx_smooth_predict_dev(m, wb.p, b_is_f64, N, (p32 || xgb) ? (float*)ctx->ws_p32.p : nullptr,
                              p64 ? (double*)ctx->ws_p64.p : nullptr, lab ? (int32_t*)ctx->ws_lab.p : nullptr);
  if (rc != GNX_OK) return rc;
  if (p32) HIPCHK(ctx, hipMemcpyAsync(p32, ctx->ws_p32.p, N * WA * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (p64) HIPCHK(ctx, hipMemcpyAsync(p64, ctx->ws_p64.p, N * WA * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (lab) HIPCHK(ctx, hipMemcpyAsync(lab, ctx->ws_lab.p, N * Wn * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GNX_OK;
}

// ---- host-pointer inference: H2D of batch i+1, kernels of batch i and D2H of batch i-1 run on three streams ------------
static int pipe_init(gnx_ctx* ctx) {
  if (ctx->s_in) return GNX_OK;
  HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking));
  HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking));
  for (int b = 0; b < 2; ++b) {
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_in[b], hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_done[b], hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_out[b], hipEventDisableTiming));
  }
  return GNX_OK;
}

// src = X (N, ld) int8, or with `packed` the 2-bit matrix (N, ld) of gnx_pack_x.  Batches alternate between the two halves of
// the staging workspaces; with page-locked host memory (gnx_host_alloc) the three streams genuinely overlap, with pageable
// memory the runtime's own staging serialises part of it (still correct).
static int infer_host(gnx_model* m, const void* src, bool packed, int64_t N, int64_t ld, float* p32, double* p64, int32_t* lab) {
  gnx_ctx* ctx = m->ctx;
  const int64_t C = m->info.C;
  GNX_BIND_DEVICE(ctx);
  const size_t WA = (size_t)m->info.W * m->info.A, Wn = (size_t)m->info.W;
  const int64_t row_bytes = packed ? (C + 3) / 4 : C;
  // batch: whole individuals, ~1 GiB of staged input at most; when overlapping, at least ~4 batches so the pipeline fills
  int64_t nb = ((int64_t)1 << 30) / std::max<int64_t>(ld, 1);
  const bool overlap = ctx->tune.h2d_overlap != 0;
  if (overlap && N >= 2048) nb = std::min<int64_t>(nb, std::max<int64_t>(512, (N + 3) / 4));
  if (ctx->tune.host_batch > 0) nb = ctx->tune.host_batch;
  nb = std::max<int64_t>(2, nb & ~(int64_t)1);
  nb = std::min(N + (N & 1), nb);
  const int nbuf = (overlap && N > nb) ? 2 : 1;
  int rc;
  if (nbuf == 2 && (rc = pipe_init(ctx)) != GNX_OK) return rc;
  const int64_t ldx_dev = packed ? ((C + 15) / 16) * 16 : ld;  // unpacked rows get a 16-byte aligned stride
  const size_t in_bytes = (((size_t)nb * ld + 64) + 255) & ~(size_t)255;
  gnx_devbuf& wsin = packed ? ctx->ws_pk : ctx->ws_x;
  if ((rc = ws_reserve(ctx, wsin, in_bytes * nbuf)) != GNX_OK) return rc;
  const bool p2 = packed && lr_p2_usable(m);  // 2-bit rows go straight into the logistic pass
  if (packed && !p2 && (rc = ws_reserve(ctx, ctx->ws_xu, (size_t)nb * ldx_dev + 256)) != GNX_OK) return rc;
  const size_t p32_b = (nb * WA * 4 + 255) & ~(size_t)255, p64_b = (nb * WA * 8 + 255) & ~(size_t)255, lab_b = (nb * Wn * 4 + 255) & ~(size_t)255;
  if ((rc = ws_reserve(ctx, ctx->ws_p32, p32_b * nbuf)) != GNX_OK) return rc;
  if (p64 && (rc = ws_reserve(ctx, ctx->ws_p64, p64_b * nbuf)) != GNX_OK) return rc;
  if (lab && (rc = ws_reserve(ctx, ctx->ws_lab, lab_b * nbuf)) != GNX_OK) return rc;
  // the workspaces gnx_infer_dev grows (B, margins, last row) must not be reallocated while a copy stream is busy: size them now
  {
    const bool f64 = (m->info.smooth_kind == GNX_SMOOTH_CRF);
    if ((rc = ws_reserve(ctx, f64 ? ctx->ws_b64 : ctx->ws_b32, nb * WA * (f64 ? 8 : 4))) != GNX_OK) return rc;
  }
  hipStream_t sc = ctx->stream, si = nbuf == 2 ? ctx->s_in : ctx->stream, so = nbuf == 2 ? ctx->s_out : ctx->stream;
  const uint8_t* hsrc = (const uint8_t*)src;
  const int64_t n_batches = (N + nb - 1) / nb;
  auto issue_h2d = [&](int64_t i) -> int {
    const int b = (int)(i % nbuf);
    const int64_t n0 = i * nb, n = std::min(nb, N - n0);
    if (nbuf == 2 && i >= 2) HIPCHK(ctx, hipStreamWaitEvent(si, ctx->ev_done[b], 0));  // kernels of batch i-2 have read this half
    HIPCHK(ctx, hipMemcpyAsync((char*)wsin.p + (size_t)b * in_bytes, hsrc + (size_t)n0 * ld, (size_t)(n - 1) * ld + row_bytes,
                               hipMemcpyHostToDevice, si));
    if (nbuf == 2) HIPCHK(ctx, hipEventRecord(ctx->ev_in[b], si));
    return GNX_OK;
  };
  if ((rc = issue_h2d(0)) != GNX_OK) return rc;
  for (int64_t i = 0; i < n_batches; ++i) {
    const int b = (int)(i % nbuf);
    const int64_t n0 = i * nb, n = std::min(nb, N - n0);
    if (nbuf == 2) {
      HIPCHK(ctx, hipStreamWaitEvent(sc, ctx->ev_in[b], 0));
      if (i >= 2) HIPCHK(ctx, hipStreamWaitEvent(sc, ctx->ev_out[b], 0));  // outputs of batch i-2 have left this half
    }
    const int8_t* dX = (const int8_t*)((char*)wsin.p + (size_t)b * in_bytes);
    int64_t ldx = ld;
    if (packed && !p2) {
      HIPCHK(ctx, gnx_launch_unpack2((const uint8_t*)dX, n, ld, C, (int8_t*)ctx->ws_xu.p, ldx_dev, sc));
      dX = (const int8_t*)ctx->ws_xu.p;
      ldx = ldx_dev;
    }
    float* dp32 = (float*)((char*)ctx->ws_p32.p + (size_t)b * p32_b);
    double* dp64 = p64 ? (double*)((char*)ctx->ws_p64.p + (size_t)b * p64_b) : nullptr;
    int32_t* dlab = lab ? (int32_t*)((char*)ctx->ws_lab.p + (size_t)b * lab_b) : nullptr;
    if ((rc = infer_any_dev(m, dX, p2, n, ldx, dp32, dp64, dlab)) != GNX_OK) return rc;
    if (nbuf == 2) HIPCHK(ctx, hipEventRecord(ctx->ev_done[b], sc));
    // the next batch's input goes out BEFORE this batch's outputs are awaited (pageable D2H blocks the host thread)
    if (i + 1 < n_batches && (rc = issue_h2d(i + 1)) != GNX_OK) return rc;
    if (nbuf == 2) HIPCHK(ctx, hipStreamWaitEvent(so, ctx->ev_done[b], 0));
    if (p32) HIPCHK(ctx, hipMemcpyAsync(p32 + n0 * WA, dp32, n * WA * 4, hipMemcpyDeviceToHost, so));
    if (p64) HIPCHK(ctx, hipMemcpyAsync(p64 + n0 * WA, dp64, n * WA * 8, hipMemcpyDeviceToHost, so));
    if (lab) HIPCHK(ctx, hipMemcpyAsync(lab + n0 * Wn, dlab, n * Wn * 4, hipMemcpyDeviceToHost, so));
    if (nbuf == 2) HIPCHK(ctx, hipEventRecord(ctx->ev_out[b], so));
    else HIPCHK(ctx, hipStreamSynchronize(sc));
  }
  if (nbuf == 2) {
    HIPCHK(ctx, hipStreamSynchronize(so));
    HIPCHK(ctx, hipStreamSynchronize(si));
  }
  HIPCHK(ctx, hipStreamSynchronize(sc));
  return GNX_OK;
}

int gnx_infer(gnx_model* m, const int8_t* X, int64_t N, int64_t ldx, float* p32, double* p64, int32_t* lab) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (N < 0 || ldx < m->info.C || (N > 0 && !X)) return fail(ctx, GNX_EINVAL, "infer: bad X / N / ldx");
  if (N == 0) return GNX_OK;
  return infer_host(m, X, false, N, ldx, p32, p64, lab);
}

// ---- 2-bit packed input ------------------------------------------------------------------------------------------------
int64_t gnx_packed_row_bytes(int64_t C) { return C <= 0 ? 0 : ((C + 15) / 16) * 4; }

int gnx_pack_x(const int8_t* X, int64_t N, int64_t ldx, int64_t C, uint8_t* P, int64_t ldp, int n_threads) {
  if (N < 0 || C <= 0 || ldx < C || ldp < (C + 3) / 4 || (N > 0 && (!X || !P))) return GNX_EINVAL;
  if (N == 0) return GNX_OK;
  int nt = n_threads > 0 ? n_threads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u);
  nt = (int)std::min<int64_t>(nt, std::max<int64_t>(1, N / 16));
  std::vector<int> bad((size_t)nt, 0);
  auto work = [&](int t) {
    const int64_t r0 = N * t / nt, r1 = N * (t + 1) / nt;
    for (int64_t n = r0; n < r1; ++n) {
      const int8_t* x = X + n * ldx;
      uint8_t* p = P + n * ldp;
      int64_t j = 0;
      uint64_t flag = 0;
      for (; j + 8 <= C; j += 8) {  // 8 SNPs -> 2 bytes
        uint64_t v;
        std::memcpy(&v, x + j, 8);
        flag |= v & 0xFCFCFCFCFCFCFCFCull;
        v |= v >> 6;                      // byte pairs: fields 0,1 in the low nibble of bytes 0, 2, 4, 6
        v &= 0x000F000F000F000Full;
        v |= v >> 12;                     // nibble pairs: one full byte in bytes 0 and 4
        const uint16_t o = (uint16_t)((v & 0xFFu) | ((v >> 24) & 0xFF00u));
        std::memcpy(p + (j >> 2), &o, 2);
      }
      for (; j < C; j += 4) {
        uint8_t o = 0;
        for (int k = 0; k < 4 && j + k < C; ++k) {
          const uint8_t v = (uint8_t)x[j + k];
          flag |= v & 0xFCu;
          o |= (uint8_t)((v & 3u) << (2 * k));
        }
        p[j >> 2] = o;
      }
      for (int64_t b = (C + 3) / 4; b < std::min<int64_t>(ldp, gnx_packed_row_bytes(C)); ++b) p[b] = 0;  // canonical stride: zero tail
      if (flag) bad[(size_t)t] = 1;
    }
  };
  if (nt == 1) work(0);
  else {
    // no exception crosses the ABI: a thread that cannot be started (resource limits) is run inline instead
    std::vector<std::thread> th;
    th.reserve((size_t)nt);
    for (int t = 0; t < nt; ++t) {
      try {
        th.emplace_back(work, t);
      } catch (...) {
        work(t);
      }
    }
    for (auto& t : th) t.join();
  }
  for (int b : bad)
    if (b) return GNX_EINVAL;  // a value outside {0,1,2,3}: not representable in 2 bits
  return GNX_OK;
}

int gnx_unpack_x_dev(gnx_ctx* ctx, const uint8_t* dP, int64_t N, int64_t ldp, int64_t C, int8_t* dX, int64_t ldx) {
  if (!ctx) return GNX_EINVAL;
  if (N < 0 || C <= 0 || ldx < C || ldp < (C + 3) / 4 || (N > 0 && (!dP || !dX))) return fail(ctx, GNX_EINVAL, "unpack_x: bad arguments");
  GNX_BIND_DEVICE(ctx);
  HIPCHK(ctx, gnx_launch_unpack2(dP, N, ldp, C, dX, ldx, ctx->stream));
  return GNX_OK;
}

int gnx_infer_packed(gnx_model* m, const uint8_t* P, int64_t N, int64_t ldp, float* p32, double* p64, int32_t* lab) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (N < 0 || ldp < (m->info.C + 3) / 4 || (N > 0 && !P)) return fail(ctx, GNX_EINVAL, "infer_packed: bad P / N / ldp");
  if (N == 0) return GNX_OK;
  return infer_host(m, P, true, N, ldp, p32, p64, lab);
}

int gnx_infer_packed_dev(gnx_model* m, const uint8_t* dP, int64_t N, int64_t ldp, float* d_p32, double* d_p64, int32_t* d_lab) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (N < 0 || ldp < (m->info.C + 3) / 4 || (N > 0 && !dP)) return fail(ctx, GNX_EINVAL, "infer_packed: bad P / N / ldp");
  if (N == 0) return GNX_OK;
  return infer_any_dev(m, dP, true, N, ldp, d_p32, d_p64, d_lab);
}

int gnx_smooth_rows(gnx_model* m, const float* rows, int64_t R, float* proba) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (m->info.smooth_kind != GNX_SMOOTH_XGB) return fail(ctx, GNX_ESTATE, "smooth_rows needs the XGB smoother");
  if (R < 0 || (R > 0 && (!rows || !proba))) return fail(ctx, GNX_EINVAL, "smooth_rows: bad arguments");
  if (R == 0) return GNX_OK;
  GNX_BIND_DEVICE(ctx);
  const int F = m->info.S * m->info.A, A = m->info.A;
  int rc;
  if ((rc = ws_reserve(ctx, ctx->ws_misc, (size_t)R * F * 4)) != GNX_OK) return rc;
  if ((rc = ws_reserve(ctx, ctx->ws_p32, (size_t)R * A * 4)) != GNX_OK) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->ws_misc.p, rows, (size_t)R * F * 4, hipMemcpyHostToDevice, ctx->stream));
  {
    ProfScope ps(ctx, GNX_K_SMOOTH_ROWS);
    HIPCHK(ctx, gnx_launch_smooth_rows(m->xgb, (const float*)ctx->ws_misc.p, R, F, A, (float*)ctx->ws_p32.p, ctx->stream));
  }
  HIPCHK(ctx, hipMemcpyAsync(proba, ctx->ws_p32.p, (size_t)R * A * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GNX_OK;
}

int gnx_calibrate_rows(gnx_model* m, const void* proba, int proba_is_f64, int64_t R, double* out) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (!m->calib_off) return fail(ctx, GNX_ESTATE, "model has no calibrator");
  if (R < 0 || (R > 0 && (!proba || !out))) return fail(ctx, GNX_EINVAL, "calibrate_rows: bad arguments");
  if (R == 0) return GNX_OK;
  GNX_BIND_DEVICE(ctx);
  const size_t bytes = (size_t)R * m->info.A * sizeof(double);
  const size_t in_bytes = (size_t)R * m->info.A * (proba_is_f64 ? 8 : 4);
  int rc = ws_reserve(ctx, ctx->ws_cal, bytes);
  if (rc != GNX_OK) return rc;
  if ((rc = ws_reserve(ctx, ctx->ws_misc, in_bytes)) != GNX_OK) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->ws_misc.p, proba, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  CalibLaunch L{};
  L.in = ctx->ws_misc.p; L.in_is_f64 = proba_is_f64 ? 1 : 0; L.R = R; L.A = m->info.A;
  L.off = m->calib_off; L.x = m->calib_x; L.y = m->calib_y; L.thr_f32 = m->calib_f32; L.out64 = (double*)ctx->ws_cal.p;
  {
    ProfScope ps(ctx, GNX_K_CALIBRATE);
    HIPCHK(ctx, gnx_launch_calibrate(L, ctx->stream));
  }
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->ws_cal.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GNX_OK;
}

// which Gnofix kernel a model runs: the rank-strip kernel (k_gnofix.hip) wherever the smoother has a rank copy, else (or with
// GNX_GNOFIX_IMPL=f32) the float32-strip kernel of rounds 1-3 (k_gnofix_f32.hip)
static int gnofix_threads(const gnx_model* m) {
  const int t = m->ctx->tune.gnofix_threads;
  return (t == 256 || t == 512 || t == 1024) ? t : 512;
}
static bool gnofix_use_rk(const gnx_model* m) {
  if (!(m->xgb.gf_packed && m->xgb.rk_thr && m->ctx->tune.gnofix_impl != 1 && m->info.C < ((int64_t)1 << 31))) return false;
  // the rank kernel keeps ~10 bytes per window next to the staged trees: where that does not fit the 160 KB (W beyond ~10 k, very
  // fine windows) the float32-strip kernel with its global-scratch strips still runs (it did before the rank kernel existed)
  const int W = (int)m->info.W, A = m->info.A, S = m->info.S, T = gnofix_threads(m);
  return gnx_gnofix_lds_bytes(W, A, S, m->xgb.gf_pitch, gnx_gnofix_cap(m->xgb.gf_max_class, m->xgb.D, S, T), m->xgb.D, T, m->xgb.n_trees) <= (size_t)160 * 1024;
}

static int gnofix_check(gnx_model* m, int64_t ldx, int64_t n_ind, int32_t max_it, bool ptrs_ok, bool* in_lds) {
  gnx_ctx* ctx = m->ctx;
  // src/model.py:194: only a smoother with .gnofix == True (XGB_Smoother) supports re-phasing
  if (m->info.smooth_kind != GNX_SMOOTH_XGB)
    return fail(ctx, GNX_ESTATE, "Type of Smoother does not currently support re-phasing");
  if (n_ind < 0 || ldx < m->info.C || max_it < 0 || (n_ind > 0 && !ptrs_ok)) return fail(ctx, GNX_EINVAL, "gnofix: bad arguments");
  if (m->calibrate_on && m->calib_off)  // smoother.predict inside the loop would be calibrated (gnofix.py:80,190); the kernel's is not
    return fail(ctx, GNX_EUNSUPPORTED, "gnofix with calibrate=True is not built: switch calibration off for re-phasing");
  const int W = (int)m->info.W, A = m->info.A, S = m->info.S;
  *in_lds = true;
  if (gnofix_use_rk(m)) {
    if (gnx_gnofix_lds_bytes(W, A, S, m->xgb.gf_pitch, gnx_gnofix_cap(m->xgb.gf_max_class, m->xgb.D, S, gnofix_threads(m)), m->xgb.D, gnofix_threads(m), m->xgb.n_trees) > (size_t)160 * 1024)
      return fail(ctx, GNX_EUNSUPPORTED, "gnofix: W too large for the LDS working set (labels: 2 bytes per window)");
    return GNX_OK;
  }
  *in_lds = gnx_gnofix_f32_lds_bytes(W, A, S, m->xgb.n_trees, m->xgb.tree_bytes, true) <= 150 * 1024;
  if (gnx_gnofix_f32_lds_bytes(W, A, S, m->xgb.n_trees, m->xgb.tree_bytes, *in_lds) > 160 * 1024)
    return fail(ctx, GNX_EUNSUPPORTED, "gnofix: model too large for the LDS working set (n_trees * 16 B + S*A*8 B)");
  return GNX_OK;
}

// scratch of one device-resident batch of n individuals (grows only).  ws_misc: [hist | par | dif | ranks] (rank kernel) or
// [hist | float32 strips] (fallback, strips that do not fit the LDS)
struct GnofixWs { size_t par = 0, dif = 0, rk = 0, pm = 0, ord = 0, bp = 0; };
static int gnofix_ws_reserve(gnx_model* m, int64_t n, int32_t max_it, bool in_lds, GnofixWs* out) {
  gnx_ctx* ctx = m->ctx;
  const int W = (int)m->info.W, A = m->info.A, S = m->info.S, pad = (S + 1) / 2;
  const size_t WA = (size_t)W * A, NWD = (size_t)(W + 31) / 32;
  int rc;
  if ((rc = ws_reserve(ctx, ctx->ws_p32, (size_t)2 * n * WA * 4)) != GNX_OK) return rc;
  if ((rc = ws_reserve(ctx, ctx->ws_y0, (size_t)2 * n * W * 4)) != GNX_OK) return rc;
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  size_t misc = up((size_t)n * std::max(max_it, 1) * NWD * 4);
  GnofixWs o;
  if (gnofix_use_rk(m)) {
    o.par = misc; misc += up((size_t)n * NWD * 4);
    o.dif = misc; misc += up((size_t)n * NWD * 4);
    o.rk = misc; misc += up((size_t)2 * n * WA * 2);
    o.pm = misc; misc += up((size_t)2 * n * W * 4);
    o.ord = misc; misc += up(((size_t)2 * n + 2 * ((size_t)W + 1)) * 4);
  } else if (!in_lds) {
    o.bp = misc; misc += (size_t)n * 2 * (W + 2 * pad) * A * 4;
  }
  if (out) *out = o;
  return ws_reserve(ctx, ctx->ws_misc, misc + 256);
}

// device-resident batch: initial labels with the batched smoother kernel, then one workgroup per individual
// side_stream: run the input-only pre-passes beside the smoother.  Not from the host-pointer pipeline: a fifth stream makes two of
// them share a hardware queue (HIP maps streams onto 4 by default), and when those two are the copy-in and copy-out streams the
// pipeline's H2D and D2H stop overlapping (measured: 13.3 k -> 9.2 k individuals/s through host pointers).
static int gnofix_run_dev(gnx_model* m, int8_t* dX, int64_t ldx, const double* dB, int64_t n, int32_t max_it, int32_t* dY,
                          int32_t* dNs, bool in_lds, bool side_stream, bool packed = false) {
  gnx_ctx* ctx = m->ctx;
  GNX_BIND_DEVICE(ctx);
  const int W = (int)m->info.W, A = m->info.A, S = m->info.S;
  int rc;
  GnofixWs ws;
  if ((rc = gnofix_ws_reserve(m, n, max_it, in_lds, &ws)) != GNX_OK) return rc;
  int32_t* dY0 = (int32_t*)ctx->ws_y0.p;
  GnofixLaunch L{};
  L.X = dX; L.ldx = ldx; L.C = m->info.C; L.B = dB; L.Y0 = dY0; L.Yout = dY; L.n_switches = dNs;
  L.x_packed = packed ? 1 : 0;
  L.W = W; L.A = A; L.S = S; L.max_it = max_it; L.d = m->xgb; L.class_tree0 = m->class_tree0;
  L.hist = (uint32_t*)ctx->ws_misc.p;
  const bool rk = gnofix_use_rk(m);
  if (rk) {
    char* base = (char*)ctx->ws_misc.p;
    L.proba0 = (const float*)ctx->ws_p32.p; L.pmax0 = (const float*)(base + ws.pm);
    L.order = (const int32_t*)(base + ws.ord);
    L.par = (uint32_t*)(base + ws.par); L.dif = (const uint32_t*)(base + ws.dif); L.R = (const uint16_t*)(base + ws.rk);
    L.gf = m->xgb.gf_packed; L.gf_pitch = m->xgb.gf_pitch; L.gf_cap = gnx_gnofix_cap(m->xgb.gf_max_class, m->xgb.D, S, gnofix_threads(m));
    // ranks of B and the SNP-difference masks of X need nothing of the smoother: they run beside it on the side stream
    side_stream = side_stream && ctx->tune.gnofix_aux;
    if (side_stream && !ctx->s_aux) {
      HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->s_aux, hipStreamNonBlocking));
      for (int b = 0; b < 2; ++b) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_aux[b], hipEventDisableTiming));
    }
    if (side_stream) {
      HIPCHK(ctx, hipEventRecord(ctx->ev_aux[0], ctx->stream));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->s_aux, ctx->ev_aux[0], 0));
      HIPCHK(ctx, gnx_launch_gnofix_prep(L, n, ctx->s_aux));
      HIPCHK(ctx, hipEventRecord(ctx->ev_aux[1], ctx->s_aux));
    } else {
      HIPCHK(ctx, gnx_launch_gnofix_prep(L, n, ctx->stream));
    }
  }
  // initial labels = smoother.predict(B) for every haplotype at once (gnofix.py:80)
  rc = gnx_smooth_predict_dev(m, dB, 1, 2 * n, (float*)ctx->ws_p32.p, nullptr, dY0);
  if (rk && side_stream) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_aux[1], 0));  // (also on failure: the side stream is joined)
  if (rc != GNX_OK) return rc;
  ProfScope ps(ctx, GNX_K_GNOFIX);
  if (rk) {
    HIPCHK(ctx, gnx_launch_gnofix(L, n, gnofix_threads(m), ctx->stream));
    return GNX_OK;
  }
  L.bp_in_lds = in_lds ? 1 : 0;
  L.bp_scratch = in_lds ? nullptr : (float*)((char*)ctx->ws_misc.p + ws.bp);
  HIPCHK(ctx, gnx_launch_gnofix_f32(L, n, ctx->stream));
  return GNX_OK;
}

int gnx_gnofix_packed_dev(gnx_model* m, uint8_t* dP, int64_t ldp, const double* dB, int64_t n_ind, int32_t max_it, int32_t* dY,
                          int32_t* d_n_switches) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (ldp < (m->info.C + 3) / 4 || (ldp & 3) || ((uintptr_t)dP & 3))
    return fail(ctx, GNX_EINVAL, "gnofix_packed: rows must be 4-byte aligned, ldp a multiple of 4 and >= ceil(C / 4)");
  bool in_lds = true;
  int rc = gnofix_check(m, m->info.C, n_ind, max_it, dP && dB && dY, &in_lds);
  if (rc != GNX_OK || n_ind == 0) return rc;
  if (!gnofix_use_rk(m)) return fail(ctx, GNX_EUNSUPPORTED, "gnofix_packed: the model's smoother has no rank-quantised copy (k_gnofix); use gnx_gnofix_dev");
  return gnofix_run_dev(m, reinterpret_cast<int8_t*>(dP), ldp, dB, n_ind, max_it, dY, d_n_switches, in_lds, true, true);
}

int gnx_gnofix_dev(gnx_model* m, int8_t* dX, int64_t ldx, const double* dB, int64_t n_ind, int32_t max_it, int32_t* dY,
                   int32_t* d_n_switches) {
  if (!m) return GNX_EINVAL;
  bool in_lds = true;
  int rc = gnofix_check(m, ldx, n_ind, max_it, dX && dB && dY, &in_lds);
  if (rc != GNX_OK || n_ind == 0) return rc;
  return gnofix_run_dev(m, dX, ldx, dB, n_ind, max_it, dY, d_n_switches, in_lds, true);
}

int gnx_gnofix(gnx_model* m, int8_t* X, int64_t ldx, const double* B, int64_t n_ind, int32_t max_it, int32_t* Y,
               int32_t* n_switches) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  bool in_lds = true;
  int rc = gnofix_check(m, ldx, n_ind, max_it, X && B && Y, &in_lds);
  if (rc != GNX_OK || n_ind == 0) return rc;
  GNX_BIND_DEVICE(ctx);
  const int W = (int)m->info.W;
  const size_t WA = (size_t)W * m->info.A;
  // Batches of whole individuals alternate between the two halves of the staging workspaces: X and B of batch i+1 go up, and X / labels
  // of batch i-1 come back, while batch i is re-phased (three streams; page-locked host memory — gnx_host_alloc — makes the overlap
  // real, pageable memory is still correct).  A batch is a multiple of the CU count near 2 GiB of X (the link, not the kernels, bounds this entry: small batches keep the pipeline's fill and drain short): the per-individual kernel keeps
  // two workgroups per CU busy and ends with its slowest individual, so a batch should hold several individuals per CU.
  const int64_t cu = std::max(ctx->n_cu, 1);
  int64_t nb = std::max<int64_t>(1, (((int64_t)2 << 30) / std::max<int64_t>(ldx, 1)) / 2);
  if (nb >= cu) nb -= nb % cu;
  if (ctx->tune.host_batch > 0) nb = std::max<int64_t>(1, ctx->tune.host_batch / 2);
  nb = std::min(nb, n_ind);
  const int nbuf = (ctx->tune.h2d_overlap != 0 && n_ind > nb) ? 2 : 1;
  if (nbuf == 2 && (rc = pipe_init(ctx)) != GNX_OK) return rc;
  const size_t x_b = (((size_t)2 * nb * ldx + 64) + 255) & ~(size_t)255, b_b = ((size_t)2 * nb * WA * 8 + 255) & ~(size_t)255;
  const size_t y_b = ((size_t)2 * nb * W * 4 + (size_t)nb * 4 + 255) & ~(size_t)255;
  if ((rc = ws_reserve(ctx, ctx->ws_x, x_b * nbuf)) != GNX_OK) return rc;
  if ((rc = ws_reserve(ctx, ctx->ws_b64, b_b * nbuf)) != GNX_OK) return rc;
  if ((rc = ws_reserve(ctx, ctx->ws_lab, y_b * nbuf)) != GNX_OK) return rc;
  // what a batch grows on the compute stream must not move while a copy stream is busy: size it now, with one individual run through
  // the smoother so that its own workspaces exist at full size too
  if ((rc = gnofix_ws_reserve(m, nb, max_it, in_lds, nullptr)) != GNX_OK) return rc;
  hipStream_t sc = ctx->stream, si = nbuf == 2 ? ctx->s_in : ctx->stream, so = nbuf == 2 ? ctx->s_out : ctx->stream;
  const int64_t n_batches = (n_ind + nb - 1) / nb;
  auto issue_h2d = [&](int64_t i) -> int {
    const int b = (int)(i % nbuf);
    const int64_t i0 = i * nb, n = std::min(nb, n_ind - i0);
    if (nbuf == 2 && i >= 2) HIPCHK(ctx, hipStreamWaitEvent(si, ctx->ev_out[b], 0));  // batch i-2 has left this half
    HIPCHK(ctx, hipMemcpyAsync((char*)ctx->ws_x.p + (size_t)b * x_b, X + 2 * i0 * ldx, (size_t)(2 * n - 1) * ldx + m->info.C,
                               hipMemcpyHostToDevice, si));
    HIPCHK(ctx, hipMemcpyAsync((char*)ctx->ws_b64.p + (size_t)b * b_b, B + (size_t)2 * i0 * WA, (size_t)2 * n * WA * 8, hipMemcpyHostToDevice, si));
    if (nbuf == 2) HIPCHK(ctx, hipEventRecord(ctx->ev_in[b], si));
    return GNX_OK;
  };
  if ((rc = issue_h2d(0)) != GNX_OK) return rc;
  for (int64_t i = 0; i < n_batches; ++i) {
    const int b = (int)(i % nbuf);
    const int64_t i0 = i * nb, n = std::min(nb, n_ind - i0);
    int8_t* dX = (int8_t*)((char*)ctx->ws_x.p + (size_t)b * x_b);
    double* dB = (double*)((char*)ctx->ws_b64.p + (size_t)b * b_b);
    int32_t* dY = (int32_t*)((char*)ctx->ws_lab.p + (size_t)b * y_b);
    int32_t* dNs = dY + (size_t)2 * nb * W;
    if (nbuf == 2) HIPCHK(ctx, hipStreamWaitEvent(sc, ctx->ev_in[b], 0));
    if ((rc = gnofix_run_dev(m, dX, ldx, dB, n, max_it, dY, dNs, in_lds, nbuf == 1)) != GNX_OK) return rc;
    if (nbuf == 2) HIPCHK(ctx, hipEventRecord(ctx->ev_done[b], sc));
    // (two halves: the next batch goes up before this one's results are awaited; one half: X is both input and output of it)
    if (nbuf == 2 && i + 1 < n_batches && (rc = issue_h2d(i + 1)) != GNX_OK) return rc;
    if (nbuf == 2) HIPCHK(ctx, hipStreamWaitEvent(so, ctx->ev_done[b], 0));
    HIPCHK(ctx, hipMemcpyAsync(X + 2 * i0 * ldx, dX, (size_t)(2 * n - 1) * ldx + m->info.C, hipMemcpyDeviceToHost, so));
    HIPCHK(ctx, hipMemcpyAsync(Y + (size_t)2 * i0 * W, dY, (size_t)2 * n * W * 4, hipMemcpyDeviceToHost, so));
    if (n_switches) HIPCHK(ctx, hipMemcpyAsync(n_switches + i0, dNs, (size_t)n * 4, hipMemcpyDeviceToHost, so));
    if (nbuf == 2) HIPCHK(ctx, hipEventRecord(ctx->ev_out[b], so));
    else {
      HIPCHK(ctx, hipStreamSynchronize(sc));
      if (i + 1 < n_batches && (rc = issue_h2d(i + 1)) != GNX_OK) return rc;
    }
  }
  if (nbuf == 2) {
    HIPCHK(ctx, hipStreamSynchronize(so));
    HIPCHK(ctx, hipStreamSynchronize(si));
  }
  HIPCHK(ctx, hipStreamSynchronize(sc));
  return GNX_OK;
}

// ---- training --------------------------------------------------------------------------------------------
int gnx_train_logistic_dev(gnx_ctx* ctx, const int8_t* dX, int64_t N, int64_t ldx, const int32_t* dy, int64_t C, int64_t M, int64_t cx,
                           int32_t A, double C_reg, double tol, int32_t max_iter, double* coef, int64_t ldc, double* intercept,
                           gnx_train_info* info) {
  if (!ctx) return GNX_EINVAL;
  if (!ctx->usable) return fail(ctx, GNX_ESTATE, "context has no device (gnx_init failed)");
  if (N <= 0 || !dX || !dy || !coef || !intercept) return fail(ctx, GNX_EINVAL, "train_logistic: bad X / y / outputs");
  if (A < 2 || A > 32) return fail(ctx, GNX_EINVAL, "A (ancestries) must be in [2, 32]");
  if (M <= 0 || C < M || cx < 0 || cx > C || ldx < C) return fail(ctx, GNX_EINVAL, "bad C / M / ctx / ldx");
  if (C % M == 0) return fail(ctx, GNX_EINVAL, "C % M == 0: the reference's window slicing (base.py:158) requires a remainder");
  if (ldc < M + 2 * cx + C % M) return fail(ctx, GNX_EINVAL, "train_logistic: ldc < M + 2*ctx + C % M");
  if (!(C_reg > 0.0) || !(tol > 0.0)) return fail(ctx, GNX_EINVAL, "train_logistic: C_reg and tol must be positive");
  GNX_BIND_DEVICE(ctx);
  const int newton = max_iter > 0 ? std::min(max_iter, 200) : 100;
  HIPCHK(ctx, gnx_train_lr_run(dX, N, ldx, dy, C, M, cx, A, C_reg, tol, newton, 250, coef, ldc, intercept, info, ctx->stream));
  return GNX_OK;
}

int gnx_train_logistic(gnx_ctx* ctx, const int8_t* X, int64_t N, int64_t ldx, const int32_t* y, int64_t C, int64_t M, int64_t cx, int32_t A,
                       double C_reg, double tol, int32_t max_iter, double* coef, int64_t ldc, double* intercept, gnx_train_info* info) {
  if (!ctx) return GNX_EINVAL;
  if (!ctx->usable) return fail(ctx, GNX_ESTATE, "context has no device (gnx_init failed)");
  if (N <= 0 || !X || !y || M <= 0 || C < M || ldx < C) return fail(ctx, GNX_EINVAL, "train_logistic: bad X / y / geometry");
  GNX_BIND_DEVICE(ctx);
  const int64_t W = C / M;
  int rc;
  if ((rc = ws_reserve(ctx, ctx->ws_x, (size_t)N * ldx + 64)) != GNX_OK) return rc;
  if ((rc = ws_reserve(ctx, ctx->ws_lab, (size_t)N * W * 4)) != GNX_OK) return rc;
  for (int64_t i = 0; i < N * W; ++i)
    if (y[i] < 0 || y[i] >= A) return fail(ctx, GNX_EINVAL, "train_logistic: label outside [0, A)");
  HIPCHK(ctx, hipMemcpyAsync(ctx->ws_x.p, X, (size_t)(N - 1) * ldx + C, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->ws_lab.p, y, (size_t)N * W * 4, hipMemcpyHostToDevice, ctx->stream));
  return gnx_train_logistic_dev(ctx, (const int8_t*)ctx->ws_x.p, N, ldx, (const int32_t*)ctx->ws_lab.p, C, M, cx, A, C_reg, tol, max_iter,
                                coef, ldc, intercept, info);
}

// ---- training the tree smoother ---------------------------------------------------------------------------------
static int gbt_check(gnx_ctx* ctx, const void* B, const int32_t* y, int64_t N, int32_t W, int32_t A, int32_t S, const gnx_gbt_params* P,
                     const void* o1, const void* o2, const void* o3, const void* o4, const void* o5, const void* o6, const void* o7) {
  if (!ctx->usable) return fail(ctx, GNX_ESTATE, "context has no device (gnx_init failed)");
  if (N <= 0 || !B || !y || !P || !o1 || !o2 || !o3 || !o4 || !o5 || !o6 || !o7) return fail(ctx, GNX_EINVAL, "train_gbt: bad B / y / params / outputs");
  if (A < 2 || A > 32) return fail(ctx, GNX_EINVAL, "A (ancestries) must be in [2, 32]");
  if (S <= 0 || S % 2 == 0) return fail(ctx, GNX_EINVAL, "S must be odd and positive (smooth.py:14)");
  if (W < 2 * S) return fail(ctx, GNX_EINVAL, "Smoother size to large for given window size. ");  // src/Smooth/models.py:13
  if (P->n_rounds < 1 || P->n_rounds > 100000 || P->max_depth < 1 || P->max_depth > 5 || P->max_bin < 2 || P->max_bin > 256)
    return fail(ctx, GNX_EINVAL, "train_gbt: n_rounds >= 1, 1 <= max_depth <= 5, 2 <= max_bin <= 256");
  if (P->tree_method != 0 && P->tree_method != 1) return fail(ctx, GNX_EINVAL, "train_gbt: tree_method is 0 (histogram) or 1 (exact greedy)");
  if (!(P->eta > 0.0) || !(P->lambda >= 0.0) || !(P->gamma >= 0.0) || !(P->min_child_weight >= 0.0))
    return fail(ctx, GNX_EINVAL, "train_gbt: eta > 0, lambda / gamma / min_child_weight >= 0");
  // rows are indexed in int32, the exact-greedy sort over the PADDED windows of a haplotype (W + 2 * ((S + 1) / 2)) as well
  if ((int64_t)N * ((int64_t)W + 2 * ((S + 1) / 2)) >= ((int64_t)1 << 31)) return fail(ctx, GNX_EINVAL, "train_gbt: N * (W + S + 1) must stay below 2^31 rows");
  return GNX_OK;
}

int gnx_train_gbt_dev(gnx_ctx* ctx, const void* dB, int32_t b_is_f64, const int32_t* dy, int64_t N, int32_t W, int32_t A, int32_t S,
                      const gnx_gbt_params* P, int32_t* tree_off, int32_t* tree_class, int32_t* left, int32_t* right, int32_t* feat,
                      float* cond, int64_t* n_nodes, double* loss) {
  if (!ctx) return GNX_EINVAL;
  int rc = gbt_check(ctx, dB, dy, N, W, A, S, P, tree_off, tree_class, left, right, feat, cond, n_nodes);
  if (rc != GNX_OK) return rc;
  GNX_BIND_DEVICE(ctx);
  HIPCHK(ctx, gnx_train_gbt_run(dB, b_is_f64, dy, N, W, A, S, *P, tree_off, tree_class, left, right, feat, cond, n_nodes, loss, ctx->n_cu,
                                ctx->stream));
  return GNX_OK;
}

int gnx_train_gbt(gnx_ctx* ctx, const void* B, int32_t b_is_f64, const int32_t* y, int64_t N, int32_t W, int32_t A, int32_t S,
                  const gnx_gbt_params* P, int32_t* tree_off, int32_t* tree_class, int32_t* left, int32_t* right, int32_t* feat,
                  float* cond, int64_t* n_nodes, double* loss) {
  if (!ctx) return GNX_EINVAL;
  int rc = gbt_check(ctx, B, y, N, W, A, S, P, tree_off, tree_class, left, right, feat, cond, n_nodes);
  if (rc != GNX_OK) return rc;
  for (int64_t i = 0; i < N * W; ++i)
    if (y[i] < 0 || y[i] >= A) return fail(ctx, GNX_EINVAL, "train_gbt: label outside [0, A)");
  GNX_BIND_DEVICE(ctx);
  const size_t nb = (size_t)N * W * A * (b_is_f64 ? 8 : 4), ny = (size_t)N * W * 4;
  gnx_devbuf& wsB = b_is_f64 ? ctx->ws_b64 : ctx->ws_b32;
  if ((rc = ws_reserve(ctx, wsB, nb)) != GNX_OK) return rc;
  if ((rc = ws_reserve(ctx, ctx->ws_lab, ny)) != GNX_OK) return rc;
  HIPCHK(ctx, hipMemcpyAsync(wsB.p, B, nb, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->ws_lab.p, y, ny, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, gnx_train_gbt_run(wsB.p, b_is_f64, (const int32_t*)ctx->ws_lab.p, N, W, A, S, *P, tree_off, tree_class, left, right, feat, cond,
                                n_nodes, loss, ctx->n_cu, ctx->stream));
  return GNX_OK;
}

// ---- fitting one isotonic map of the calibrator (host arithmetic: a sort and two linear passes over a few thousand points) -------
// sklearn.isotonic.IsotonicRegression(out_of_bounds="clip").fit(x, y) as the reference's Calibrator.fit calls it per class
// (src/Smooth/Calibration.py:55) on float32 probabilities: sort by (x, y); merge x closer than float32's resolution (1e-6) to the
// first x of their group, y = float32 running mean; pool adjacent violators (block means in float64, as scipy's PAVA behind
// sklearn >= 1.4 does, result cast to float32); drop interior points whose y equals both neighbours'.
int gnx_fit_isotonic_f32(const float* x, const float* y, int64_t n, float* x_thr, float* y_thr, int64_t* n_thr) {
  if (n <= 0 || !x || !y || !x_thr || !y_thr || !n_thr) return GNX_EINVAL;
  try {
    std::vector<int64_t> order((size_t)n);
    for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return x[a] < x[b] || (x[a] == x[b] && y[a] < y[b]); });
    // _make_unique (sklearn/_isotonic.pyx), float32 throughout
    std::vector<float> ux, uy, uw;
    const float eps = 1e-6f;  // np.finfo(np.float32).resolution
    float cx = x[order[0]], cy = 0.f, cw = 0.f;
    for (int64_t j = 0; j < n; ++j) {
      const float xv = x[order[(size_t)j]], yv = y[order[(size_t)j]];
      if (xv - cx >= eps) {
        ux.push_back(cx); uw.push_back(cw); uy.push_back(cy / cw);
        cx = xv; cw = 1.f; cy = yv * 1.f;
      } else {
        cw += 1.f;
        cy += yv * 1.f;
      }
    }
    ux.push_back(cx); uw.push_back(cw); uy.push_back(cy / cw);
    // pool adjacent violators on (uy, uw): blocks as (weighted sum, weight, first index), means in float64
    const size_t m = ux.size();
    std::vector<double> bs(m), bw(m);
    std::vector<size_t> b0(m);
    size_t nb = 0;
    for (size_t i = 0; i < m; ++i) {
      bs[nb] = (double)uw[i] * (double)uy[i]; bw[nb] = (double)uw[i]; b0[nb] = i; ++nb;
      while (nb > 1 && bs[nb - 2] / bw[nb - 2] > bs[nb - 1] / bw[nb - 1]) {
        bs[nb - 2] += bs[nb - 1]; bw[nb - 2] += bw[nb - 1]; --nb;
      }
    }
    std::vector<float> fy(m);
    for (size_t b = 0; b < nb; ++b) {
      const float v = (float)(bs[b] / bw[b]);
      const size_t e = (b + 1 < nb) ? b0[b + 1] : m;
      for (size_t i = b0[b]; i < e; ++i) fy[i] = v;
    }
    int64_t k = 0;
    for (size_t i = 0; i < m; ++i) {
      const bool keep = (i == 0 || i + 1 == m) ? true : (fy[i] != fy[i - 1] || fy[i] != fy[i + 1]);
      if (keep) { x_thr[k] = ux[i]; y_thr[k] = fy[i]; ++k; }
    }
    *n_thr = k;
  } catch (...) {
    return GNX_ENOMEM;
  }
  return GNX_OK;
}

// ---- profiling ---------------------------------------------------------------------------------------
int gnx_profile_enable(gnx_ctx* ctx, int on) {
  if (!ctx) return GNX_EINVAL;
  GNX_BIND_DEVICE(ctx);
  if (!on) prof_drain(ctx);
  ctx->prof = on != 0;
  return GNX_OK;
}

int gnx_profile_reset(gnx_ctx* ctx) {
  if (!ctx) return GNX_EINVAL;
  GNX_BIND_DEVICE(ctx);
  prof_drain(ctx);
  for (int k = 0; k < GNX_K_COUNT; ++k) { ctx->prof_ms[k] = 0; ctx->prof_n[k] = 0; }
  return GNX_OK;
}

int gnx_profile_get(gnx_ctx* ctx, int kid, double* total_ms, int64_t* launches) {
  if (!ctx || kid < 0 || kid >= GNX_K_COUNT) return GNX_EINVAL;
  GNX_BIND_DEVICE(ctx);
  prof_drain(ctx);
  if (total_ms) *total_ms = ctx->prof_ms[kid];
  if (launches) *launches = ctx->prof_n[kid];
  return GNX_OK;
}

}  // extern "C"

int gnx_pipe_init(gnx_ctx* ctx) { return pipe_init(ctx); }
bool gnx_gnofix_packed_ok(const gnx_model* m) { return m->info.smooth_kind == GNX_SMOOTH_XGB && gnofix_use_rk(m); }
bool gnx_lr_p2_usable(const gnx_model* m) { return lr_p2_usable(m); }

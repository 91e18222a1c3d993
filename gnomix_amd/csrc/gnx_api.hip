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

template <typename T>
static int dev_upload(gnx_model* m, const std::vector<T>& h, const T** out, size_t pad_bytes = 0) {
  gnx_ctx* ctx = m->ctx;
  void* p = nullptr;
  const size_t bytes = h.size() * sizeof(T);
  hipError_t e = hipMalloc(&p, bytes + pad_bytes + 16);
  if (e != hipSuccess) return fail(ctx, GNX_ENOMEM, std::string("hipMalloc model: ") + hipGetErrorString(e));
  m->dev_allocs.push_back(p);
  m->info.device_bytes += (int64_t)(bytes + pad_bytes + 16);
  if (bytes) HIPCHK(ctx, hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice));
  if (pad_bytes + 16) HIPCHK(ctx, hipMemset((char*)p + bytes, 0, pad_bytes + 16));
  *out = (const T*)p;
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

// ------------------------------------------------------------------------------------------------
// model preparation: logistic base
// ------------------------------------------------------------------------------------------------
static int build_lr(gnx_model* m, const gnx_model_desc* d) {
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
  std::vector<double> V(n_chunks * 16 * (size_t)NT * 64, 0.0);
  std::vector<int32_t> Vwin(m->lr_i8 ? V.size() : 0, -1);
  std::vector<double> maxabs((size_t)W, 0.0);
  const double* coef = d->lr_coef;
  const int64_t ldc = d->lr_ldc;
  for (size_t c = 0; c < n_chunks; ++c)
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

  std::vector<double> icpt(d->lr_intercept, d->lr_intercept + (size_t)W * A);
  int rc;
  if (m->lr_i8) {
    // exact fixed point: q = round(c * 2^f_w), |q| < 2^54, seven balanced base-256 digits per weight
    std::vector<int> fexp((size_t)W, 0);
    std::vector<double> wscale((size_t)W, 1.0);
    for (int64_t i = 0; i < W; ++i) {
      if (!(maxabs[(size_t)i] < 1e300)) return fail(ctx, GNX_EINVAL, "logistic base: non-finite coefficient");
      if (maxabs[(size_t)i] > 0.0) fexp[(size_t)i] = 53 - std::ilogb(maxabs[(size_t)i]);
      wscale[(size_t)i] = std::ldexp(1.0, -fexp[(size_t)i]);
    }
    std::vector<int8_t> V8(n_chunks * (size_t)NT * 7 * 64 * 16, 0);
    for (size_t c = 0; c < n_chunks; ++c)
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
    if ((rc = dev_upload(m, V8, &m->lr.V8, 64)) != GNX_OK) return rc;
    if ((rc = dev_upload(m, wscale, &m->lr.wscale)) != GNX_OK) return rc;
  } else if ((rc = dev_upload(m, V, &m->lr.V)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, icpt, &m->lr.icpt)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, chunk_j0, &m->lr.chunk_j0)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, chunk_flush0, &m->lr.chunk_flush0)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, chunk_nflush, &m->lr.chunk_nflush)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, win_chunk0, &m->lr.win_chunk0)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, win_chunk1, &m->lr.win_chunk1)) != GNX_OK) return rc;
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
  if ((rc = dev_upload(m, packed, &m->xgb.rk_packed, 64)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, U, &m->xgb.rk_thr, 64)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, lut, &m->xgb.rk_lut)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, group_tree0, &m->xgb.rk_group_tree0)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, group_class, &m->xgb.rk_group_class)) != GNX_OK) return rc;
  m->xgb.rk_K = K; m->xgb.rk_steps = steps; m->xgb.rk_stride = stride; m->xgb.rk_tree_bytes = tree_bytes;
  m->xgb.rk_n_groups = (int32_t)group_class.size(); m->xgb.rk_max_group = G; m->xgb.rk_rpl = rpl;
  // measured (chr22, 10 000 haplotypes, MI355X): rk 1.83 ms; h64 2.22 ms + 0.23 ms of rank pre-pass — conflict-free, and slower
  // (DESIGN.md 4.2b): the rank kernel stays the default, h64 is what GNX_SMOOTH_IMPL=h64 selects
  // (round 3, same inputs: rk 1.83 ms, pointer nodes 1.78 ms, bit-identical: default where the shape allows, "rk" = heap-index nodes)
  m->xgb.impl = (impl && std::string(impl) == "h64") ? 2 : (impl && std::string(impl) == "rk") ? 1 : 3;
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
    if ((rc = dev_upload(m, pp, &m->xgb.rp_packed, 64)) != GNX_OK) return rc;
    if ((rc = dev_upload(m, gp_tree0, &m->xgb.rp_group_tree0)) != GNX_OK) return rc;
    if ((rc = dev_upload(m, gp_class, &m->xgb.rp_group_class)) != GNX_OK) return rc;
    m->xgb.rp_tree_bytes = tbp; m->xgb.rp_n_groups = (int32_t)gp_class.size(); m->xgb.rp_max_group = Gp;
  }
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
    if ((rc = dev_upload(m, p8, &m->xgb.h8_packed, 64)) != GNX_OK) return rc;
    if ((rc = dev_upload(m, g8_tree0, &m->xgb.h8_group_tree0)) != GNX_OK) return rc;
    if ((rc = dev_upload(m, g8_class, &m->xgb.h8_group_class)) != GNX_OK) return rc;
    m->xgb.h8_tree_bytes = tb8; m->xgb.h8_n_groups = (int32_t)g8_class.size(); m->xgb.h8_max_group = G8;
  }
  return GNX_OK;
}

static int build_xgb(gnx_model* m, const gnx_model_desc* d) {
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
  if ((rc = dev_upload(m, packed, &m->xgb.packed, 64)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, group_tree0, &m->xgb.group_tree0)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, group_class, &m->xgb.group_class)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, class_tree0, &m->class_tree0)) != GNX_OK) return rc;
  m->xgb.n_groups = (int32_t)group_class.size();
  m->xgb.n_trees = d->n_trees;
  m->xgb.D = D;
  m->xgb.tree_bytes = tree_bytes;
  m->xgb.max_group = G;
  m->xgb.base_score = d->base_score;
  m->info.n_trees = d->n_trees;
  m->info.tree_depth = D;
  return build_xgb_rk(m, d, order, D);
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

static int build_forest(gnx_model* m, const gnx_model_desc* d) {
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
  if ((rc = dev_upload(m, packed, &m->forest.packed, 64)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, nodes2, &m->forest.nodes2, 64)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, win_tree0, &m->forest.win_tree0)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, wct, &m->forest.win_class_tree0)) != GNX_OK) return rc;
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

static int build_rforest(gnx_model* m, const gnx_model_desc* d) {
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
  if ((rc = dev_upload(m, packed, &m->forest.packed, 64)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, leafval, &m->forest.rf_leafval)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, win_tree0, &m->forest.win_tree0)) != GNX_OK) return rc;
  m->forest.D = D; m->forest.tree_bytes = tree_bytes; m->forest.max_trees = max_trees; m->forest.max_words = max_words;
  return GNX_OK;
}

// ------------------------------------------------------------------------------------------------
// model preparation: CovRSK / SVC base — support vectors as bit-planes, run-length table g
// ------------------------------------------------------------------------------------------------
static int build_covrsk(gnx_model* m, const gnx_model_desc* d) {
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
  if ((rc = dev_upload(m, wins, &m->svc.win)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, svbits, &m->svc.svbits, 64)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, coef, &m->svc.coef)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, gtab, &m->svc.gtab)) != GNX_OK) return rc;
  m->svc.max_nw = max_nw;
  m->svc.max_width = max_width;
  return GNX_OK;
}

static int build_crf(gnx_model* m, const gnx_model_desc* d) {
  gnx_ctx* ctx = m->ctx;
  const int A = d->A;
  if (!d->crf_state || !d->crf_trans) return fail(ctx, GNX_EINVAL, "crf smoother: crf_state / crf_trans is NULL");
  std::vector<double> st(d->crf_state, d->crf_state + (size_t)A * A), et((size_t)A * A);
  for (int i = 0; i < A * A; ++i) et[(size_t)i] = std::exp(d->crf_trans[i]);
  int rc;
  if ((rc = dev_upload(m, st, &m->crf_state)) != GNX_OK) return rc;
  if ((rc = dev_upload(m, et, &m->crf_etrans)) != GNX_OK) return rc;
  return GNX_OK;
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
  t.lr_ws = geti("GNX_LR_WS", 0);
  t.lr_w512 = geti("GNX_LR_W512", 0);
  t.lr_ws_pw = geti("GNX_LR_WS_PW", 2);
  t.lr_nbuf = geti("GNX_LR_NBUF", 0);
  t.sm_nw = geti("GNX_SM_NW", 0);
  t.sm_pair = geti("GNX_SM_PAIR", 1);
  if (const char* e = std::getenv("GNX_SM_TUNE")) std::sscanf(e, "%d,%d", &t.smf_rpl, &t.smf_nw);
  if (const char* e = std::getenv("GNX_CRF_FLAGS")) t.crf_flags = atoi(e);
  if (const char* e = std::getenv("GNX_LDS_PAD")) std::sscanf(e, "%d,%d", &t.lr_lds_pad, &t.sm_lds_pad);
  if (const char* e = std::getenv("GNX_CRF_IMPL")) t.crf_impl = !std::strcmp(e, "scan") ? 1 : !std::strcmp(e, "row") ? 2 : !std::strcmp(e, "lanes") ? 3 : 0;
  t.forest_threads = geti("GNX_FOREST_T", 0);
  t.forest_wrun = geti("GNX_FOREST_WRUN", 0);
  t.forest_halves = geti("GNX_FOREST_H", 0);
  t.forest_flags = geti("GNX_FOREST_FLAGS", 0);
  t.forest_impl = geti("GNX_FOREST_IMPL", 0);
  t.forest_skew = geti("GNX_FOREST_SKEW", -1);
  if (const char* e = std::getenv("GNX_HOST_BATCH")) t.host_batch = std::atoll(e);
  t.h2d_overlap = geti("GNX_H2D_OVERLAP", 1);
  t.debug = std::getenv("GNX_DEBUG") ? atoi(std::getenv("GNX_DEBUG")) : 0;
}

extern "C" {

int gnx_abi_version(void) { return GNX_ABI_VERSION; }

int gnx_init(int device, gnx_ctx** out) {
  if (!out) return GNX_EINVAL;
  *out = nullptr;
  gnx_ctx* ctx = new (std::nothrow) gnx_ctx();
  if (!ctx) return GNX_ENOMEM;
  ctx->device = device;
  hipError_t e = hipSetDevice(device);
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
  for (gnx_devbuf* b : {&ctx->ws_pk, &ctx->ws_xu, &ctx->ws_psi, &ctx->ws_gt2, &ctx->ws_src, &ctx->ws_gt2o, &ctx->ws_rank})
    if (b->p) (void)hipFree(b->p);
  for (gnx_devbuf* b : {&ctx->ws_x, &ctx->ws_b32, &ctx->ws_b64, &ctx->ws_p32, &ctx->ws_p64, &ctx->ws_lab, &ctx->ws_misc, &ctx->ws_scale, &ctx->ws_bits, &ctx->ws_lastrow, &ctx->ws_rpair, &ctx->ws_y0, &ctx->ws_cal, &ctx->ws_marg})
    if (b->p) (void)hipFree(b->p);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* gnx_last_error(const gnx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int gnx_set_stream(gnx_ctx* ctx, void* hip_stream) {
  if (!ctx) return GNX_EINVAL;
  if (!ctx->own_stream && ctx->stream == (hipStream_t)hip_stream) return GNX_OK;
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
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  ctx->own_stream = true;
  return GNX_OK;
}

int gnx_host_alloc(gnx_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return GNX_EINVAL;
  *out = nullptr;
  if (!ctx->usable) return fail(ctx, GNX_ESTATE, "context has no device (gnx_init failed)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return GNX_OK;
}

int gnx_host_free(gnx_ctx* ctx, void* p) {
  if (!ctx) return GNX_EINVAL;
  if (p) HIPCHK(ctx, hipHostFree(p));
  return GNX_OK;
}

int gnx_synchronize(gnx_ctx* ctx) {
  if (!ctx) return GNX_EINVAL;
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
  HIPCHK(ctx, hipSetDevice(ctx->device));
  gnx_model* m = new (std::nothrow) gnx_model();
  if (!m) return fail(ctx, GNX_ENOMEM, "host allocation failed");
  m->ctx = ctx;
  m->info.C = d->C; m->info.M = d->M; m->info.ctx = d->ctx; m->info.W = d->C / d->M;
  m->info.A = d->A; m->info.S = d->S; m->info.base_kind = d->base_kind; m->info.smooth_kind = d->smooth_kind;
  int rc = GNX_OK;
  switch (d->base_kind) {
    case GNX_BASE_NONE: break;
    case GNX_BASE_LOGISTIC: rc = build_lr(m, d); break;
    case GNX_BASE_COVRSK_SVC: rc = build_covrsk(m, d); break;
    case GNX_BASE_FOREST: rc = build_forest(m, d); break;
    case GNX_BASE_RFOREST: rc = build_rforest(m, d); break;
    default: rc = fail(ctx, GNX_EINVAL, "unknown base_kind");
  }
  if (rc == GNX_OK) switch (d->smooth_kind) {
    case GNX_SMOOTH_NONE: break;
    case GNX_SMOOTH_XGB:
      if (d->S <= 0 || d->S % 2 == 0) rc = fail(ctx, GNX_EINVAL, "S must be odd and positive (smooth.py:14)");
      else rc = build_xgb(m, d);
      break;
    case GNX_SMOOTH_CRF: rc = build_crf(m, d); break;
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
        if ((rc = dev_upload(m, wv, &m->cnn_weight)) == GNX_OK) rc = dev_upload(m, bv, &m->cnn_bias);
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
        if ((rc = dev_upload(m, off, &m->calib_off)) == GNX_OK && (rc = dev_upload(m, cx, &m->calib_x)) == GNX_OK)
          rc = dev_upload(m, cy, &m->calib_y);
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
    // boosted trees: the two-blocks-per-CU kernel wherever its 128-haplotype tile fits the LDS; random forest and GNX_FOREST_IMPL=1:
    // the 256-haplotype kernel
    const bool v2 = !L.rf_leafval && m->forest.nodes2 && ctx->tune.forest_impl != 1 &&
                    gnx_forest2_lds_bytes(L.A, gnx_forest_ring_words(L.width_last), L.max_trees, L.D) <= (size_t)160 * 1024;
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
    hipError_t e = ctx->tune.lr_ws ? gnx_launch_base_logistic_i8_ws(L, ctx->n_cu, ctx->tune, ctx->stream) : hipErrorNotSupported;
    if (e == hipErrorNotSupported && ctx->tune.lr_w512) e = gnx_launch_base_logistic_i8_w512(L, ctx->n_cu, ctx->tune, ctx->stream);
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
    const bool h64 = m->xgb.rk_packed && m->xgb.impl == 2 && gnx_smooth_h64_waves(m->xgb, m->info.A, m->info.S) > 0;
    if (m->xgb.impl == 2 && !h64) return fail(ctx, GNX_EUNSUPPORTED, "GNX_SMOOTH_IMPL=h64: the model's strip does not fit the LDS");
    if (m->xgb.rk_packed) {
      const size_t n_pad = (size_t)((N + 63) / 64 * 64) * m->info.W * m->info.A;   // h64 parks whole 64-haplotype lines
      int rc = ws_reserve(ctx, ctx->ws_marg, n_pad * sizeof(float));
      if (rc != GNX_OK) return rc;
      L.marg = (float*)ctx->ws_marg.p;
      if (h64 && (rc = ws_reserve(ctx, ctx->ws_rank, gnx_smooth_h64_rank_bytes(N, L.W, L.A, L.S))) != GNX_OK) return rc;
    }
    ProfScope ps(ctx, GNX_K_SMOOTH_XGB);
    if (h64) HIPCHK(ctx, gnx_launch_smooth_xgb_h64(L, (uint16_t*)ctx->ws_rank.p, ctx->tune, ctx->stream));
    else if (m->xgb.rk_packed) HIPCHK(ctx, gnx_launch_smooth_xgb_rk(L, ctx->tune, ctx->stream));
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
    L.state = m->crf_state; L.etrans = m->crf_etrans;
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

int gnx_infer_dev(gnx_model* m, const int8_t* dX, int64_t N, int64_t ldx, float* d_p32, double* d_p64, int32_t* d_lab) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (N <= 0) return N == 0 ? GNX_OK : fail(ctx, GNX_EINVAL, "infer: N < 0");
  const size_t n = (size_t)N * m->info.W * m->info.A;
  const bool f64 = (m->info.smooth_kind == GNX_SMOOTH_CRF);  // CRF consumes float64 base probabilities
  int rc = ws_reserve(ctx, f64 ? ctx->ws_b64 : ctx->ws_b32, n * (f64 ? 8 : 4));
  if (rc != GNX_OK) return rc;
  rc = gnx_base_predict_dev(m, dX, N, ldx, f64 ? nullptr : (float*)ctx->ws_b32.p, f64 ? (double*)ctx->ws_b64.p : nullptr);
  if (rc != GNX_OK) return rc;
  return gnx_smooth_predict_dev(m, f64 ? ctx->ws_b64.p : ctx->ws_b32.p, f64, N, d_p32, d_p64, d_lab);
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
  HIPCHK(ctx, hipSetDevice(ctx->device));
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
  HIPCHK(ctx, hipSetDevice(ctx->device));
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
  HIPCHK(ctx, hipSetDevice(ctx->device));
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
  if (packed && (rc = ws_reserve(ctx, ctx->ws_xu, (size_t)nb * ldx_dev + 256)) != GNX_OK) return rc;
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
    if (packed) {
      HIPCHK(ctx, gnx_launch_unpack2((const uint8_t*)dX, n, ld, C, (int8_t*)ctx->ws_xu.p, ldx_dev, sc));
      dX = (const int8_t*)ctx->ws_xu.p;
      ldx = ldx_dev;
    }
    float* dp32 = (float*)((char*)ctx->ws_p32.p + (size_t)b * p32_b);
    double* dp64 = p64 ? (double*)((char*)ctx->ws_p64.p + (size_t)b * p64_b) : nullptr;
    int32_t* dlab = lab ? (int32_t*)((char*)ctx->ws_lab.p + (size_t)b * lab_b) : nullptr;
    if ((rc = gnx_infer_dev(m, dX, n, ldx, dp32, dp64, dlab)) != GNX_OK) return rc;
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
  const int64_t ldx = ((m->info.C + 15) / 16) * 16;
  int rc = ws_reserve(ctx, ctx->ws_xu, (size_t)N * ldx + 256);
  if (rc != GNX_OK) return rc;
  HIPCHK(ctx, gnx_launch_unpack2(dP, N, ldp, m->info.C, (int8_t*)ctx->ws_xu.p, ldx, ctx->stream));
  return gnx_infer_dev(m, (const int8_t*)ctx->ws_xu.p, N, ldx, d_p32, d_p64, d_lab);
}

int gnx_smooth_rows(gnx_model* m, const float* rows, int64_t R, float* proba) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  if (m->info.smooth_kind != GNX_SMOOTH_XGB) return fail(ctx, GNX_ESTATE, "smooth_rows needs the XGB smoother");
  if (R < 0 || (R > 0 && (!rows || !proba))) return fail(ctx, GNX_EINVAL, "smooth_rows: bad arguments");
  if (R == 0) return GNX_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
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
  HIPCHK(ctx, hipSetDevice(ctx->device));
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

static int gnofix_check(gnx_model* m, int64_t ldx, int64_t n_ind, int32_t max_it, bool ptrs_ok, bool* in_lds) {
  gnx_ctx* ctx = m->ctx;
  // src/model.py:194: only a smoother with .gnofix == True (XGB_Smoother) supports re-phasing
  if (m->info.smooth_kind != GNX_SMOOTH_XGB)
    return fail(ctx, GNX_ESTATE, "Type of Smoother does not currently support re-phasing");
  if (n_ind < 0 || ldx < m->info.C || max_it < 0 || (n_ind > 0 && !ptrs_ok)) return fail(ctx, GNX_EINVAL, "gnofix: bad arguments");
  if (m->calibrate_on && m->calib_off)  // smoother.predict inside the loop would be calibrated (gnofix.py:80,190); the kernel's is not
    return fail(ctx, GNX_EUNSUPPORTED, "gnofix with calibrate=True is not built: switch calibration off for re-phasing");
  const int W = (int)m->info.W, A = m->info.A, S = m->info.S;
  *in_lds = gnx_gnofix_lds_bytes(W, A, S, m->xgb.n_trees, m->xgb.tree_bytes, true) <= 150 * 1024;
  if (gnx_gnofix_lds_bytes(W, A, S, m->xgb.n_trees, m->xgb.tree_bytes, *in_lds) > 160 * 1024)
    return fail(ctx, GNX_EUNSUPPORTED, "gnofix: model too large for the LDS working set (n_trees * 16 B + S*A*8 B)");
  return GNX_OK;
}

// scratch of one device-resident batch of n individuals (grows only)
static int gnofix_ws_reserve(gnx_model* m, int64_t n, int32_t max_it, bool in_lds, size_t* bp_off_out) {
  gnx_ctx* ctx = m->ctx;
  const int W = (int)m->info.W, A = m->info.A, S = m->info.S, pad = (S + 1) / 2;
  const size_t WA = (size_t)W * A, NWD = (size_t)(W + 31) / 32;
  int rc;
  if ((rc = ws_reserve(ctx, ctx->ws_p32, (size_t)2 * n * WA * 4)) != GNX_OK) return rc;
  if ((rc = ws_reserve(ctx, ctx->ws_y0, (size_t)2 * n * W * 4)) != GNX_OK) return rc;
  size_t misc = (size_t)n * std::max(max_it, 1) * NWD * 4;
  const size_t bp_off = (misc + 255) & ~(size_t)255;
  if (!in_lds) misc = bp_off + (size_t)n * 2 * (W + 2 * pad) * A * 4;
  if (bp_off_out) *bp_off_out = bp_off;
  return ws_reserve(ctx, ctx->ws_misc, misc + 256);
}

// device-resident batch: initial labels with the batched smoother kernel, then one workgroup per individual
static int gnofix_run_dev(gnx_model* m, int8_t* dX, int64_t ldx, const double* dB, int64_t n, int32_t max_it, int32_t* dY,
                          int32_t* dNs, bool in_lds) {
  gnx_ctx* ctx = m->ctx;
  const int W = (int)m->info.W, A = m->info.A, S = m->info.S;
  int rc;
  size_t bp_off = 0;
  if ((rc = gnofix_ws_reserve(m, n, max_it, in_lds, &bp_off)) != GNX_OK) return rc;
  int32_t* dY0 = (int32_t*)ctx->ws_y0.p;
  // initial labels = smoother.predict(B) for every haplotype at once (gnofix.py:80)
  rc = gnx_smooth_predict_dev(m, dB, 1, 2 * n, (float*)ctx->ws_p32.p, nullptr, dY0);
  if (rc != GNX_OK) return rc;
  GnofixLaunch L{};
  L.X = dX; L.ldx = ldx; L.C = m->info.C; L.B = dB; L.Y0 = dY0; L.Yout = dY; L.n_switches = dNs;
  L.W = W; L.A = A; L.S = S; L.max_it = max_it; L.d = m->xgb; L.class_tree0 = m->class_tree0;
  L.bp_in_lds = in_lds ? 1 : 0;
  L.hist = (uint32_t*)ctx->ws_misc.p;
  L.bp_scratch = in_lds ? nullptr : (float*)((char*)ctx->ws_misc.p + bp_off);
  ProfScope ps(ctx, GNX_K_GNOFIX);
  HIPCHK(ctx, gnx_launch_gnofix(L, n, ctx->stream));
  return GNX_OK;
}

int gnx_gnofix_dev(gnx_model* m, int8_t* dX, int64_t ldx, const double* dB, int64_t n_ind, int32_t max_it, int32_t* dY,
                   int32_t* d_n_switches) {
  if (!m) return GNX_EINVAL;
  bool in_lds = true;
  int rc = gnofix_check(m, ldx, n_ind, max_it, dX && dB && dY, &in_lds);
  if (rc != GNX_OK || n_ind == 0) return rc;
  return gnofix_run_dev(m, dX, ldx, dB, n_ind, max_it, dY, d_n_switches, in_lds);
}

int gnx_gnofix(gnx_model* m, int8_t* X, int64_t ldx, const double* B, int64_t n_ind, int32_t max_it, int32_t* Y,
               int32_t* n_switches) {
  if (!m) return GNX_EINVAL;
  gnx_ctx* ctx = m->ctx;
  bool in_lds = true;
  int rc = gnofix_check(m, ldx, n_ind, max_it, X && B && Y, &in_lds);
  if (rc != GNX_OK || n_ind == 0) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int W = (int)m->info.W;
  const size_t WA = (size_t)W * m->info.A;
  // Batches of whole individuals alternate between the two halves of the staging workspaces: X and B of batch i+1 go up, and X / labels
  // of batch i-1 come back, while batch i is re-phased (three streams; page-locked host memory — gnx_host_alloc — makes the overlap
  // real, pageable memory is still correct).  A batch is a multiple of the CU count (one workgroup per individual) near 1 GiB of X.
  const int64_t cu = std::max(ctx->n_cu, 1);
  int64_t nb = std::max<int64_t>(1, (((int64_t)1 << 30) / std::max<int64_t>(ldx, 1)) / 2);
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
    if ((rc = gnofix_run_dev(m, dX, ldx, dB, n, max_it, dY, dNs, in_lds)) != GNX_OK) return rc;
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
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int newton = max_iter > 0 ? std::min(max_iter, 200) : 100;
  HIPCHK(ctx, gnx_train_lr_run(dX, N, ldx, dy, C, M, cx, A, C_reg, tol, newton, 250, coef, ldc, intercept, info, ctx->stream));
  return GNX_OK;
}

int gnx_train_logistic(gnx_ctx* ctx, const int8_t* X, int64_t N, int64_t ldx, const int32_t* y, int64_t C, int64_t M, int64_t cx, int32_t A,
                       double C_reg, double tol, int32_t max_iter, double* coef, int64_t ldc, double* intercept, gnx_train_info* info) {
  if (!ctx) return GNX_EINVAL;
  if (!ctx->usable) return fail(ctx, GNX_ESTATE, "context has no device (gnx_init failed)");
  if (N <= 0 || !X || !y || M <= 0 || C < M || ldx < C) return fail(ctx, GNX_EINVAL, "train_logistic: bad X / y / geometry");
  HIPCHK(ctx, hipSetDevice(ctx->device));
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
  if (!(P->eta > 0.0) || !(P->lambda >= 0.0) || !(P->gamma >= 0.0) || !(P->min_child_weight >= 0.0))
    return fail(ctx, GNX_EINVAL, "train_gbt: eta > 0, lambda / gamma / min_child_weight >= 0");
  if ((int64_t)N * W >= ((int64_t)1 << 31)) return fail(ctx, GNX_EINVAL, "train_gbt: N * W must stay below 2^31 rows");
  return GNX_OK;
}

int gnx_train_gbt_dev(gnx_ctx* ctx, const void* dB, int32_t b_is_f64, const int32_t* dy, int64_t N, int32_t W, int32_t A, int32_t S,
                      const gnx_gbt_params* P, int32_t* tree_off, int32_t* tree_class, int32_t* left, int32_t* right, int32_t* feat,
                      float* cond, int64_t* n_nodes, double* loss) {
  if (!ctx) return GNX_EINVAL;
  int rc = gbt_check(ctx, dB, dy, N, W, A, S, P, tree_off, tree_class, left, right, feat, cond, n_nodes);
  if (rc != GNX_OK) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
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
  HIPCHK(ctx, hipSetDevice(ctx->device));
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
  if (!on) prof_drain(ctx);
  ctx->prof = on != 0;
  return GNX_OK;
}

int gnx_profile_reset(gnx_ctx* ctx) {
  if (!ctx) return GNX_EINVAL;
  prof_drain(ctx);
  for (int k = 0; k < GNX_K_COUNT; ++k) { ctx->prof_ms[k] = 0; ctx->prof_n[k] = 0; }
  return GNX_OK;
}

int gnx_profile_get(gnx_ctx* ctx, int kid, double* total_ms, int64_t* launches) {
  if (!ctx || kid < 0 || kid >= GNX_K_COUNT) return GNX_EINVAL;
  prof_drain(ctx);
  if (total_ms) *total_ms = ctx->prof_ms[kid];
  if (launches) *launches = ctx->prof_n[kid];
  return GNX_OK;
}

}  // extern "C"

int gnx_pipe_init(gnx_ctx* ctx) { return pipe_init(ctx); }

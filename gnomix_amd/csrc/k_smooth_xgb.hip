// k_smooth_xgb.hip — sliding-window gradient-boosted-tree smoother on gfx950 (CDNA4).
//
// Replaces slide_window + XGBClassifier.predict_proba + argmax (reference src/Smooth/utils.py:4-29,
// src/Smooth/smooth.py:40-65, src/Smooth/models.py:8-24; xgboost 1.1.1 multi:softprob semantics as
// restated in the CPU oracle).
//
// Design:
//  * the (N*W, S*A) feature matrix is never built: with the base probabilities of one haplotype
//    laid out [padded window][class] in LDS, the S*A features of row (n,w) are the CONTIGUOUS
//    floats starting at (w - w0)*A, so feature f of a row is strip[(w-w0)*A + f].
//  * one lane owns RPL rows (64 consecutive windows of RPL haplotypes per wave): consecutive lanes
//    read LDS addresses A dwords apart — conflict-free whenever A is odd and lanes sit on the same
//    node (always at the root).
//  * trees are expanded to complete depth-D heaps, 8-byte nodes {feature byte offset, threshold}
//    + 2^D float leaves, stored class-major; groups of one class are staged through a
//    double-buffered LDS window (global -> VGPR -> LDS overlapped with the walk of the previous
//    group), node fetch = one ds_read_b64 (the <=8 distinct nodes of a level sit on distinct banks).
//  * per class the float32 margin is accumulated in tree order from 0 and added to base_score, exactly
//    the order of the restated xgboost predictor; margins are parked in the output buffer, then
//    one softmax/argmax pass (expf evaluated as float(exp(double)), which agrees with glibc's
//    correctly-rounded expf).
#include <cstdio>
#include <cstdlib>

#include "gnx_internal.h"
#include "gnx_exp.h"

namespace {

constexpr int WS = 64;  // windows per segment = one wave width

__device__ __forceinline__ int slide_src(int j, int W, int pad) {
  // reflect padding of slide_window (src/Smooth/utils.py:14-17)
  if (j < pad) return pad - 1 - j;
  if (j < pad + W) return j - pad;
  return W - 1 - (j - pad - W);
}

template <int D>
__device__ __forceinline__ float walk(const uint8_t* tb, const uint8_t* rowbase) {
  constexpr uint32_t half = 1u << (D - 1);
  uint32_t j = 1;
#pragma unroll
  for (int d = 0; d < D - 1; ++d) {
    const uint2 nd = *reinterpret_cast<const uint2*>(tb + half * 16 + (j - 1) * 8);
    const float fv = *reinterpret_cast<const float*>(rowbase + nd.x);
    j = 2 * j + ((fv < __uint_as_float(nd.y)) ? 0u : 1u);
  }
  const uint4 n4 = *reinterpret_cast<const uint4*>(tb + (j - half) * 16);  // last split + both leaves in one read
  const float fv = *reinterpret_cast<const float*>(rowbase + n4.x);
  return (fv < __uint_as_float(n4.y)) ? __uint_as_float(n4.z) : __uint_as_float(n4.w);
}

// One wave = one haplotype x RPL consecutive 64-window segments (RPL rows per lane, all on the same LDS strip);
// NWAVE haplotypes per block.  DT = compile-time depth (0 = runtime).
template <int RPL, int NWAVE, int DT>
__global__ __launch_bounds__(NWAVE * 64) void k_smooth_xgb(SmoothXGBLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int THREADS = NWAVE * 64;
  const int A = L.A, W = L.W, S = L.S, pad = (S + 1) / 2;
  const int D = DT ? DT : L.d.D;
  const int tree_bytes = L.d.tree_bytes;
  const int strip_w = RPL * WS + S - 1;          // padded windows held per haplotype
  const int strip_bytes = strip_w * A * 4;
  const int buf_bytes = L.d.max_group * tree_bytes;  // multiple of 16
  uint8_t* strip = lds;                          // [NWAVE][strip_w][A] float
  uint8_t* tbuf0 = lds + (((size_t)NWAVE * strip_bytes + 15) & ~(size_t)15);
  uint8_t* tbuf1 = tbuf0 + buf_bytes;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t h0 = (int64_t)blockIdx.y * NWAVE;
  const int w0 = blockIdx.x * (RPL * WS);

  // ---- stage the reflected base-probability strips ----
  {
    const int per_h = strip_w * A;
    for (int e = tid; e < NWAVE * per_h; e += THREADS) {
      const int hl = e / per_h, r = e - hl * per_h;
      const int q = r / A, a = r - q * A;
      const int64_t n = h0 + hl;
      const int j = w0 + q;
      float v = 0.f;
      if (n < L.N && j < W + 2 * pad) {
        const size_t idx = ((size_t)n * W + slide_src(j, W, pad)) * A + a;
        v = L.b_is_f64 ? (float)reinterpret_cast<const double*>(L.B)[idx] : reinterpret_cast<const float*>(L.B)[idx];
      }
      reinterpret_cast<float*>(strip)[e] = v;
    }
  }

  const int64_t n = h0 + wave;
  const uint8_t* rowbase[RPL];
  bool valid[RPL];
  size_t orow[RPL];
#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    rowbase[k] = strip + (size_t)wave * strip_bytes + (size_t)(k * WS + lane) * A * 4;
    const int w = w0 + k * WS + lane;
    valid[k] = (n < L.N) && (w < W);
    orow[k] = ((size_t)(valid[k] ? n : 0) * W + (valid[k] ? w : 0)) * A;
  }

  // ---- tree groups through the double-buffered LDS window ----
  const int ng = L.d.n_groups;
  constexpr int MAXV = (16384 / 16 + THREADS - 1) / THREADS;  // 16-byte staging pieces per thread (buf <= 16 KB)
  uint4 stg[MAXV];
  const int nv = (buf_bytes / 16 + THREADS - 1) / THREADS;
  // unconditional clamped loads: no branch around a load, so hipcc keeps its waits counted
#define GNX_G_LOAD(g)                                                                               \
  {                                                                                                 \
    const int t0_ = L.d.group_tree0[g], t1_ = L.d.group_tree0[(g) + 1];                             \
    const int last_ = (t1_ - t0_) * tree_bytes / 16 - 1;                                            \
    const uint4* src_ = reinterpret_cast<const uint4*>(L.d.packed + (size_t)t0_ * tree_bytes);      \
    _Pragma("unroll") for (int v = 0; v < MAXV; ++v) if (v < nv) stg[v] = src_[min(v * THREADS + tid, last_)]; \
  }
#define GNX_G_STORE(dst)                                                                            \
  {                                                                                                 \
    _Pragma("unroll") for (int v = 0; v < MAXV; ++v) {                                              \
      const int e_ = v * THREADS + tid;                                                             \
      if (v < nv && e_ * 16 < buf_bytes) *reinterpret_cast<uint4*>((dst) + (size_t)e_ * 16) = stg[v]; \
    }                                                                                               \
  }

  float psum[RPL];
#pragma unroll
  for (int k = 0; k < RPL; ++k) psum[k] = 0.f;

  GNX_G_LOAD(0);
  GNX_G_STORE(tbuf0);
  __syncthreads();

  int cur_class = L.d.group_class[0];
  for (int g = 0; g < ng; ++g) {
    uint8_t* cur = (g & 1) ? tbuf1 : tbuf0;
    uint8_t* nxt = (g & 1) ? tbuf0 : tbuf1;
    const int gn = min(g + 1, ng - 1);  // clamped: the last iteration re-fetches its own group
    GNX_G_LOAD(gn);

    const int cls = L.d.group_class[g];
    if (cls != cur_class) {  // class finished: park its margin (base_score + psum)
#pragma unroll
      for (int k = 0; k < RPL; ++k) {
        if (valid[k]) L.proba[orow[k] + cur_class] = L.d.base_score + psum[k];
        psum[k] = 0.f;
      }
      cur_class = cls;
    }
    const int nt = L.d.group_tree0[g + 1] - L.d.group_tree0[g];
    for (int t = 0; t < nt; ++t) {
      const uint8_t* tb = cur + (size_t)t * tree_bytes;
#pragma unroll
      for (int k = 0; k < RPL; ++k) {
        float leaf;
        if constexpr (DT > 0) leaf = walk<DT>(tb, rowbase[k]);
        else leaf = gnx_walk(tb, rowbase[k], D);
        psum[k] += leaf;
      }
    }
    GNX_G_STORE(nxt);
    __syncthreads();
  }
#undef GNX_G_LOAD
#undef GNX_G_STORE
#pragma unroll
  for (int k = 0; k < RPL; ++k)
    if (valid[k]) L.proba[orow[k] + cur_class] = L.d.base_score + psum[k];

  // ---- softmax (xgboost common/math.h Softmax) + argmax, per row, by the lane that wrote the margins ----
#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    if (!valid[k]) continue;
    float* o = L.proba + orow[k];
    float wmax = o[0];
    for (int a = 1; a < A; ++a) wmax = fmaxf(o[a], wmax);
    double wsum = 0.0;
    for (int a = 0; a < A; ++a) {
      const float e = gnx_softmax_exp(o[a] - wmax);
      o[a] = e;
      wsum += (double)e;
    }
    const float fs = (float)wsum;
    int best = 0;
    float bv = -1.f;
    for (int a = 0; a < A; ++a) {
      const float p = o[a] / fs;
      o[a] = p;
      if (L.proba64) L.proba64[orow[k] + a] = (double)p;
      if (p > bv) { bv = p; best = a; }
    }
    if (L.labels) L.labels[orow[k] / A] = best;
  }
}

// smoother.model.predict_proba on explicit rows: one block = 8 rows, thread = (row, class)
__global__ __launch_bounds__(256) void k_smooth_rows(SmoothXGBDev d, const float* rows, int64_t R, int F, int A,
                                                       float* proba) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  float* feats = reinterpret_cast<float*>(lds);  // [8][F]
  float* marg = feats + 8 * F;                   // [8][A]
  const int64_t r0 = (int64_t)blockIdx.x * 8;
  for (int e = threadIdx.x; e < 8 * F; e += blockDim.x) {
    const int64_t r = r0 + e / F;
    feats[e] = (r < R) ? rows[r * F + (e % F)] : 0.f;
  }
  __syncthreads();
  const int rl = threadIdx.x / A, cls = threadIdx.x % A;
  if (rl < 8 && threadIdx.x < 8 * A) {
    float psum = 0.f;
    const uint8_t* rb = reinterpret_cast<const uint8_t*>(feats + rl * F);
    for (int g = 0; g < d.n_groups; ++g) {
      if (d.group_class[g] != cls) continue;
      for (int t = d.group_tree0[g]; t < d.group_tree0[g + 1]; ++t)
        psum += gnx_walk(d.packed + (size_t)t * d.tree_bytes, rb, d.D);
    }
    marg[rl * A + cls] = d.base_score + psum;
  }
  __syncthreads();
  if (threadIdx.x < 8 && r0 + threadIdx.x < R) {
    float* m = marg + threadIdx.x * A;
    float wmax = m[0];
    for (int a = 1; a < A; ++a) wmax = fmaxf(m[a], wmax);
    double wsum = 0.0;
    for (int a = 0; a < A; ++a) { m[a] = gnx_softmax_exp(m[a] - wmax); wsum += (double)m[a]; }
    const float fs = (float)wsum;
    for (int a = 0; a < A; ++a) proba[(r0 + threadIdx.x) * A + a] = m[a] / fs;
  }
}

template <int RPL, int NWAVE>
size_t lds_bytes(const SmoothXGBDev& d, int A, int S) {
  const size_t strip = (size_t)NWAVE * (RPL * WS + S - 1) * A * 4;
  return ((strip + 15) & ~(size_t)15) + 2 * (size_t)d.max_group * d.tree_bytes;
}

template <int RPL, int NWAVE>
hipError_t launch(const SmoothXGBLaunch& L, hipStream_t s) {
  const dim3 grid((unsigned)((L.W + RPL * WS - 1) / (RPL * WS)), (unsigned)((L.N + NWAVE - 1) / NWAVE));
  const size_t lds = lds_bytes<RPL, NWAVE>(L.d, L.A, L.S);
  if (L.d.D == 4) {
    GNX_LDS_OPTIN(lds, k_smooth_xgb<RPL, NWAVE, 4>);
    hipLaunchKernelGGL((k_smooth_xgb<RPL, NWAVE, 4>), grid, dim3(NWAVE * 64), lds, s, L);
  } else {
    GNX_LDS_OPTIN(lds, k_smooth_xgb<RPL, NWAVE, 0>);
    hipLaunchKernelGGL((k_smooth_xgb<RPL, NWAVE, 0>), grid, dim3(NWAVE * 64), lds, s, L);
  }
  return hipGetLastError();
}

}  // namespace

size_t gnx_smooth_xgb_lds_bytes(const SmoothXGBDev& d, int A, int S) { return lds_bytes<1, 1>(d, A, S); }

hipError_t gnx_launch_smooth_xgb(const SmoothXGBLaunch& L, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  // rows per lane = 64-window segments one wave walks on its haplotype's strip; more segments per strip = less halo
  // and more independent chains per lane, bounded by LDS (want >= 2-3 blocks per CU)
  const int nseg = (L.W + WS - 1) / WS;
  int rpl = tune.smf_rpl, nw = tune.smf_nw;
  if (!rpl) {
    // measured on chr22 (W=370, A=7, 700 trees): 3 segments x 4 waves is the sweet spot (2.07 ms / 10k haplotypes);
    // 6-8 segments per lane lose to LDS/issue pressure, 1 segment pays 2.2x halo
    rpl = nseg >= 3 ? 3 : nseg;
    nw = 4;
    auto fits = [&](int r, int w) {
      return (size_t)w * (r * WS + L.S - 1) * L.A * 4 + 2 * (size_t)L.d.max_group * L.d.tree_bytes + 64 <= 80 * 1024;
    };
    while (!fits(rpl, nw) && nw > 1) nw = (nw == 4) ? 3 : nw - 1;
    while (!fits(rpl, nw) && rpl > 1) --rpl;
    if (nw == 2 && rpl == 3) nw = 1;  // not instantiated
  }
#define GNX_SM_CASE(R_, W_) if (rpl == R_ && nw == W_) return launch<R_, W_>(L, s);
  GNX_SM_CASE(3, 4) GNX_SM_CASE(3, 3) GNX_SM_CASE(3, 1) GNX_SM_CASE(2, 4) GNX_SM_CASE(2, 3) GNX_SM_CASE(2, 2) GNX_SM_CASE(2, 1)
  GNX_SM_CASE(1, 4) GNX_SM_CASE(1, 3) GNX_SM_CASE(1, 2) GNX_SM_CASE(1, 1) GNX_SM_CASE(4, 4) GNX_SM_CASE(6, 4)
#undef GNX_SM_CASE
  return hipErrorInvalidValue;
}

hipError_t gnx_launch_smooth_rows(const SmoothXGBDev& d, const float* rows, int64_t R, int32_t F, int32_t A,
                                  float* proba, hipStream_t s) {
  if (R <= 0) return hipSuccess;
  if (8 * A > 256) return hipErrorInvalidValue;
  const size_t lds = (size_t)(8 * F + 8 * A) * 4;
  hipLaunchKernelGGL(k_smooth_rows, dim3((unsigned)((R + 7) / 8)), dim3(256), lds, s, d, rows, R, (int)F, (int)A, proba);
  return hipGetLastError();
}

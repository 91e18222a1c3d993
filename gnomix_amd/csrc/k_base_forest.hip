// k_base_forest.hip — per-window gradient-boosted-tree base classifiers ("forest" bases) on gfx950.
//
// Replaces XGBBase.predict_proba (reference src/Base/models.py:24-35: per window
// XGBClassifier(n_estimators=20, max_depth=4, missing=missing_encoding); xgboost semantics as restated in the CPU
// oracle: a SNP equal to the missing code takes the node's default child, otherwise left iff float(v) < threshold;
// multi:softprob for A >= 3, binary:logistic for A == 2).  No reference mode selects these bases (README.md:120-146
// shows how to plug them in), they are the "forest-of-stumps" family of the north star.
//
// One thread = one haplotype of one window; a block = T = 64*waves haplotypes of that window.
//  * staging: every wave reads ITS OWN 64 haplotypes' SNP bytes of the window straight from X (16 SNPs = one
//    unaligned 16-byte load per lane, 8 lanes along a row = one 128-byte line, LB loads in flight per lane),
//    squeezes them to 2 bits per SNP and stores them in LDS as xw[word][hap]: during the walks a lane's bank is then
//    fixed by its haplotype, so lanes that sit on different split SNPs never conflict.  Reflect padding
//    (base.py:146-151) only touches the first / last ctx SNPs of a row: those words take a per-byte path.
//  * the window's trees sit next to the tile as compact heaps: one 32-bit word per node = (SNP index << 4) | left-mask
//    (bit v = "value v goes left": SNPs only take the values 0..3, so the float compare and the missing code's
//    default direction were folded into 4 bits by the loader), then the 2^D float leaves.  A lane walks TP trees at
//    a time level by level (TP independent LDS chains), adds the leaves of each class in tree order (float32, as
//    the restated predictor does), parks the margins in LDS and applies softmax / sigmoid in float32.
#include "gnx_internal.h"

#include <cstdlib>

namespace {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int64_t pad_src(int64_t p, int64_t C, int64_t ctx) {
  if (p < ctx) return ctx - 1 - p;
  if (p < ctx + C) return p - ctx;
  return C - 1 - (p - ctx - C);
}

// 4 SNP bytes (values 0..3) -> 8 bits
__device__ __forceinline__ uint32_t squeeze4(uint32_t d) {
  const uint32_t x = d & 0x03030303u;
  const uint32_t y = x | (x >> 6);
  return (y | (y >> 12)) & 0xffu;
}

constexpr int TP = 8;  // trees walked concurrently per lane
constexpr int LB = 16;  // 16-byte loads in flight per lane while staging

// left iff bit v of the node's mask is set, v = the 2-bit SNP value
__device__ __forceinline__ uint32_t step(uint32_t j, uint32_t nd, uint32_t xv) {
  const uint32_t v = (xv >> ((nd >> 3) & 30u)) & 3u;
  return 2 * j + (((nd >> v) & 1u) ^ 1u);
}

template <int D, int NT>
__device__ __forceinline__ void walk(const uint8_t* tb, int tree_bytes, const uint32_t* xcol, int T, float* leaf) {
  uint32_t j[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) j[k] = 1;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    uint32_t nd[NT], xv[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) nd[k] = reinterpret_cast<const uint32_t*>(tb + k * tree_bytes)[j[k]];
#pragma unroll
    for (int k = 0; k < NT; ++k) xv[k] = xcol[(nd[k] >> 8) * T];
#pragma unroll
    for (int k = 0; k < NT; ++k) j[k] = step(j[k], nd[k], xv[k]);
  }
#pragma unroll
  for (int k = 0; k < NT; ++k) leaf[k] = reinterpret_cast<const float*>(tb + k * tree_bytes)[j[k]];  // leaves follow the 2^D nodes
}

// Stage the calling wave's 64 haplotypes of window w: 8 haplotypes x 8 words per wave instruction.
__device__ __forceinline__ void stage_window(const ForestLaunch& L, int w, int64_t width, uint32_t* xw, int T) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t C = L.C, ctx = L.ctx, Cp = C + 2 * ctx;
  {
    const int64_t s = (int64_t)w * L.M;  // padded coordinate of the window's first SNP
    const int nw = (int)((width + 15) >> 4);
    const int wsub = lane & 7, hsub = lane >> 3;
    const int64_t cmax = C - 16;  // the model loader guarantees C >= 16
    for (int hb = 0; hb < 8; ++hb) {
      const int h = wv * 64 + hb * 8 + hsub;
      int64_t n = (int64_t)blockIdx.x * T + h;
      n = n < L.N ? n : L.N - 1;
      const int8_t* row = L.X + n * L.ldx;
      uint32_t* col = xw + h;
      for (int wb = 0; wb < nw; wb += 8 * LB) {
        v4u v[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {  // unconditional loads at clamped addresses: all LB are in flight together
          int64_t c0 = s + 16 * (int64_t)(wb + 8 * u + wsub) - ctx;
          c0 = c0 < 0 ? 0 : (c0 > cmax ? cmax : c0);
          __builtin_memcpy(&v[u], row + c0, 16);
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          const int wd = wb + 8 * u + wsub;
          const int64_t p0 = s + 16 * (int64_t)wd;
          uint32_t q = squeeze4(v[u].x) | (squeeze4(v[u].y) << 8) | (squeeze4(v[u].z) << 16) | (squeeze4(v[u].w) << 24);
          if (wd < nw && (p0 < ctx || p0 + 16 > ctx + C)) {  // rare: the word touches the reflect padding / the row's end
            q = 0;
            for (int b = 0; b < 16; ++b) {
              const int64_t p = p0 + b;
              if (p < Cp) q |= ((uint32_t)(uint8_t)row[pad_src(p, C, ctx)] & 3u) << (2 * b);
            }
          }
          if (wd < nw) col[(size_t)wd * T] = q;
        }
      }
    }
  }
}

template <int D>
__global__ __launch_bounds__(256) void k_base_forest(ForestLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x, T = blockDim.x;
  const int w = L.w_first + blockIdx.y;
  const int A = L.A, tree_bytes = L.tree_bytes;
  const int64_t width = (w == L.W - 1) ? L.width_last : L.width;
  const int t0 = L.win_tree0[w], nt = L.win_tree0[w + 1] - t0;

  uint32_t* xw = reinterpret_cast<uint32_t*>(lds);                                    // [max_words][T]
  uint8_t* tr = lds + (size_t)L.max_words * T * 4;                                    // trees of this window
  float* marg = reinterpret_cast<float*>(tr + (((size_t)L.max_trees * tree_bytes + 15) & ~(size_t)15)) + tid;  // [A][T]

  stage_window(L, w, width, xw, T);
  for (int e = tid; e < nt * tree_bytes / 16; e += T)
    reinterpret_cast<uint4*>(tr)[e] = reinterpret_cast<const uint4*>(L.packed + (size_t)t0 * tree_bytes)[e];
  __syncthreads();

  // ---- walks: class-major packing per window, trees of class c contiguous and in model order ----------------
  const uint32_t* xcol = xw + tid;
  const int32_t* cls0 = L.win_class_tree0 + (size_t)w * (A + 1);  // [A+1] offsets relative to t0
  const int n_groups = (A == 2) ? 1 : A;
  for (int c = 0; c < n_groups; ++c) {
    const int a0 = (A == 2) ? 0 : cls0[c], a1 = (A == 2) ? nt : cls0[c + 1];
    float psum = 0.f;
    int t = a0;
    for (; t + TP <= a1; t += TP) {
      float leaf[TP];
      walk<D, TP>(tr + (size_t)t * tree_bytes, tree_bytes, xcol, T, leaf);
#pragma unroll
      for (int k = 0; k < TP; ++k) psum += leaf[k];  // tree order
    }
    for (; t < a1; ++t) {
      float leaf[1];
      walk<D, 1>(tr + (size_t)t * tree_bytes, tree_bytes, xcol, T, leaf);
      psum += leaf[0];
    }
    marg[c * T] = psum;
  }
  const int64_t n = (int64_t)blockIdx.x * T + tid;
  if (n >= L.N) return;
  const size_t o = ((size_t)n * L.W + w) * A;
  if (A == 2) {
    const float margin = logf(L.base_score / (1.0f - L.base_score)) + marg[0];  // ProbToMargin of binary:logistic
    const float p1 = 1.0f / (1.0f + (float)exp((double)(-margin)));
    const float p[2] = {1.0f - p1, p1};
    for (int a = 0; a < 2; ++a) {
      if (L.b32) L.b32[o + a] = p[a];
      if (L.b64) L.b64[o + a] = (double)p[a];
    }
    return;
  }
  float wmax = L.base_score + marg[0];
  for (int a = 1; a < A; ++a) wmax = fmaxf(L.base_score + marg[a * T], wmax);
  double wsum = 0.0;
  for (int a = 0; a < A; ++a) {
    const float e = (float)exp((double)((L.base_score + marg[a * T]) - wmax));
    marg[a * T] = e;
    wsum += (double)e;
  }
  const float fs = (float)wsum;
  for (int a = 0; a < A; ++a) {
    const float p = marg[a * T] / fs;
    if (L.b32) L.b32[o + a] = p;
    if (L.b64) L.b64[o + a] = (double)p;
  }
}

// ---- random-forest variant (RFBase, src/Base/models.py:54-66) --------------------------------------------------------
// Same tile, same mask nodes (mask bit v = "float32(v) <= threshold"); a leaf contributes its class-probability row
// (float64, expanded per heap slot in global memory: 20 trees x 16 leaves x A doubles per window stay in L1/L2), the
// rows are added in estimator order and divided by the tree count, as ForestClassifier.predict_proba does.
template <int AMAX>
__global__ __launch_bounds__(256) void k_base_rforest(ForestLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x, T = blockDim.x;
  const int w = L.w_first + blockIdx.y;
  const int A = L.A, D = L.D, tree_bytes = L.tree_bytes;
  const int64_t width = (w == L.W - 1) ? L.width_last : L.width;
  const int t0 = L.win_tree0[w], nt = L.win_tree0[w + 1] - t0;
  uint32_t* xw = reinterpret_cast<uint32_t*>(lds);   // [max_words][T]
  uint8_t* tr = lds + (size_t)L.max_words * T * 4;   // node words of this window's trees
  stage_window(L, w, width, xw, T);
  for (int e = tid; e < nt * tree_bytes / 16; e += T)
    reinterpret_cast<uint4*>(tr)[e] = reinterpret_cast<const uint4*>(L.packed + (size_t)t0 * tree_bytes)[e];
  __syncthreads();

  const uint32_t* xcol = xw + tid;
  double acc[AMAX];
#pragma unroll
  for (int a = 0; a < AMAX; ++a) acc[a] = 0.0;
  const size_t leaves = (size_t)1 << D;
  for (int t = 0; t < nt; ++t) {
    const uint32_t* nodes = reinterpret_cast<const uint32_t*>(tr + (size_t)t * tree_bytes);
    uint32_t j = 1;
    for (int d = 0; d < D; ++d) {
      const uint32_t nd = nodes[j];
      j = step(j, nd, xcol[(nd >> 8) * T]);
    }
    const double* v = L.rf_leafval + ((size_t)(t0 + t) * leaves + (j - (uint32_t)leaves)) * A;
#pragma unroll
    for (int a = 0; a < AMAX; ++a)
      if (a < A) acc[a] += v[a];  // estimator order
  }
  const int64_t n = (int64_t)blockIdx.x * T + tid;
  if (n >= L.N) return;
  const size_t o = ((size_t)n * L.W + w) * A;
  const double cnt = (double)nt;
#pragma unroll
  for (int a = 0; a < AMAX; ++a)
    if (a < A) {
      const double p = acc[a] / cnt;
      if (L.b64) L.b64[o + a] = p;
      if (L.b32) L.b32[o + a] = (float)p;
    }
}

template <int AMAX>
hipError_t launch_rf(const ForestLaunch& L, int n_windows, int threads, size_t lds, hipStream_t s) {
  GNX_LDS_OPTIN((size_t)160 * 1024, k_base_rforest<AMAX>);
  hipLaunchKernelGGL(k_base_rforest<AMAX>, dim3((unsigned)((L.N + threads - 1) / threads), (unsigned)n_windows), dim3(threads), lds, s, L);
  return hipGetLastError();
}

template <int D>
hipError_t launch_d(const ForestLaunch& L, int n_windows, int threads, size_t lds, hipStream_t s) {
  GNX_LDS_OPTIN((size_t)160 * 1024, k_base_forest<D>);
  hipLaunchKernelGGL(k_base_forest<D>, dim3((unsigned)((L.N + threads - 1) / threads), (unsigned)n_windows), dim3(threads), lds, s, L);
  return hipGetLastError();
}

// windows [w_first, w_first + n_windows) with an X tile of max_words words per haplotype
hipError_t launch_range(ForestLaunch L, int w_first, int n_windows, int max_words, const gnx_tune& tune, hipStream_t s) {
  if (n_windows <= 0) return hipSuccess;
  // the walks are issue-bound with one wave per SIMD and staging does not overlap them inside a block: take as many
  // waves per CU as the LDS holds (the X tile is 4 * max_words bytes per haplotype)
  constexpr size_t kLds = (size_t)160 * 1024;
  L.w_first = w_first;
  L.max_words = max_words;
  int threads = tune.forest_threads / 64 * 64;
  if (threads < 64 || threads > 256) threads = 256;
  while (threads > 64 && gnx_forest_lds_bytes(L.A, max_words, L.max_trees, L.tree_bytes, threads) > kLds) threads -= 64;
  while (threads > 64 && (int64_t)(threads - 64) >= L.N) threads -= 64;
  const size_t lds = gnx_forest_lds_bytes(L.A, max_words, L.max_trees, L.tree_bytes, threads);
  if (lds > kLds) return hipErrorInvalidValue;
  if (L.rf_leafval) {
    if (L.A <= 8) return launch_rf<8>(L, n_windows, threads, lds, s);
    if (L.A <= 16) return launch_rf<16>(L, n_windows, threads, lds, s);
    return launch_rf<32>(L, n_windows, threads, lds, s);
  }
  switch (L.D) {
    case 1: return launch_d<1>(L, n_windows, threads, lds, s);
    case 2: return launch_d<2>(L, n_windows, threads, lds, s);
    case 3: return launch_d<3>(L, n_windows, threads, lds, s);
    case 4: return launch_d<4>(L, n_windows, threads, lds, s);
    case 5: return launch_d<5>(L, n_windows, threads, lds, s);
    case 6: return launch_d<6>(L, n_windows, threads, lds, s);
    case 7: return launch_d<7>(L, n_windows, threads, lds, s);
    case 8: return launch_d<8>(L, n_windows, threads, lds, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

size_t gnx_forest_lds_bytes(int A, int max_words, int max_trees, int tree_bytes, int threads) {
  return (size_t)max_words * threads * 4 + (((size_t)max_trees * tree_bytes + 15) & ~(size_t)15) + (size_t)A * threads * 4;
}

hipError_t gnx_launch_base_forest(const ForestLaunch& L, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  // the last window is wider by C mod M (base.py:163-164): its own launch, so that the others get the smaller tile
  const int words = (int)((L.width + 15) >> 4), words_last = (int)((L.width_last + 15) >> 4);
  hipError_t e = launch_range(L, 0, L.W - 1, words, tune, s);
  if (e != hipSuccess) return e;
  return launch_range(L, L.W - 1, 1, words_last, tune, s);
}

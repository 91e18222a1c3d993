// k_base_forest.hip — per-window gradient-boosted-tree base classifiers ("forest" bases) on gfx950.
//
// Replaces XGBBase.predict_proba (reference src/Base/models.py:24-35: per window
// XGBClassifier(n_estimators=20, max_depth=4, missing=missing_encoding); xgboost semantics as restated in the CPU
// oracle: a SNP equal to the missing code takes the node's default child, otherwise left iff float(v) < threshold;
// multi:softprob for A >= 3, binary:logistic for A == 2).  No reference mode selects these bases (README.md:120-146
// shows how to plug them in), they are the "forest-of-stumps" family of the north star.
//
//  * pass 1 (k_pack2): X int8 {0,1,2} -> 2 bits per SNP over the reflect-PADDED coordinate, 16 SNPs per word;
//  * pass 2 (k_base_forest): one wave = 64 haplotypes of one window.  The window's 2-bit words sit in LDS as
//    [word][lane] (lanes on the same split read consecutive banks), its trees — complete depth-D heaps in the packed
//    layout of the smoother, with the default direction in bit 31 of the feature word and the feature as a SNP index —
//    next to them; a lane walks TP trees at a time (independent chains), sums each class in tree order (float32, as
//    the restated predictor does), parks the margins in LDS, then applies softmax / sigmoid in float32.
#include "gnx_internal.h"

namespace {

__device__ __forceinline__ int64_t pad_src(int64_t p, int64_t C, int64_t ctx) {
  if (p < ctx) return ctx - 1 - p;
  if (p < ctx + C) return p - ctx;
  return C - 1 - (p - ctx - C);
}

// one thread = one 16-SNP word of the padded 2-bit matrix
__global__ __launch_bounds__(256) void k_pack2(const int8_t* X, int64_t N, int64_t ldx, int64_t C, int64_t ctx, int64_t nwq,
                                               uint32_t* q) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * nwq) return;
  const int64_t n = idx / nwq, wd = idx - n * nwq;
  const int64_t Cp = C + 2 * ctx;
  const int8_t* x = X + n * ldx;
  uint32_t v = 0;
  for (int b = 0; b < 16; ++b) {
    const int64_t p = wd * 16 + b;
    if (p < Cp) v |= ((uint32_t)(uint8_t)x[pad_src(p, C, ctx)] & 3u) << (2 * b);
  }
  q[n * nwq + wd] = v;
}

constexpr int TP = 4;  // trees walked concurrently per lane

__device__ __forceinline__ float forest_walk(const uint8_t* tb, const uint32_t* xw, int T, int D, int missing) {
  const uint32_t half = 1u << (D - 1);
  uint32_t j = 1;
  for (int d = 0; d < D - 1; ++d) {
    const uint2 nd = *reinterpret_cast<const uint2*>(tb + half * 16 + (j - 1) * 8);
    const uint32_t f = nd.x & 0x7fffffffu;
    const int v = (int)((xw[(f >> 4) * T] >> (2 * (f & 15))) & 3u);
    const bool left = (v == missing) ? (nd.x >> 31) != 0 : ((float)v < __uint_as_float(nd.y));
    j = 2 * j + (left ? 0u : 1u);
  }
  const uint4 n4 = *reinterpret_cast<const uint4*>(tb + (j - half) * 16);
  const uint32_t f = n4.x & 0x7fffffffu;
  const int v = (int)((xw[(f >> 4) * T] >> (2 * (f & 15))) & 3u);
  const bool left = (v == missing) ? (n4.x >> 31) != 0 : ((float)v < __uint_as_float(n4.y));
  return left ? __uint_as_float(n4.z) : __uint_as_float(n4.w);
}

__global__ __launch_bounds__(256) void k_base_forest(ForestLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x, T = blockDim.x;  // T = 64 * waves, one thread = one haplotype
  const int w = blockIdx.y;
  const int A = L.A, D = L.D, tree_bytes = L.tree_bytes;
  const int64_t width = (w == L.W - 1) ? L.width_last : L.width;
  const int t0 = L.win_tree0[w], nt = L.win_tree0[w + 1] - t0;

  uint32_t* xw = reinterpret_cast<uint32_t*>(lds) + lane;                            // [max_words][T], this lane's column
  uint8_t* tr = lds + (size_t)L.max_words * T * 4;                                   // trees of this window
  float* marg = reinterpret_cast<float*>(tr + (((size_t)L.max_trees * tree_bytes + 15) & ~(size_t)15)) + lane;  // [A][T]

  const int64_t n = (int64_t)blockIdx.x * T + lane;
  const int64_t nc = n < L.N ? n : L.N - 1;

  // window bits: 2-bit funnel shift to the window start (padded SNP w*M), 16 SNPs per word
  {
    const int64_t s = (int64_t)w * L.M;
    const int64_t w0 = s >> 4;
    const int sh = (int)(s & 15) * 2;
    const uint32_t* src = L.q + nc * L.nwq + w0;
    const int nw = (int)((width + 15) >> 4);
    for (int i = 0; i < nw; ++i) {
      const uint32_t a = src[i], b = src[i + 1];  // the packed rows carry 2 spare words
      xw[(size_t)i * T] = sh ? ((a >> sh) | (b << (32 - sh))) : a;
    }
  }
  for (int e = lane; e < nt * tree_bytes / 16; e += T)
    reinterpret_cast<uint4*>(tr)[e] = reinterpret_cast<const uint4*>(L.packed + (size_t)t0 * tree_bytes)[e];
  __syncthreads();

  // class-major packing per window: trees of class c are contiguous, in model order
  const int32_t* cls0 = L.win_class_tree0 + (size_t)w * (A + 1);  // [A+1] offsets relative to t0
  const int n_groups = (A == 2) ? 1 : A;
  for (int c = 0; c < n_groups; ++c) {
    const int a0 = (A == 2) ? 0 : cls0[c], a1 = (A == 2) ? nt : cls0[c + 1];
    float psum = 0.f;
    int t = a0;
    for (; t + TP <= a1; t += TP) {
      float leaf[TP];
#pragma unroll
      for (int k = 0; k < TP; ++k) leaf[k] = forest_walk(tr + (size_t)(t + k) * tree_bytes, xw, T, D, L.missing);
#pragma unroll
      for (int k = 0; k < TP; ++k) psum += leaf[k];  // tree order
    }
    for (; t < a1; ++t) psum += forest_walk(tr + (size_t)t * tree_bytes, xw, T, D, L.missing);
    marg[c * T] = psum;
  }
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

}  // namespace

size_t gnx_forest_lds_bytes(int A, int max_words, int max_trees, int tree_bytes, int threads) {
  return (size_t)max_words * threads * 4 + (((size_t)max_trees * tree_bytes + 15) & ~(size_t)15) + (size_t)A * threads * 4;
}

hipError_t gnx_launch_pack2(const int8_t* X, int64_t N, int64_t ldx, int64_t C, int64_t ctx, int64_t nwq, uint32_t* q,
                            hipStream_t s) {
  if (N <= 0) return hipSuccess;
  const int64_t total = N * nwq;
  hipLaunchKernelGGL(k_pack2, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, N, ldx, C, ctx, nwq, q);
  return hipGetLastError();
}

hipError_t gnx_launch_base_forest(const ForestLaunch& L, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  int threads = 256;  // as many waves as fit the 160 KB LDS next to the window's trees
  while (threads > 64 && gnx_forest_lds_bytes(L.A, L.max_words, L.max_trees, L.tree_bytes, threads) > (size_t)160 * 1024) threads -= 64;
  while (threads > 64 && (int64_t)(threads - 64) >= L.N) threads -= 64;
  const size_t lds = gnx_forest_lds_bytes(L.A, L.max_words, L.max_trees, L.tree_bytes, threads);
  if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_base_forest), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_base_forest, dim3((unsigned)((L.N + threads - 1) / threads), (unsigned)L.W), dim3(threads), lds, s, L);
  return hipGetLastError();
}

// k_train_gbt.hip — training the tree smoother on gfx950 (second half).
//
// Replaces Smoother.train of XGB_Smoother (reference src/Smooth/smooth.py:28-38 -> src/Smooth/models.py:14-20:
// XGBClassifier(n_estimators=100, max_depth=4, learning_rate=0.1, reg_lambda=1, objective='multi:softprob',
// num_class=A).fit(slide_window(B, S), y)).  xgboost is a third-party fitter that is not part of the reference tree, so what
// is rebuilt is the algorithm that call asks for — second-order boosting of A regression trees per round on the softmax
// objective — in the histogram form (xgboost's tree_method="hist"), MI355X-first:
//   * the (N*W, S*A) matrix of slide_window is never built: feature s*A + a of row (n, w) is byte (w+s)*A + a of the
//     haplotype's reflect-padded strip of 8-bit BINS (<= 256 quantile bins per class column, cut on a 1/65536 grid);
//   * gradients are rounded to multiples of 2^-30 and every sum is an int64, so histograms built with LDS atomics in any
//     order are exact: the result does not depend on scheduling, and the CPU oracle (gnxo_train_gbt under oracle/)
//     produces IDENTICAL trees — that is the parity test of this path (tests/test_train_gbt.py);
//   * all A trees of a round grow together, level by level: one histogram launch (block = a group of features x one class
//     x a slice of the rows, its 128 KB of LDS holding [feature][node][bin] sums) that visits only the rows of the SMALLER child of
//     every split (the sibling's histogram is parent - built, exact in int64), one split search (a wave per feature: prefix sums
//     over the bins, gains in float64, ties to the lowest feature then the lowest bin), one partition pass;
//   * the softmax goes through det_exp (plain IEEE operations in a fixed order) because the oracle must reproduce p exactly.
// The trees come back in the layout gnx_model_desc takes (tree_off / left / right / feat / cond / tree_class), thresholds on
// the 1/65536 grid, so the trained smoother runs on k_smooth_xgb_rk like any other.
#include <algorithm>
#include <limits>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

#include "gnx_internal.h"

namespace {

constexpr double FIX = 1073741824.0;  // 2^30
constexpr int MAXN = 63;               // heap positions of a tree of depth <= 5
constexpr int HIST_ENTRIES = 8192;     // (g, h) int64 pairs of LDS per histogram block = 128 KB

struct Geom {
  int64_t N, R;
  int32_t W, A, S, pad, Wp, F;
};

__device__ __forceinline__ int slide_src(int j, int W, int pad) {
  if (j < pad) return pad - 1 - j;
  if (j < pad + W) return j - pad;
  return W - 1 - (j - pad - W);
}
__device__ __forceinline__ int bucket_of(float v) {
  int b = (v > 0.0f) ? (int)(v * 65536.0f) : 0;
  return (v >= 1.0f) ? 65535 : b;
}

__device__ __forceinline__ double det_exp(double x) {  // x <= 0; the same operations, in the same order, as the oracle's
  if (!(x > -745.0)) return 0.0;
  const double LOG2E = 1.4426950408889634, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  const double kf = rint(x * LOG2E);
  const double r = (x - kf * LN2_HI) - kf * LN2_LO;
  double p = 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  return ldexp(p, (int)kf);
}

template <bool IS64>
__global__ __launch_bounds__(256) void k_gbt_cast_count(const void* B, float* Bf, uint32_t* cnt, Geom G) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= G.R * G.A) return;
  const float v = IS64 ? (float)reinterpret_cast<const double*>(B)[i] : reinterpret_cast<const float*>(B)[i];
  Bf[i] = v;
  atomicAdd(&cnt[(size_t)(i % G.A) * 65536 + bucket_of(v)], 1u);
}

__global__ __launch_bounds__(256) void k_gbt_quantise(const float* Bf, const uint8_t* lut, uint8_t* Bq, Geom G) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= G.N * G.Wp * G.A) return;
  const int a = (int)(e % G.A);
  const int64_t nj = e / G.A;
  const int j = (int)(nj % G.Wp);
  const int64_t n = nj / G.Wp;
  const float v = Bf[(n * G.W + slide_src(j, G.W, G.pad)) * G.A + a];
  Bq[e] = lut[(size_t)a * 65536 + bucket_of(v)];
}

__global__ __launch_bounds__(256) void k_gbt_fill(float* p, float v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

// one softmax per row for the A trees of the round; root sums per class; log loss (fixed point, informational)
template <int AMAX>
__global__ __launch_bounds__(256) void k_gbt_grad(const float* Fm, const int32_t* y, long long* gq, long long* hq, long long* rootG,
                                                   long long* rootH, long long* loss_fix, uint8_t* pos, int32_t* st, Geom G) {
  __shared__ long long sg[AMAX], sh[AMAX], sl;
  const int A = G.A;
  if (threadIdx.x < AMAX) { sg[threadIdx.x] = 0; sh[threadIdx.x] = 0; }
  if (threadIdx.x == 0) sl = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < G.R) {
    float m = Fm[i * A];
    for (int c = 1; c < A; ++c) m = Fm[i * A + c] > m ? Fm[i * A + c] : m;
    double e[AMAX], sum = 0.0;
#pragma unroll
    for (int c = 0; c < AMAX; ++c)
      if (c < A) { e[c] = det_exp((double)(Fm[i * A + c] - m)); sum += e[c]; }
    const int yi = y[i];
    double py = 0.0;
#pragma unroll
    for (int c = 0; c < AMAX; ++c)
      if (c < A) {
        const double p = e[c] / sum;
        const double g = p - (yi == c ? 1.0 : 0.0);
        double h = 2.0 * p * (1.0 - p);
        if (h < 1e-16) h = 1e-16;
        const long long gi = llrint(g * FIX), hi = llrint(h * FIX);
        gq[(size_t)c * G.R + i] = gi;
        hq[(size_t)c * G.R + i] = hi;
        pos[(size_t)c * G.R + i] = 0;
        atomicAdd(reinterpret_cast<unsigned long long*>(&sg[c]), (unsigned long long)gi);
        atomicAdd(reinterpret_cast<unsigned long long*>(&sh[c]), (unsigned long long)hi);
        if (yi == c) py = p;
      }
    atomicAdd(reinterpret_cast<unsigned long long*>(&sl), (unsigned long long)llrint(-log(py > 1e-300 ? py : 1e-300) * 16777216.0));
  }
  __syncthreads();
  if (threadIdx.x < A) {
    atomicAdd(reinterpret_cast<unsigned long long*>(&rootG[threadIdx.x * MAXN]), (unsigned long long)sg[threadIdx.x]);
    atomicAdd(reinterpret_cast<unsigned long long*>(&rootH[threadIdx.x * MAXN]), (unsigned long long)sh[threadIdx.x]);
    st[threadIdx.x * MAXN] = 1;  // (every block writes the same value)
  }
  if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long*>(loss_fix), (unsigned long long)sl);
}

// histograms of one level for all classes: block = (feature group, class, row slice); LDS [feature][node of level][bin] (g, h)
constexpr int HIST_T = 1024;  // one block per CU (its histogram fills the LDS): 16 waves keep enough row loads in flight
__global__ __launch_bounds__(HIST_T) void k_gbt_hist(const uint8_t* Bq, const long long* gq, const long long* hq, const uint8_t* pos,
                                                   const int32_t* st, const int32_t* build, long long* part, int d, int fpb, int n_slices, Geom G) {
  extern __shared__ __attribute__((aligned(16))) long long sh[];
  const int nl = 1 << d, base = nl - 1;
  const int f0 = blockIdx.x * fpb, c = blockIdx.y, slice = blockIdx.z;
  const int nf = min(fpb, G.F - f0);
  // two planes, [feature][node][bin] of g and the same of h: 8-byte cells, so the 64 lanes of an atomic spread over 32 bank pairs
  // (16-byte (g, h) cells would leave 16: a 4-way conflict at best)
  const int cells = fpb * nl * 256;
  unsigned long long* shg = reinterpret_cast<unsigned long long*>(sh);
  unsigned long long* shh = shg + cells;
  for (int e = threadIdx.x; e < 2 * cells; e += HIST_T) sh[e] = 0;
  __syncthreads();
  const int64_t r0 = G.R * slice / n_slices, r1 = G.R * (slice + 1) / n_slices;
  const uint8_t* posc = pos + (size_t)c * G.R;
  const int32_t* stc = st + c * MAXN;
  const int32_t* bldc = build + c * MAXN;
  for (int64_t i = r0 + threadIdx.x; i < r1; i += HIST_T) {
    const int node = posc[i], k = node - base;
    if (k < 0 || k >= nl || stc[node] != 1 || !bldc[node]) continue;  // (the sibling of a built node is derived: parent - built)
    const int64_t n = i / G.W;
    const int w = (int)(i - n * G.W);
    const unsigned long long g = (unsigned long long)gq[(size_t)c * G.R + i], h = (unsigned long long)hq[(size_t)c * G.R + i];
    const uint8_t* q = Bq + ((size_t)n * G.Wp + w) * G.A + f0;
    for (int f4 = 0; f4 < nf; f4 += 4) {  // four bins per (unaligned) dword load; bytes past the block's features are not used
      uint32_t qv;
      __builtin_memcpy(&qv, q + f4, 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (f4 + u < nf) {
          const int cell = ((f4 + u) * nl + k) * 256 + (int)((qv >> (8 * u)) & 0xffu);
          atomicAdd(shg + cell, g);
          atomicAdd(shh + cell, h);
        }
      }
    }
  }
  __syncthreads();
  // part[slice][c][k][f][bin][2]
  for (int e = threadIdx.x; e < nf * nl * 256; e += HIST_T) {
    const int fi = e / (nl * 256), r = e - fi * nl * 256, k = r / 256, b = r - k * 256;
    long long* dst = part + ((((size_t)slice * G.A + c) * nl + k) * G.F + (f0 + fi)) * 512 + b * 2;
    dst[0] = (long long)shg[e];
    dst[1] = (long long)shh[e];
  }
}

struct Best {
  double gain;
  int f, j;
  long long GL, HL;
};
constexpr int SPLIT_CHUNKS = 16;// the features of a node are searched by this many blocks; k_gbt_split_pick takes the best of them

// best split of every open node of the level: grid (node of level, class, feature chunk); a wave per feature, lanes over the bins
__global__ __launch_bounds__(256) void k_gbt_split(const long long* part, const int32_t* ncut, const long long* nG, const long long* nH,
                                                    const int32_t* st, const int32_t* build, const long long* Hprev, long long* Hcur, Best* cand, int d,
                                                    int n_slices, double lambda, double gamma, double mcw, Geom G) {
  const int nl = 1 << d, base = nl - 1;
  const int k = blockIdx.x, c = blockIdx.y, node = base + k;
  if (st[c * MAXN + node] != 1) return;
  const int fchunk = (G.F + SPLIT_CHUNKS - 1) / SPLIT_CHUNKS, fbeg = blockIdx.z * fchunk, fend = min(G.F, fbeg + fchunk);
  const bool built = build[c * MAXN + node] != 0;
  const int ks = built ? k : (k ^ 1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long Gn = nG[c * MAXN + node], Hn = nH[c * MAXN + node];
  const double Gd = (double)Gn / FIX, Hd = (double)Hn / FIX;
  const double root_term = Gd * Gd / (Hd + lambda);
  // the running best as scalars (see k_gbt_split_pick: no struct selects); a candidate exists iff bf >= 0
  double bg = gamma > 1e-6 ? gamma : 1e-6;
  int bf = -1, bj = -1;
  long long bGL = 0, bHL = 0;
  auto beats = [](double g1, int f1, int j1, double g2, int f2, int j2) {  // larger gain, then lower feature, then lower bin
    return g1 > g2 || (g1 == g2 && (f1 < f2 || (f1 == f2 && j1 < j2)));
  };
  for (int f = fbeg + wave; f < fend; f += 4) {
    const int nc = ncut[f % G.A];
    long long g4[4], h4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { g4[q] = 0; h4[q] = 0; }
    for (int s = 0; s < n_slices; ++s) {
      const long long* src = part + ((((size_t)s * G.A + c) * nl + ks) * G.F + f) * 512 + lane * 8;
#pragma unroll
      for (int q = 0; q < 4; ++q) { g4[q] += src[q * 2]; h4[q] += src[q * 2 + 1]; }
    }
    if (!built) {  // sibling subtraction: this node's histogram = its parent's - its (built) sibling's, exact in int64
      const long long* par = Hprev + (((size_t)c * (nl >> 1) + (k >> 1)) * G.F + f) * 512 + lane * 8;
#pragma unroll
      for (int q = 0; q < 4; ++q) { g4[q] = par[q * 2] - g4[q]; h4[q] = par[q * 2 + 1] - h4[q]; }
    }
    {
      long long* dst = Hcur + (((size_t)c * nl + k) * G.F + f) * 512 + lane * 8;
#pragma unroll
      for (int q = 0; q < 4; ++q) { dst[q * 2] = g4[q]; dst[q * 2 + 1] = h4[q]; }
    }
    // inclusive prefix over the 256 bins: inside the lane, then across lanes
#pragma unroll
    for (int q = 1; q < 4; ++q) { g4[q] += g4[q - 1]; h4[q] += h4[q - 1]; }
    long long og = g4[3], oh = h4[3];
    for (int o = 1; o < 64; o <<= 1) {
      const long long tg = __shfl_up(og, o), th = __shfl_up(oh, o);
      if (lane >= o) { og += tg; oh += th; }
    }
    const long long eg = og - g4[3], eh = oh - h4[3];  // exclusive prefix of this lane
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = lane * 4 + q;
      if (j >= nc) continue;
      const long long GL = eg + g4[q], HL = eh + h4[q];
      const double gl = (double)GL / FIX, hl = (double)HL / FIX;
      const double gr = (double)(Gn - GL) / FIX, hr = (double)(Hn - HL) / FIX;
      if (hl < mcw || hr < mcw) continue;
      const double gain = (gl * gl / (hl + lambda) + gr * gr / (hr + lambda)) - root_term;
      if (bf < 0 ? gain > bg : beats(gain, f, j, bg, bf, bj)) { bg = gain; bf = f; bj = j; bGL = GL; bHL = HL; }
    }
  }
  // wave reduction, then the four waves through LDS
  for (int o = 32; o > 0; o >>= 1) {
    const double tg = __shfl_down(bg, o);
    const int tf = __shfl_down(bf, o), tj = __shfl_down(bj, o);
    const long long tGL = __shfl_down(bGL, o), tHL = __shfl_down(bHL, o);
    if (tf >= 0 && (bf < 0 || beats(tg, tf, tj, bg, bf, bj))) { bg = tg; bf = tf; bj = tj; bGL = tGL; bHL = tHL; }
  }
  __shared__ double wg[4];
  __shared__ int wf[4], wj[4];
  __shared__ long long wGL[4], wHL[4];
  if (lane == 0) { wg[wave] = bg; wf[wave] = bf; wj[wave] = bj; wGL[wave] = bGL; wHL[wave] = bHL; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int bw = -1;
    for (int w = 0; w < 4; ++w)
      if (wf[w] >= 0 && (bw < 0 || beats(wg[w], wf[w], wj[w], wg[bw], wf[bw], wj[bw]))) bw = w;
    Best* dst = cand + ((size_t)c * nl + k) * SPLIT_CHUNKS + blockIdx.z;
    dst->gain = bw >= 0 ? wg[bw] : 0.0;
    dst->f = bw >= 0 ? wf[bw] : -1;
    dst->j = bw >= 0 ? wj[bw] : -1;
    dst->GL = bw >= 0 ? wGL[bw] : 0;
    dst->HL = bw >= 0 ? wHL[bw] : 0;
  }
}

__global__ void k_gbt_split_pick(const Best* cand, long long* nG, long long* nH, int32_t* nF, int32_t* nB, int32_t* st, int d, int A) {
  const int nl = 1 << d, base = nl - 1;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= A * nl) return;
  const int c = e / nl, k = e - c * nl, node = base + k, o = c * MAXN;
  if (st[o + node] != 1) return;
  // (scalars and an index, not struct copies: hipcc 7.2 mixed the fields of two `Best` values in the select it built for `b = t`)
  const Best* cq = cand + (size_t)e * SPLIT_CHUNKS;
  int bi = -1, bf = -1, bj = -1;
  double bg = 0.0;
  for (int q = 0; q < SPLIT_CHUNKS; ++q) {
    const double tg = cq[q].gain;
    const int tf = cq[q].f, tj = cq[q].j;
    if (tf < 0) continue;
    if (bi < 0 || tg > bg || (tg == bg && (tf < bf || (tf == bf && tj < bj)))) { bi = q; bg = tg; bf = tf; bj = tj; }
  }
  if (bi >= 0) {
    const long long GL = cq[bi].GL, HL = cq[bi].HL;
    const long long Gn = nG[o + node], Hn = nH[o + node];
    st[o + node] = 2; nF[o + node] = bf; nB[o + node] = bj;
    const int l = 2 * node + 1, r = 2 * node + 2;
    st[o + l] = 1; st[o + r] = 1;
    nG[o + l] = GL; nH[o + l] = HL; nG[o + r] = Gn - GL; nH[o + r] = Hn - HL;
  } else {
    st[o + node] = 3;
  }
}

__global__ __launch_bounds__(256) void k_gbt_partition(const uint8_t* Bq, uint8_t* pos, const int32_t* st, const int32_t* nF, const int32_t* nB,
                                                        int32_t* cnt, int d, Geom G) {
  __shared__ int lc[64];
  if (threadIdx.x < 64) lc[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (i < G.R) {
    const int node = pos[(size_t)c * G.R + i];
    if (node >= (1 << d) - 1 && st[c * MAXN + node] == 2) {
      const int64_t n = i / G.W;
      const int w = (int)(i - n * G.W);
      const uint8_t q = Bq[((size_t)n * G.Wp + w) * G.A + nF[c * MAXN + node]];
      const int child = q <= nB[c * MAXN + node] ? 2 * node + 1 : 2 * node + 2;
      pos[(size_t)c * G.R + i] = (uint8_t)child;
      atomicAdd(&lc[child], 1);  // rows per child: the smaller child of a pair gets its histogram built, the other derived
    }
  }
  __syncthreads();
  if (threadIdx.x < MAXN && lc[threadIdx.x]) atomicAdd(&cnt[c * MAXN + threadIdx.x], lc[threadIdx.x]);
}

// which child of every node split at level d gets its histogram built at level d + 1 (the one with fewer rows; ties: the left)
__global__ void k_gbt_choose(const int32_t* st, const int32_t* cnt, int32_t* build, int d, int A) {
  const int nl = 1 << d, base = nl - 1;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= A * nl) return;
  const int c = e / nl, node = base + (e - c * nl), o = c * MAXN;
  if (st[o + node] != 2) return;
  const int l = 2 * node + 1, r = 2 * node + 2;
  const bool left = cnt[o + l] <= cnt[o + r];
  build[o + l] = left ? 1 : 0;
  build[o + r] = left ? 0 : 1;
}

__global__ void k_gbt_close(const long long* nG, const long long* nH, int32_t* st, float* nV, int A, double eta, double lambda) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= A * MAXN) return;
  if (st[e] == 1) st[e] = 3;
  if (st[e] == 3) {
    const double Gd = (double)nG[e] / FIX, Hd = (double)nH[e] / FIX;
    nV[e] = (float)(eta * (-Gd / (Hd + lambda)));
  }
}

__global__ __launch_bounds__(256) void k_gbt_margin(float* Fm, const uint8_t* pos, const float* nV, Geom G) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= G.R * G.A) return;
  const int c = (int)(e % G.A);
  const int64_t i = e / G.A;
  Fm[e] += nV[c * MAXN + pos[(size_t)c * G.R + i]];
}

// ---- exact greedy (gnx_gbt_params::tree_method = 1) ------------------------------------------------------------------------------
// xgboost's tree_method="exact" (what XGBClassifier(...) of src/Smooth/models.py:14-20 uses for data of this size; ColMaker's column
// scan restated, NOT pinned to xgboost: absent from the image): per node and feature the node's rows in ascending feature value,
// a candidate between every two consecutive DISTINCT values v0 < v1: rows up to v0 left, threshold (v0 + v1) * 0.5f (v1 when that
// rounds down to v0).  All features s*A + a of one class column a read the same values — the haplotypes' padded strips of that column —
// so ONE sorted list of strip positions per column (sorted on the host once per training) serves its S features: feature s sees
// position (n, j) as row (n, j - s).  k_gbt_exact_scan: one wave per (feature, class tree), 64 sorted positions per step; for every
// node present in the step a masked wave scan gives each of its rows the node's sums before it and the previous value of the node
// (the last row of an earlier step is carried in LDS); gains in float64 from exact int64 sums, first-best-wins in scan order — the
// CPU oracle (gnxo_train_gbt, tree_method 1: a qsort per node and feature) grows IDENTICAL trees (tests/test_train_gbt.py).
struct ExBest {
  double gain;   // < 0: no admissible split of this node on this feature
  float thr;
  int32_t pad_;
  long long GL, HL;
};

__device__ __forceinline__ float gbt_mid(float v0, float v1) {
  const float m = (v0 + v1) * 0.5f;
  return m > v0 ? m : v1;
}

__global__ __launch_bounds__(64) void k_gbt_exact_scan(const int32_t* __restrict__ ordn, const int32_t* __restrict__ ordj, const float* __restrict__ val,
                                                      const long long* __restrict__ gq, const long long* __restrict__ hq, const uint8_t* __restrict__ pos,
                                                      const int32_t* __restrict__ st, const long long* __restrict__ nG, const long long* __restrict__ nH,
                                                      ExBest* __restrict__ cand, int d, double lambda, double gamma, double mcw, Geom G) {
  __shared__ long long accG[32], accH[32], bGL[32], bHL[32], nodeG[32], nodeH[32];
  __shared__ double bgain[32];
  __shared__ float lastv[32], bthr[32];
  __shared__ int seen[32], open_[32], found[32];
  const int nl = 1 << d, base = nl - 1, lane = threadIdx.x;
  const int f = blockIdx.x, c = blockIdx.y, sft = f / G.A, a = f - sft * G.A;
  if (lane < nl) {
    const int node = base + lane;
    accG[lane] = 0; accH[lane] = 0; seen[lane] = 0; found[lane] = 0;
    open_[lane] = st[c * MAXN + node] == 1;
    nodeG[lane] = nG[c * MAXN + node]; nodeH[lane] = nH[c * MAXN + node];
    bgain[lane] = gamma > 1e-6 ? gamma : 1e-6;
  }
  __syncthreads();
  const int64_t NP = G.N * G.Wp, col = (int64_t)a * NP;
  const long long* gc = gq + (size_t)c * G.R;
  const long long* hc = hq + (size_t)c * G.R;
  const uint8_t* pc = pos + (size_t)c * G.R;
  for (int64_t p0 = 0; p0 < NP; p0 += 64) {
    const int64_t p = p0 + lane;
    const bool in = p < NP;
    const int n = in ? ordn[col + p] : 0, j = in ? ordj[col + p] : 0;
    const float v = in ? val[col + p] : 0.f;
    const int w = j - sft;
    const bool valid = in && w >= 0 && w < G.W;
    const int64_t row = valid ? (int64_t)n * G.W + w : 0;
    const int k = valid ? (int)pc[row] - base : -1;
    const bool live = valid && k >= 0 && k < nl && open_[k];
    const long long g = live ? gc[row] : 0, h = live ? hc[row] : 0;
    unsigned long long rem = __ballot(live);
    while (rem) {
      const int kcur = __shfl(k, __builtin_ctzll(rem));
      const bool mask = live && k == kcur;
      const unsigned long long bal = __ballot(mask);
      rem &= ~bal;
      long long sg = mask ? g : 0, sh = mask ? h : 0;
      const long long ig = sg, ih = sh;
      for (int o = 1; o < 64; o <<= 1) {
        const long long tg = __shfl_up(sg, o), th = __shfl_up(sh, o);
        if (lane >= o) { sg += tg; sh += th; }
      }
      const unsigned long long below = bal & ((1ull << lane) - 1ull);
      bool hasp = below != 0;
      float vp = __shfl(v, hasp ? 63 - __builtin_clzll(below) : 0);
      if (!hasp) { vp = lastv[kcur]; hasp = seen[kcur] != 0; }
      bool ok = false;
      double gain = 0.0;
      long long GL = 0, HL = 0;
      if (mask && hasp && vp < v) {
        GL = accG[kcur] + (sg - ig);
        HL = accH[kcur] + (sh - ih);
        const double gl = (double)GL / FIX, hl = (double)HL / FIX;
        const double gr = (double)(nodeG[kcur] - GL) / FIX, hr = (double)(nodeH[kcur] - HL) / FIX;
        if (!(hl < mcw || hr < mcw)) {
          const double Gd = (double)nodeG[kcur] / FIX, Hd = (double)nodeH[kcur] / FIX;
          gain = (gl * gl / (hl + lambda) + gr * gr / (hr + lambda)) - Gd * Gd / (Hd + lambda);
          ok = gain > bgain[kcur];   // strictly better than every earlier candidate of this node on this feature
        }
      }
      const unsigned long long cb = __ballot(ok);
      if (cb) {  // the largest gain of the step, the earliest lane among equals
        double mx = ok ? gain : -1.0;
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
        const int win = __builtin_ctzll(__ballot(ok && gain == mx));
        if (lane == win) { bgain[kcur] = gain; bthr[kcur] = gbt_mid(vp, v); bGL[kcur] = GL; bHL[kcur] = HL; found[kcur] = 1; }
      }
      if (lane == 63 - __builtin_clzll(bal)) { accG[kcur] += sg; accH[kcur] += sh; lastv[kcur] = v; seen[kcur] = 1; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __syncthreads();
  if (lane < nl) {
    ExBest* dst = cand + ((size_t)c * nl + lane) * G.F + f;
    dst->gain = found[lane] ? bgain[lane] : -1.0;
    dst->thr = bthr[lane];
    dst->pad_ = 0;
    dst->GL = bGL[lane];
    dst->HL = bHL[lane];
  }
}

// best feature of every open node: features in ascending order, strictly larger gain replaces (first best wins)
__global__ void k_gbt_exact_pick(const ExBest* cand, long long* nG, long long* nH, int32_t* nF, int32_t* nB, int32_t* st, int d, int A, int F) {
  const int nl = 1 << d, base = nl - 1;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= A * nl) return;
  const int c = e / nl, k = e - c * nl, node = base + k, o = c * MAXN;
  if (st[o + node] != 1) return;
  const ExBest* cq = cand + (size_t)e * F;
  int bf = -1;
  double bg = 0.0;
  for (int f = 0; f < F; ++f) {
    const double tg = cq[f].gain;
    if (tg < 0.0) continue;
    if (bf < 0 || tg > bg) { bf = f; bg = tg; }
  }
  if (bf >= 0) {
    const long long GL = cq[bf].GL, HL = cq[bf].HL;
    const long long Gn = nG[o + node], Hn = nH[o + node];
    st[o + node] = 2; nF[o + node] = bf; nB[o + node] = __float_as_int(cq[bf].thr);
    const int l = 2 * node + 1, r = 2 * node + 2;
    st[o + l] = 1; st[o + r] = 1;
    nG[o + l] = GL; nH[o + l] = HL; nG[o + r] = Gn - GL; nH[o + r] = Hn - HL;
  } else {
    st[o + node] = 3;
  }
}

__global__ __launch_bounds__(256) void k_gbt_exact_partition(const float* Bf, uint8_t* pos, const int32_t* st, const int32_t* nF, const int32_t* nB, int d, Geom G) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (i >= G.R) return;
  const int node = pos[(size_t)c * G.R + i];
  if (node >= (1 << d) - 1 && st[c * MAXN + node] == 2) {
    const int64_t n = i / G.W;
    const int w = (int)(i - n * G.W);
    const int f = nF[c * MAXN + node], sft = f / G.A, a = f - sft * G.A;
    const float v = Bf[(n * G.W + slide_src(w + sft, G.W, G.pad)) * G.A + a];
    pos[(size_t)c * G.R + i] = (uint8_t)(v < __int_as_float(nB[c * MAXN + node]) ? 2 * node + 1 : 2 * node + 2);
  }
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

}  // namespace

#define GBT_HIP(x)                      \
  do {                                  \
    const hipError_t e_ = (x);          \
    if (e_ != hipSuccess) return e_;    \
  } while (0)

// dB (N, W, A) float32 / float64 and dy (N, W) int32 on the device.  Host outputs as in gnx_train_gbt (include/gnomix_hip.h).
hipError_t gnx_train_gbt_run(const void* dB, int b_is_f64, const int32_t* dy, int64_t N, int32_t W, int32_t A, int32_t S,
                             const gnx_gbt_params& P, int32_t* tree_off, int32_t* tree_class, int32_t* left, int32_t* right,
                             int32_t* feat, float* cond, int64_t* n_nodes_out, double* loss_out, int n_cu, hipStream_t s) {
  Geom G;
  G.N = N; G.W = W; G.A = A; G.S = S; G.pad = (S + 1) / 2; G.Wp = W + 2 * G.pad; G.F = S * A; G.R = N * (int64_t)W;
  const int D = P.max_depth, T = P.n_rounds * A;
  const int64_t RA = G.R * A;
  DevBuf bBf, bCnt, bLut, bBq, bFm, bG, bH, bPos, bPart, bTab, bNcut, bLoss, bCand, bRows, bLev;
  GBT_HIP(bBf.alloc((size_t)RA * 4));
  GBT_HIP(bCnt.alloc((size_t)A * 65536 * 4));
  GBT_HIP(bLut.alloc((size_t)A * 65536));
  GBT_HIP(bBq.alloc((size_t)N * G.Wp * A + 64));
  GBT_HIP(bFm.alloc((size_t)RA * 4));
  GBT_HIP(bG.alloc((size_t)RA * 8));
  GBT_HIP(bH.alloc((size_t)RA * 8));
  GBT_HIP(bPos.alloc((size_t)RA));
  GBT_HIP(bNcut.alloc((size_t)A * 4));
  GBT_HIP(bLoss.alloc((size_t)(P.n_rounds + 1) * 8));
  GBT_HIP(bCand.alloc((size_t)A * 32 * SPLIT_CHUNKS * sizeof(Best)));
  // node tables of every tree: [T][63] x {G, H (int64), F, B, st (int32), V (float)}
  const size_t tab_n = (size_t)T * MAXN;
  GBT_HIP(bTab.alloc(tab_n * (8 + 8 + 4 + 4 + 4 + 4)));
  long long* tG = bTab.as<long long>();
  long long* tH = tG + tab_n;
  int32_t* tF = reinterpret_cast<int32_t*>(tH + tab_n);
  int32_t* tB = tF + tab_n;
  int32_t* tS = tB + tab_n;
  float* tV = reinterpret_cast<float*>(tS + tab_n);
  GBT_HIP(hipMemsetAsync(bTab.p, 0, tab_n * 32, s));
  GBT_HIP(hipMemsetAsync(bCnt.p, 0, (size_t)A * 65536 * 4, s));
  GBT_HIP(hipMemsetAsync(bLoss.p, 0, (size_t)(P.n_rounds + 1) * 8, s));

  // ---- cuts ----
  const unsigned gRA = (unsigned)((RA + 255) / 256);
  if (b_is_f64) hipLaunchKernelGGL(k_gbt_cast_count<true>, dim3(gRA), dim3(256), 0, s, dB, bBf.as<float>(), bCnt.as<uint32_t>(), G);
  else hipLaunchKernelGGL(k_gbt_cast_count<false>, dim3(gRA), dim3(256), 0, s, dB, bBf.as<float>(), bCnt.as<uint32_t>(), G);
  GBT_HIP(hipGetLastError());
  std::vector<uint32_t> cnt((size_t)A * 65536);
  GBT_HIP(hipMemcpyAsync(cnt.data(), bCnt.p, cnt.size() * 4, hipMemcpyDeviceToHost, s));
  GBT_HIP(hipStreamSynchronize(s));
  std::vector<float> cuts((size_t)A * 256, 0.f);
  std::vector<int32_t> ncut((size_t)A, 0);
  std::vector<uint8_t> lut((size_t)A * 65536);
  for (int a = 0; a < A; ++a) {  // the k-th cut is the upper edge of the first bucket whose cumulative count reaches k*R/max_bin
    int k = 1, nc = 0;
    uint64_t cum = 0;
    for (int u = 0; u < 65536 && k < P.max_bin; ++u) {
      cum += cnt[(size_t)a * 65536 + u];
      bool hit = false;
      while (k < P.max_bin && cum >= (uint64_t)k * (uint64_t)G.R / (uint64_t)P.max_bin) { ++k; hit = true; }
      if (hit && u < 65535) cuts[(size_t)a * 256 + nc++] = (float)(u + 1) / 65536.0f;
    }
    ncut[(size_t)a] = nc;
    int c = 0;
    for (int u = 0; u < 65536; ++u) {
      while (c < nc && (int)(cuts[(size_t)a * 256 + c] * 65536.0f) <= u) ++c;
      lut[(size_t)a * 65536 + u] = (uint8_t)c;
    }
  }
  GBT_HIP(hipMemcpyAsync(bLut.p, lut.data(), lut.size(), hipMemcpyHostToDevice, s));
  GBT_HIP(hipMemcpyAsync(bNcut.p, ncut.data(), (size_t)A * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_gbt_quantise, dim3((unsigned)((N * G.Wp * A + 255) / 256)), dim3(256), 0, s, bBf.as<float>(), bLut.as<uint8_t>(),
                     bBq.as<uint8_t>(), G);
  hipLaunchKernelGGL(k_gbt_fill, dim3(gRA), dim3(256), 0, s, bFm.as<float>(), (float)P.base_score, RA);
  GBT_HIP(hipGetLastError());

  // ---- exact greedy: one sorted list of padded strip positions per class column (host sort, once per training) ----
  const bool exact = P.tree_method == 1;
  DevBuf bOrdN, bOrdJ, bVal, bEx;
  if (exact) {
    const int64_t NP = N * G.Wp;
    std::vector<float> hB((size_t)RA);
    GBT_HIP(hipMemcpyAsync(hB.data(), bBf.p, (size_t)RA * 4, hipMemcpyDeviceToHost, s));
    GBT_HIP(hipStreamSynchronize(s));
    std::vector<int32_t> on((size_t)A * NP), oj((size_t)A * NP);
    std::vector<float> ov((size_t)A * NP);
    auto sort_col = [&](int a) {
      std::vector<float> v((size_t)NP);
      for (int64_t n = 0; n < N; ++n)
        for (int64_t j = 0; j < G.Wp; ++j) {
          const int64_t src = j < G.pad ? G.pad - 1 - j : (j < G.pad + W ? j - G.pad : W - 1 - (j - G.pad - W));
          const float f = hB[(size_t)((n * W + src) * A + a)];
          v[(size_t)(n * G.Wp + j)] = f == f ? f : std::numeric_limits<float>::infinity();  // a NaN would break the comparator's strict weak order
        }
      std::vector<int32_t> idx((size_t)NP);
      std::iota(idx.begin(), idx.end(), 0);
      std::sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return v[(size_t)x] < v[(size_t)y] || (v[(size_t)x] == v[(size_t)y] && x < y); });
      for (int64_t p = 0; p < NP; ++p) {
        const int32_t q = idx[(size_t)p];
        on[(size_t)a * NP + p] = (int32_t)(q / G.Wp);
        oj[(size_t)a * NP + p] = (int32_t)(q % G.Wp);
        ov[(size_t)a * NP + p] = v[(size_t)q];
      }
    };
    {
      std::vector<std::thread> th;
      for (int a = 0; a < A; ++a) {
        try { th.emplace_back(sort_col, a); } catch (...) { sort_col(a); }
      }
      for (auto& t : th) t.join();
    }
    GBT_HIP(bOrdN.alloc((size_t)A * NP * 4));
    GBT_HIP(bOrdJ.alloc((size_t)A * NP * 4));
    GBT_HIP(bVal.alloc((size_t)A * NP * 4));
    GBT_HIP(bEx.alloc((size_t)A * 16 * G.F * sizeof(ExBest)));
    GBT_HIP(hipMemcpyAsync(bOrdN.p, on.data(), (size_t)A * NP * 4, hipMemcpyHostToDevice, s));
    GBT_HIP(hipMemcpyAsync(bOrdJ.p, oj.data(), (size_t)A * NP * 4, hipMemcpyHostToDevice, s));
    GBT_HIP(hipMemcpyAsync(bVal.p, ov.data(), (size_t)A * NP * 4, hipMemcpyHostToDevice, s));
    GBT_HIP(hipStreamSynchronize(s));   // (the host vectors leave scope)
  }

  // ---- histogram geometry per level: features per block from the LDS budget, row slices to fill the chip ----
  int fpb[8], nsl[8];
  size_t part_entries = 0;
  for (int d = 0; d < D; ++d) {
    const int nl = 1 << d;
    fpb[d] = std::max(1, HIST_ENTRIES / (nl * 256));
    fpb[d] = std::min(fpb[d], (int)G.F);
    const int fg = (G.F + fpb[d] - 1) / fpb[d];
    int ns = (int)std::max<int64_t>(1, ((int64_t)8 * n_cu + (int64_t)fg * A - 1) / ((int64_t)fg * A));
    ns = (int)std::min<int64_t>(ns, std::max<int64_t>(1, G.R / 1024));
    nsl[d] = ns;
    part_entries = std::max(part_entries, (size_t)ns * A * nl * G.F * 512);
  }
  GBT_HIP(bPart.alloc(part_entries * 8));
  // rows per node and "build this node's histogram" flags of the round's trees; the level's summed histograms, this level's and the
  // previous one's (a derived node reads its parent's)
  GBT_HIP(bRows.alloc((size_t)2 * A * MAXN * 4));
  int32_t* dCnt = bRows.as<int32_t>();
  int32_t* dBuild = dCnt + (size_t)A * MAXN;
  const size_t lev_entries = (size_t)A * ((size_t)1 << (D - 1)) * G.F * 512;
  GBT_HIP(bLev.alloc(2 * lev_entries * 8));
  GNX_LDS_OPTIN((size_t)HIST_ENTRIES * 16, k_gbt_hist);

  const unsigned gR = (unsigned)((G.R + 255) / 256);
  for (int r = 0; r <= P.n_rounds; ++r) {
    const size_t o = (size_t)std::min(r, P.n_rounds - 1) * A * MAXN;  // (the extra pass after the last round only reports the loss)
    long long* rG = tG + o;
    long long* rH = tH + o;
    int32_t* rF = tF + o;
    int32_t* rB = tB + o;
    int32_t* rS = tS + o;
    float* rV = tV + o;
    if (r == P.n_rounds) {  // loss after the last round: gradients into scratch tables that are not read again
      DevBuf scratch;
      GBT_HIP(scratch.alloc((size_t)A * MAXN * 24));
      GBT_HIP(hipMemsetAsync(scratch.p, 0, (size_t)A * MAXN * 24, s));
      long long* sG = scratch.as<long long>();
      if (A <= 8) hipLaunchKernelGGL(k_gbt_grad<8>, dim3(gR), dim3(256), 0, s, bFm.as<float>(), dy, bG.as<long long>(), bH.as<long long>(), sG, sG + A * MAXN, bLoss.as<long long>() + r, bPos.as<uint8_t>(), reinterpret_cast<int32_t*>(sG + 2 * A * MAXN), G);
      else if (A <= 16) hipLaunchKernelGGL(k_gbt_grad<16>, dim3(gR), dim3(256), 0, s, bFm.as<float>(), dy, bG.as<long long>(), bH.as<long long>(), sG, sG + A * MAXN, bLoss.as<long long>() + r, bPos.as<uint8_t>(), reinterpret_cast<int32_t*>(sG + 2 * A * MAXN), G);
      else hipLaunchKernelGGL(k_gbt_grad<32>, dim3(gR), dim3(256), 0, s, bFm.as<float>(), dy, bG.as<long long>(), bH.as<long long>(), sG, sG + A * MAXN, bLoss.as<long long>() + r, bPos.as<uint8_t>(), reinterpret_cast<int32_t*>(sG + 2 * A * MAXN), G);
      GBT_HIP(hipGetLastError());
      GBT_HIP(hipStreamSynchronize(s));
      break;
    }
    if (A <= 8) hipLaunchKernelGGL(k_gbt_grad<8>, dim3(gR), dim3(256), 0, s, bFm.as<float>(), dy, bG.as<long long>(), bH.as<long long>(), rG, rH, bLoss.as<long long>() + r, bPos.as<uint8_t>(), rS, G);
    else if (A <= 16) hipLaunchKernelGGL(k_gbt_grad<16>, dim3(gR), dim3(256), 0, s, bFm.as<float>(), dy, bG.as<long long>(), bH.as<long long>(), rG, rH, bLoss.as<long long>() + r, bPos.as<uint8_t>(), rS, G);
    else hipLaunchKernelGGL(k_gbt_grad<32>, dim3(gR), dim3(256), 0, s, bFm.as<float>(), dy, bG.as<long long>(), bH.as<long long>(), rG, rH, bLoss.as<long long>() + r, bPos.as<uint8_t>(), rS, G);
    GBT_HIP(hipMemsetAsync(dCnt, 0, (size_t)A * MAXN * 4, s));
    GBT_HIP(hipMemsetAsync(dBuild, 0xff, (size_t)A * MAXN * 4, s));  // (the roots are built; children are decided by k_gbt_choose)
    for (int d = 0; exact && d < D; ++d) {
      const int nl = 1 << d;
      hipLaunchKernelGGL(k_gbt_exact_scan, dim3((unsigned)G.F, (unsigned)A), dim3(64), 0, s, bOrdN.as<int32_t>(), bOrdJ.as<int32_t>(), bVal.as<float>(),
                         bG.as<long long>(), bH.as<long long>(), bPos.as<uint8_t>(), rS, rG, rH, bEx.as<ExBest>(), d, P.lambda, P.gamma, P.min_child_weight, G);
      hipLaunchKernelGGL(k_gbt_exact_pick, dim3((unsigned)((A * nl + 63) / 64)), dim3(64), 0, s, bEx.as<ExBest>(), rG, rH, rF, rB, rS, d, A, (int)G.F);
      hipLaunchKernelGGL(k_gbt_exact_partition, dim3(gR, (unsigned)A), dim3(256), 0, s, bBf.as<float>(), bPos.as<uint8_t>(), rS, rF, rB, d, G);
    }
    for (int d = 0; !exact && d < D; ++d) {
      const int nl = 1 << d, fg = (G.F + fpb[d] - 1) / fpb[d];
      long long* Hcur = bLev.as<long long>() + (size_t)(d & 1) * lev_entries;
      const long long* Hprev = bLev.as<long long>() + (size_t)((d & 1) ^ 1) * lev_entries;
      hipLaunchKernelGGL(k_gbt_hist, dim3((unsigned)fg, (unsigned)A, (unsigned)nsl[d]), dim3(HIST_T), (size_t)fpb[d] * nl * 512 * 8, s, bBq.as<uint8_t>(),
                         bG.as<long long>(), bH.as<long long>(), bPos.as<uint8_t>(), rS, dBuild, bPart.as<long long>(), d, fpb[d], nsl[d], G);
      hipLaunchKernelGGL(k_gbt_split, dim3((unsigned)nl, (unsigned)A, SPLIT_CHUNKS), dim3(256), 0, s, bPart.as<long long>(), bNcut.as<int32_t>(), rG,
                         rH, rS, dBuild, Hprev, Hcur, bCand.as<Best>(), d, nsl[d], P.lambda, P.gamma, P.min_child_weight, G);
      hipLaunchKernelGGL(k_gbt_split_pick, dim3((unsigned)((A * nl + 63) / 64)), dim3(64), 0, s, bCand.as<Best>(), rG, rH, rF, rB, rS, d, A);
      hipLaunchKernelGGL(k_gbt_partition, dim3(gR, (unsigned)A), dim3(256), 0, s, bBq.as<uint8_t>(), bPos.as<uint8_t>(), rS, rF, rB, dCnt, d, G);
      if (d + 1 < D) hipLaunchKernelGGL(k_gbt_choose, dim3((unsigned)((A * nl + 63) / 64)), dim3(64), 0, s, rS, dCnt, dBuild, d, A);
    }
    hipLaunchKernelGGL(k_gbt_close, dim3((unsigned)((A * MAXN + 255) / 256)), dim3(256), 0, s, rG, rH, rS, rV, A, P.eta, P.lambda);
    hipLaunchKernelGGL(k_gbt_margin, dim3(gRA), dim3(256), 0, s, bFm.as<float>(), bPos.as<uint8_t>(), rV, G);
    GBT_HIP(hipGetLastError());
  }

  // ---- trees back to the host, nodes in heap order ----
  std::vector<int32_t> hF(tab_n), hB(tab_n), hS(tab_n);
  std::vector<float> hV(tab_n);
  std::vector<long long> hL((size_t)P.n_rounds + 1);
  GBT_HIP(hipMemcpyAsync(hF.data(), tF, tab_n * 4, hipMemcpyDeviceToHost, s));
  GBT_HIP(hipMemcpyAsync(hB.data(), tB, tab_n * 4, hipMemcpyDeviceToHost, s));
  GBT_HIP(hipMemcpyAsync(hS.data(), tS, tab_n * 4, hipMemcpyDeviceToHost, s));
  GBT_HIP(hipMemcpyAsync(hV.data(), tV, tab_n * 4, hipMemcpyDeviceToHost, s));
  GBT_HIP(hipMemcpyAsync(hL.data(), bLoss.p, hL.size() * 8, hipMemcpyDeviceToHost, s));
  GBT_HIP(hipStreamSynchronize(s));
  int64_t nn = 0;
  tree_off[0] = 0;
  for (int t = 0; t < T; ++t) {
    const size_t o = (size_t)t * MAXN;
    int32_t idx[MAXN];
    int32_t cntn = 0;
    for (int node = 0; node < MAXN; ++node) idx[node] = (hS[o + node] >= 2) ? cntn++ : -1;
    for (int node = 0; node < MAXN; ++node) {
      if (hS[o + node] < 2) continue;
      const int64_t q = nn + idx[node];
      if (hS[o + node] == 2) {
        left[q] = idx[2 * node + 1]; right[q] = idx[2 * node + 2]; feat[q] = hF[o + node];
        if (exact) std::memcpy(&cond[q], &hB[o + node], 4);   // the threshold's float32 bits
        else cond[q] = cuts[(size_t)(hF[o + node] % A) * 256 + hB[o + node]];
      } else {
        left[q] = -1; right[q] = -1; feat[q] = 0; cond[q] = hV[o + node];
      }
    }
    nn += cntn;
    tree_off[t + 1] = (int32_t)nn;
    tree_class[t] = t % A;
  }
  *n_nodes_out = nn;
  if (loss_out)
    for (int r = 0; r <= P.n_rounds; ++r) loss_out[r] = ((double)hL[(size_t)r] / 16777216.0) / (double)G.R;
  return hipSuccess;
}

// k_base_covrsk.hip — covering-random-string-kernel SVC base classifiers on gfx950.
//
// Replaces CovRSKBase.predict_proba (reference src/Base/models.py:195-215): per window
//   K = CovRSK_DP_triangular_numbers(Xw, Xfit)        src/Base/string_kernel.py:91-110
//   sklearn SVC(kernel=callable, probability=True).predict_proba  -> libsvm predict_values /
//   sigmoid_predict / multiclass_probability (sklearn/svm/src/libsvm/svm.cpp, third-party).
//
// The kernel value is an exact integer: over a maximal run of L equal symbols the reference adds
// g(L) = sum_{m in Ms, m<=L} (L-m+1)  (cov_tri counts the m <= run-so-far at every matched position), so
//   K(x,y) = sum over maximal match runs of g(run length).
// Design:
//  * pass 1 (k_pack_bits): X int8 {0,1,2} -> two bit-planes over the reflect-PADDED coordinate (base.py:41-44),
//    so a window is a bit range and symbol equality is ~((xl^yl)|(xh^yh)): 32 SNP compares per 3 VALU ops;
//  * pass 2 (k_covrsk_svc): one wave = 64 query haplotypes of one window; the query's window bits sit in LDS
//    [word][lane]; the support vector is WAVE-UNIFORM, so its bit-planes and dual coefficients come through
//    the scalar unit; match runs are peeled with ctz and looked up in an LDS copy of g; the pairwise decision
//    values accumulate in float64 in libsvm's own order (class-major, SV order inside a class), then the
//    Platt sigmoids and the Wu-Lin-Weng coupling iteration run per lane on LDS-resident [index][lane] arrays.
// Integer-ALU bound: W * n_sv * width symbol compares per haplotype.
#include "gnx_internal.h"

namespace {

__device__ __forceinline__ int64_t pad_src(int64_t p, int64_t C, int64_t ctx) {
  if (p < ctx) return ctx - 1 - p;
  if (p < ctx + C) return p - ctx;
  return C - 1 - (p - ctx - C);
}

// one thread = one 32-SNP word of the padded bit-planes
__global__ __launch_bounds__(256) void k_pack_bits(const int8_t* X, int64_t N, int64_t ldx, int64_t C, int64_t ctx,
                                                    int64_t nwp, uint32_t* planes) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * nwp) return;
  const int64_t n = idx / nwp, wd = idx - n * nwp;
  const int64_t Cp = C + 2 * ctx;
  const int8_t* x = X + n * ldx;
  uint32_t lo = 0, hi = 0;
  for (int b = 0; b < 32; ++b) {
    const int64_t p = wd * 32 + b;
    if (p < Cp) {
      const uint32_t v = (uint32_t)(uint8_t)x[pad_src(p, C, ctx)];
      lo |= (v & 1u) << b;
      hi |= ((v >> 1) & 1u) << b;
    }
  }
  planes[(n * 2 + 0) * nwp + wd] = lo;
  planes[(n * 2 + 1) * nwp + wd] = hi;
}

__device__ __forceinline__ int pair_index(int i, int j, int A) { return i * (2 * A - i - 1) / 2 + (j - i - 1); }

__device__ __forceinline__ double sigmoid_predict(double dec, double pa, double pb) {
  const double f = dec * pa + pb;
  if (f >= 0) return exp(-f) / (1.0 + exp(-f));
  return 1.0 / (1.0 + exp(f));
}

__global__ __launch_bounds__(64) void k_covrsk_svc(CovRSKLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x;
  const int w = blockIdx.y;
  const int A = L.A, P = A * (A - 1) / 2;
  const SvcWinDev win = L.win[w];
  const int NW = win.nw, width = win.width, n_sv = win.n_sv;

  size_t off = 0;
  auto carve = [&](size_t bytes) { uint8_t* p = lds + off; off += (bytes + 15) & ~(size_t)15; return p; };
  uint32_t* xq = reinterpret_cast<uint32_t*>(carve((size_t)2 * L.max_nw * 64 * 4));  // [plane][word][lane]
  uint32_t* gl = reinterpret_cast<uint32_t*>(carve((size_t)(L.max_width + 2) * 4));   // g[0..width]
  double* dec = reinterpret_cast<double*>(carve((size_t)P * 64 * 8));                 // [pair][lane]
  double* Q = reinterpret_cast<double*>(carve((size_t)A * A * 64 * 8));               // [t][j][lane]
  double* Qp = reinterpret_cast<double*>(carve((size_t)A * 64 * 8));
  double* pr = reinterpret_cast<double*>(carve((size_t)A * 64 * 8));

  const int64_t n = (int64_t)blockIdx.x * 64 + lane;
  const int64_t nc = n < L.N ? n : L.N - 1;

  // ---- query window bits: funnel-shift the padded planes to the window start (w*M) ----
  {
    const int64_t s = (int64_t)w * L.M;
    const int64_t w0 = s >> 5;
    const int sh = (int)(s & 31);
    for (int pl = 0; pl < 2; ++pl) {
      const uint32_t* src = L.planes + (nc * 2 + pl) * L.nwp + w0;
      for (int i = 0; i < NW; ++i) {
        const uint32_t a = src[i], b = src[i + 1];  // planes are padded with 2 zero words
        uint32_t v = sh ? ((a >> sh) | (b << (32 - sh))) : a;
        if (i == NW - 1 && (width & 31)) v &= (1u << (width & 31)) - 1u;
        xq[((size_t)pl * L.max_nw + i) * 64 + lane] = v;
      }
    }
  }
  for (int i = lane; i <= width; i += 64) gl[i] = L.gtab[win.g_off + i];
  for (int p = 0; p < P; ++p) dec[p * 64 + lane] = 0.0;
  __syncthreads();

  const uint32_t tail_mask = (width & 31) ? ((1u << (width & 31)) - 1u) : 0xffffffffu;
  const double* dual = L.coef + win.coef_off;  // (A-1, n_sv)

  for (int c = 0; c < A; ++c) {
    for (int sv = win.cls_start[c]; sv < win.cls_start[c + 1]; ++sv) {
      const uint32_t* yb = L.svbits + win.sv_off + (size_t)sv * 2 * NW;  // wave-uniform -> scalar loads
      uint32_t K = 0, run = 0;
      for (int i = 0; i < NW; ++i) {
        const uint32_t yl = yb[i], yh = yb[NW + i];
        const uint32_t xl = xq[(size_t)i * 64 + lane], xh = xq[((size_t)L.max_nw + i) * 64 + lane];
        uint32_t e = ~((xl ^ yl) | (xh ^ yh));
        if (i == NW - 1) e &= tail_mask;
        if (e == 0xffffffffu) { run += 32; continue; }
        // trailing ones continue the carried run
        uint32_t t = (uint32_t)__builtin_ctz(~e);
        run += t;
        K += gl[run];
        run = 0;
        e >>= t;
        uint32_t rem = 32 - t;
        while (e) {
          const uint32_t z = (uint32_t)__builtin_ctz(e);
          e >>= z;
          rem -= z;
          const uint32_t o = (uint32_t)__builtin_ctz(~e);  // e has zeros above bit rem-1, so o <= rem
          if (o == rem) { run = o; break; }                // the run touches the end of the word: carry
          K += gl[o];
          e >>= o;
          rem -= o;
        }
      }
      K += gl[run];
      const double Kd = (double)K;
      // libsvm predict_values order: every pair (i<j) sums its class-i SVs (coef row j-1) then its class-j SVs (row i)
      for (int o = 0; o < A; ++o) {
        if (o == c) continue;
        const int row = (o > c) ? o - 1 : o;
        const int p = (o > c) ? pair_index(c, o, A) : pair_index(o, c, A);
        dec[p * 64 + lane] += dual[(size_t)row * n_sv + sv] * Kd;
      }
    }
  }

  // ---- Platt sigmoids (svm_predict_probability) ----
  const double* icpt = dual + (size_t)(A - 1) * n_sv;
  const double* pA = icpt + P;
  const double* pB = pA + P;
  const double min_prob = 1e-7;
  for (int p = 0; p < P; ++p) {
    const double d = dec[p * 64 + lane] + icpt[p];  // sklearn _intercept_ = -rho
    double v = sigmoid_predict(d, pA[p], pB[p]);
    v = fmin(fmax(v, min_prob), 1 - min_prob);
    dec[p * 64 + lane] = v;  // r[i][j], i<j; r[j][i] = 1 - v
  }
  auto r = [&](int i, int j) -> double {
    return (i < j) ? dec[pair_index(i, j, A) * 64 + lane] : 1.0 - dec[pair_index(j, i, A) * 64 + lane];
  };
  // ---- multiclass_probability (Wu, Lin, Weng 2004) ----
  const int k = A;
  const int max_iter = k > 100 ? k : 100;
  const double eps = 0.005 / k;
#define QQ(t, j) Q[((size_t)(t) * k + (j)) * 64 + lane]
#define QP(t) Qp[(size_t)(t) * 64 + lane]
#define PP(t) pr[(size_t)(t) * 64 + lane]
  for (int t = 0; t < k; ++t) {
    PP(t) = 1.0 / k;
    double qtt = 0.0;
    for (int j = 0; j < t; ++j) { qtt += r(j, t) * r(j, t); QQ(t, j) = QQ(j, t); }
    for (int j = t + 1; j < k; ++j) { qtt += r(j, t) * r(j, t); QQ(t, j) = -r(j, t) * r(t, j); }
    QQ(t, t) = qtt;
  }
  for (int iter = 0; iter < max_iter; ++iter) {
    double pQp = 0.0;
    for (int t = 0; t < k; ++t) {
      double q = 0.0;
      for (int j = 0; j < k; ++j) q += QQ(t, j) * PP(j);
      QP(t) = q;
      pQp += PP(t) * q;
    }
    double max_error = 0.0;
    for (int t = 0; t < k; ++t) { const double e = fabs(QP(t) - pQp); if (e > max_error) max_error = e; }
    if (max_error < eps) break;
    for (int t = 0; t < k; ++t) {
      const double diff = (-QP(t) + pQp) / QQ(t, t);
      PP(t) += diff;
      pQp = (pQp + diff * (diff * QQ(t, t) + 2 * QP(t))) / (1 + diff) / (1 + diff);
      for (int j = 0; j < k; ++j) { QP(j) = (QP(j) + diff * QQ(t, j)) / (1 + diff); PP(j) /= (1 + diff); }
    }
  }
  if (n < L.N) {
    const size_t o = ((size_t)n * L.W + w) * A;
    for (int a = 0; a < A; ++a) {
      const double v = PP(a);
      if (L.b64) L.b64[o + a] = v;
      if (L.b32) L.b32[o + a] = (float)v;
    }
  }
#undef QQ
#undef QP
#undef PP
}

}  // namespace

size_t gnx_covrsk_lds_bytes(int A, int max_nw, int max_width) {
  auto r16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const int P = A * (A - 1) / 2;
  return r16((size_t)2 * max_nw * 64 * 4) + r16((size_t)(max_width + 2) * 4) + r16((size_t)P * 64 * 8) +
         r16((size_t)A * A * 64 * 8) + 2 * r16((size_t)A * 64 * 8);
}

hipError_t gnx_launch_pack_bits(const int8_t* X, int64_t N, int64_t ldx, int64_t C, int64_t ctx, int64_t nwp,
                                uint32_t* planes, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  const int64_t total = N * nwp;
  hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, N, ldx, C, ctx, nwp, planes);
  return hipGetLastError();
}

hipError_t gnx_launch_covrsk(const CovRSKLaunch& L, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  const size_t lds = gnx_covrsk_lds_bytes(L.A, L.max_nw, L.max_width);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_covrsk_svc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_covrsk_svc, dim3((unsigned)((L.N + 63) / 64), (unsigned)L.W), dim3(64), lds, s, L);
  return hipGetLastError();
}

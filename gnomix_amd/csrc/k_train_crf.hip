// k_train_crf.hip — training the linear-chain CRF smoother on gfx950.
//
// Replaces CRF.fit (reference src/Smooth/crf.py:51-58: sklearn_crfsuite.CRF(algorithm="lbfgs", max_iterations=10000,
// all_possible_transitions=True, all_possible_states=True).fit, called by Smoother.train, src/Smooth/smooth.py:28-38, for
// CRF_Smoother, src/Smooth/models.py:27-32).  CRFsuite 0.12 (python-crfsuite behind sklearn-crfsuite 0.3.6, requirements.txt:9) is a
// third-party dependency that is absent here; its published model and objective are restated:
//   attributes "0" .. "A-1" with the window's base probabilities as values (crf.py:17-32), state features (attribute a, label l)
//   for ALL pairs, transition features (l', l) for ALL pairs, no other features                                (crf1d_feature.c)
//   p(y | x) = exp( sum_t sum_a state[a][y_t] x[t][a]  +  sum_{t>=1} trans[y_{t-1}][y_t] ) / Z(x)              (crf1d_context.c)
//   f(w) = - sum over sequences of log p(y | x)  +  c2 |w|^2,   c1 = 0, c2 = 1 (the lbfgs trainer's defaults)   (train_lbfgs.c)
// f is smooth and strictly convex (c2 > 0): it has ONE minimiser.  CRFsuite stops libLBFGS near it (epsilon 1e-5, or a relative
// improvement below delta 1e-5 over 10 iterations); this trainer runs L-BFGS further (|g| / max(1, |w|) < epsilon, default 1e-8),
// so the two fits differ by CRFsuite's stopping error.  Parity is stated like the logistic base's (tests/test_train_crf.py): the
// device's objective and gradient equal the oracle's restatement to 1e-10, the fitted weights equal the oracle's independent
// optimiser to 5e-6, and on a host with sklearn_crfsuite tests/golden/make_golden.py G13 records CRFsuite's own fit and its
// training set for the "objective never worse, weights within the stopping error" comparison (tests/test_pins_thirdparty.py;
// skipped here: parity unpinned).
//
// One evaluation = k_crf_eval (one wave per sequence: state potentials, scaled forward pass, then the backward pass with the
// node and edge marginals folded straight into the lane's gradient accumulators: lane = (attribute, label) / (from, to) pair)
// + k_crf_reduce1 / 2 (per-sequence partials summed in a fixed order: no atomics, two runs give the same bits).  All float64.
// The L-BFGS recursion over the 2 A^2 parameters runs on the host between evaluations (98 numbers at A = 7).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gnx_internal.h"

#define HIPCHK(ctx, expr)                                                                          \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return gnx_fail((ctx), GNX_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
  } while (0)

namespace {

constexpr int SEQ_PER_BLOCK = 4;  // one wave each

// the chains are per wave (one sequence each): LDS traffic between the lanes of ONE wave only needs program order (DS operations of
// a wave execute in order) — a wavefront-scope fence for the compiler instead of a block barrier that would tie four chains together
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return __shfl(v, 0, 64);
}

// KP = pairs per lane = ceil(A * A / 64)
template <int KP>
__global__ __launch_bounds__(SEQ_PER_BLOCK * 64) void k_crf_eval(const double* __restrict__ X, const int32_t* __restrict__ y, int64_t N, int W, int A,
                                                                 const double* __restrict__ wv /* state (A, A) then trans (A, A) */,
                                                                 double* __restrict__ scratch /* per sequence: psi, alpha (W, A) each, scale (W) */,
                                                                 double* __restrict__ fpart, double* __restrict__ gpart) {
  extern __shared__ double sm[];
  const int AA = A * A;
  double* st = sm;                 // state weights [a][l]
  double* et = sm + AA;            // exp(trans) [i][j]
  double* tr = sm + 2 * AA;        // trans
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* cur = sm + 3 * AA + wave * 3 * A;  // per wave: a vector of A (alpha_{t-1} / beta_t), the node marginals, psi_t * beta_t
  double* marg = cur + A;
  double* pb = marg + A;
  for (int e = threadIdx.x; e < AA; e += blockDim.x) {
    st[e] = wv[e];
    tr[e] = wv[AA + e];
    et[e] = exp(wv[AA + e]);
  }
  __syncthreads();
  const int64_t n = (int64_t)blockIdx.x * SEQ_PER_BLOCK + wave;
  const bool live = n < N;
  const int64_t nn = live ? n : N - 1;
  const double* x = X + (size_t)nn * W * A;
  const int32_t* yy = y + (size_t)nn * W;
  double* psi = scratch + (size_t)nn * (2 * (size_t)W * A + W);
  double* alpha = psi + (size_t)W * A;
  double* scale = alpha + (size_t)W * A;

  // ---- state potentials and the observed path's score (lanes over (t, l) / over t) ----
  double score = 0.0;
  if (live) {
    for (int e = lane; e < W * A; e += 64) {
      const int t = e / A, l = e - t * A;
      double s = 0.0;
      for (int a = 0; a < A; ++a) s += st[a * A + l] * x[t * A + a];
      psi[e] = exp(s);
    }
    for (int t = lane; t < W; t += 64) {
      const int l = yy[t];
      double s = 0.0;
      for (int a = 0; a < A; ++a) s += st[a * A + l] * x[t * A + a];
      if (t > 0) s += tr[yy[t - 1] * A + l];
      score += s;
    }
  }
  score = wave_sum(score);
  __syncthreads();  // psi is read by other lanes below (same wave; the block barrier also orders the global stores)

  // ---- scaled forward pass: lane l < A ----
  // (every per-step operand that lives in global memory is fetched one step ahead: the steps are a chain of dependent barriers,
  //  a load issued inside a step would add its whole latency to it)
  double logz = 0.0;
  const bool vec = live && lane < A;
  double psi_next = vec ? psi[lane] : 0.0;
  for (int t = 0; t < W; ++t) {
    double v = 0.0;
    const double psi_t = psi_next;
    if (vec && t + 1 < W) psi_next = psi[(t + 1) * A + lane];
    if (vec) {
      if (t == 0) v = psi_t;
      else {
        double acc = 0.0;
        for (int i = 0; i < A; ++i) acc += cur[i] * et[i * A + lane];
        v = acc * psi_t;
      }
    }
    const double sum = wave_sum(v);
    const double sc = (sum != 0.0) ? 1.0 / sum : 1.0;
    wave_sync();  // everybody has read cur
    if (live && lane < A) {
      cur[lane] = v * sc;
      alpha[t * A + lane] = v * sc;
    }
    if (live && lane == 0) scale[t] = sc;
    logz -= log(sc);
    wave_sync();
  }

  __threadfence_block();
  __syncthreads();  // alpha and scale (global scratch) were stored by other lanes: complete before anybody reads them back

  // ---- backward pass with the gradient: pair p = lane + 64 k = (i, j) ----
  double gs[KP], gt[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) gs[k] = gt[k] = 0.0;
  // operands of step t, fetched during step t + 1: alpha_t, alpha_{t-1}, psi_t (lane l < A), scale_t, the labels, the pair's x
  double al_t = vec ? alpha[(W - 1) * A + lane] : 0.0, al_p = (vec && W > 1) ? alpha[(W - 2) * A + lane] : 0.0;
  double ps_t = vec ? psi[(W - 1) * A + lane] : 0.0, sc_t = live ? scale[W - 1] : 1.0;
  int yt = live ? yy[W - 1] : 0, yp = (live && W > 1) ? yy[W - 2] : -1;
  double xk[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const int p = lane + 64 * k;
    xk[k] = (live && p < AA) ? x[(W - 1) * A + p / A] : 0.0;
  }
  for (int t = W - 1; t >= 0; --t) {
    double n_alp = 0.0, n_ps = 0.0, n_sc = 1.0, n_xk[KP];
    int n_yp = -1;
    if (vec && t > 1) n_alp = alpha[(t - 2) * A + lane];
    if (vec && t > 0) n_ps = psi[(t - 1) * A + lane];
    if (live && t > 0) n_sc = scale[t - 1];
    if (live && t > 1) n_yp = yy[t - 2];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int p = lane + 64 * k;
      n_xk[k] = (live && t > 0 && p < AA) ? x[(t - 1) * A + p / A] : 0.0;
    }
    // beta_t (into cur), from beta_{t+1} * psi_{t+1} (pb)
    double bt = 0.0;
    if (vec) {
      if (t == W - 1) bt = sc_t;
      else {
        double acc = 0.0;
        for (int j = 0; j < A; ++j) acc += et[lane * A + j] * pb[j];
        bt = acc * sc_t;
      }
    }
    wave_sync();  // pb (of t + 1) has been read
    if (vec) {
      cur[lane] = al_p;  // alpha_{t-1}: the edge marginals' left factor
      marg[lane] = al_t * bt / sc_t;
      pb[lane] = ps_t * bt;
    }
    wave_sync();
    if (live) {
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int p = lane + 64 * k;
        if (p < AA) {
          const int i = p / A, j = p - i * A;
          gs[k] += (marg[j] - (j == yt ? 1.0 : 0.0)) * xk[k];                                           // attribute i, label j
          if (t > 0) gt[k] += cur[i] * et[p] * pb[j] - ((i == yp && j == yt) ? 1.0 : 0.0);              // edge i -> j
        }
      }
    }
    al_t = al_p; al_p = n_alp; ps_t = n_ps; sc_t = n_sc;
    yt = yp; yp = n_yp;
#pragma unroll
    for (int k = 0; k < KP; ++k) xk[k] = n_xk[k];
  }
  if (live) {
    if (lane == 0) fpart[n] = logz - score;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int p = lane + 64 * k;
      if (p < AA) {
        gpart[(size_t)n * 2 * AA + p] = gs[k];
        gpart[(size_t)n * 2 * AA + AA + p] = gt[k];
      }
    }
  }
}

// f = sum_n fpart + c2 |w|^2, g = sum_n gpart + 2 c2 w in two fixed-order stages: chunks of RCH sequences (thread = parameter,
// rows read along the parameters), then the chunks
constexpr int RCH = 32;
__global__ void k_crf_reduce1(int64_t N, int P, const double* __restrict__ fpart, const double* __restrict__ gpart, double* __restrict__ part /* [chunk][P + 1] */) {
  const int64_t n0 = (int64_t)blockIdx.x * RCH, n1 = n0 + RCH < N ? n0 + RCH : N;
  for (int i = threadIdx.x; i <= P; i += blockDim.x) {
    double acc = 0.0;
    if (i < P)
      for (int64_t n = n0; n < n1; ++n) acc += gpart[(size_t)n * P + i];
    else
      for (int64_t n = n0; n < n1; ++n) acc += fpart[n];
    part[(size_t)blockIdx.x * (P + 1) + i] = acc;
  }
}
__global__ void k_crf_reduce2(int n_chunks, int P, const double* __restrict__ part, const double* __restrict__ wv, double c2, double* __restrict__ out /* g[P], f */) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > P) return;
  double acc = 0.0;
  for (int c = 0; c < n_chunks; ++c) acc += part[(size_t)c * (P + 1) + i];
  if (i < P) out[i] = acc + 2.0 * c2 * wv[i];
  else {
    double nrm = 0.0;
    for (int k = 0; k < P; ++k) nrm += wv[k] * wv[k];
    out[P] = acc + c2 * nrm;
  }
}

__global__ void k_crf_to_f64(const float* __restrict__ in, int64_t n, double* __restrict__ out) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) out[e] = (double)in[e];
}

double dot(const std::vector<double>& a, const std::vector<double>& b) {
  double s = 0.0;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}

}  // namespace

extern "C" int gnx_train_crf(gnx_ctx* ctx, const void* B, int32_t b_is_f64, const int32_t* y, int64_t N, int32_t W, int32_t A, const gnx_crf_params* P,
                             double* state, double* trans, gnx_crf_info* info) {
  if (!ctx) return GNX_EINVAL;
  if (!ctx->usable) return gnx_fail(ctx, GNX_EINVAL, "train_crf: context is not usable");
  if (!B || !y || !P || !state || !trans) return gnx_fail(ctx, GNX_EINVAL, "train_crf: NULL argument");
  if (N <= 0 || W <= 0 || A < 2 || A > 32) return gnx_fail(ctx, GNX_EINVAL, "train_crf: need N, W > 0 and 2 <= A <= 32");
  if (P->c1 != 0.0) return gnx_fail(ctx, GNX_EUNSUPPORTED, "train_crf: c1 != 0 (OWL-QN) is not built; the reference trains with CRFsuite's default c1 = 0");
  if (!(P->c2 > 0.0) || !(P->epsilon > 0.0) || P->max_iterations < 0 || P->memory < 1 || P->memory > 64)
    return gnx_fail(ctx, GNX_EINVAL, "train_crf: need c2 > 0, epsilon > 0, max_iterations >= 0, 1 <= memory <= 64");
  for (int64_t i = 0; i < N * W; ++i)
    if (y[i] < 0 || y[i] >= A) return gnx_fail(ctx, GNX_EINVAL, "train_crf: label outside [0, A)");
  GNX_BIND_DEVICE(ctx);
  hipStream_t s = ctx->stream;
  const int AA = A * A, NP = 2 * AA;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t nx = (size_t)N * W * A;
  const size_t o_raw = take(b_is_f64 ? 0 : nx * 4), o_x = take(nx * 8), o_y = take((size_t)N * W * 4), o_w = take((size_t)NP * 8);
  const size_t o_scr = take((size_t)N * (2 * (size_t)W * A + W) * 8), o_fp = take((size_t)N * 8), o_gp = take((size_t)N * NP * 8), o_out = take((size_t)(NP + 1) * 8);
  const int n_chunks = (int)((N + RCH - 1) / RCH);
  const size_t o_part = take((size_t)n_chunks * (NP + 1) * 8);
  int rc = gnx_ws_reserve(ctx, ctx->ws_misc, off);
  if (rc != GNX_OK) return rc;
  uint8_t* base = (uint8_t*)ctx->ws_misc.p;
  double* d_x = (double*)(base + o_x);
  int32_t* d_y = (int32_t*)(base + o_y);
  double *d_w = (double*)(base + o_w), *d_scr = (double*)(base + o_scr), *d_fp = (double*)(base + o_fp), *d_gp = (double*)(base + o_gp), *d_out = (double*)(base + o_out), *d_part = (double*)(base + o_part);
  if (b_is_f64) HIPCHK(ctx, hipMemcpyAsync(d_x, B, nx * 8, hipMemcpyHostToDevice, s));
  else {
    HIPCHK(ctx, hipMemcpyAsync(base + o_raw, B, nx * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_crf_to_f64, dim3(1024), dim3(256), 0, s, (const float*)(base + o_raw), (int64_t)nx, d_x);
  }
  HIPCHK(ctx, hipMemcpyAsync(d_y, y, (size_t)N * W * 4, hipMemcpyHostToDevice, s));
  const size_t lds = (size_t)(3 * AA + SEQ_PER_BLOCK * 3 * A) * sizeof(double);
  const int grid = (int)((N + SEQ_PER_BLOCK - 1) / SEQ_PER_BLOCK);
  std::vector<double> out((size_t)NP + 1);
  int n_eval = 0;
  // f(w), g(w)
  auto eval = [&](const std::vector<double>& w, double& f, std::vector<double>& g) -> int {
    ++n_eval;
    HIPCHK(ctx, hipMemcpyAsync(d_w, w.data(), (size_t)NP * 8, hipMemcpyHostToDevice, s));
    if (AA <= 64) hipLaunchKernelGGL(k_crf_eval<1>, dim3(grid), dim3(SEQ_PER_BLOCK * 64), lds, s, (const double*)d_x, (const int32_t*)d_y, N, (int)W, (int)A, (const double*)d_w, d_scr, d_fp, d_gp);
    else if (AA <= 256) hipLaunchKernelGGL(k_crf_eval<4>, dim3(grid), dim3(SEQ_PER_BLOCK * 64), lds, s, (const double*)d_x, (const int32_t*)d_y, N, (int)W, (int)A, (const double*)d_w, d_scr, d_fp, d_gp);
    else hipLaunchKernelGGL(k_crf_eval<16>, dim3(grid), dim3(SEQ_PER_BLOCK * 64), lds, s, (const double*)d_x, (const int32_t*)d_y, N, (int)W, (int)A, (const double*)d_w, d_scr, d_fp, d_gp);
    hipLaunchKernelGGL(k_crf_reduce1, dim3(n_chunks), dim3(128), 0, s, N, NP, (const double*)d_fp, (const double*)d_gp, d_part);
    hipLaunchKernelGGL(k_crf_reduce2, dim3((NP + 1 + 127) / 128), dim3(128), 0, s, n_chunks, NP, (const double*)d_part, (const double*)d_w, P->c2, d_out);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out.data(), d_out, (size_t)(NP + 1) * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    std::copy(out.begin(), out.begin() + NP, g.begin());
    f = out[(size_t)NP];
    return GNX_OK;
  };

  // ---- L-BFGS (two-loop recursion, `memory` pairs) with a backtracking / expanding line search on the Armijo and curvature
  //      conditions; f is strictly convex, so every accepted step has s.y > 0 ----
  std::vector<double> w((size_t)NP), g((size_t)NP), wn((size_t)NP), gn((size_t)NP), d((size_t)NP);
  std::copy(state, state + AA, w.begin());
  std::copy(trans, trans + AA, w.begin() + AA);
  double f = 0.0;
  if ((rc = eval(w, f, g)) != GNX_OK) return rc;
  const int m = P->memory;
  std::vector<std::vector<double>> S_, Y_;
  std::vector<double> rho;
  int it = 0;
  auto gnorm_rel = [&]() { return std::sqrt(dot(g, g)) / std::max(1.0, std::sqrt(dot(w, w))); };
  while (it < P->max_iterations && gnorm_rel() >= P->epsilon) {
    // direction
    std::vector<double> q = g;
    const int k = (int)S_.size();
    std::vector<double> al((size_t)k);
    for (int i = k - 1; i >= 0; --i) {
      al[(size_t)i] = rho[(size_t)i] * dot(S_[(size_t)i], q);
      for (int j = 0; j < NP; ++j) q[(size_t)j] -= al[(size_t)i] * Y_[(size_t)i][(size_t)j];
    }
    double h0 = k ? dot(S_.back(), Y_.back()) / dot(Y_.back(), Y_.back()) : 1.0 / std::max(1e-300, std::sqrt(dot(g, g)));
    for (int j = 0; j < NP; ++j) q[(size_t)j] *= h0;
    for (int i = 0; i < k; ++i) {
      const double be = rho[(size_t)i] * dot(Y_[(size_t)i], q);
      for (int j = 0; j < NP; ++j) q[(size_t)j] += (al[(size_t)i] - be) * S_[(size_t)i][(size_t)j];
    }
    for (int j = 0; j < NP; ++j) d[(size_t)j] = -q[(size_t)j];
    double dg = dot(d, g);
    if (!(dg < 0.0)) {  // not a descent direction (rounding): restart from steepest descent
      S_.clear(); Y_.clear(); rho.clear();
      for (int j = 0; j < NP; ++j) d[(size_t)j] = -g[(size_t)j];
      dg = -dot(g, g);
    }
    // line search: Armijo (1e-4) + weak curvature (0.9); step 1 first, halve on failure of the former, double on the latter
    double step = 1.0, lo = 0.0, hi = 0.0, fn = f;
    bool ok = false;
    for (int ls = 0; ls < 60; ++ls) {
      for (int j = 0; j < NP; ++j) wn[(size_t)j] = w[(size_t)j] + step * d[(size_t)j];
      if ((rc = eval(wn, fn, gn)) != GNX_OK) return rc;
      // near the minimiser the decrease of f drowns in its rounding error (1e-16 |f| against |g|^2): there the Armijo test is
      // replaced by its derivative form (Hager & Zhang's approximate Wolfe conditions), which only needs the gradient
      const double dgn = dot(gn, d);
      const bool armijo = fn <= f + 1e-4 * step * dg || (fn <= f + 1e-13 * std::fabs(f) && dgn <= (2e-4 - 1.0) * dg);
      if (!armijo) {
        hi = step;
        step = 0.5 * (lo + hi);
      } else if (dgn < 0.9 * dg) {
        lo = step;
        step = hi > 0.0 ? 0.5 * (lo + hi) : 2.0 * step;
      } else {
        ok = true;
        break;
      }
    }
    if (!ok && !(dot(gn, gn) < dot(g, g))) break;  // no progress possible at this precision
    std::vector<double> sv((size_t)NP), yv((size_t)NP);
    for (int j = 0; j < NP; ++j) { sv[(size_t)j] = wn[(size_t)j] - w[(size_t)j]; yv[(size_t)j] = gn[(size_t)j] - g[(size_t)j]; }
    const double sy = dot(sv, yv);
    if (sy > 1e-300) {
      if ((int)S_.size() == m) { S_.erase(S_.begin()); Y_.erase(Y_.begin()); rho.erase(rho.begin()); }
      S_.push_back(sv); Y_.push_back(yv); rho.push_back(1.0 / sy);
    }
    w.swap(wn); g.swap(gn); f = fn;
    ++it;
  }
  std::copy(w.begin(), w.begin() + AA, state);
  std::copy(w.begin() + AA, w.end(), trans);
  if (info) {
    info->iterations = it;
    info->evaluations = n_eval;
    info->objective = f;
    info->grad_norm = std::sqrt(dot(g, g));
    info->converged = gnorm_rel() < P->epsilon ? 1 : 0;
    info->reserved = 0;
  }
  return GNX_OK;
}

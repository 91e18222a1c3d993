// k_smooth_crf.hip — linear-chain CRF smoother (marginals) on gfx950.
//
// Replaces CRF_Smoother.predict_proba (reference src/Smooth/models.py:27-32, src/Smooth/crf.py:17-67 ->
// sklearn_crfsuite.CRF.predict_marginals; CRFsuite semantics as restated in the CPU oracle: attributes
// "0".."A-1" with values B[n,t,a], dense state weights theta[a][y] and transitions tau[y'][y]).
//
// One group of A lanes per haplotype (lane = label y), floor(64/A) haplotypes per wave; the chain is walked
// forward (scaled alpha, parked in the output buffer) and backward (beta on the fly) in float64 with the
// same left-to-right association as the oracle; cross-label terms travel by ds_bpermute shuffles.
// The W-step recurrence is inherently sequential per haplotype: parallelism = haplotypes.
#include "gnx_internal.h"

namespace {

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

__global__ __launch_bounds__(256) void k_smooth_crf(SmoothCRFLaunch L) {
  extern __shared__ __attribute__((aligned(16))) double lds_d[];
  const int A = L.A, W = L.W;
  double* th = lds_d;           // theta[a][y]
  double* et = lds_d + A * A;   // exp(tau)[y'][y]
  for (int i = threadIdx.x; i < A * A; i += blockDim.x) { th[i] = L.state[i]; et[i] = L.etrans[i]; }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = 64 / A;                 // haplotypes per wave
  const int g = lane / A, y = lane - g * A;
  const int gbase = g * A;
  const int64_t n = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * G + g;
  const bool active = (g < G) && (n < L.N);
  const int64_t nn = active ? n : 0;
  const size_t row0 = (size_t)nn * W * A;
  double* alpha = L.alpha + row0;       // (W, A) parked scaled alphas, overwritten by the marginals
  double* scale = L.scale + (size_t)nn * W;

  auto loadB = [&](int t) -> double {
    const size_t idx = row0 + (size_t)t * A + y;
    return L.b_is_f64 ? reinterpret_cast<const double*>(L.B)[idx] : (double)reinterpret_cast<const float*>(L.B)[idx];
  };
  auto psi_of = [&](double myB) -> double {  // exp(sum_a theta[a][y] * B[t][a])
    double s = 0.0;
    for (int a = 0; a < A; ++a) s += th[a * A + y] * shfl_d(myB, gbase + a);
    return exp(s);
  };

  // ---- forward ----
  double a_prev = 0.0;
  for (int t = 0; t < W; ++t) {
    const double psi = psi_of(active ? loadB(t) : 0.0);
    double v;
    if (t == 0) v = psi;
    else {
      double acc = 0.0;
      for (int yp = 0; yp < A; ++yp) acc += shfl_d(a_prev, gbase + yp) * et[yp * A + y];
      v = acc * psi;
    }
    double sum = 0.0;
    for (int yy = 0; yy < A; ++yy) sum += shfl_d(v, gbase + yy);
    const double sc = (sum != 0.0) ? 1.0 / sum : 1.0;
    a_prev = v * sc;
    if (active) {
      alpha[(size_t)t * A + y] = a_prev;
      if (y == 0) scale[t] = sc;
    }
  }

  // ---- backward + marginals ----
  // scale[t] was written by lane y==0 of this group: that lane reads its own store back and broadcasts it
  auto scale_at = [&](int t) -> double { return shfl_d((active && y == 0) ? scale[t] : 1.0, gbase); };
  double sc_t = scale_at(W - 1);
  double beta = sc_t;
  double psi_next = 0.0;
  for (int t = W - 1; t >= 0; --t) {
    if (t < W - 1) {
      // beta_t(y') = c_t * sum_y exp(tau)[y'][y] * psi_{t+1}(y) * beta_{t+1}(y)     (this lane: y' = y)
      const double pb_psi = psi_next, pb_beta = beta;
      double acc = 0.0;
      for (int yy = 0; yy < A; ++yy) acc += et[y * A + yy] * shfl_d(pb_psi, gbase + yy) * shfl_d(pb_beta, gbase + yy);
      sc_t = scale_at(t);
      beta = acc * sc_t;
    }
    const double myB = active ? loadB(t) : 0.0;
    psi_next = psi_of(myB);  // psi_t, consumed by step t-1
    const double al = active ? alpha[(size_t)t * A + y] : 0.0;
    const double m = al * beta / sc_t;
    // argmax over the group, first max wins
    int best = 0;
    double bv = shfl_d(m, gbase);
    for (int yy = 1; yy < A; ++yy) {
      const double o = shfl_d(m, gbase + yy);
      if (o > bv) { bv = o; best = yy; }
    }
    if (active) {
      const size_t o = row0 + (size_t)t * A + y;
      if (L.proba64) L.proba64[o] = m;          // may alias alpha: alpha[t] was read above
      if (L.proba32) L.proba32[o] = (float)m;
      if (L.labels && y == 0) L.labels[(size_t)nn * W + t] = best;
    }
  }
}

}  // namespace

hipError_t gnx_launch_smooth_crf(const SmoothCRFLaunch& L, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  const int G = 64 / L.A;
  const int waves = 4;
  const int64_t per_block = (int64_t)G * waves;
  const unsigned grid = (unsigned)((L.N + per_block - 1) / per_block);
  const size_t lds = (size_t)2 * L.A * L.A * sizeof(double);
  hipLaunchKernelGGL(k_smooth_crf, dim3(grid), dim3(64 * waves), lds, s, L);
  return hipGetLastError();
}

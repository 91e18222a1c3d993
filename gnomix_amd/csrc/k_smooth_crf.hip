// k_smooth_crf.hip — linear-chain CRF smoother (marginals) on gfx950.
//
// Replaces CRF_Smoother.predict_proba (reference src/Smooth/models.py:27-32, src/Smooth/crf.py:17-67 ->
// sklearn_crfsuite.CRF.predict_marginals; CRFsuite semantics as restated in the CPU oracle: attributes
// "0".."A-1" with values B[n,t,a], dense state weights theta[a][y] and transitions tau[y'][y]).
//
// One group of A lanes per haplotype (lane = label y), floor(64/A) haplotypes per wave; the chain is walked
// forward (scaled alpha, parked in the output buffer) and backward (beta on the fly) in float64 with the
// same left-to-right association as the oracle; cross-label terms travel by ds_bpermute shuffles.
// The W-step recurrence is inherently sequential per haplotype: parallelism = haplotypes.  Memory latency is kept off that
// chain (the base probabilities and, on the way back, the parked alphas and scales of the next PFD steps are requested
// while the current PFD steps are computed); what remains per step is float64 VALU work: two exp, a division and
// ~4A shuffles (measured: 17 ms per 25 000 haplotypes x 1431 windows x 12 classes with or without the prefetch).
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

  constexpr int PFD = 4;  // steps per software-pipeline stage
  auto clampt = [&](int t) { return t < 0 ? 0 : (t > W - 1 ? W - 1 : t); };

  // ---- forward ----
  double a_prev = 0.0;
  double bn[PFD];
#pragma unroll
  for (int k = 0; k < PFD; ++k) bn[k] = loadB(clampt(k));
  for (int t0 = 0; t0 < W; t0 += PFD) {
    double bc[PFD];
#pragma unroll
    for (int k = 0; k < PFD; ++k) bc[k] = bn[k];
#pragma unroll
    for (int k = 0; k < PFD; ++k) bn[k] = loadB(clampt(t0 + PFD + k));  // unconditional, clamped: in flight during this stage
#pragma unroll
    for (int k = 0; k < PFD; ++k) {
      const int t = t0 + k;
      if (t < W) {
        const double psi = psi_of(active ? bc[k] : 0.0);
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
    }
  }
  __threadfence_block();  // the backward pass reads this wave's own alpha / scale stores back through global memory

  // ---- backward + marginals ----
  // scale[t] was written by lane y==0 of this group: that lane reads its own store back and broadcasts it
  auto loadS = [&](int t) -> double { return (active && y == 0) ? scale[t] : 1.0; };
  auto loadA = [&](int t) -> double { return active ? alpha[(size_t)t * A + y] : 0.0; };
  double sc_t = 1.0, beta = 0.0, psi_next = 0.0;
  double an[PFD], sn[PFD];
#pragma unroll
  for (int k = 0; k < PFD; ++k) {
    const int t = clampt(W - 1 - k);
    bn[k] = loadB(t); an[k] = loadA(t); sn[k] = loadS(t);
  }
  for (int t0 = W - 1; t0 >= 0; t0 -= PFD) {
    double bc[PFD], ac[PFD], scur[PFD];
#pragma unroll
    for (int k = 0; k < PFD; ++k) { bc[k] = bn[k]; ac[k] = an[k]; scur[k] = sn[k]; }
#pragma unroll
    for (int k = 0; k < PFD; ++k) {
      const int t = clampt(t0 - PFD - k);
      bn[k] = loadB(t); an[k] = loadA(t); sn[k] = loadS(t);
    }
#pragma unroll
    for (int k = 0; k < PFD; ++k) {
      const int t = t0 - k;
      if (t >= 0) {
        const double sct = shfl_d(scur[k], gbase);
        if (t < W - 1) {
          // beta_t(y') = c_t * sum_y exp(tau)[y'][y] * psi_{t+1}(y) * beta_{t+1}(y)     (this lane: y' = y)
          const double pb_psi = psi_next, pb_beta = beta;
          double acc = 0.0;
          for (int yy = 0; yy < A; ++yy) acc += et[y * A + yy] * shfl_d(pb_psi, gbase + yy) * shfl_d(pb_beta, gbase + yy);
          beta = acc * sct;
        } else {
          beta = sct;
        }
        sc_t = sct;
        psi_next = psi_of(active ? bc[k] : 0.0);  // psi_t, consumed by step t-1
        const double m = ac[k] * beta / sc_t;
        // argmax over the group, first max wins
        int best = 0;
        double bv = shfl_d(m, gbase);
        for (int yy = 1; yy < A; ++yy) {
          const double o = shfl_d(m, gbase + yy);
          if (o > bv) { bv = o; best = yy; }
        }
        if (active) {
          const size_t o = row0 + (size_t)t * A + y;
          if (L.proba64) L.proba64[o] = m;          // may alias alpha: alpha[t] was read PFD steps ago at the latest
          if (L.proba32) L.proba32[o] = (float)m;
          if (L.labels && y == 0) L.labels[(size_t)nn * W + t] = best;
        }
      }
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

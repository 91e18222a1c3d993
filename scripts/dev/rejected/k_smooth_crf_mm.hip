// k_smooth_crf_mm.hip — the CRF smoother's forward-backward recurrence with its cross-label products on the float64 matrix cores
// (gfx950: v_mfma_f64_16x16x4_f64), 16 haplotypes per wave, up to 16 labels.
//
// Replaces CRF_Smoother.predict_proba (reference src/Smooth/models.py:27-32, src/Smooth/crf.py:17-67 ->
// sklearn_crfsuite.CRF.predict_marginals), same arithmetic as k_smooth_crf.hip (scaled forward-backward with the scale taken every
// (norm_mask + 1)-th window, alphas parked per segment and recomputed on the way back, marginals alpha beta c).
//
// Why.  k_smooth_crf_ck (one haplotype per 16-lane DPP row) spends its time on float64 DPP multiply-adds: the only DPP form float64
// has (row_newbcast) issues far below the float64 rate, on rows that are a quarter padding at 12 labels, and a wave carries four
// haplotypes.  A cross-label product of 16 haplotypes IS a 16 x 16 x 16 matrix product, and written TRANSPOSED it needs no data
// movement between windows at all:
//     alpha_new^T [label m][haplotype n] = sum_k  E^T [m][k] * alpha^T [k][n]
//   * the constant matrix is the MFMA's A operand (lane i supplies A[m = i % 16][k = i / 16] of the 4-wide K slice of chunk c),
//   * alpha^T is the B operand (lane i supplies B[k = i / 16][n = i % 16]): one double per lane and chunk,
//   * the result lands as D[m = 4 (i / 16) + r][n = i % 16], r = 0..3: FOUR doubles per lane.
//   So lane i = (haplotype h = i % 16, label group g = i / 16) holds the labels of storage slots 4 g + r in its result registers, and
//   chunk c of the NEXT product takes exactly register c of every lane as its B operand (K index k = lane group): slot 4 k + c.  The
//   layout is a fixed point of the product — alpha never leaves its registers, there is no shuffle, DPP or LDS step on the chain.
//   * Labels fill the slots with r < RL = ceil(A / 4) first (label = RL g + r): the product takes RL chunks (3 at 12 labels).
//   * psi_t = exp(theta' B_t), beta_t = c_t E (psi beta): the same product with theta^T / E as the A operand; B_t arrives in the
//     layout the product wants (a lane loads the RL consecutive labels of its group); a row sum (the forward scale) is the product
//     with a matrix of ones, which leaves the haplotype's sum in every register of its four lanes.
//   * psi is computed in both sweeps from B (no psi traffic): 2 B + marginals + parked alphas per launch.
// 16 haplotypes per wave mean 1 563 waves for config 5a's 25 000 haplotypes — one or two per SIMD; what makes that enough here is
// that a window of 16 haplotypes is ~20 MFMAs and ~150 other instructions (the four-lanes-per-haplotype VALU kernel of
// scripts/dev/rejected needed 830).  Summation order differs from the oracle's left-to-right (the MFMA adds its four products of a
// chunk in hardware order).
//
// PARKED (round 4, `make EXPERIMENTS=1`, GNX_CRF_IMPL=mm): measured no faster, and not debugged to parity (the first long-chain test
// fails).  Config 5a 4.07-4.10 ms against k_smooth_crf_ck's 3.9; chr22 0.57 against 0.38 ms; a lone wave needs 4 800 cycles per
// window.  scripts/dev/f64_rate_probe.hip says why: v_mfma_f64_16x16x4_f64 occupies the matrix pipe for 64 clock ticks dependent or
// not (no overlap inside a wave), 40-44 per instruction with four waves — the vector float64 rate, as the guide says — while
// v_fmac_f64_dpp issues at the FULL float64 rate (the same figures as v_fma_f64: 2.9 ticks, 8.8 dependent).  Three MFMAs per 16
// haplotypes are what twelve DPP multiply-adds per 4 haplotypes are: the same cycles per haplotype, with a quarter of the waves.
#include "gnx_internal.h"
#include "gnx_exp.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ v4d mfma_f64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
#else   // host pass: never called
__device__ inline v4d mfma_f64(double, double, v4d c) { return c; }
#endif

template <int RL, bool BF64>
__global__ __launch_bounds__(64) void k_smooth_crf_mm(SmoothCRFLaunch L) {
  constexpr int SEG = 4;
  __shared__ double la[SEG][RL][64];     // recomputed alpha_t of the segment (a lane reads back its own)
  __shared__ double lpsi[SEG][RL][64];   // psi_t of the segment
  __shared__ double2 lsc[SEG][64];       // (1/c_t, c_t)
  const int A = L.A, W = L.W;
  const int lane = threadIdx.x, h = lane & 15, g = lane >> 4;
  const int64_t n = (int64_t)blockIdx.x * 16 + h;
  const bool live = n < L.N;
  const int64_t nn = live ? n : 0;
  const size_t row0 = (size_t)nn * W * A;
  const int NSEG = (W + SEG - 1) / SEG;
  double* ck = L.alpha + (size_t)nn * NSEG * A;  // [segment][label]: alpha at the segment's last window
  const int y0 = g * RL;                          // the lane's labels: y0 + r, r < RL
  const int norm_mask = L.norm_mask;

  // the constant matrices as A operands: lane i supplies, for chunk c, the entry [output slot m = i % 16][K index k = i / 16];
  // slot m = 4 g' + r' is label RL g' + r' (r' < RL), chunk c's K index k is slot 4 k + c = label RL k + c.  Padding = 0.
  double EfA[RL], EbA[RL], ThA[RL], OnA[RL];
  {
    const int m = lane & 15, k = lane >> 4;
    const int gm = m >> 2, rm = m & 3;
    const int yout = RL * gm + rm;
    const bool out_ok = rm < RL && yout < A;
#pragma unroll
    for (int c = 0; c < RL; ++c) {
      const int yin = RL * k + c;
      const bool ok = out_ok && yin < A;
      EfA[c] = ok ? L.etrans[yin * A + yout] : 0.0;   // forward: alpha_new(y) = sum_y' alpha(y') E[y'][y]
      EbA[c] = ok ? L.etrans[yout * A + yin] : 0.0;   // backward: beta(y') = sum_y E[y'][y] g(y)
      ThA[c] = ok ? L.state[yin * A + yout] : 0.0;    // psi: s(y) = sum_a theta[a][y] B(a)
      OnA[c] = 1.0;                                   // row sum (padding slots of the B operand are zero)
    }
  }
  bool valid[RL];
  int yl[RL];
#pragma unroll
  for (int r = 0; r < RL; ++r) {
    valid[r] = y0 + r < A;
    yl[r] = valid[r] ? y0 + r : 0;
  }
  auto prod = [&](const double (&x)[RL], const double (&M)[RL]) -> v4d {
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < RL; ++c) acc = mfma_f64(M[c], x[c], acc);
    return acc;
  };

  auto clampt = [&](int t) { return t < 0 ? 0 : (t > W - 1 ? W - 1 : t); };
  // unconditional loads (padding labels read label 0 and drop it): a load under a divergent branch serialises the prefetch
  auto loadB = [&](int t, double (&b)[RL]) {
#pragma unroll
    for (int r = 0; r < RL; ++r) {
      const size_t idx = row0 + (size_t)t * A + yl[r];
      double v;
      if constexpr (BF64) v = reinterpret_cast<const double*>(L.B)[idx];
      else v = (double)reinterpret_cast<const float*>(L.B)[idx];
      b[r] = valid[r] ? v : 0.0;
    }
  };
  auto psi1 = [&](double (&b)[RL]) {  // b (a window's base probabilities) -> psi in place
    const v4d s = prod(b, ThA);
#pragma unroll
    for (int r = 0; r < RL; ++r) b[r] = valid[r] ? gnx_exp_sc(s[r]) : 0.0;
  };
  // one step of the scaled forward recurrence on the lane's labels; sc = 1/c_t, sum = c_t (1 between two scaled windows)
  auto fwd_step = [&](double (&a)[RL], const double (&psi)[RL], int t, double& sc, double& sum) {
    if (t != 0) {
      const v4d d = prod(a, EfA);
#pragma unroll
      for (int r = 0; r < RL; ++r) a[r] = d[r] * psi[r];
    } else {
#pragma unroll
      for (int r = 0; r < RL; ++r) a[r] = psi[r];
    }
    if ((t & norm_mask) != norm_mask && t != W - 1) {  // wave-uniform
      sc = 1.0;
      sum = 1.0;
      return;
    }
    const double s = prod(a, OnA)[0];   // the haplotype's sum, the same bits in every register of its four lanes
    const bool nz = s != 0.0;
    sum = nz ? s : 1.0;
    sc = __builtin_amdgcn_rcp(sum);
    sc = fma(fma(-sum, sc, 1.0), sc, sc);
    sc = fma(fma(-sum, sc, 1.0), sc, sc);
    sc = nz ? sc : 1.0;
#pragma unroll
    for (int r = 0; r < RL; ++r) a[r] *= sc;
  };

  // ---- forward: alpha parked once per segment; B two segments ahead ----
  double a[RL];
#pragma unroll
  for (int r = 0; r < RL; ++r) a[r] = 0.0;
  double bn1[SEG][RL], bn2[SEG][RL];
#pragma unroll
  for (int k = 0; k < SEG; ++k) { loadB(clampt(k), bn1[k]); loadB(clampt(SEG + k), bn2[k]); }
  for (int sg = 0; sg < NSEG; ++sg) {
    const int t0 = sg * SEG;
    double bc[SEG][RL];
#pragma unroll
    for (int k = 0; k < SEG; ++k)
#pragma unroll
      for (int r = 0; r < RL; ++r) { bc[k][r] = bn1[k][r]; bn1[k][r] = bn2[k][r]; }
#pragma unroll
    for (int k = 0; k < SEG; ++k) loadB(clampt(t0 + 2 * SEG + k), bn2[k]);
#pragma unroll
    for (int k = 0; k < SEG; ++k) psi1(bc[k]);   // off the chain
#pragma unroll
    for (int k = 0; k < SEG; ++k) {
      const int t = t0 + k;
      if (t < W) {
        double sc, sum;
        fwd_step(a, bc[k], t, sc, sum);
      }
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < RL; ++r)
        if (valid[r]) ck[(size_t)sg * A + y0 + r] = a[r];
    }
  }
  __threadfence_block();

  // ---- backward: per segment recompute alpha into LDS, then beta and the marginals ----
  double beta[RL], psi_next[RL], an[RL];
#pragma unroll
  for (int r = 0; r < RL; ++r) { beta[r] = 0.0; psi_next[r] = 0.0; an[r] = 0.0; }
  auto loadCk = [&](int sg_prev, double (&v)[RL]) {  // alpha entering segment sg_prev + 1 (zeros before the first)
#pragma unroll
    for (int r = 0; r < RL; ++r) {
      const double x = ck[(size_t)(sg_prev > 0 ? sg_prev : 0) * A + yl[r]];
      v[r] = (valid[r] && sg_prev >= 0) ? x : 0.0;
    }
  };
#pragma unroll
  for (int k = 0; k < SEG; ++k) { loadB(clampt((NSEG - 1) * SEG + k), bn1[k]); loadB(clampt((NSEG - 2) * SEG + k), bn2[k]); }
  loadCk(NSEG - 2, an);
  for (int sg = NSEG - 1; sg >= 0; --sg) {
    const int t0 = sg * SEG;
    double bc[SEG][RL], a_in[RL];
#pragma unroll
    for (int k = 0; k < SEG; ++k)
#pragma unroll
      for (int r = 0; r < RL; ++r) { bc[k][r] = bn1[k][r]; bn1[k][r] = bn2[k][r]; }
#pragma unroll
    for (int r = 0; r < RL; ++r) a_in[r] = an[r];
#pragma unroll
    for (int k = 0; k < SEG; ++k) loadB(clampt((sg - 2) * SEG + k), bn2[k]);
    loadCk(sg - 2, an);
#pragma unroll
    for (int k = 0; k < SEG; ++k) psi1(bc[k]);
#pragma unroll
    for (int k = 0; k < SEG; ++k) {
      const int t = t0 + k;
      if (t < W) {
        double sc, sum;
        fwd_step(a_in, bc[k], t, sc, sum);
#pragma unroll
        for (int r = 0; r < RL; ++r) { la[k][r][lane] = a_in[r]; lpsi[k][r][lane] = bc[k][r]; }
        lsc[k][lane] = make_double2(sc, sum);
      }
    }
#pragma unroll
    for (int k = SEG - 1; k >= 0; --k) {
      const int t = t0 + k;
      if (t < W) {
        const double2 scur = lsc[k][lane];
        if (t < W - 1) {
          double gv[RL];
#pragma unroll
          for (int r = 0; r < RL; ++r) gv[r] = psi_next[r] * beta[r];
          const v4d d = prod(gv, EbA);
#pragma unroll
          for (int r = 0; r < RL; ++r) beta[r] = d[r] * scur.x;
        } else {
#pragma unroll
          for (int r = 0; r < RL; ++r) beta[r] = valid[r] ? scur.x : 0.0;
        }
        double m[RL];
#pragma unroll
        for (int r = 0; r < RL; ++r) {
          psi_next[r] = lpsi[k][r][lane];
          m[r] = la[k][r][lane] * beta[r] * scur.y;
        }
        if (live) {
#pragma unroll
          for (int r = 0; r < RL; ++r)
            if (valid[r]) {
              const size_t o = row0 + (size_t)t * A + y0 + r;
              if (L.proba64) L.proba64[o] = m[r];
              if (L.proba32) L.proba32[o] = (float)m[r];
            }
        }
        if (L.labels) {  // arg-max, first maximum wins: the lane's best, then across the haplotype's four lanes (ties -> the lower label)
          double bv = valid[0] ? m[0] : -1.0;
          int bi = y0;
#pragma unroll
          for (int r = 1; r < RL; ++r) {
            const bool up = valid[r] && m[r] > bv;
            bv = up ? m[r] : bv;
            bi = up ? y0 + r : bi;
          }
#pragma unroll
          for (int sh = 16; sh <= 32; sh <<= 1) {
            const double ov = __shfl_xor(bv, sh);
            const int oi = __shfl_xor(bi, sh);
            const bool up = ov > bv || (ov == bv && oi < bi);
            bv = up ? ov : bv;
            bi = up ? oi : bi;
          }
          if (live && g == 0) L.labels[(size_t)nn * W + t] = bi;
        }
      }
    }
  }
}

template <int RL>
void launch_mm(const SmoothCRFLaunch& L, hipStream_t s) {
  const dim3 grid((unsigned)((L.N + 15) / 16)), block(64);
  if (L.b_is_f64) hipLaunchKernelGGL((k_smooth_crf_mm<RL, true>), grid, block, 0, s, L);
  else hipLaunchKernelGGL((k_smooth_crf_mm<RL, false>), grid, block, 0, s, L);
}

}  // namespace

// up to 16 labels
hipError_t gnx_launch_smooth_crf_mm(const SmoothCRFLaunch& L, hipStream_t s) {
  if (L.A > 16) return hipErrorInvalidValue;
  if (L.A <= 4) launch_mm<1>(L, s);
  else if (L.A <= 8) launch_mm<2>(L, s);
  else if (L.A <= 12) launch_mm<3>(L, s);
  else launch_mm<4>(L, s);
  return hipGetLastError();
}

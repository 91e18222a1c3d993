// k_smooth_crf_quad.hip — the CRF smoother's forward-backward recurrence with FOUR LANES PER HAPLOTYPE (up to 12 labels), gfx950.
//
// Replaces CRF_Smoother.predict_proba (reference src/Smooth/models.py:27-32, src/Smooth/crf.py:17-67 ->
// sklearn_crfsuite.CRF.predict_marginals), same arithmetic as k_smooth_crf.hip (scaled forward-backward, marginals alpha beta c).
//
// Why another layout.  k_smooth_crf_ck puts one haplotype on a 16-lane DPP row, one label per lane: every cross-label sum is a chain
// of v_fmac_f64_dpp row_newbcast — the only DPP form float64 has — which were thought to issue at HALF the float64 rate (scripts/dev/f64_rate_probe.hip later measured the full rate), on rows that are
// a quarter padding at 12 labels.  The counters say the kernel is bound by exactly those instructions (VALU-busy, HBM at 3 TB/s),
// ~250 VALU cycles per haplotype and window.  Here a haplotype is a QUAD of lanes and a lane owns LPL = ceil(A/4) consecutive labels:
//   * a cross-label product is LPL x 4 LPL plain v_fma_f64 per lane (full rate, no padding lanes) against coefficients in registers,
//     plus a quad exchange done with 32-bit quad_perm DPP moves (full rate);
//   * forward (alpha_t(y) = psi_t(y) sum_y' alpha(y') E[y'][y]): the lane multiplies ITS alpha(y') into partial sums for every y and
//     a two-step reduce-scatter (xor 2, xor 1) leaves each lane with the sums of its own labels: 9 LPL moves+adds;
//   * backward (beta_t(y') = c_t sum_y E[y'][y] psi(y) beta(y)): the lane gathers psi beta of the other three lanes (6 LPL moves) and
//     multiplies with the SAME coefficient registers — E[own y'][all y], kept in the lane-relative order block (q ^ j) so that the
//     register a lane reads across the quad does not depend on the lane;
//   * psi_t = exp(theta' B_t) is computed in the kernel in both sweeps (scatter form like the forward product, theta from LDS, a
//     segment at a time, off the chain): B is read twice and psi never travels — 2 B + marginals + alpha checkpoints per launch.
// 16 haplotypes per wave, one wave per workgroup.  Alphas are parked every SEG windows and a segment is recomputed on the way back
// (as in k_smooth_crf_ck); the forward scale is taken every (norm_mask + 1)-th window (gnx_build_crf).  Summation order differs from
// the oracle's left-to-right: marginals within 1e-11 (tests/test_gpu_parity.py), labels identical.
//
// PARKED (round 4, `make EXPERIMENTS=1`, GNX_CRF_IMPL=quad).  All 76 CRF parity tests pass with it; measured on MI355X:
// chr1 / A = 12 / 25 000 haplotypes 4.64 ms against k_smooth_crf_ck's 4.42, chr22 / A = 7 / 10 000 haplotypes 0.60 against 0.39 ms.
// It issues ~52 VALU instructions per haplotype and window against the row kernel's ~62 — psi twice (the 36-multiply-add product,
// its reduce-scatter, three exp) is 60 % of them — but sixteen haplotypes per wave leave 1 563 waves for 1 024 SIMDs at the chr1
// batch (625 at chr22): one or two waves per SIMD cannot hide the float64 dependency stalls that three waves of the row kernel do,
// and half the SIMDs carry twice the work of the other half.  The layout would win at >= 50 000 resident haplotypes.
#include "gnx_internal.h"

namespace {

constexpr int QX1 = 0xB1;  // quad_perm [1,0,3,2]: lane reads lane ^ 1
constexpr int QX2 = 0x4E;  // quad_perm [2,3,0,1]: lane ^ 2
constexpr int QX3 = 0x1B;  // quad_perm [3,2,1,0]: lane ^ 3

template <int CTRL>
__device__ __forceinline__ double quad_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int quad_mov_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false); }

// w[j][i]: this lane's partial sums for label i of block (q ^ j)  ->  w[0][i] = the quad's sum for the lane's own labels
template <int LPL>
__device__ __forceinline__ void quad_reduce_scatter(double (&w)[4][LPL]) {
#pragma unroll
  for (int i = 0; i < LPL; ++i) {
    w[0][i] += quad_mov<QX2>(w[2][i]);
    w[1][i] += quad_mov<QX2>(w[3][i]);
  }
#pragma unroll
  for (int i = 0; i < LPL; ++i) w[0][i] += quad_mov<QX1>(w[1][i]);
}

template <int LPL, bool BF64>
__global__ __launch_bounds__(64) void k_smooth_crf_quad(SmoothCRFLaunch L) {
  constexpr int SEG = 4;
  __shared__ double lth[4][LPL][4][LPL];   // theta in lane-relative order, one table per quad position
  __shared__ double la[SEG][LPL][64];      // recomputed alpha_t of the segment (a lane reads back its own)
  __shared__ double lpsi[SEG][LPL][64];    // psi_t of the segment
  __shared__ double2 lsc[SEG][64];         // (1/c_t, c_t)
  const int A = L.A, W = L.W;
  const int lane = threadIdx.x, q = lane & 3;
  const int64_t n = (int64_t)blockIdx.x * 16 + (lane >> 2);
  const bool live = n < L.N;
  const int64_t nn = live ? n : 0;
  const size_t row0 = (size_t)nn * W * A;
  const int NSEG = (W + SEG - 1) / SEG;
  double* ck = L.alpha + (size_t)nn * NSEG * A;  // [segment][label]: alpha at the segment's last window
  const int y0 = q * LPL;
  const int norm_mask = L.norm_mask;

  // E[own label i'][label i of block q ^ j]; zero for padding labels
  double Er[LPL][4][LPL];
#pragma unroll
  for (int ip = 0; ip < LPL; ++ip)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < LPL; ++i) {
        const int yi = y0 + ip, yo = (q ^ j) * LPL + i;
        Er[ip][j][i] = (yi < A && yo < A) ? L.etrans[yi * A + yo] : 0.0;
      }
  if (lane < 4) {
    for (int ip = 0; ip < LPL; ++ip)
      for (int j = 0; j < 4; ++j)
        for (int i = 0; i < LPL; ++i) {
          const int yi = y0 + ip, yo = (q ^ j) * LPL + i;
          lth[q][ip][j][i] = (yi < A && yo < A) ? L.state[yi * A + yo] : 0.0;
        }
  }
  __syncthreads();
  bool valid[LPL];
  int yl[LPL];
#pragma unroll
  for (int i = 0; i < LPL; ++i) {
    valid[i] = y0 + i < A;
    yl[i] = valid[i] ? y0 + i : 0;
  }

  auto clampt = [&](int t) { return t < 0 ? 0 : (t > W - 1 ? W - 1 : t); };
  // unconditional loads (padding labels read label 0 and drop it): a load under a divergent branch serialises the prefetch
  auto loadB = [&](int t, double (&b)[LPL]) {
#pragma unroll
    for (int i = 0; i < LPL; ++i) {
      const size_t idx = row0 + (size_t)t * A + yl[i];
      double v;
      if constexpr (BF64) v = reinterpret_cast<const double*>(L.B)[idx];
      else v = (double)reinterpret_cast<const float*>(L.B)[idx];
      b[i] = valid[i] ? v : 0.0;
    }
  };
  // psi of one window, off the chain: b -> psi in place (theta from the LDS table: 4 distinct addresses per wave, broadcast)
  auto psi1 = [&](double (&b)[LPL]) {
    double w[4][LPL];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < LPL; ++i) {
        double acc = b[0] * lth[q][0][j][i];
#pragma unroll
        for (int ip = 1; ip < LPL; ++ip) acc = fma(b[ip], lth[q][ip][j][i], acc);
        w[j][i] = acc;
      }
    quad_reduce_scatter<LPL>(w);
#pragma unroll
    for (int i = 0; i < LPL; ++i) b[i] = valid[i] ? exp(w[0][i]) : 0.0;
  };
  // one step of the scaled forward recurrence on the lane's labels; sc = 1/c_t, sum = c_t (1 between two scaled windows)
  auto fwd_step = [&](double (&a)[LPL], const double (&psi)[LPL], int t, double& sc, double& sum) {
    if (t != 0) {
      double w[4][LPL];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < LPL; ++i) {
          double acc = a[0] * Er[0][j][i];
#pragma unroll
          for (int ip = 1; ip < LPL; ++ip) acc = fma(a[ip], Er[ip][j][i], acc);
          w[j][i] = acc;
        }
      quad_reduce_scatter<LPL>(w);
#pragma unroll
      for (int i = 0; i < LPL; ++i) a[i] = w[0][i] * psi[i];
    } else {
#pragma unroll
      for (int i = 0; i < LPL; ++i) a[i] = psi[i];
    }
    if ((t & norm_mask) != norm_mask && t != W - 1) {  // wave-uniform
      sc = 1.0;
      sum = 1.0;
      return;
    }
    double s = a[0];
#pragma unroll
    for (int i = 1; i < LPL; ++i) s += a[i];
    s += quad_mov<QX1>(s);
    s += quad_mov<QX2>(s);   // commutative at both levels: the four lanes hold the same bits
    const bool nz = s != 0.0;
    sum = nz ? s : 1.0;
    sc = __builtin_amdgcn_rcp(sum);
    sc = fma(fma(-sum, sc, 1.0), sc, sc);
    sc = fma(fma(-sum, sc, 1.0), sc, sc);
    sc = nz ? sc : 1.0;
#pragma unroll
    for (int i = 0; i < LPL; ++i) a[i] *= sc;
  };

  // ---- forward: alpha parked once per segment; B two segments ahead ----
  double a[LPL];
#pragma unroll
  for (int i = 0; i < LPL; ++i) a[i] = 0.0;
  double bn1[SEG][LPL], bn2[SEG][LPL];
#pragma unroll
  for (int k = 0; k < SEG; ++k) { loadB(clampt(k), bn1[k]); loadB(clampt(SEG + k), bn2[k]); }
  for (int sg = 0; sg < NSEG; ++sg) {
    const int t0 = sg * SEG;
    double bc[SEG][LPL];
#pragma unroll
    for (int k = 0; k < SEG; ++k)
#pragma unroll
      for (int i = 0; i < LPL; ++i) { bc[k][i] = bn1[k][i]; bn1[k][i] = bn2[k][i]; }
#pragma unroll
    for (int k = 0; k < SEG; ++k) loadB(clampt(t0 + 2 * SEG + k), bn2[k]);
#pragma unroll
    for (int k = 0; k < SEG; ++k) {
      const int t = t0 + k;
      if (t < W) {
        double sc, sum;
        psi1(bc[k]);
        fwd_step(a, bc[k], t, sc, sum);
      }
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < LPL; ++i)
        if (valid[i]) ck[(size_t)sg * A + y0 + i] = a[i];
    }
  }
  __threadfence_block();

  // ---- backward: per segment recompute alpha into LDS, then beta and the marginals ----
  double beta[LPL], psi_next[LPL], an[LPL];
#pragma unroll
  for (int i = 0; i < LPL; ++i) { beta[i] = 0.0; psi_next[i] = 0.0; an[i] = 0.0; }
  auto loadCk = [&](int sg_prev, double (&v)[LPL]) {  // alpha entering segment sg_prev + 1 (zeros before the first)
#pragma unroll
    for (int i = 0; i < LPL; ++i) {
      const double x = ck[(size_t)(sg_prev > 0 ? sg_prev : 0) * A + yl[i]];
      v[i] = (valid[i] && sg_prev >= 0) ? x : 0.0;
    }
  };
#pragma unroll
  for (int k = 0; k < SEG; ++k) { loadB(clampt((NSEG - 1) * SEG + k), bn1[k]); loadB(clampt((NSEG - 2) * SEG + k), bn2[k]); }
  loadCk(NSEG - 2, an);
  for (int sg = NSEG - 1; sg >= 0; --sg) {
    const int t0 = sg * SEG;
    double bc[SEG][LPL], a_in[LPL];
#pragma unroll
    for (int k = 0; k < SEG; ++k)
#pragma unroll
      for (int i = 0; i < LPL; ++i) { bc[k][i] = bn1[k][i]; bn1[k][i] = bn2[k][i]; }
#pragma unroll
    for (int i = 0; i < LPL; ++i) a_in[i] = an[i];
#pragma unroll
    for (int k = 0; k < SEG; ++k) loadB(clampt((sg - 2) * SEG + k), bn2[k]);
    loadCk(sg - 2, an);
#pragma unroll
    for (int k = 0; k < SEG; ++k) {
      const int t = t0 + k;
      if (t < W) {
        double sc, sum;
        psi1(bc[k]);
        fwd_step(a_in, bc[k], t, sc, sum);
#pragma unroll
        for (int i = 0; i < LPL; ++i) { la[k][i][lane] = a_in[i]; lpsi[k][i][lane] = bc[k][i]; }
        lsc[k][lane] = make_double2(sc, sum);
      }
    }
#pragma unroll
    for (int k = SEG - 1; k >= 0; --k) {
      const int t = t0 + k;
      if (t < W) {
        const double2 scur = lsc[k][lane];
        if (t < W - 1) {
          double g[4][LPL];
#pragma unroll
          for (int i = 0; i < LPL; ++i) g[0][i] = psi_next[i] * beta[i];
#pragma unroll
          for (int i = 0; i < LPL; ++i) {
            g[1][i] = quad_mov<QX1>(g[0][i]);
            g[2][i] = quad_mov<QX2>(g[0][i]);
            g[3][i] = quad_mov<QX3>(g[0][i]);
          }
#pragma unroll
          for (int ip = 0; ip < LPL; ++ip) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int i = 0; i < LPL; ++i) acc = fma(Er[ip][j][i], g[j][i], acc);
            beta[ip] = acc * scur.x;
          }
        } else {
#pragma unroll
          for (int i = 0; i < LPL; ++i) beta[i] = scur.x;
        }
        double m[LPL];
#pragma unroll
        for (int i = 0; i < LPL; ++i) {
          psi_next[i] = lpsi[k][i][lane];
          m[i] = la[k][i][lane] * beta[i] * scur.y;
        }
        if (live) {
#pragma unroll
          for (int i = 0; i < LPL; ++i)
            if (valid[i]) {
              const size_t o = row0 + (size_t)t * A + y0 + i;
              if (L.proba64) L.proba64[o] = m[i];
              if (L.proba32) L.proba32[o] = (float)m[i];
            }
        }
        if (L.labels) {  // arg-max, first maximum wins: the lane's best, then the quad's (ties -> the lower label)
          double bv = valid[0] ? m[0] : -1.0;
          int bi = y0;
#pragma unroll
          for (int i = 1; i < LPL; ++i) {
            const bool up = valid[i] && m[i] > bv;
            bv = up ? m[i] : bv;
            bi = up ? y0 + i : bi;
          }
          {
            const double ov = quad_mov<QX1>(bv);
            const int oi = quad_mov_i<QX1>(bi);
            const bool up = ov > bv || (ov == bv && oi < bi);
            bv = up ? ov : bv;
            bi = up ? oi : bi;
          }
          {
            const double ov = quad_mov<QX2>(bv);
            const int oi = quad_mov_i<QX2>(bi);
            const bool up = ov > bv || (ov == bv && oi < bi);
            bv = up ? ov : bv;
            bi = up ? oi : bi;
          }
          if (live && q == 0) L.labels[(size_t)nn * W + t] = bi;
        }
      }
    }
  }
}

}  // namespace

template <int LPL>
static void launch_quad(const SmoothCRFLaunch& L, hipStream_t s) {
  const dim3 grid((unsigned)((L.N + 15) / 16)), block(64);
  if (L.b_is_f64) hipLaunchKernelGGL((k_smooth_crf_quad<LPL, true>), grid, block, 0, s, L);
  else hipLaunchKernelGGL((k_smooth_crf_quad<LPL, false>), grid, block, 0, s, L);
}

// up to 12 labels; the caller keeps k_smooth_crf_ck for 13..16
hipError_t gnx_launch_smooth_crf_quad(const SmoothCRFLaunch& L, hipStream_t s) {
  if (L.A > 12) return hipErrorInvalidValue;
  if (L.A <= 4) launch_quad<1>(L, s);
  else if (L.A <= 8) launch_quad<2>(L, s);
  else launch_quad<3>(L, s);
  return hipGetLastError();
}

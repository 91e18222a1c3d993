// k_smooth_crf.hip — linear-chain CRF smoother (marginals) on gfx950.
//
// Replaces CRF_Smoother.predict_proba (reference src/Smooth/models.py:27-32, src/Smooth/crf.py:17-67 ->
// sklearn_crfsuite.CRF.predict_marginals; CRFsuite semantics as restated in the CPU oracle: attributes
// "0".."A-1" with values B[n,t,a], dense state weights theta[a][y] and transitions tau[y'][y]).
//
// The forward-backward recurrence is sequential over the W windows of a haplotype, so what decides the run time is the
// LENGTH OF ONE STEP on its critical path.  Round 1 spread a haplotype over A lanes (lane = label) and paid ~7A float64
// cross-lane shuffles (14A ds_bpermute) per step, plus an exp and a theta'B product twice per step (LDS pipe 56 % busy,
// 16.6 ms for 25 000 haplotypes x 1431 windows x 12 labels).  Round 2:
//   * k_crf_psi: psi_t(y) = exp(sum_a theta[a][y] B[t][a]) does not belong to the recurrence — one fully parallel pass
//     (thread = (haplotype, window)) computes it once for both directions;
//   * k_crf_scan: ONE LANE PER HAPLOTYPE holds the whole label vector in registers (alpha, beta: A doubles each): a step is
//     A^2 multiply-adds against exp(tau) rows broadcast from LDS, no cross-lane traffic at all; the lane's rows of psi and of
//     the parked alphas stream through an LDS ring filled by LDS-direct loads two chunks ahead, so HBM latency is off the chain.
//   * k_smooth_crf_row16 (the default up to 16 labels, further down): ONE HAPLOTYPE PER 16-LANE DPP ROW, the cross-label sums as
//     v_fmac_f64 with a row_newbcast DPP operand — no LDS and no shuffles on the chain; k_crf_scan (<= 8 labels) and
//     k_smooth_crf_lanes (any label count; the only kernel for 17..32) stay selectable with GNX_CRF_IMPL and serve as cross-checks.
// k_crf_scan / k_smooth_crf_lanes keep the oracle's left-to-right association (float64, no FMA contraction); the row kernel fuses
// its multiply-adds (marginals within 1e-11 of the oracle's, labels identical).
#include <cstdlib>
#include "gnx_internal.h"
#include "gnx_exp.h"

namespace {

// EXACT: A == AT, so a row is a compile-time number of doubles and the loads / stores vectorise (2 doubles per access)
template <int AT, bool EXACT>
__global__ __launch_bounds__(256) void k_crf_psi(SmoothCRFLaunch L) {
  __shared__ double th[AT * AT];  // theta[a][y]
  const int A = EXACT ? AT : L.A;
  for (int i = threadIdx.x; i < AT * AT; i += blockDim.x) {
    const int a = i / AT, y = i - a * AT;
    th[i] = (a < A && y < A) ? L.state[a * A + y] : 0.0;
  }
  __syncthreads();
  const int64_t total = L.N * L.W;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  double b[AT], o[AT];
  if (L.b_is_f64) {
    const double* src = reinterpret_cast<const double*>(L.B) + e * A;
#pragma unroll
    for (int a = 0; a < AT; ++a) b[a] = (a < A) ? src[a] : 0.0;
  } else {
    const float* src = reinterpret_cast<const float*>(L.B) + e * A;
#pragma unroll
    for (int a = 0; a < AT; ++a) b[a] = (a < A) ? (double)src[a] : 0.0;
  }
  double* dst = L.psi + e * A;
  if constexpr (EXACT) {
#pragma unroll
    for (int y = 0; y < AT; ++y) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < AT; ++a) s += th[a * AT + y] * b[a];
      o[y] = exp(s);
    }
#pragma unroll
    for (int y = 0; y < AT; ++y) dst[y] = o[y];
  } else {
    // run-time label count below the template bound (17..32 labels, or 9..15 through the 16-wide instance): one label at a
    // time, stored as it is ready.  Fully unrolled, AT inlined exp() bodies and the o[] vector cost 52 B .. 6 KB of scratch per
    // lane; the sums run over a = 0..A-1 in the same order either way.
    (void)o;
#pragma unroll 1
    for (int y = 0; y < A; ++y) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < AT; ++a)
        if (a < A) s += th[a * AT + y] * b[a];
      dst[y] = exp(s);
    }
  }
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// exp(tau): read from LDS once — the compiler keeps all 2 A^2 doubles (both orientations) in registers across the W steps (one
// wave per SIMD: 512 VGPRs are there to be used; re-reading them per step put 2 A^2 LDS latencies on the chain: 40 ms vs 16 ms)
__device__ __forceinline__ double ld_et(const double* p) { return *p; }

// One lane per haplotype, one wave per block, A = AT exactly.  The lane's rows of psi (and, on the way back, of the parked
// alphas) arrive through an LDS ring filled by `global_load_lds_dwordx4`: chunk c = the TC windows [c TC, (c+1) TC) of every
// lane's row = TC*A doubles = PC 16-byte pieces per lane, piece p of all 64 lanes in ring slot bytes [p*1024, (p+1)*1024)
// (LDS-direct loads write lane-linear, which is exactly "every lane its own row").  Two chunks are in flight while one is
// computed and no register holds data on its way in, so HBM latency never sits on the W-step chain; the only waits are
// `s_waitcnt vmcnt(loads of the youngest chunk)` (loads retire in order).
template <int AT>
__global__ __launch_bounds__(64) void k_crf_scan(SmoothCRFLaunch L) {
  constexpr int TC = 4, NS = 3;
  constexpr int PC = TC * AT / 2;              // 16-byte pieces per lane and chunk (TC * AT is even)
  constexpr int SLOT = PC * 1024;              // bytes per ring slot and array
  static_assert(2 * PC < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  double* et = reinterpret_cast<double*>(lds);  // exp(tau)[y'][y]
  double* etT = et + AT * AT;                   // transposed: [y][y']
  uint8_t* ring = lds + ((2 * AT * AT * 8 + 15) & ~15);  // [2 arrays][NS][SLOT]
  const int W = L.W, lane = threadIdx.x;
  for (int i = lane; i < AT * AT; i += 64) {
    const int r = i / AT, c = i - r * AT;
    et[r * AT + c] = L.etrans[r * AT + c];
    etT[c * AT + r] = L.etrans[r * AT + c];
  }
  __syncthreads();
  const int64_t n = (int64_t)blockIdx.x * 64 + lane;
  const bool active = n < L.N;
  const int64_t nn = active ? n : L.N - 1;
  const size_t row0 = (size_t)nn * W * AT;
  const uint8_t* psi_row = reinterpret_cast<const uint8_t*>(L.psi + row0);
  const uint8_t* alpha_row = reinterpret_cast<const uint8_t*>(L.alpha + row0);
  double* alpha = L.alpha + row0;  // parked scaled alphas: always the context's scratch (its tail is padded: chunks may overrun a row)
  double* scale = L.scale + (size_t)nn * W;
  const int n_chunks = (W + TC - 1) / TC;
  auto issue = [&](const uint8_t* rowp, int arr, int c) {  // unconditional, clamped chunk index: PC loads, always
    const int cc = c < 0 ? 0 : (c > n_chunks - 1 ? n_chunks - 1 : c);
    const uint8_t* src = rowp + (size_t)cc * (TC * AT * 8);
    uint8_t* dst = ring + (size_t)(arr * NS + ((c % NS) + NS) % NS) * SLOT;
#pragma unroll
    for (int p = 0; p < PC; ++p) __builtin_amdgcn_global_load_lds((gptr_t)(src + p * 16), (lptr_t)(dst + p * 1024), 16, 0, 0);
  };
  auto slot_val = [&](int arr, int c, int idx) -> double {  // value idx (= step-in-chunk * AT + label) of this lane's chunk c
    const uint8_t* base = ring + (size_t)(arr * NS + ((c % NS) + NS) % NS) * SLOT;
    return *reinterpret_cast<const double*>(base + (idx >> 1) * 1024 + lane * 16 + (idx & 1) * 8);
  };

  // ---- forward: alpha_0 = psi_0, alpha_t(y) = psi_t(y) * sum_y' alpha_{t-1}(y') exp(tau)[y'][y], each alpha_t scaled to sum 1 ----
  double ap[AT];
#pragma unroll
  for (int y = 0; y < AT; ++y) ap[y] = 0.0;
  issue(psi_row, 0, 0);
  issue(psi_row, 0, 1);
  for (int c = 0; c < n_chunks; ++c) {
    wait_vm<PC>();                 // chunk c has landed (at most the PC loads of chunk c+1 are still out)
    issue(psi_row, 0, c + 2);      // into the slot chunk c-1 left
#pragma unroll
    for (int k = 0; k < TC; ++k) {
      const int t = c * TC + k;
      if (t < W) {
        double pc[AT], v[AT];
#pragma unroll
        for (int y = 0; y < AT; ++y) pc[y] = slot_val(0, c, k * AT + y);
        if (t == 0) {
#pragma unroll
          for (int y = 0; y < AT; ++y) v[y] = pc[y];
        } else {
          double acc[AT];
#pragma unroll
          for (int y = 0; y < AT; ++y) acc[y] = 0.0;
#pragma unroll
          for (int yp = 0; yp < AT; ++yp)
#pragma unroll
            for (int y = 0; y < AT; ++y) acc[y] += ap[yp] * ld_et(et + yp * AT + y);  // per y: y' = 0, 1, ... in order, as the oracle
#pragma unroll
          for (int y = 0; y < AT; ++y) v[y] = acc[y] * pc[y];
        }
        double sum = 0.0;
#pragma unroll
        for (int y = 0; y < AT; ++y) sum += v[y];
        const double sc = (sum != 0.0) ? 1.0 / sum : 1.0;
#pragma unroll
        for (int y = 0; y < AT; ++y) ap[y] = v[y] * sc;
        if (active) {
#pragma unroll
          for (int y = 0; y < AT; ++y) alpha[(size_t)t * AT + y] = ap[y];
          scale[t] = sc;
        }
      }
    }
  }
  wait_vm<0>();            // the stores of this lane's alphas / scales are done ...
  __threadfence_block();   // ... before the backward pass reads them back

  // ---- backward: beta_{W-1} = c_{W-1}, beta_t(y') = c_t sum_y exp(tau)[y'][y] psi_{t+1}(y) beta_{t+1}(y); marginal = alpha beta / c ----
  double beta[AT], psin[AT];
#pragma unroll
  for (int y = 0; y < AT; ++y) { beta[y] = 0.0; psin[y] = 0.0; }
  auto issue2 = [&](int c) { issue(psi_row, 0, c); issue(alpha_row, 1, c); };
  issue2(n_chunks - 1);
  issue2(n_chunks - 2);
  for (int c = n_chunks - 1; c >= 0; --c) {
    wait_vm<2 * PC>();
    issue2(c - 2);
    double scs[TC];
#pragma unroll
    for (int k = 0; k < TC; ++k) scs[k] = scale[min(c * TC + k, W - 1)];
#pragma unroll
    for (int k = TC - 1; k >= 0; --k) {
      const int t = c * TC + k;
      if (t < W) {
        const double sct = scs[k];
        if (t < W - 1) {
          double acc[AT];
#pragma unroll
          for (int yp = 0; yp < AT; ++yp) acc[yp] = 0.0;
#pragma unroll
          for (int y = 0; y < AT; ++y)
#pragma unroll
            for (int yp = 0; yp < AT; ++yp) acc[yp] += ld_et(etT + y * AT + yp) * psin[y] * beta[y];  // per y': y = 0, 1, ... in order
#pragma unroll
          for (int yp = 0; yp < AT; ++yp) beta[yp] = acc[yp] * sct;
        } else {
#pragma unroll
          for (int y = 0; y < AT; ++y) beta[y] = sct;
        }
        double m[AT];
#pragma unroll
        for (int y = 0; y < AT; ++y) {
          psin[y] = slot_val(0, c, k * AT + y);  // psi_t, consumed by step t-1
          m[y] = slot_val(1, c, k * AT + y) * beta[y] / sct;
        }
        int best = 0;  // argmax, first max wins
        double bv = m[0];
#pragma unroll
        for (int y = 1; y < AT; ++y)
          if (m[y] > bv) { bv = m[y]; best = y; }
        if (active) {
          const size_t o = row0 + (size_t)t * AT;
#pragma unroll
          for (int y = 0; y < AT; ++y) {
            if (L.proba64) L.proba64[o + y] = m[y];
            if (L.proba32) L.proba32[o + y] = (float)m[y];
          }
          if (L.labels) L.labels[(size_t)nn * W + t] = best;
        }
      }
    }
  }
  wait_vm<0>();  // nothing of this block may still be writing LDS when it retires
}

template <int AT>
hipError_t launch(const SmoothCRFLaunch& L, hipStream_t s) {
  const int64_t total = L.N * L.W;
  hipLaunchKernelGGL((k_crf_psi<AT, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L);
  constexpr size_t lds = ((2 * AT * AT * 8 + 15) & ~15) + (size_t)2 * 3 * (4 * AT / 2) * 1024;
  GNX_LDS_OPTIN(lds, k_crf_scan<AT>);
  hipLaunchKernelGGL(k_crf_scan<AT>, dim3((unsigned)((L.N + 63) / 64)), dim3(64), lds, s, L);
  return hipGetLastError();
}

// ---- more than 8 labels: A lanes per haplotype (lane = label), cross-label terms by ds_bpermute shuffles -------------------------
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

template <bool PRE>
__global__ __launch_bounds__(256) void k_smooth_crf_lanes(SmoothCRFLaunch L) {
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

  // PRE: psi_t(y) comes from k_crf_psi's pass (it is not part of the recurrence: computed once for both directions, off the chain)
  auto loadB = [&](int t) -> double {
    const size_t idx = row0 + (size_t)t * A + y;
    if (PRE) return L.psi[idx];
    return L.b_is_f64 ? reinterpret_cast<const double*>(L.B)[idx] : (double)reinterpret_cast<const float*>(L.B)[idx];
  };
  auto psi_of = [&](double myB) -> double {  // exp(sum_a theta[a][y] * B[t][a])
    if (PRE) return myB;
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


// ---- 9..16 labels: one haplotype per 16-lane DPP row (lane = label) ---------------------------------------------------------------
// A step of the lanes kernel above costs 60-85 LDS-pipe operations (every shuffle of a double is two ds_bpermute) on a strictly
// dependent chain: ~9.6 k clocks per window at A = 12.  gfx950's double-precision ALU takes a DPP operand on v_fmac_f64 with
// row_newbcast:i ("every lane of a 16-lane row reads lane i of its row"), so with one haplotype per row
//     acc += alpha[i] * E[i][y]          is ONE instruction per label i (the coefficient E[i][y] sits in a register of lane y),
// in label order, and a row sum is the same instruction against 1.0.  No LDS and no separate moves on the chain.  Differences to the
// oracle's arithmetic, all inside the 1e-11 the tests allow on marginals (labels identical): the multiply-adds are fused, 1/sum is a
// refined v_rcp_f64 instead of a division, and alpha*beta/c_t multiplies by the stored 1/c_t.
template <int I>
__device__ __forceinline__ void fmac_bcast(double& acc, double v, double c) {  // acc += (lane I of this row's v) * c
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(c), "n"(I));
}
// The AT multiply-adds of a row product as ONE asm statement: between separate statements the compiler (which cannot see into
// them) puts a wait state after every float64 DPP operation — ~100 s_nop per window, each an issue slot of the wave's in-order
// stream.  The chain needs none: the accumulator is not the DPP source, and plain VALU read-after-write is interlocked in hardware.
// (scripts/dev/f64_rate_probe.hip: v_fmac_f64_dpp issues at the full float64 rate, 4 cycles, 12 cycles dependent — like v_fma_f64.)
template <int AT>
__device__ __forceinline__ void row_dot(double& acc, double v, const double (&c)[AT]) {
  static_assert(AT == 8 || AT == 12 || AT == 16, "row widths the launchers use");
  if constexpr (AT == 8) {
    asm(
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf"
      : "+v"(acc) : "v"(v), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
  } else if constexpr (AT == 12) {
    asm(
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf"
      : "+v"(acc) : "v"(v), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]), "v"(c[10]), "v"(c[11]));
  } else {
    asm(
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf"
      : "+v"(acc) : "v"(v), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]), "v"(c[9]), "v"(c[10]), "v"(c[11]),
                    "v"(c[12]), "v"(c[13]), "v"(c[14]), "v"(c[15]));
  }
}
template <int AT>
__device__ __forceinline__ void row_sum(double& acc, double v) {
  static_assert(AT == 8 || AT == 12 || AT == 16, "row widths the launchers use");
  const double one = 1.0;   // (the DPP encoding takes registers only)
  if constexpr (AT == 8) {
    asm(
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf"
      : "+v"(acc) : "v"(v), "v"(one));
  } else if constexpr (AT == 12) {
    asm(
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xf"
      : "+v"(acc) : "v"(v), "v"(one));
  } else {
    asm(
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf"
      : "+v"(acc) : "v"(v), "v"(one));
  }
}
// gfx9 wants two wait states between the VALU write of a VGPR and a DPP read of it; inline asm is opaque to the hazard recogniser
__device__ __forceinline__ double dpp_ready(double v) {
  asm volatile("s_nop 1" : "+v"(v));
  return v;
}
// maximum of a 32-bit value over the 16 lanes of a DPP row (every lane gets it): v_max_u32 with the rotated value as its DPP operand
__device__ __forceinline__ uint32_t row_max_u32(uint32_t v) {
  uint32_t r;
  asm("s_nop 1\n\t"
      "v_max_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
      : "=&v"(r) : "v"(v));
  return r;
}
template <int K>
__device__ __forceinline__ double ror16(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x120 + K, 0xf, 0xf, false);  // row_ror:K
  hi = __builtin_amdgcn_mov_dpp(hi, 0x120 + K, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// FUSE (GNX_CRF_FLAGS=1, not the default): psi_t(y) = exp(sum_a theta[a][y] B[t][a]) computed here from the prefetched B row, off the
// chain, in both directions — no psi pass and no psi buffer (a third less HBM traffic).  Measured: chr22 / A = 7 0.57 ms against
// 0.09 + 0.38 ms, chr1 / A = 12 5.71 against 5.65 ms: two float64 exp per window cost the issue slots the saved bytes would buy.
template <int AT, bool FUSE>  // labels padded to AT (8, 12 or 16) inside the 16-lane row; padded coefficients are zero
__global__ __launch_bounds__(256) void k_smooth_crf_row16(SmoothCRFLaunch L) {
  const int A = L.A, W = L.W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y = lane & 15;
  const int64_t n = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * 4 + (lane >> 4);
  const bool label = y < A, active = label && n < L.N;
  const int64_t nn = (n < L.N) ? n : 0;
  const size_t row0 = (size_t)nn * W * A;
  double* alpha = L.alpha + row0;
  double* scale = L.scale + (size_t)nn * W * 2;  // [t] = (c_t, 1/c_t)

  double Ef[AT], Eb[AT], Th[FUSE ? AT : 1];  // forward: from label i to y; backward: from y to i; theta[i][y]
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    const bool ok = label && i < A;
    Ef[i] = ok ? L.etrans[i * A + y] : 0.0;
    Eb[i] = ok ? L.etrans[y * A + i] : 0.0;
    if constexpr (FUSE) Th[i] = ok ? L.state[i * A + y] : 0.0;
  }

  constexpr int PFD = 8;  // steps per software-pipeline stage (loads run one stage ahead: ~1 us of chain hides an HBM round trip)
  auto clampt = [&](int t) { return t < 0 ? 0 : (t > W - 1 ? W - 1 : t); };
  // loads are unconditional (a padding lane reads label 0's entry and drops it): a load under a divergent branch makes the compiler
  // wait for ALL outstanding loads before the next use, which would put the prefetches of the next stage on the chain
  const int yl = label ? y : 0;
  auto loadP = [&](int t) -> double {  // psi, or with FUSE the base probability it is computed from
    const size_t idx = row0 + (size_t)t * A + yl;
    double v;
    if constexpr (FUSE) v = L.b_is_f64 ? reinterpret_cast<const double*>(L.B)[idx] : (double)reinterpret_cast<const float*>(L.B)[idx];
    else v = L.psi[idx];
    return label ? v : 0.0;
  };
  auto psi_of = [&](double b) -> double {
    if constexpr (FUSE) {
      double sdot = 0.0;
      row_dot<AT>(sdot, dpp_ready(b), Th);
      return label ? exp(sdot) : 0.0;
    } else {
      return b;
    }
  };

  // ---- forward ----
  double a_prev = 0.0;
  double bn[PFD];
#pragma unroll
  for (int k = 0; k < PFD; ++k) bn[k] = loadP(clampt(k));
  for (int t0 = 0; t0 < W; t0 += PFD) {
    double bc[PFD];
#pragma unroll
    for (int k = 0; k < PFD; ++k) bc[k] = bn[k];
#pragma unroll
    for (int k = 0; k < PFD; ++k) bn[k] = loadP(clampt(t0 + PFD + k));  // unconditional, clamped: in flight during this stage
#pragma unroll
    for (int k = 0; k < PFD; ++k) {
      const int t = t0 + k;
      if (t < W) {
        const double psi = psi_of(bc[k]);
        double v = psi;
        if (t > 0) {
          double acc = 0.0;
          row_dot<AT>(acc, dpp_ready(a_prev), Ef);
          v = acc * psi;
        }
        double sum = 0.0;
        row_sum<AT>(sum, dpp_ready(v));
        const bool nz = sum != 0.0;
        sum = nz ? sum : 1.0;
        double sc = __builtin_amdgcn_rcp(sum);
        sc = fma(fma(-sum, sc, 1.0), sc, sc);
        sc = fma(fma(-sum, sc, 1.0), sc, sc);
        sc = nz ? sc : 1.0;
        a_prev = v * sc;
        if (active) {
          alpha[(size_t)t * A + y] = a_prev;
          // every label lane writes the same pair and later reads back its OWN store
          *reinterpret_cast<double2*>(scale + (size_t)t * 2) = make_double2(sc, sum);
        }
      }
    }
  }
  __threadfence_block();

  // ---- backward + marginals ----
  auto loadS = [&](int t) -> double2 { return *reinterpret_cast<const double2*>(scale + (size_t)t * 2); };  // (rows beyond N: row 0's)
  auto loadA = [&](int t) -> double { const double v = alpha[(size_t)t * A + yl]; return label ? v : 0.0; };
  double beta = 0.0, psi_next = 0.0;
  double an[PFD];
  double2 sn[PFD];
#pragma unroll
  for (int k = 0; k < PFD; ++k) {
    const int t = clampt(W - 1 - k);
    bn[k] = loadP(t); an[k] = loadA(t); sn[k] = loadS(t);
  }
  for (int t0 = W - 1; t0 >= 0; t0 -= PFD) {
    double bc[PFD], ac[PFD];
    double2 scur[PFD];
#pragma unroll
    for (int k = 0; k < PFD; ++k) { bc[k] = bn[k]; ac[k] = an[k]; scur[k] = sn[k]; }
#pragma unroll
    for (int k = 0; k < PFD; ++k) {
      const int t = clampt(t0 - PFD - k);
      bn[k] = loadP(t); an[k] = loadA(t); sn[k] = loadS(t);
    }
#pragma unroll
    for (int k = 0; k < PFD; ++k) {
      const int t = t0 - k;
      if (t >= 0) {
        const double sct = scur[k].x;
        // beta_t(y') = c_t * sum_y exp(tau)[y'][y] * psi_{t+1}(y) * beta_{t+1}(y)     (this lane: y' = y)
        if (t < W - 1) {
          double acc = 0.0;
          row_dot<AT>(acc, dpp_ready(psi_next * beta), Eb);
          beta = acc * sct;
        } else {
          beta = sct;
        }
        psi_next = psi_of(bc[k]);  // psi_t, consumed by step t-1
        const double m = ac[k] * beta * scur[k].y;
        // arg-max over the row, first maximum wins: the row maximum by rotations, then the lowest label that attains it
        double mx = label ? m : -1.0;
        mx = fmax(mx, ror16<8>(mx));
        mx = fmax(mx, ror16<4>(mx));
        mx = fmax(mx, ror16<2>(mx));
        mx = fmax(mx, ror16<1>(mx));
        const unsigned long long hit = __ballot(label && m == mx);
        const int best = __builtin_ctz((unsigned)(hit >> (lane & 48)) & 0xffffu);
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

// ---- round 4: the same recurrence with CHECKPOINTED alphas (default) ---------------------------------------------------------------
// k_smooth_crf_row16 is HBM-bound at chr1 / A = 12 (16.9 GB per launch: psi twice, every alpha and (c_t, 1/c_t) pair out and back,
// marginals out).  Here the forward sweep parks alpha only at the end of every SEG-window segment; the backward sweep takes a segment's
// psi (the registers of its prefetch stage), recomputes the segment's alphas and scales from the parked vector into wave-private LDS
// (same instruction sequence: same bits) and runs beta through it.  Per launch: psi twice + marginals + 1/SEG of the alphas twice =
// 11.2 GB instead of 16.9.  FWDPSI: the forward sweep computes psi from B itself (off the chain) and writes it for the backward
// sweep — no k_crf_psi pass (B once + psi out instead of B once + psi out + psi in).
// RECOMP (with FWDPSI): psi is never written — the backward sweep reads B again and recomputes its segment's psi with the same
// instruction sequence (same bits): B twice + marginals + 1/SEG of the alphas twice = 11.2 GB -> the psi array (N W A float64) is not
// needed at all; the price is the psi phase (A multiply-adds + one exp per window and lane) a second time and 146 registers instead of
// 122 (three waves per SIMD instead of four).  MEASURED at config 5a (25 000 chr1 haplotypes, A = 12): 3.87-3.91 ms against 3.43-3.47 —
// a quarter of the traffic less and 13 % slower: the kernel is bound by the length of its instruction stream, not by its bytes
// (DESIGN.md 4.3).  Built in `make EXPERIMENTS=1` only (GNX_CRF_FLAGS=8); outputs bit-identical (the CRF parity tests pass with it).
template <int AT, bool FWDPSI, int NWB, bool RECOMP = false>   // NWB waves per workgroup
__global__ __launch_bounds__(64 * NWB) void k_smooth_crf_ck(SmoothCRFLaunch L) {
  static_assert(!RECOMP || FWDPSI, "recomputing psi needs the in-kernel psi phase");
  constexpr int SEG = 8, EXPW = 4;
  __shared__ double la[NWB][SEG][64];    // [wave][step][lane] recomputed alpha_t(y)
  __shared__ double2 lsc[NWB][SEG][4];   // [wave][step][row] (1/c_t, c_t)
  __shared__ double lth[FWDPSI ? AT : 1][16];  // theta[i][y]: read per segment in the psi phase only (24 registers less held through the chain)
  const int A = L.A, W = L.W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int y = lane & 15, row = lane >> 4;
  const int64_t n = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * 4 + row;
  const bool label = y < A, active = label && n < L.N;
  const int64_t nn = (n < L.N) ? n : 0;
  const size_t row0 = (size_t)nn * W * A;
  const int NSEG = (W + SEG - 1) / SEG;
  double* ck = L.alpha + (size_t)nn * NSEG * A;  // [segment][label] alpha at the segment's last window

  double Ef[AT];  // (Eb, the transposed coefficients of the backward product, are loaded after the forward sweep: 24 registers less in it)
#pragma unroll
  for (int i = 0; i < AT; ++i) Ef[i] = (label && i < A) ? L.etrans[i * A + y] : 0.0;
  if constexpr (FWDPSI) {
    for (int e = threadIdx.x; e < AT * 16; e += blockDim.x) {
      const int i = e >> 4, yy = e & 15;
      lth[i][yy] = (i < A && yy < A) ? L.state[i * A + yy] : 0.0;
    }
    __syncthreads();
  }
  auto clampt = [&](int t) { return t < 0 ? 0 : (t > W - 1 ? W - 1 : t); };
  const int yl = label ? y : 0;
  auto loadB = [&](int t) -> double {
    const size_t idx = row0 + (size_t)t * A + yl;
    const double v = L.b_is_f64 ? reinterpret_cast<const double*>(L.B)[idx] : (double)reinterpret_cast<const float*>(L.B)[idx];
    return label ? v : 0.0;
  };
  auto loadPsi = [&](int t) -> double { const double v = L.psi[row0 + (size_t)t * A + yl]; return label ? v : 0.0; };
  // one step of the scaled forward recurrence: alpha_t = psi_t * (alpha_{t-1} . E) / c_t; returns alpha_t, sets (1/c_t, c_t).
  // The scales only keep the numbers in range: ANY positive c_t give the same marginals as long as both sweeps use the same ones and
  // the last window's is the true row sum (then prod c_t = Z).  A row sum is 12 dependent DPP multiply-adds, a reciprocal with two
  // refinement steps and a multiply on the critical path of two of the three sweeps: the sum and its reciprocal are taken every
  // (norm_mask + 1)-th window only (and at the last one), c_t = 1 in between (config 5a: 5.19 -> 4.42 ms at every 4th).  gnx_build_crf picks
  // 8, 4, 2 or 1 windows from the weights' range so that the unscaled stretch stays inside float64 (8 for any trained model).
  const int norm_mask = L.norm_mask;
  auto norm_at = [&](int t) { return (t & norm_mask) == norm_mask || t == W - 1; };
  // a segment whose windows need no checks: all inside the chain, not the first, the scale exactly at its last window (interval 8 = SEG)
  auto plain_seg = [&](int t0) { return norm_mask == SEG - 1 && t0 > 0 && t0 + SEG < W; };
  // (first / norm are wave-uniform; the interior segments pass compile-time constants: the window's checks cost issue slots like
  //  everything else — scripts/dev/f64_rate_probe.hip, DESIGN.md 4.3: the kernel is bound by the length of its instruction stream)
  auto fwd_step = [&](double a_prev, double psi, bool first, bool norm, double& sc, double& sum) -> double {
    double v = psi;
    if (!first) {
      double acc = 0.0;
      row_dot<AT>(acc, dpp_ready(a_prev), Ef);
      v = acc * psi;
    }
    if (!norm) {
      sc = 1.0;
      sum = 1.0;
      return v;
    }
    sum = 0.0;
    row_sum<AT>(sum, dpp_ready(v));
    const bool nz = sum != 0.0;
    sum = nz ? sum : 1.0;
    sc = __builtin_amdgcn_rcp(sum);
    sc = fma(fma(-sum, sc, 1.0), sc, sc);
    sc = fma(fma(-sum, sc, 1.0), sc, sc);
    sc = nz ? sc : 1.0;
    return v * sc;
  };

  // ---- forward: alpha parked once per segment ----
  size_t rowyf = row0 + (size_t)yl;
  asm volatile("" : "+v"(rowyf));
  double a_prev = 0.0;
  double bn[SEG];
#pragma unroll
  for (int k = 0; k < SEG; ++k) bn[k] = FWDPSI ? loadB(clampt(k)) : loadPsi(clampt(k));
  for (int sg = 0; sg < NSEG; ++sg) {
    const int t0 = sg * SEG;
    const bool plain = plain_seg(t0);
    double bc[SEG];
#pragma unroll
    for (int k = 0; k < SEG; ++k) bc[k] = bn[k];
    if constexpr (!FWDPSI) {
#pragma unroll
      for (int k = 0; k < SEG; ++k) bn[k] = loadPsi(clampt(t0 + SEG + k));
    }
    if constexpr (FWDPSI) {  // psi of the whole segment first: nothing of it is on the chain
      double Th[AT];
      {
        int yo = y;
        asm volatile("" : "+v"(yo));  // (an address the optimiser cannot prove loop-invariant: the reads stay inside the segment)
#pragma unroll
        for (int i = 0; i < AT; ++i) Th[i] = lth[i][yo];
      }
#pragma unroll
      for (int k0 = 0; k0 < SEG; k0 += EXPW) {   // EXPW windows at a time: the exponential's constants are moved into scalar pairs once for all of them
        double sd[EXPW];
#pragma unroll
        for (int i = 0; i < EXPW; ++i) {
          sd[i] = 0.0;
          row_dot<AT>(sd[i], dpp_ready(bc[k0 + i]), Th);
        }
        gnx_exp_scN<EXPW>(sd);
#pragma unroll
        for (int i = 0; i < EXPW; ++i) {
          bc[k0 + i] = label ? sd[i] : 0.0;
          if (!RECOMP && active && (plain || t0 + k0 + i < W)) L.psi[row0 + (size_t)(t0 + k0 + i) * A + y] = bc[k0 + i];
        }
      }
    }
    if constexpr (FWDPSI) {  // the next segment's B: requested AFTER the psi phase (its registers are free in there) — the 8 chain steps
      if (t0 + 2 * SEG <= W) {  // that follow are several microseconds, more than the loads need.  Inside the chain: no clamps
        const size_t e0 = rowyf + (size_t)((t0 + SEG) * A);
        if (L.b_is_f64) {
          const double* pp = reinterpret_cast<const double*>(L.B) + e0;
#pragma unroll
          for (int k = 0; k < SEG; ++k) { const double v = pp[k * A]; bn[k] = label ? v : 0.0; }
        } else {
          const float* pp = reinterpret_cast<const float*>(L.B) + e0;
#pragma unroll
          for (int k = 0; k < SEG; ++k) { const double v = (double)pp[k * A]; bn[k] = label ? v : 0.0; }
        }
      } else {
#pragma unroll
        for (int k = 0; k < SEG; ++k) bn[k] = loadB(clampt(t0 + SEG + k));
      }
    }
    if (plain) {
#pragma unroll
      for (int k = 0; k < SEG; ++k) {
        double sc, sum;
        a_prev = fwd_step(a_prev, bc[k], false, k == SEG - 1, sc, sum);
      }
    } else {
#pragma unroll
      for (int k = 0; k < SEG; ++k) {
        const int t = t0 + k;
        if (t < W) {
          double sc, sum;
          a_prev = fwd_step(a_prev, bc[k], t == 0, norm_at(t), sc, sum);
        }
      }
    }
    if (active) ck[(size_t)sg * A + y] = a_prev;
  }
  __threadfence_block();

  // ---- backward: per segment recompute alpha into LDS, then beta and the marginals ----
  double Eb[AT];
  {
    int yo = y;
    asm volatile("" : "+v"(yo));  // (keeps these loads below the forward sweep)
#pragma unroll
    for (int i = 0; i < AT; ++i) Eb[i] = (label && i < A) ? L.etrans[yo * A + i] : 0.0;
  }
  size_t oy = row0 + (size_t)y;
  asm volatile("" : "+v"(oy));
  size_t rowy = row0 + (size_t)yl;
  asm volatile("" : "+v"(rowy));
  double beta = 0.0, psi_next = 0.0;
  double an = 0.0;
  {
    const int sg = NSEG - 1;
#pragma unroll
    for (int k = 0; k < SEG; ++k) bn[k] = RECOMP ? loadB(clampt(sg * SEG + k)) : loadPsi(clampt(sg * SEG + k));
    an = (active && sg > 0) ? ck[(size_t)(sg - 1) * A + y] : 0.0;
  }
  for (int sg = NSEG - 1; sg >= 0; --sg) {
    const int t0 = sg * SEG;
    double bc[SEG];
#pragma unroll
    for (int k = 0; k < SEG; ++k) bc[k] = bn[k];
    double a_in = an;
    {
      const int sp = sg > 0 ? sg - 1 : 0;
      if constexpr (RECOMP) {   // this segment's psi from its B (the forward sweep's psi phase, instruction for instruction), then the next segment's B
        double Th[AT];
        {
          int yo = y;
          asm volatile("" : "+v"(yo));
#pragma unroll
          for (int i = 0; i < AT; ++i) Th[i] = lth[i][yo];
        }
#pragma unroll
        for (int k0 = 0; k0 < SEG; k0 += EXPW) {
          double sd[EXPW];
#pragma unroll
          for (int i = 0; i < EXPW; ++i) {
            sd[i] = 0.0;
            row_dot<AT>(sd[i], dpp_ready(bc[k0 + i]), Th);
          }
          gnx_exp_scN<EXPW>(sd);
#pragma unroll
          for (int i = 0; i < EXPW; ++i) bc[k0 + i] = label ? sd[i] : 0.0;
        }
        const size_t e0 = rowy + (size_t)(sp * SEG * A);
        if (L.b_is_f64) {
          const double* pp = reinterpret_cast<const double*>(L.B) + e0;
#pragma unroll
          for (int k = 0; k < SEG; ++k) { const double v = pp[k * A]; bn[k] = label ? v : 0.0; }
        } else {
          const float* pp = reinterpret_cast<const float*>(L.B) + e0;
#pragma unroll
          for (int k = 0; k < SEG; ++k) { const double v = (double)pp[k * A]; bn[k] = label ? v : 0.0; }
        }
      } else {  // segment sp always lies inside the chain: one per-lane address and scalar offsets k * A — no clamp, no 64-bit multiply per window
        const double* pp = L.psi + (rowy + (size_t)(sp * SEG * A));
#pragma unroll
        for (int k = 0; k < SEG; ++k) { const double v = pp[k * A]; bn[k] = label ? v : 0.0; }
      }
      const double v = ck[(size_t)(sp > 0 ? sp - 1 : 0) * A + yl];  // unconditional; dropped for the first segment / padding lanes
      an = (label && sp > 0) ? v : 0.0;
    }
    const bool plain = plain_seg(t0);
    if (plain) {
#pragma unroll
      for (int k = 0; k < SEG; ++k) {
        double sc, sum;
        a_in = fwd_step(a_in, bc[k], false, k == SEG - 1, sc, sum);
        la[wave][k][lane] = a_in;
        if (k == SEG - 1 && y == 0) lsc[wave][0][row] = make_double2(sc, sum);   // (the segment's one scale pair; the others are (1, 1))
      }
    } else {
#pragma unroll
      for (int k = 0; k < SEG; ++k) {
        const int t = t0 + k;
        if (t < W) {
          double sc, sum;
          a_in = fwd_step(a_in, bc[k], t == 0, norm_at(t), sc, sum);
          la[wave][k][lane] = a_in;
          if (y == 0) lsc[wave][k][row] = make_double2(sc, sum);
        }
      }
    }
    // (wave-private LDS: a lane reads back its own alpha; the row's pair is written by its lane 0 — LDS operations of a wave run in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // one window of the backward sweep; `chk`: the window may lie beyond W - 1 or be W - 1 itself; (sx, sy) = its (1/c_t, c_t)
    unsigned long long hits[SEG];   // plain segments: the windows' arg-max ballots, turned into labels and stored once per segment
    auto beta_step = [&](int k, bool chk, double sx, double sy) {
      const int t = t0 + k;
      if (chk && !(t < W)) return;
      if (!chk || t < W - 1) {
        double acc = 0.0;
        row_dot<AT>(acc, dpp_ready(psi_next * beta), Eb);
        beta = acc * sx;
      } else {
        beta = sx;
      }
      psi_next = bc[k];
      const double m = la[wave][k][lane] * beta * sy;
      // arg-max over the row, first maximum wins.  Marginals are >= 0, so float64 order is the order of the bit patterns: the row
      // maximum of the HIGH words by four 32-bit DPP rotations (v_max_u32 with a row_ror operand: 4 instructions instead of the
      // 8 moves + 8 v_max_f64 of a float64 butterfly), then the maximum of the LOW words among the lanes that hold it.
      const uint32_t mh = label ? (uint32_t)__double2hiint(m) : 0u, ml = (uint32_t)__double2loint(m);
      const uint32_t hmax = row_max_u32(mh);
      const bool top = label && mh == hmax;
      const uint32_t lmax = row_max_u32(top ? ml : 0u);
      const unsigned long long hit = __ballot(top && ml == lmax);
      if (!chk) hits[k] = hit;
      if (active) {
        const size_t o = oy + (size_t)(t * A);   // (t * A: one scalar multiply; oy = row0 + y is opaque to the optimiser, which otherwise rebuilds ((n W + t) A + y) in 64-bit vector arithmetic every window)
        if (L.proba64) L.proba64[o] = m;
        if (L.proba32) L.proba32[o] = (float)m;
        if (chk && L.labels && y == 0) L.labels[(size_t)nn * W + t] = __builtin_ctz((unsigned)(hit >> (lane & 48)) & 0xffffu);
      }
    };
    if (plain) {
      const double2 s7 = lsc[wave][0][row];
#pragma unroll
      for (int k = SEG - 1; k >= 0; --k) beta_step(k, false, k == SEG - 1 ? s7.x : 1.0, k == SEG - 1 ? s7.y : 1.0);
      if (L.labels && active && y == 0) {   // the segment's 8 labels of this row: two 16-byte stores instead of eight masked 4-byte ones
        int lb[SEG];
#pragma unroll
        for (int k = 0; k < SEG; ++k) lb[k] = __builtin_ctz((unsigned)(hits[k] >> (lane & 48)) & 0xffffu);
        int* lp = L.labels + ((size_t)nn * W + t0);
#pragma unroll
        for (int k = 0; k < SEG; ++k) lp[k] = lb[k];
      }
    } else {
#pragma unroll
      for (int k = SEG - 1; k >= 0; --k) {
        const double2 scur = lsc[wave][k][row];
        beta_step(k, true, scur.x, scur.y);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

template <int AT>
hipError_t launch_lanes_pre(const SmoothCRFLaunch& L, hipStream_t s) {
  const int64_t total = L.N * L.W;
  if (L.A == AT) hipLaunchKernelGGL((k_crf_psi<AT, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L);
  else hipLaunchKernelGGL((k_crf_psi<AT, false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L);
  const int G = 64 / L.A, waves = 4;
  const int64_t per_block = (int64_t)G * waves;
  hipLaunchKernelGGL(k_smooth_crf_lanes<true>, dim3((unsigned)((L.N + per_block - 1) / per_block)), dim3(64 * waves),
                     (size_t)2 * L.A * L.A * sizeof(double), s, L);
  return hipGetLastError();
}

static void launch_psi(const SmoothCRFLaunch& L, hipStream_t s) {
  const dim3 grid((unsigned)((L.N * L.W + 255) / 256)), block(256);
  switch (L.A) {
    case 2: hipLaunchKernelGGL((k_crf_psi<2, true>), grid, block, 0, s, L); break;
    case 3: hipLaunchKernelGGL((k_crf_psi<3, true>), grid, block, 0, s, L); break;
    case 4: hipLaunchKernelGGL((k_crf_psi<4, true>), grid, block, 0, s, L); break;
    case 5: hipLaunchKernelGGL((k_crf_psi<5, true>), grid, block, 0, s, L); break;
    case 6: hipLaunchKernelGGL((k_crf_psi<6, true>), grid, block, 0, s, L); break;
    case 7: hipLaunchKernelGGL((k_crf_psi<7, true>), grid, block, 0, s, L); break;
    case 8: hipLaunchKernelGGL((k_crf_psi<8, true>), grid, block, 0, s, L); break;
    case 12: hipLaunchKernelGGL((k_crf_psi<12, true>), grid, block, 0, s, L); break;
    case 16: hipLaunchKernelGGL((k_crf_psi<16, true>), grid, block, 0, s, L); break;
    default: hipLaunchKernelGGL((k_crf_psi<16, false>), grid, block, 0, s, L); break;  // 9..15 labels
  }
}

hipError_t gnx_launch_smooth_crf(const SmoothCRFLaunch& L, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  if (!L.psi) return hipErrorInvalidValue;
  // Up to 16 labels: one haplotype per 16-lane DPP row (k_smooth_crf_row16).  GNX_CRF_IMPL=scan keeps the round-2a kernel for up to 8
  // labels (one lane per haplotype, LDS ring), =lanes the shuffle kernel (A lanes per haplotype), which also serves 17..32 labels.
  const int impl = tune.crf_impl;  // 0 auto, 1 scan, 2 row, 3 lanes, 4 quad
#ifdef GNX_EXPERIMENTS
  // the row products on the float64 matrix cores (scripts/dev/rejected/k_smooth_crf_mm.hip): no faster, see its header
  if (L.A <= 16 && impl == 5) return gnx_launch_smooth_crf_mm(L, s);
  // four lanes per haplotype (scripts/dev/rejected/k_smooth_crf_quad.hip): correct, 5 % slower than the row kernel at chr1 / A = 12
  if (L.A <= 12 && impl == 4) return gnx_launch_smooth_crf_quad(L, s);
#endif
  if (L.A <= 16 && (impl == 0 || impl == 2 || (impl == 1 && L.A > 8))) {
    const int waves = 4;
    const dim3 grid((unsigned)((L.N + 4 * waves - 1) / (4 * waves)));
    if (!(tune.crf_flags & 3)) {  // default (round 4): checkpointed alphas; GNX_CRF_FLAGS=4 keeps k_crf_psi as a separate pass
      const bool fwdpsi = !(tune.crf_flags & 4);
      if (!fwdpsi) launch_psi(L, s);
      // ONE wave per workgroup: the waves of the last, partly filled round spread over the SIMDs one by one instead of four by four
      // (measured: 1-2 % at 25 000 chr1 haplotypes = 6 250 waves on 4 096 slots, nothing at chr22; scripts/dev/crf_nsweep.py shows the
      // rounds: 4 096 waves 2.42 ms, 6 144 3.63 ms, 6 250 4.05 ms, 8 192 4.76 ms — a SIMD with three waves in the second round is the
      // critical path of the launch)
      static const int nwb_env = [] { const char* e = std::getenv("GNX_CRF_NWB"); return e ? atoi(e) : 1; }();
      const int nwb = nwb_env == 4 ? 4 : 1;   // (GNX_CRF_NWB=4: the four-wave workgroups of the first version, kept for comparison)
      const dim3 gridw((unsigned)((L.N + 4 * nwb - 1) / (4 * nwb)));
#define GNX_CK1(AT_, F_, W_) hipLaunchKernelGGL((k_smooth_crf_ck<AT_, F_, W_>), gridw, dim3(64 * W_), 0, s, L)
#ifdef GNX_EXPERIMENTS   /* GNX_CRF_FLAGS=8: psi recomputed by the backward sweep instead of written and read (RECOMP; measured slower, see the kernel's header) */
#define GNX_CK_RECOMP(AT_) if (fwdpsi && (tune.crf_flags & 8)) { hipLaunchKernelGGL((k_smooth_crf_ck<AT_, true, 1, true>), dim3((unsigned)((L.N + 3) / 4)), dim3(64), 0, s, L); } else
#else
#define GNX_CK_RECOMP(AT_)
#endif
#define GNX_CK(AT_) \
      GNX_CK_RECOMP(AT_) \
      if (fwdpsi) { if (nwb == 1) GNX_CK1(AT_, true, 1); else GNX_CK1(AT_, true, 4); } \
      else { if (nwb == 1) GNX_CK1(AT_, false, 1); else GNX_CK1(AT_, false, 4); }
      if (L.A <= 8) { GNX_CK(8) } else if (L.A <= 12) { GNX_CK(12) } else { GNX_CK(16) }
#undef GNX_CK
#undef GNX_CK_RECOMP
#undef GNX_CK1
    } else if (tune.crf_flags & 2) {  // GNX_CRF_FLAGS=2: round 3's kernel (every alpha and scale pair parked), psi from the separate pass
      launch_psi(L, s);
      if (L.A <= 8) hipLaunchKernelGGL((k_smooth_crf_row16<8, false>), grid, dim3(64 * waves), 0, s, L);
      else if (L.A <= 12) hipLaunchKernelGGL((k_smooth_crf_row16<12, false>), grid, dim3(64 * waves), 0, s, L);
      else hipLaunchKernelGGL((k_smooth_crf_row16<16, false>), grid, dim3(64 * waves), 0, s, L);
    } else {  // GNX_CRF_FLAGS=1: psi computed in both sweeps of round 3's kernel
      if (L.A <= 8) hipLaunchKernelGGL((k_smooth_crf_row16<8, true>), grid, dim3(64 * waves), 0, s, L);
      else if (L.A <= 12) hipLaunchKernelGGL((k_smooth_crf_row16<12, true>), grid, dim3(64 * waves), 0, s, L);
      else hipLaunchKernelGGL((k_smooth_crf_row16<16, true>), grid, dim3(64 * waves), 0, s, L);
    }
    return hipGetLastError();
  }
  if (impl == 1) {
    switch (L.A) {
      case 2: return launch<2>(L, s);
      case 3: return launch<3>(L, s);
      case 4: return launch<4>(L, s);
      case 5: return launch<5>(L, s);
      case 6: return launch<6>(L, s);
      case 7: return launch<7>(L, s);
      case 8: return launch<8>(L, s);
      default: break;
    }
  }
  if (L.A <= 12) return launch_lanes_pre<12>(L, s);
  if (L.A <= 16) return launch_lanes_pre<16>(L, s);
  if (L.A <= 24) return launch_lanes_pre<24>(L, s);
  return launch_lanes_pre<32>(L, s);
}

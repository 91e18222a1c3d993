// k_gnofix.hip — the Gnofix re-phasing loop on gfx950: one workgroup per individual.
//
// Replaces Gnomix.phase -> gnofix() with its default arguments (reference src/model.py:188-214,
// src/Gnofix/gnofix.py:58-208: check_criterion="disc_smooth", max_center_offset=0, non_lin_s=0,
// prob_comp="max", prior_switch_prob=0.5, padding=True, no naive switch) and track_switch /
// correct_phase_error (src/Gnofix/phasing.py:182-198).
//
// The loop is sequential per individual (every accepted switch changes B from window w to the end), so
// parallelism is across individuals (grid) and inside one smoother evaluation (threads):
//  * both haplotypes' float32 base probabilities live reflect-padded in LDS (global scratch when an
//    individual does not fit), so the S*A features of any row are one contiguous slice;
//  * a candidate switch = 4 rows x n_trees walks spread over the block (leaf values to LDS, then per
//    (row, class) an IN-ORDER float32 sum — bit-identical to the sequential predictor), softmax, max;
//  * an accepted switch swaps the two padded strips from w on, flips the per-window parity, swaps the labels of
//    rows that only see windows >= w and re-evaluates the <= S+1 rows per haplotype whose sliding window
//    straddles w (exactly what a full smoother.predict(B) would return, at ~1/5 of the work);
//  * convergence (gnofix.py:108-113 compares whole X_m vectors) is tracked as a per-window signature
//    parity & (block of SNPs differs between the two haplotypes), which is equal iff the X_m vectors are;
//  * SNPs are swapped once at the end from the final parity (correct_phase_error applied cumulatively).
#include "gnx_internal.h"

namespace {

constexpr int THREADS = 256;

__device__ __forceinline__ int slide_src(int j, int W, int pad) {
  if (j < pad) return pad - 1 - j;
  if (j < pad + W) return j - pad;
  return W - 1 - (j - pad - W);
}

__device__ __forceinline__ void softmax_row(float* m, int A) {  // xgboost Softmax, in place
  float wmax = m[0];
  for (int a = 1; a < A; ++a) wmax = fmaxf(m[a], wmax);
  double wsum = 0.0;
  for (int a = 0; a < A; ++a) { m[a] = (float)exp((double)(m[a] - wmax)); wsum += (double)m[a]; }
  const float fs = (float)wsum;
  for (int a = 0; a < A; ++a) m[a] = m[a] / fs;
}

__global__ __launch_bounds__(THREADS) void k_gnofix(GnofixLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int W = L.W, A = L.A, S = L.S, pad = (S + 1) / 2, half = (S - 1) / 2;
  const int Wp = W + 2 * pad, F = S * A, D = L.d.D, NT = L.d.n_trees, NWD = (W + 31) / 32;
  const int tid = threadIdx.x;
  const int64_t ind = blockIdx.x;

  // ---- carve LDS ----
  size_t off = 0;
  auto carve = [&](size_t bytes) { uint8_t* p = lds + off; off += (bytes + 15) & ~(size_t)15; return p; };
  float* bp = L.bp_in_lds ? reinterpret_cast<float*>(carve((size_t)2 * Wp * A * 4))
                          : L.bp_scratch + (size_t)ind * 2 * Wp * A;   // [2][Wp][A]
  float* swrows = reinterpret_cast<float*>(carve((size_t)2 * F * 4));                // switched rows m', p'
  float* leafbuf = reinterpret_cast<float*>(carve((size_t)4 * NT * 4));              // [4][NT]
  float* marg = reinterpret_cast<float*>(carve((size_t)2 * (S + 2) * A * 4));        // margins of re-evaluated rows
  uint8_t* Y = carve((size_t)2 * W);                                                 // labels [2][W]
  uint32_t* par = reinterpret_cast<uint32_t*>(carve((size_t)NWD * 4));               // switch parity per window
  uint32_t* dif = reinterpret_cast<uint32_t*>(carve((size_t)NWD * 4));               // SNP block differs m vs p
  int* relist = reinterpret_cast<int*>(carve((size_t)W * 4));                        // rows to re-evaluate
  int* flags = reinterpret_cast<int*>(carve(64));                                    // [0]=accept [1]=converged [2]=n_re
  float* rowprob = reinterpret_cast<float*>(carve(64));
  uint32_t* hist = L.hist + (size_t)ind * L.max_it * NWD;

  int8_t* Xm = L.X + (2 * ind) * L.ldx;
  int8_t* Xp = Xm + L.ldx;
  const int64_t C = L.C;
  const int64_t ws = C / W;  // gnofix.py:74 window_size = len(M)//W

  // ---- load: padded float32 strips, initial labels, per-window SNP difference ----
  for (int e = tid; e < 2 * Wp * A; e += THREADS) {
    const int h = e / (Wp * A), r = e - h * Wp * A;
    const int j = r / A, a = r - j * A;
    bp[e] = (float)L.B[(((size_t)(2 * ind + h)) * W + slide_src(j, W, pad)) * A + a];
  }
  for (int e = tid; e < 2 * W; e += THREADS) Y[e] = (uint8_t)L.Y0[(size_t)2 * ind * W + e];
  for (int e = tid; e < NWD; e += THREADS) { par[e] = 0; dif[e] = 0; }
  __syncthreads();
  for (int u = tid; u < W; u += THREADS) {
    const int64_t j0 = (int64_t)u * ws, j1 = (u == W - 1) ? C : j0 + ws;
    bool d = false;
    for (int64_t j = j0; j < j1 && !d; ++j) d = Xm[j] != Xp[j];
    if (d) atomicOr(&dif[u >> 5], 1u << (u & 31));
  }
  __syncthreads();

  auto row_ptr = [&](int h, int w) -> const float* { return bp + ((size_t)h * Wp + w) * A; };  // features of row (h,w)

  int n_switch = 0;
  for (int it = 0; it < L.max_it; ++it) {
    // ---- convergence: has this X_m been seen at the start of an earlier sweep? (gnofix.py:108-113) ----
    if (tid == 0) {
      int conv = 0;
      for (int k = 0; k < it && !conv; ++k) {
        bool same = true;
        for (int q = 0; q < NWD && same; ++q) same = hist[(size_t)k * NWD + q] == (par[q] & dif[q]);
        conv = same;
      }
      flags[1] = conv;
      if (!conv) for (int q = 0; q < NWD; ++q) hist[(size_t)it * NWD + q] = par[q] & dif[q];
    }
    __syncthreads();
    if (flags[1]) break;

    for (int w = 1; w < W; ++w) {
      // check(): "disc_smooth" (gnofix.py:32) — uniform across the block
      if (Y[w] == Y[w - 1] && Y[W + w] == Y[W + w - 1]) continue;
      const int center = min(max(w, half), W - 1 - half);
      const int lo = center - half;  // scope = windows [lo, lo+S)
      // switched rows: m' = [B0[lo:w], B1[w:hi]], p' = [B1[lo:w], B0[w:hi]]   (gnofix.py:144-153)
      for (int e = tid; e < 2 * F; e += THREADS) {
        const int r = e / F, f = e - r * F;
        const int u = lo + f / A;
        const int h = (u < w) ? r : (1 - r);
        swrows[e] = bp[((size_t)h * Wp + pad + u) * A + (f % A)];
      }
      __syncthreads();
      // 4 rows x NT tree walks; rows 0,1 = original scope slices of the padded strips (unpadded window u
      // sits at padded index u+pad), rows 2,3 = switched copies
      for (int e = tid; e < 4 * NT; e += THREADS) {
        const int r = e & 3, t = e >> 2;
        const float* row = (r < 2) ? (bp + ((size_t)r * Wp + pad + lo) * A) : (swrows + (size_t)(r - 2) * F);
        leafbuf[(size_t)r * NT + t] = gnx_walk(L.d.packed + (size_t)t * L.d.tree_bytes, reinterpret_cast<const uint8_t*>(row), D);
      }
      __syncthreads();
      if (tid < 4 * A) {  // per (row, class): in-order float32 sum of that class's trees (class-major packing)
        const int r = tid / A, c = tid - r * A;
        float ps = 0.f;
        for (int t = L.class_tree0[c]; t < L.class_tree0[c + 1]; ++t) ps += leafbuf[(size_t)r * NT + t];
        marg[r * A + c] = L.d.base_score + ps;
      }
      __syncthreads();
      if (tid < 4) {
        softmax_row(marg + tid * A, A);
        float mx = marg[tid * A];
        for (int a = 1; a < A; ++a) mx = fmaxf(mx, marg[tid * A + a]);
        rowprob[tid] = mx;
      }
      __syncthreads();
      if (tid == 0) {
        const float p_orig = fmaxf(rowprob[0], rowprob[1]);   // prob_comp="max" over hap and ancestry
        const float p_sw = fmaxf(rowprob[2], rowprob[3]);
        flags[0] = (p_sw * 0.5f > p_orig * 0.5f) ? 1 : 0;     // prior_switch_prob = 0.5 (gnofix.py:171)
      }
      __syncthreads();
      if (!flags[0]) continue;

      // ---- accept: swap the strips from window w on (incl. reflected pads), flip parity, relabel ----
      ++n_switch;
      for (int e = tid; e < Wp * A; e += THREADS) {
        const int j = e / A;
        if (slide_src(j, W, pad) >= w) {
          const float t0 = bp[e], t1 = bp[(size_t)Wp * A + e];
          bp[e] = t1;
          bp[(size_t)Wp * A + e] = t0;
        }
      }
      for (int q = tid; q < NWD; q += THREADS) {
        const int b0 = q * 32;
        uint32_t m = 0;
        if (w <= b0) m = 0xffffffffu;
        else if (w < b0 + 32) m = 0xffffffffu << (w - b0);
        par[q] ^= m;
      }
      __syncthreads();
      // rows whose sliding window only sees windows >= w: the two haplotypes' rows are exchanged
      // rows that see windows on both sides of w: re-evaluate.  Row w' sees unpadded windows
      // {slide_src(w'+s)} = [max(0,w'-pad) .. min(W-1,w'+S-1-pad)] plus reflections that stay inside it
      // except at the edges, where the reflected part can reach further: handled by the explicit min/max.
      int n_re = 0;
      // (block-uniform scan; W iterations of trivial work per thread would be wasteful, so thread 0 lists rows)
      if (tid == 0) {
        for (int wr = 0; wr < W; ++wr) {
          int mn = W, mx = -1;
          // sources: j = wr .. wr+S-1
          const int j0 = wr, j1 = wr + S - 1;
          // interior part
          const int a0 = max(j0, pad), a1 = min(j1, pad + W - 1);
          if (a0 <= a1) { mn = min(mn, a0 - pad); mx = max(mx, a1 - pad); }
          if (j0 < pad) { const int b1 = min(j1, pad - 1); mn = min(mn, pad - 1 - b1); mx = max(mx, pad - 1 - j0); }
          if (j1 >= pad + W) { const int b0 = max(j0, pad + W); mn = min(mn, W - 1 - (j1 - pad - W)); mx = max(mx, W - 1 - (b0 - pad - W)); }
          if (mn >= w) {  // pure swap
            const uint8_t t0 = Y[wr];
            Y[wr] = Y[W + wr];
            Y[W + wr] = t0;
          } else if (mx >= w) {
            relist[n_re++] = wr;
          }
        }
        flags[2] = n_re;
      }
      __syncthreads();
      n_re = flags[2];
      // re-evaluate rows (h, relist[k]): thread = (row, class), in-order sum over the class's trees
      for (int base = 0; base < 2 * n_re; base += 2 * (S + 2)) {
        const int nrow = min(2 * (S + 2), 2 * n_re - base);
        for (int e = tid; e < nrow * A; e += THREADS) {
          const int rr = e / A, c = e - rr * A;
          const int k = (base + rr) >> 1, h = (base + rr) & 1;
          const float* row = row_ptr(h, relist[k]);
          float ps = 0.f;
          for (int t = L.class_tree0[c]; t < L.class_tree0[c + 1]; ++t)
            ps += gnx_walk(L.d.packed + (size_t)t * L.d.tree_bytes, reinterpret_cast<const uint8_t*>(row), D);
          marg[rr * A + c] = L.d.base_score + ps;
        }
        __syncthreads();
        for (int rr = tid; rr < nrow; rr += THREADS) {
          softmax_row(marg + rr * A, A);
          int best = 0;
          float bv = marg[rr * A];
          for (int a = 1; a < A; ++a) if (marg[rr * A + a] > bv) { bv = marg[rr * A + a]; best = a; }
          const int k = (base + rr) >> 1, h = (base + rr) & 1;
          Y[h * W + relist[k]] = (uint8_t)best;
        }
        __syncthreads();
      }
    }
  }

  // ---- outputs: labels, switch count, SNP swap from the final parity (phasing.py:188-198) ----
  for (int e = tid; e < 2 * W; e += THREADS) L.Yout[(size_t)2 * ind * W + e] = Y[e];
  if (tid == 0 && L.n_switches) L.n_switches[ind] = n_switch;
  for (int64_t j = tid; j < C; j += THREADS) {
    int64_t u = j / ws;
    if (u > W - 1) u = W - 1;
    if ((par[u >> 5] >> (u & 31)) & 1u) {
      const int8_t t0 = Xm[j];
      Xm[j] = Xp[j];
      Xp[j] = t0;
    }
  }
}

}  // namespace

size_t gnx_gnofix_lds_bytes(int W, int A, int S, int n_trees, bool bp_in_lds) {
  const int pad = (S + 1) / 2, Wp = W + 2 * pad, F = S * A, NWD = (W + 31) / 32;
  auto r16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
  size_t t = 0;
  if (bp_in_lds) t += r16((size_t)2 * Wp * A * 4);
  t += r16((size_t)2 * F * 4) + r16((size_t)4 * n_trees * 4) + r16((size_t)2 * (S + 2) * A * 4) + r16((size_t)2 * W) +
       2 * r16((size_t)NWD * 4) + r16((size_t)W * 4) + 64 + 64;
  return t;
}

hipError_t gnx_launch_gnofix(const GnofixLaunch& L, int64_t n_ind, hipStream_t s) {
  if (n_ind <= 0) return hipSuccess;
  const size_t lds = gnx_gnofix_lds_bytes(L.W, L.A, L.S, L.d.n_trees, L.bp_in_lds);
  GNX_LDS_OPTIN(lds, k_gnofix);
  hipLaunchKernelGGL(k_gnofix, dim3((unsigned)n_ind), dim3(THREADS), lds, s, L);
  return hipGetLastError();
}

// k_gnofix_f32.hip — the Gnofix re-phasing loop on float32 strips: the FALLBACK of k_gnofix.hip (rounds 1-3's kernel).
// Reached when the ensemble has no rank copy (more than 65 000 distinct thresholds, or GNX_SMOOTH_IMPL=f32 at model load) or when
// GNX_GNOFIX_IMPL=f32 asks for it (the parity tests run both).  One workgroup per individual.
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
// Round 2: nothing on the individual's critical path is left to one thread or to a window-by-window scan:
//  * the next window whose labels change is found by ballot, THREADS windows at a time (the reference's `for w in range(1, W)`
//    only ever acts on those);
//  * a candidate's 4 x n_trees walks go NWALK_C per thread side by side (chains of dependent L2-latency loads: the 1200 trees,
//    230 KB, stay in global memory), its softmax one (row, class) per lane;
//  * after an accepted switch all threads classify the rows (swap / re-evaluate list by LDS atomic); the re-evaluation goes class
//    by class with the class's trees staged in LDS, P lanes per row splitting them and handing the running float32 sum down the
//    lanes in tree order (bit-identical to the sequential predictor);
//  * the per-window "SNP blocks differ" flags stop at the first difference; the final SNP swap touches only windows of odd parity,
//    one byte range per run of such windows, unaligned 16-byte pieces;
//  * the convergence history is compared one past sweep per thread.
// -DGNX_GNOFIX_CLOCKS turns n_switches into per-phase clock counts (scripts/dev/gnofix_phases.py).
#include "gnx_internal.h"
#include "gnx_exp.h"

namespace {

constexpr int THREADS = 512;   // one workgroup per CU (the strips fill its LDS); 8 waves with up to 256 VGPRs each
constexpr int NWAVES = THREADS / 64;

__device__ __forceinline__ int slide_src(int j, int W, int pad) {
  if (j < pad) return pad - 1 - j;
  if (j < pad + W) return j - pad;
  return W - 1 - (j - pad - W);
}

// NW independent walks side by side (generic pointers: trees in global memory, rows in LDS or global); same arithmetic as gnx_walk
template <int NW>
__device__ __forceinline__ void walk_n(const uint8_t* const (&tb)[NW], const uint8_t* const (&row)[NW], int D, float (&out)[NW]) {
  const uint32_t half = 1u << (D - 1);
  uint32_t j[NW];
#pragma unroll
  for (int k = 0; k < NW; ++k) j[k] = 1;
  for (int d = 0; d < D - 1; ++d) {
    uint2 nd[NW];
    float fv[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) __builtin_memcpy(&nd[k], tb[k] + half * 16 + (j[k] - 1) * 8, 8);
#pragma unroll
    for (int k = 0; k < NW; ++k) __builtin_memcpy(&fv[k], row[k] + nd[k].x, 4);
#pragma unroll
    for (int k = 0; k < NW; ++k) j[k] = 2 * j[k] + ((fv[k] < __uint_as_float(nd[k].y)) ? 0u : 1u);
  }
  uint4 n4[NW];
  float fv[NW];
#pragma unroll
  for (int k = 0; k < NW; ++k) __builtin_memcpy(&n4[k], tb[k] + (j[k] - half) * 16, 16);
#pragma unroll
  for (int k = 0; k < NW; ++k) __builtin_memcpy(&fv[k], row[k] + n4[k].x, 4);
#pragma unroll
  for (int k = 0; k < NW; ++k) out[k] = (fv[k] < __uint_as_float(n4[k].y)) ? __uint_as_float(n4[k].z) : __uint_as_float(n4[k].w);
}

// ---- 16 SNPs of a haplotype row at any byte address (gfx950 global memory takes unaligned dwordx4; the rows of an individual are
// ldx bytes apart with no alignment promise) ----
struct __attribute__((packed, aligned(1))) snp16 { uint32_t x, y, z, w; };
__device__ __forceinline__ snp16 ld16(const int8_t* p) { snp16 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st16(int8_t* p, const snp16& v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ bool neq(const snp16& a, const snp16& b) { return ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) != 0; }

__host__ __device__ inline size_t gnofix_swrows_bytes(int S, int A, bool strips_in_lds) {
  const size_t a = (size_t)2 * (S + 2) * A * 4, b = (size_t)(strips_in_lds ? 2 : 4) * S * A * 4;
  return a > b ? a : b;
}
__host__ __device__ inline size_t gnofix_leafbuf_bytes(int n_trees, int tree_bytes) {
  const size_t a = (size_t)4 * n_trees * 4, b = (size_t)8 * tree_bytes;  // at least 8 staged trees
  return a > b ? a : b;
}

constexpr int NWALK = 4;   // re-evaluation: trees of one (row, class) walked side by side
constexpr int RE_PER = 40; // re-evaluation: most trees one lane walks per staged chunk (their leaves stay in registers)
constexpr int NWALK_C = 10; // candidate: 4 rows x NT walks over the block (4 x 1200 = 4800 <= 10 x 512)

template <bool SL>  // SL: the two padded strips live in LDS (else in global scratch: very long chromosomes)
__global__ __launch_bounds__(THREADS) void k_gnofix(GnofixLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int W = L.W, A = L.A, S = L.S, pad = (S + 1) / 2, half = (S - 1) / 2;
  const int Wp = W + 2 * pad, F = S * A, D = L.d.D, NT = L.d.n_trees, NWD = (W + 31) / 32;
  const int tid = threadIdx.x;
  const int64_t ind = blockIdx.x;

  // ---- carve LDS ----
  size_t off = 0;
  auto carve = [&](size_t bytes) { uint8_t* p = lds + off; off += (bytes + 15) & ~(size_t)15; return p; };
  float* bp;  // [2][Wp][A]
  if constexpr (SL) bp = reinterpret_cast<float*>(carve((size_t)2 * Wp * A * 4));
  else bp = L.bp_scratch + (size_t)ind * 2 * Wp * A;
  // candidate rows [2][F] (SL: the switched pair; the original pair is read from the strips) or [4][F] (strips in global memory:
  // original pair + switched pair); also the exp() of re-evaluated rows
  float* swrows = reinterpret_cast<float*>(carve(gnofix_swrows_bytes(S, A, SL)));
  // strips in global memory: the slice of both strips that a group of re-evaluated rows reads, [2][SEGW][A]
  const int SEGW = 2 * S + 2;
  float* seg = SL ? nullptr : reinterpret_cast<float*>(carve((size_t)2 * SEGW * A * 4));
  float* leafbuf = reinterpret_cast<float*>(carve(gnofix_leafbuf_bytes(NT, L.d.tree_bytes)));  // [4][NT] leaves / staged trees
  float* marg = reinterpret_cast<float*>(carve((size_t)2 * (S + 2) * A * 4));        // margins of re-evaluated rows
  uint8_t* Y = carve((size_t)2 * W + 16);                                                 // labels [2][W]
  uint32_t* par = reinterpret_cast<uint32_t*>(carve((size_t)NWD * 4));               // switch parity per window
  uint32_t* dif = reinterpret_cast<uint32_t*>(carve((size_t)NWD * 4));               // SNP block differs m vs p
  int* flags = reinterpret_cast<int*>(carve(128));                                   // [0]=accept [1]=converged [2],[3]=rows to re-evaluate [r0,r1) [4..4+NWAVES)=first change per wave
  uint32_t* hist = L.hist + (size_t)ind * L.max_it * NWD;

#ifdef GNX_GNOFIX_CLOCKS  // development aid: per-phase shader clocks, individual i reports phase (i & 7) in n_switches (units of 64 clocks)
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
  const long long tstart = tprev;
#define TICK(i) { __syncthreads(); const long long tn = clock64(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define TICK(i)
#endif
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
  TICK(0)
  const int wv = tid >> 6, ln = tid & 63;
  // "does this window's SNP block differ between the two haplotypes": four lanes per window, 64 bytes per step, stopping at the first
  // difference (heterozygous sites are dense, so a window is normally decided by its first step; identical blocks read it all)
  for (int u0 = wv * 16; u0 < W; u0 += NWAVES * 16) {
    const int u = u0 + (ln >> 2), q = ln & 3;
    const bool live = u < W;
    const int64_t j0 = live ? (int64_t)u * ws : 0, j1 = !live ? 0 : (u == W - 1) ? C : j0 + ws;
    bool d = false;
    for (int64_t j = j0; ; j += 64) {
      const int64_t a0 = j + q * 16, a1 = min(a0 + 16, j1);
      if (!d && a0 < j1) {
        if (a1 - a0 == 16) {
          d = neq(ld16(Xm + a0), ld16(Xp + a0));
        } else {
          for (int64_t i = a0; i < a1; ++i) d |= Xm[i] != Xp[i];
        }
      }
      const unsigned long long bal = __ballot(d);
      d = ((bal >> (ln & ~3)) & 0xfull) != 0;  // the window's four lanes agree
      if (__ballot(!d && j + 64 < j1) == 0) break;
    }
    if (d && q == 0) atomicOr(&dif[u >> 5], 1u << (u & 31));
  }
  __syncthreads();
  TICK(1)

  int n_switch = 0;
  for (int it = 0; it < L.max_it; ++it) {
    // ---- convergence: has this X_m been seen at the start of an earlier sweep? (gnofix.py:108-113) ----
    if (tid == 0) flags[1] = 0;
    __syncthreads();
    for (int k = tid; k < it; k += THREADS) {  // one past sweep per thread
      bool same = true;
      for (int q = 0; q < NWD && same; ++q) same = hist[(size_t)k * NWD + q] == (par[q] & dif[q]);
      if (same) atomicOr(&flags[1], 1);
    }
    for (int q = tid; q < NWD; q += THREADS) hist[(size_t)it * NWD + q] = par[q] & dif[q];  // (harmless when converged: never read again)
    __syncthreads();
    if (flags[1]) break;

    // check(): "disc_smooth" (gnofix.py:32): the reference walks w = 1 .. W-1 and acts only where a label changes; the next such
    // window at or after `from` is found THREADS windows at a time (labels of later windows may change while the sweep advances, so
    // the search restarts after every candidate)
    auto next_change = [&](int from) -> int {
      for (int base = from; base < W; base += THREADS) {
        const int wq = base + tid;
        const bool hit = wq < W && (Y[wq] != Y[wq - 1] || Y[W + wq] != Y[W + wq - 1]);
        const unsigned long long bal = __ballot(hit);
        if ((tid & 63) == 0) flags[4 + (tid >> 6)] = bal ? base + (tid & ~63) + __builtin_ctzll(bal) : W;
        __syncthreads();
        int first = W;
#pragma unroll
        for (int k = 0; k < NWAVES; ++k) first = min(first, flags[4 + k]);
        __syncthreads();
        if (first < W) return first;
      }
      return W;
    };
    for (int w = next_change(1); w < W; w = next_change(w + 1)) {
      TICK(2)
      const int center = min(max(w, half), W - 1 - half);
      const int lo = center - half;  // scope = windows [lo, lo+S)
      // switched rows: m' = [B0[lo:w], B1[w:hi]], p' = [B1[lo:w], B0[w:hi]]   (gnofix.py:144-153)
      constexpr int R0 = SL ? 2 : 0;  // first row kept in swrows
      for (int e = tid; e < (4 - R0) * F; e += THREADS) {
        const int r = R0 + e / F, f = e % F;
        const int u = lo + f / A;
        const int h = (r < 2) ? r : (u < w) ? (r - 2) : (3 - r);
        swrows[e] = bp[((size_t)h * Wp + pad + u) * A + (f % A)];
      }
      __syncthreads();
      // 4 rows x NT tree walks; rows 0,1 = original scope slices of the padded strips (unpadded window u
      // sits at padded index u+pad; copied to LDS when the strips are in global memory), rows 2,3 = switched copies
      for (int e0 = tid; e0 < 4 * NT; e0 += NWALK_C * THREADS) {
        const uint8_t* tb[NWALK_C];
        const uint8_t* rw[NWALK_C];
        float lf[NWALK_C];
#pragma unroll
        for (int k = 0; k < NWALK_C; ++k) {
          const int e = min(e0 + k * THREADS, 4 * NT - 1);  // clamped: the tail repeats the last walk and drops it
          const int r = e & 3, t = e >> 2;
          tb[k] = L.d.packed + (size_t)t * L.d.tree_bytes;
          if constexpr (SL) rw[k] = reinterpret_cast<const uint8_t*>((r < 2) ? (bp + ((size_t)r * Wp + pad + lo) * A) : (swrows + (size_t)(r - 2) * F));
          else rw[k] = reinterpret_cast<const uint8_t*>(swrows + (size_t)r * F);
        }
        walk_n<NWALK_C>(tb, rw, D, lf);
#pragma unroll
        for (int k = 0; k < NWALK_C; ++k) {
          const int e = e0 + k * THREADS;
          if (e < 4 * NT) leafbuf[(size_t)(e & 3) * NT + (e >> 2)] = lf[k];
        }
      }
      __syncthreads();
      if (tid < 4 * A) {  // per (row, class): in-order float32 sum of that class's trees (class-major packing)
        const int r = tid / A, c = tid - r * A;
        float ps = 0.f;
        const int t1 = L.class_tree0[c + 1];
        int t = L.class_tree0[c];
        for (; t + 8 <= t1; t += 8) {  // loads first, then the adds in tree order
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = leafbuf[(size_t)r * NT + t + k];
#pragma unroll
          for (int k = 0; k < 8; ++k) ps += v[k];
        }
        for (; t < t1; ++t) ps += leafbuf[(size_t)r * NT + t];
        marg[r * A + c] = L.d.base_score + ps;
      }
      __syncthreads();
      // xgboost Softmax of the 4 rows, one (row, class) per lane: exp((double)(m - max)) -> float, in-order double sum, divide
      float* ex = marg + 4 * A;
      if (tid < 4 * A) {
        const int r = tid / A;
        float wmax = marg[r * A];
        for (int a = 1; a < A; ++a) wmax = fmaxf(marg[r * A + a], wmax);
        ex[tid] = gnx_softmax_exp(marg[tid] - wmax);
      }
      __syncthreads();
      if (tid < 64) {  // lanes 0..3 = rows; the decision is taken by lane 0
        float mx = 0.f;
        if (tid < 4) {
          double wsum = 0.0;
          for (int a = 0; a < A; ++a) wsum += (double)ex[tid * A + a];
          const float fs = (float)wsum;
          mx = ex[tid * A] / fs;
          for (int a = 1; a < A; ++a) mx = fmaxf(mx, ex[tid * A + a] / fs);
        }
        const float p_orig = fmaxf(__shfl(mx, 0), __shfl(mx, 1));   // prob_comp="max" over hap and ancestry
        const float p_sw = fmaxf(__shfl(mx, 2), __shfl(mx, 3));
        if (tid == 0) flags[0] = (p_sw * 0.5f > p_orig * 0.5f) ? 1 : 0;  // prior_switch_prob = 0.5 (gnofix.py:171)
      }
      __syncthreads();
      TICK(3)
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
      // (The rows to re-evaluate form one contiguous range [r0, r1): re-evaluating a row of that range that needed nothing, or one
      // that was also swapped, just recomputes its label from the current strips.)
      if (tid == 0) { flags[2] = W; flags[3] = 0; }
      __syncthreads();
      for (int wr = tid; wr < W; wr += THREADS) {  // every row classified by its own thread
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
          atomicMin(&flags[2], wr);
          atomicMax(&flags[3], wr + 1);
        }
      }
      __syncthreads();
      const int r0 = flags[2], r1 = flags[3];
      TICK(4)
      // re-evaluate rows (h, r0..r1-1), S+2 windows at a time, class by class: the class's trees are staged in LDS (leafbuf is idle here), P lanes share a
      // row and split the staged trees, and the row's sum is taken in tree order by handing the running sum down those lanes
      const int tree_bytes = L.d.tree_bytes, cap = min((int)(gnofix_leafbuf_bytes(NT, tree_bytes) / tree_bytes), 64 * RE_PER);
      uint8_t* stage_t = reinterpret_cast<uint8_t*>(leafbuf);
      for (int gb = r0; gb < r1; gb += S + 2) {
        const int nwin = min(S + 2, r1 - gb), nrow = 2 * nwin;  // row rr = (window gb + (rr >> 1), haplotype rr & 1)
        if constexpr (!SL) {  // the rows read padded windows [gb, gb + nwin + S - 1) of both strips
          const int nj = nwin + S - 1;
          for (int e = tid; e < 2 * nj * A; e += THREADS) {
            const int h = e / (nj * A), r = e - h * nj * A;
            seg[(size_t)h * SEGW * A + r] = bp[((size_t)h * Wp + gb) * A + r];
          }
        }
        for (int c = 0; c < A; ++c) {
          const int t0 = L.class_tree0[c], t1 = L.class_tree0[c + 1];
          for (int ts = t0; ts < t1; ts += cap) {
            const int n_st = min(cap, t1 - ts);
            __syncthreads();
            {
              const uint4* src = reinterpret_cast<const uint4*>(L.d.packed + (size_t)ts * tree_bytes);
              uint4* dst = reinterpret_cast<uint4*>(stage_t);
              for (int q = tid; q < n_st * (tree_bytes / 16); q += THREADS) dst[q] = src[q];
            }
            __syncthreads();
            // P lanes per row, rows never straddling a wave: the most lanes that still cover all rows in one pass of the block
            // (at least enough for a lane's share of the staged trees to fit its registers)
            int P = (n_st + RE_PER - 1) / RE_PER;
            while (P < 64 && (P + 1) * 4 <= n_st && (nrow + 64 / (P + 1) - 1) / (64 / (P + 1)) <= NWAVES) ++P;
            const int rpw = 64 / P;                      // rows per wave
            const int per = (n_st + P - 1) / P;
            const int n_pass = (nrow + rpw * NWAVES - 1) / (rpw * NWAVES);
            for (int pass = 0; pass < n_pass; ++pass) {
              const int p = ln % P, rr_ = (pass * NWAVES + wv) * rpw + ln / P;
              const bool live = ln < rpw * P && rr_ < nrow;
              const int rr = live ? rr_ : 0;
              const int k = rr >> 1, h = rr & 1;
              const uint8_t* row = reinterpret_cast<const uint8_t*>(SL ? bp + ((size_t)h * Wp + gb + k) * A : seg + ((size_t)h * SEGW + k) * A);
              const int lo_t = min(p * per, n_st), cnt = live ? min(per, n_st - lo_t) : 0;
              float lf[RE_PER];
#pragma unroll
              for (int b = 0; b < RE_PER; b += NWALK) {
                if (b < per && b < cnt) {
                  const uint8_t* tb[NWALK];
                  const uint8_t* rw[NWALK];
                  float o[NWALK];
#pragma unroll
                  for (int i = 0; i < NWALK; ++i) { tb[i] = stage_t + (size_t)min(lo_t + b + i, n_st - 1) * tree_bytes; rw[i] = row; }
                  walk_n<NWALK>(tb, rw, D, o);
#pragma unroll
                  for (int i = 0; i < NWALK; ++i) lf[b + i] = o[i];
                }
              }
              float ps = (ts == t0) ? 0.f : marg[rr * A + c];  // (only lane p == 0 uses it)
              for (int step = 0; step < P; ++step) {
                const float up = __shfl_up(ps, 1);
                if (p == step) {
                  if (step > 0) ps = up;
#pragma unroll
                  for (int i = 0; i < RE_PER; ++i) if (i < per && i < cnt) ps += lf[i];  // tree order
                }
              }
              if (live && p == P - 1) marg[rr * A + c] = ps;
            }
          }
        }
        __syncthreads();
        for (int e = tid; e < nrow * A; e += THREADS) {  // margin -> exp(margin - row max), one (row, class) per thread
          const int rr = e / A;
          float wmax = L.d.base_score + marg[rr * A];
          for (int a = 1; a < A; ++a) wmax = fmaxf(L.d.base_score + marg[rr * A + a], wmax);
          swrows[e] = gnx_softmax_exp((L.d.base_score + marg[e]) - wmax);  // (swrows is idle here)
        }
        __syncthreads();
        for (int rr = tid; rr < nrow; rr += THREADS) {
          const float* m = swrows + rr * A;
          double wsum = 0.0;
          for (int a = 0; a < A; ++a) wsum += (double)m[a];
          const float fs = (float)wsum;
          int best = 0;
          float bv = m[0] / fs;
          for (int a = 1; a < A; ++a) { const float v = m[a] / fs; if (v > bv) { bv = v; best = a; } }
          Y[(rr & 1) * W + gb + (rr >> 1)] = (uint8_t)best;
        }
        __syncthreads();
      }
      TICK(5)
    }
    TICK(2)
  }

  // ---- outputs: labels, switch count, SNP swap from the final parity (phasing.py:188-198) ----
  for (int e = tid; e < 2 * W; e += THREADS) L.Yout[(size_t)2 * ind * W + e] = Y[e];
  TICK(2)
  // Only windows of odd parity are touched; they come in a few long runs (every accepted switch flips "from w to the end"), so the
  // block sweeps each run as one byte range, four 16-byte pieces per thread in flight for each row.
  __syncthreads();                                  // Y is free from here: it takes the run starts
  int* runs = reinterpret_cast<int*>(Y);            // <= W/2 starts, 2W bytes
  if (tid == 0) flags[2] = 0;
  __syncthreads();
  auto flipped = [&](int u) { return ((par[u >> 5] >> (u & 31)) & 1u) != 0; };
  for (int u = tid; u < W; u += THREADS)
    if (flipped(u) && (u == 0 || !flipped(u - 1))) runs[atomicAdd(&flags[2], 1)] = u;
  __syncthreads();
  const int n_runs = flags[2];
  for (int r = 0; r < n_runs; ++r) {
    const int ua = runs[r];
    int ub = ua + 1;
    while (ub < W && flipped(ub)) ++ub;             // (block-uniform scan; runs are few)
    const int64_t j0 = (int64_t)ua * ws, j1 = (ub == W) ? C : (int64_t)ub * ws;
    const int64_t n16 = (j1 - j0) / 16;
    for (int64_t k0 = 0; k0 < n16; k0 += 4 * THREADS) {
      snp16 xa[4], xb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t k = k0 + q * THREADS + tid;
        if (k < n16) { xa[q] = ld16(Xm + j0 + k * 16); xb[q] = ld16(Xp + j0 + k * 16); }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t k = k0 + q * THREADS + tid;
        if (k < n16) { st16(Xm + j0 + k * 16, xb[q]); st16(Xp + j0 + k * 16, xa[q]); }
      }
    }
    for (int64_t j = j0 + n16 * 16 + tid; j < j1; j += THREADS) {
      const int8_t t0 = Xm[j];
      Xm[j] = Xp[j];
      Xp[j] = t0;
    }
  }
  TICK(6)
#ifdef GNX_GNOFIX_CLOCKS
  tacc[7] = tprev - tstart;
  if (tid == 0 && L.n_switches) L.n_switches[ind] = (int)(tacc[ind & 7] >> 6);
#else
  if (tid == 0 && L.n_switches) L.n_switches[ind] = n_switch;
#endif
}

}  // namespace

size_t gnx_gnofix_f32_lds_bytes(int W, int A, int S, int n_trees, int tree_bytes, bool bp_in_lds) {
  const int pad = (S + 1) / 2, Wp = W + 2 * pad, NWD = (W + 31) / 32;
  auto r16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
  size_t t = 0;
  if (bp_in_lds) t += r16((size_t)2 * Wp * A * 4);
  t += r16(gnofix_swrows_bytes(S, A, bp_in_lds)) + (bp_in_lds ? 0 : r16((size_t)2 * (2 * S + 2) * A * 4)) +
       r16(gnofix_leafbuf_bytes(n_trees, tree_bytes)) + r16((size_t)2 * (S + 2) * A * 4) + r16((size_t)2 * W + 16) +
       2 * r16((size_t)NWD * 4) + 128;
  return t;
}

hipError_t gnx_launch_gnofix_f32(const GnofixLaunch& L, int64_t n_ind, hipStream_t s) {
  if (n_ind <= 0) return hipSuccess;
  const size_t lds = gnx_gnofix_f32_lds_bytes(L.W, L.A, L.S, L.d.n_trees, L.d.tree_bytes, L.bp_in_lds != 0);
  if (L.bp_in_lds) {
    GNX_LDS_OPTIN(lds, k_gnofix<true>);
    hipLaunchKernelGGL(k_gnofix<true>, dim3((unsigned)n_ind), dim3(THREADS), lds, s, L);
  } else {
    GNX_LDS_OPTIN(lds, k_gnofix<false>);
    hipLaunchKernelGGL(k_gnofix<false>, dim3((unsigned)n_ind), dim3(THREADS), lds, s, L);
  }
  return hipGetLastError();
}

// k_base_forest2.hip — the boosted-tree window base (XGBBase, reference src/Base/models.py:24-35) with TWO blocks per CU.
//
// Same contract, tile and walk as k_base_forest.hip (2-bit SNP fields in an LDS ring anchored on the padded chromosome
// coordinate, mask nodes, margins per class in float32 in tree order, softmax / sigmoid as the restated predictor).  What
// round 2's kernel paid for: one 256-haplotype tile filled the LDS, so ONE block (two waves per SIMD) had to hide the HBM
// latency of its own staging with a hand-rolled register prefetch — 64 raw + 32 squeezed + 16 tree registers next to 8 walk
// chains: 256 VGPRs + 54 spilled + 148 SGPR spills, 312 B of scratch per lane (scripts/scratch_audit.py) — and rewrote every
// node word for the window and block shape on its way into LDS.  Here
//   * a tile is 128 haplotypes (64 KB of ring) and only the NODE words of the window's trees go to LDS (64 B per depth-4 tree;
//     the leaves, read once per walk, come from global memory / L1): 76 KB per block -> two blocks per CU, 16 waves, and the
//     staging of one block runs under the walks of the other.  Staging is synchronous and simple: no prefetch registers.
//   * node words are baked per window at model load (ring slot byte offset << 15 | 2 * field << 4 | right-mask: the word the
//     walk consumes), a window's nodes are a straight copy.
//   * four wave groups share a tile and split the classes (margins meet in LDS), group 0 writes the probabilities.
// The random-forest base runs here too (RF = true, round 3 close): its float64 class rows are summed in estimator order PER CLASS,
// so the four wave groups split the classes (each walks all 20 trees: 80 walks per haplotype and window against the boosted
// base's 140) — 2.84 ms in k_base_forest (one wave per SIMD, 506 VGPRs, nothing busy) -> see the launcher's note.
#include "gnx_internal.h"

namespace {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#ifndef GNX_F2_T
#define GNX_F2_T 128
#endif
constexpr int T2 = GNX_F2_T;   // haplotypes per tile (128: two blocks per CU at chr22's window width; 64: three)
constexpr int H2 = 4;     // wave groups per tile
constexpr int NTHR2 = T2 * H2, NWV2 = NTHR2 / 64;
constexpr int WPS2 = T2 == 128 ? 4 : 3;  // waves per SIMD the kernel is compiled for (HIP's second launch bound): 128 / 170 VGPRs
#ifndef GNX_F2_TPW
#define GNX_F2_TPW 10
#endif
constexpr int TPW2 = GNX_F2_TPW;  // trees walked side by side per lane (XGBBase: 20 rounds -> 20 trees per class = two batches)
constexpr int LB2 = 4;    // 16-byte loads in flight per lane while staging (16 waves per CU: 64 KB in flight; more spills registers)

__device__ __forceinline__ int64_t pad_src2(int64_t p, int64_t C, int64_t ctx) {
  if (p < ctx) return ctx - 1 - p;
  if (p < ctx + C) return p - ctx;
  return C - 1 - (p - ctx - C);
}
__device__ __forceinline__ uint32_t step2(uint32_t j, uint32_t nd, uint32_t xv) {
  const uint32_t v = __builtin_amdgcn_ubfe(xv, __builtin_amdgcn_ubfe(nd, 4, 5), 2);  // the 2-bit SNP value
  return 2 * j + __builtin_amdgcn_ubfe(nd, v, 1);                                     // + 1 iff value v goes right
}

// NT trees side by side, trees t .. t + NT - 1 of the window clamped to t_last (a short last batch walks its last tree more
// than once; the caller ignores those leaves): nodes in LDS (2^D words per tree), the lane's ring column xcol, leaves from the
// loader's records in global memory.  The leaf loads are ISSUED here and awaited where the caller adds them up.
template <int D, int NT>
__device__ __forceinline__ void walk2(const uint32_t* nodes, const uint8_t* xcol, const uint8_t* rec, int tree_bytes, int t, int t_last,
                                      float* leaf) {
  uint32_t j[NT];
  int tk[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    j[k] = 1;
    tk[k] = min(t + k, t_last);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    uint32_t nd[NT], xv[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) nd[k] = nodes[(tk[k] << D) + j[k]];
#pragma unroll
    for (int k = 0; k < NT; ++k) xv[k] = *reinterpret_cast<const uint32_t*>(xcol + (nd[k] >> 15));
#pragma unroll
    for (int k = 0; k < NT; ++k) j[k] = step2(j[k], nd[k], xv[k]);
  }
#pragma unroll
  for (int k = 0; k < NT; ++k) leaf[k] = reinterpret_cast<const float*>(rec + (size_t)tk[k] * tree_bytes)[j[k]];  // leaves follow the 2^D node words
}

// the same walk for the random-forest base: the heap index of the leaf (2^D <= j < 2^(D+1)); its class row lives in global memory
template <int D, int NT>
__device__ __forceinline__ void walk2_idx(const uint32_t* nodes, const uint8_t* xcol, int t, int t_last, uint32_t* j) {
  int tk[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    j[k] = 1;
    tk[k] = min(t + k, t_last);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    uint32_t nd[NT], xv[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) nd[k] = nodes[(tk[k] << D) + j[k]];
#pragma unroll
    for (int k = 0; k < NT; ++k) xv[k] = *reinterpret_cast<const uint32_t*>(xcol + (nd[k] >> 15));
#pragma unroll
    for (int k = 0; k < NT; ++k) j[k] = step2(j[k], nd[k], xv[k]);
  }
}

__device__ __forceinline__ void window_words2(const ForestLaunch& L, int w, int64_t& g0, int64_t& g1) {
  const int64_t s = (int64_t)w * L.M, width = (w == L.W - 1) ? L.width_last : L.width;
  g0 = s >> 4;
  g1 = ((s + width - 1) >> 4) + 1;
}

// RF = true: the random-forest base (RFBase, reference src/Base/models.py:54-66) on the same tile: every wave group walks ALL trees
// of the window and adds, for ITS classes, the leaf's class-probability row (float64, global memory / L1) in estimator order —
// the order of ForestClassifier.predict_proba's sums is a per-class matter, so the classes can be split between the groups without
// changing a bit — the sums meet in LDS (float64 margins) and group 0 divides by the tree count.
constexpr int CG2 = 8;  // classes a wave group may own (A <= 4 * CG2 = 32)
template <int D, bool RF>
// (HIP reads the second launch bound as waves per SIMD: two 8-wave blocks per CU = 4, i.e. at most 128 VGPRs)
__global__ __launch_bounds__(NTHR2, WPS2) void k_base_forest2(ForestLaunch L, const uint32_t* __restrict__ nodes2) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = tid / T2, hap = tid - half * T2;
  const int A = L.A, tree_bytes = L.tree_bytes;
  const uint32_t ring = (uint32_t)L.ring;
  uint32_t* xw = reinterpret_cast<uint32_t*>(lds);                                   // [ring][T2]
  uint32_t* nlds = xw + (size_t)ring * T2;                                           // node words of the current window
  float* marg = reinterpret_cast<float*>(nlds + ((size_t)L.max_trees << D)) + hap;   // [A][T2]

  const int wa = L.w_first + blockIdx.y * L.wrun, wb = min(L.w_first + L.n_windows, wa + L.wrun);
  const int64_t blk0 = (int64_t)blockIdx.x * T2;
  const int64_t n = blk0 + hap;
  const int64_t C = L.C, ctx = L.ctx, cmax = C - 16;
  const int wsub = lane & 7, hsub = lane >> 3;  // 8 consecutive lanes along a row: 128 contiguous bytes per haplotype

  // words [ga, gb) of the tile's haplotypes -> ring.  A wave owns RPW groups of 8 haplotypes for the whole kernel (their row
  // pointers are computed once), a load instruction covers 8 haplotypes x 8 consecutive words, LB2 loads in flight per lane.
  // 16 SNP bytes become one 32-bit word with four v_dot4_u32_u8 against the weights (1, 4, 16, 64): the staging of a window costs
  // about as many VALU instructions as a third of its walks (the shift-and-or squeeze of k_base_forest cost as many as all of
  // them, so staging one block could not hide under the walks of the other).  The ring is [word slot][haplotype]: lanes that
  // store different words of ONE haplotype hit one bank (8-way conflicted stores); shapes that avoid it fetch rows in 32- or
  // 64-byte pieces and measured slower (2 x 32: 1.95 ms, 4 x 16: 1.72 ms against 1.61 ms for this 8 x 8 shape, chr22 / 10 000
  // haplotypes): the fetch pattern decides, not the store conflicts.
  constexpr int RPW = (T2 / 8) / NWV2;  // haplotype groups per wave (2)
  const int8_t* rowp[RPW];
  int hpw[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    hpw[r] = (wave * RPW + r) * 8 + hsub;
    int64_t nn = blk0 + hpw[r];
    nn = nn < L.N ? nn : L.N - 1;
    rowp[r] = L.X + nn * L.ldx;
  }
  auto squeeze16 = [](const v4u& v) -> uint32_t {
    constexpr uint32_t M = 0x03030303u, K = 0x40100401u;  // bytes (1, 4, 16, 64): b0 + 4 b1 + 16 b2 + 64 b3
    const uint32_t q0 = __builtin_amdgcn_udot4(v.x & M, K, 0u, false), q1 = __builtin_amdgcn_udot4(v.y & M, K, 0u, false);
    const uint32_t q2 = __builtin_amdgcn_udot4(v.z & M, K, 0u, false), q3 = __builtin_amdgcn_udot4(v.w & M, K, 0u, false);
    return q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
  };
  auto stage = [&](int64_t ga, int64_t gb) {
    constexpr int WB = LB2 / RPW;  // word groups (8 words each) per batch
    for (int64_t gw = ga; gw < gb; gw += 8 * WB) {
      v4u v[WB * RPW];
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int64_t g = min(gw + 8 * u + wsub, gb - 1);  // clamped: loads stay unconditional
        int64_t c0 = 16 * g - ctx;
        c0 = c0 < 0 ? 0 : (c0 > cmax ? cmax : c0);
#pragma unroll
        for (int r = 0; r < RPW; ++r) __builtin_memcpy(&v[u * RPW + r], rowp[r] + c0, 16);
      }
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int64_t g = gw + 8 * u + wsub;
        if (g >= gb) continue;
        const int64_t p0 = 16 * g;
        const bool edge = p0 < ctx || p0 + 16 > ctx + C;  // rare: the word touches the reflect padding / the row's end
        uint32_t* dst = xw + (size_t)((uint32_t)g & (ring - 1u)) * T2;  // ring: a power of two
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          uint32_t q = squeeze16(v[u * RPW + r]);
          if (edge) {
            q = 0;
            for (int b = 0; b < 16; ++b) {
              const int64_t p = p0 + b;
              if (p < C + 2 * ctx) q |= ((uint32_t)(uint8_t)rowp[r][pad_src2(p, C, ctx)] & 3u) << (2 * b);
            }
          }
          dst[hpw[r]] = q;
        }
      }
    }
  };

  int64_t g0, g1, pg1 = 0;
  for (int w = wa; w < wb; ++w) {
    window_words2(L, w, g0, g1);
    // the window's new words (all of them for the first window of the run) and its node words
    if (w == wa || !(L.flags & 4)) stage(w == wa ? g0 : (g0 > pg1 ? g0 : pg1), g1);  // (flags: development ablation switches)
    pg1 = g1;
    const int t0 = L.win_tree0[w], nt = L.win_tree0[w + 1] - t0;
    {
      const uint4* src = reinterpret_cast<const uint4*>(nodes2 + ((size_t)t0 << D));
      const int np = D >= 2 ? (nt << D) / 4 : 0;  // 16-byte pieces (a tree = 2^D words; stumps are copied word by word below)
      for (int e = tid; e < np; e += NTHR2) reinterpret_cast<uint4*>(nlds)[e] = src[e];
      for (int e = np * 4 + tid; e < (nt << D); e += NTHR2) nlds[e] = nodes2[((size_t)t0 << D) + e];
    }
    __syncthreads();  // tile + nodes of window w complete; everybody has read the previous window's margins

    const uint8_t* xcol = reinterpret_cast<const uint8_t*>(xw + hap);
    const int32_t* cls0 = L.win_class_tree0 + (size_t)w * (A + 1);  // [A+1] offsets relative to t0
    const int n_groups = (A == 2) ? 1 : A;
    const int c_lo = (half * n_groups) / H2, c_hi = ((half + 1) * n_groups) / H2;
    const uint8_t* rec0 = L.packed + (size_t)t0 * tree_bytes;
    if constexpr (RF) {
      double* margd = reinterpret_cast<double*>(nlds + ((size_t)L.max_trees << D)) + hap;  // [A][T2] float64
      const int r_lo = (half * A) / H2, r_hi = ((half + 1) * A) / H2;                      // this group's classes
      double acc[CG2];
#pragma unroll
      for (int q = 0; q < CG2; ++q) acc[q] = 0.0;
      constexpr int TPW = D <= 4 ? TPW2 : (D <= 6 ? TPW2 - 2 : TPW2 - 4);
      const double* lv0 = L.rf_leafval + (size_t)t0 * ((size_t)A << D) + r_lo;
      for (int t = 0; t < nt && !(L.flags & 1); t += TPW) {
        uint32_t jj[TPW];
        walk2_idx<D, TPW>(nlds, xcol, t, nt - 1, jj);
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          if (t + k < nt) {
            const double* v = lv0 + ((size_t)(t + k) * ((size_t)1 << D) + (jj[k] - (1u << D))) * A;
#pragma unroll
            for (int q = 0; q < CG2; ++q)
              if (r_lo + q < r_hi) acc[q] += v[q];  // estimator order, class by class
          }
        }
      }
#pragma unroll
      for (int q = 0; q < CG2; ++q)
        if (r_lo + q < r_hi) margd[(r_lo + q) * T2] = acc[q];
      __syncthreads();  // the sums of all classes are in LDS; every lane is done with window w's tile and nodes
      if (half == 0 && n < L.N) {
        const size_t o = ((size_t)n * L.W + w) * A;
        const double cnt = (double)nt;
        for (int a = 0; a < A; ++a) {
          const double p = margd[a * T2] / cnt;
          if (L.b64) L.b64[o + a] = p;
          if (L.b32) L.b32[o + a] = (float)p;
        }
      }
      continue;  // (the next iteration's first barrier comes after group 0 has read the sums: as for the margins below)
    }
    for (int c = c_lo; c < c_hi && !(L.flags & 1); ++c) {
      const int a0 = (A == 2) ? 0 : cls0[c], a1 = (A == 2) ? nt : cls0[c + 1];
      constexpr int TPW = D <= 4 ? TPW2 : (D <= 6 ? TPW2 - 2 : TPW2 - 4);
      // batches of TPW trees (10 up to depth 4, fewer for deeper trees: registers), two per trip: the leaves of a batch (global memory: an L1 / L2 round trip) are added up only
      // after the NEXT batch has been walked, in tree order all the same (A, B, A', B', ...)
      float psum = 0.f;
      float lA[TPW], lB[TPW];
      int nB = 0;
      for (int t = a0; t < a1; t += 2 * TPW) {
        const int nA = min(TPW, a1 - t);
        walk2<D, TPW>(nlds, xcol, rec0, tree_bytes, t, a1 - 1, lA);
        if (nB) {
#pragma unroll
          for (int k = 0; k < TPW; ++k)
            if (k < nB) psum += lB[k];
        }
        nB = max(0, min(TPW, a1 - (t + TPW)));
        if (nB) walk2<D, TPW>(nlds, xcol, rec0, tree_bytes, t + TPW, a1 - 1, lB);
#pragma unroll
        for (int k = 0; k < TPW; ++k)
          if (k < nA) psum += lA[k];
      }
      if (nB) {
#pragma unroll
        for (int k = 0; k < TPW; ++k)
          if (k < nB) psum += lB[k];
      }
      marg[c * T2] = psum;
    }
    __syncthreads();  // every lane is done with window w's tile and nodes; the margins of all classes are in LDS

    if (half == 0 && n < L.N) {
      const size_t o = ((size_t)n * L.W + w) * A;
      if (A == 2) {
        const float margin = logf(L.base_score / (1.0f - L.base_score)) + marg[0];  // ProbToMargin of binary:logistic
        const float p1 = 1.0f / (1.0f + (float)exp((double)(-margin)));
        const float p[2] = {1.0f - p1, p1};
        for (int a = 0; a < 2; ++a) {
          if (L.b32) L.b32[o + a] = p[a];
          if (L.b64) L.b64[o + a] = (double)p[a];
        }
      } else {
        float wmax = L.base_score + marg[0];
        for (int a = 1; a < A; ++a) wmax = fmaxf(L.base_score + marg[a * T2], wmax);
        // the exponentials are parked in the lane's own margin slots (LDS) between the two loops instead of in A registers: the
        // kernel has to stay inside 128 VGPRs for two blocks per CU
        double wsum = 0.0;
        for (int a = 0; a < A; ++a) {
          const float e = (float)exp((double)((L.base_score + marg[a * T2]) - wmax));
          marg[a * T2] = e;
          wsum += (double)e;
        }
        const float fs = (float)wsum;
        for (int a = 0; a < A; ++a) {
          const float p = marg[a * T2] / fs;
          if (L.b32) L.b32[o + a] = p;
          if (L.b64) L.b64[o + a] = (double)p;
        }
      }
    }
    // (the next iteration's staging rewrites ring slots and node words every lane is done with; the margins are rewritten
    //  only behind the next iteration's first barrier, after group 0 has read them)
  }
}

template <int D>
hipError_t launch2_d(const ForestLaunch& L, const uint32_t* nodes2, size_t lds, hipStream_t s) {
  const int n_runs = (L.n_windows + L.wrun - 1) / L.wrun;
  const dim3 grid((unsigned)((L.N + T2 - 1) / T2), (unsigned)n_runs);
  if (L.rf_leafval) {
    GNX_LDS_OPTIN(lds, k_base_forest2<D, true>);
    hipLaunchKernelGGL((k_base_forest2<D, true>), grid, dim3(NTHR2), lds, s, L, nodes2);
  } else {
    GNX_LDS_OPTIN(lds, k_base_forest2<D, false>);
    hipLaunchKernelGGL((k_base_forest2<D, false>), grid, dim3(NTHR2), lds, s, L, nodes2);
  }
  return hipGetLastError();
}

hipError_t launch_range2(ForestLaunch L, const uint32_t* nodes2, int w_first, int n_windows, int64_t width, int n_cu, const gnx_tune& tune,
                         hipStream_t s) {
  if (n_windows <= 0) return hipSuccess;
  L.w_first = w_first;
  L.n_windows = n_windows;
  L.ring = gnx_forest_ring_words(width);
  L.flags = tune.forest_flags;
  const size_t lds = gnx_forest2_lds_bytes(L.A, L.ring, L.max_trees, L.D, L.rf_leafval != nullptr);
  if (lds > (size_t)160 * 1024 || (L.rf_leafval && L.A > H2 * CG2)) return hipErrorInvalidValue;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)160 * 1024 / lds, (size_t)(4 * WPS2) / NWV2));
  // windows per block: as k_base_forest — long runs re-use the shared half of every window (the first window of a run is staged
  // in full: about two window-steps), short runs fill the chip; blocks run in rounds of per_cu per CU
  const int64_t tiles = (L.N + T2 - 1) / T2;
  int64_t wrun = tune.forest_wrun;
  if (wrun <= 0) {
    int64_t best = INT64_MAX;
    for (int64_t r = 1; r <= std::min<int64_t>(24, n_windows); ++r) {
      const int64_t blocks = tiles * ((n_windows + r - 1) / r);
      const int64_t cost = ((blocks + (int64_t)n_cu * per_cu - 1) / ((int64_t)n_cu * per_cu)) * (r + 1);
      if (cost < best) { best = cost; wrun = r; }
    }
  }
  L.wrun = (int)std::min<int64_t>(std::max<int64_t>(1, wrun), n_windows);
  switch (L.D) {
    case 1: return launch2_d<1>(L, nodes2, lds, s);
    case 2: return launch2_d<2>(L, nodes2, lds, s);
    case 3: return launch2_d<3>(L, nodes2, lds, s);
    case 4: return launch2_d<4>(L, nodes2, lds, s);
    case 5: return launch2_d<5>(L, nodes2, lds, s);
    case 6: return launch2_d<6>(L, nodes2, lds, s);
    case 7: return launch2_d<7>(L, nodes2, lds, s);
    case 8: return launch2_d<8>(L, nodes2, lds, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

size_t gnx_forest2_lds_bytes(int A, int ring_words, int max_trees, int D, bool rf) {  // rf: float64 class sums instead of float32 margins
  return (size_t)ring_words * T2 * 4 + (((size_t)max_trees << D) * 4 + 15 & ~(size_t)15) + (size_t)A * T2 * (rf ? 8 : 4);
}

// node word of k_base_forest2 for a loader word (position << 4 | left-mask) of a window whose first word is g0, in a ring of
// `ring` word slots (a power of two): (byte offset of the slot's row of 128 haplotypes) << 15 | (2 * field) << 4 | right-mask
uint32_t gnx_forest2_node(uint32_t nd, uint32_t g0, uint32_t ring) {
  const uint32_t pos = nd >> 4;
  const uint32_t slot = ((pos >> 4) + g0) & (ring - 1u);
  return ((slot * (uint32_t)T2 * 4u) << 15) | ((2u * (pos & 15u)) << 4) | (~nd & 15u);
}

hipError_t gnx_launch_base_forest2(const ForestLaunch& L, const uint32_t* nodes2, int n_cu, const gnx_tune& tune, hipStream_t s,
                                   hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join) {
  if (L.N <= 0) return hipSuccess;
  // The last window is wider by C mod M (base.py:163-164): its own launch with the larger ring — N / 128 blocks, a fraction of the
  // chip — on the side stream beside the main grid (they write disjoint windows of B): fork after whatever produced X, join before
  // whatever reads B.
  const bool side = aux && ev_fork && ev_join && L.W > 1;
  hipError_t e;
  if (side) {
    if ((e = hipEventRecord(ev_fork, s)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(aux, ev_fork, 0)) != hipSuccess) return e;
    if ((e = launch_range2(L, nodes2, L.W - 1, 1, L.width_last, n_cu, tune, aux)) != hipSuccess) return e;
    if ((e = hipEventRecord(ev_join, aux)) != hipSuccess) return e;
  }
  if ((e = launch_range2(L, nodes2, 0, L.W - 1, L.width, n_cu, tune, s)) != hipSuccess) return e;
  if (side) return hipStreamWaitEvent(s, ev_join, 0);
  return launch_range2(L, nodes2, L.W - 1, 1, L.width_last, n_cu, tune, s);
}

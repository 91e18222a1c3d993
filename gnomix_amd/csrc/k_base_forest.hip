// k_base_forest.hip — per-window gradient-boosted-tree base classifiers ("forest" bases) on gfx950.
//
// Replaces XGBBase.predict_proba (reference src/Base/models.py:24-35: per window
// XGBClassifier(n_estimators=20, max_depth=4, missing=missing_encoding); xgboost semantics as restated in the CPU
// oracle: a SNP equal to the missing code takes the node's default child, otherwise left iff float(v) < threshold;
// multi:softprob for A >= 3, binary:logistic for A == 2).  No reference mode selects these bases (README.md:120-146
// shows how to plug them in), they are the "forest-of-stumps" family of the north star.
//
// One thread = one haplotype of one window; a block = T = 64*waves haplotypes and a RUN of consecutive windows.
//  * the SNPs live in LDS as 2-bit fields, 16 per word, xw[word slot][hap]: during the walks a lane's bank is fixed by its
//    haplotype, so lanes that sit on different split SNPs never conflict.  Words are anchored on the PADDED chromosome
//    coordinate (word g = padded SNPs 16g .. 16g+15) and kept in a power-of-two ring of slots (slot = g mod RING), so the
//    half of a window that the next window shares (context: windows overlap by 2*ctx of M + 2*ctx SNPs, base.py:146-151)
//    stays where it is: a block walks windows w, w+1, ... of its haplotypes and only fetches each window's NEW words —
//    every byte of X is read about once instead of (M + 2 ctx) / M times.
//  * those new words are fetched while the trees of the current window are walked: the loads are issued in four batches
//    between tree groups (16 x 16-byte loads in flight per lane), squeezed to 2 bits per SNP into registers as they
//    arrive, and stored into the ring slots the current window no longer needs after the block's barrier.  (One block
//    fills the LDS, i.e. one wave per SIMD: nothing else would hide the HBM latency.)
//  * staging proper: every wave reads ITS OWN 64 haplotypes' bytes straight from X (one unaligned 16-byte load per lane
//    = one word, 8 lanes along a row = one 128-byte line).  Reflect padding only touches the first / last ctx SNPs of a
//    row: those words take a per-byte path.
//  * the window's trees sit next to the tile as compact heaps: one 32-bit word per node = (position << 4) | left-mask
//    (position = SNP index within the window + the window start's offset inside its first word; bit v of the mask =
//    "value v goes left": SNPs only take the values 0..3, so the float compare and the missing code's default direction
//    were folded into 4 bits by the loader), then the 2^D float leaves.  A lane walks TP trees at a time level by level
//    (TP independent LDS chains), adds the leaves of each class in tree order (float32, as the restated predictor does),
//    parks the margins in LDS and applies softmax / sigmoid in float32.
#include "gnx_internal.h"

#include <cstdint>
#include <cstdlib>

#ifndef GNX_FOREST_PART
#define GNX_FOREST_PART 0
#endif
// This file is compiled four times (Makefile: k_base_forest.o, k_base_forest_p1/p2/p3.o with -DGNX_FOREST_PART=1..3), each pass
// instantiating two tree depths: one pass with all eight took 69 s, longer than the rest of the library together.
hipError_t gnx_forest_v1_part1(const ForestLaunch& L, int haps, size_t lds, bool two, hipStream_t s);
hipError_t gnx_forest_v1_part2(const ForestLaunch& L, int haps, size_t lds, bool two, hipStream_t s);
hipError_t gnx_forest_v1_part3(const ForestLaunch& L, int haps, size_t lds, bool two, hipStream_t s);

namespace {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int64_t pad_src(int64_t p, int64_t C, int64_t ctx) {
  if (p < ctx) return ctx - 1 - p;
  if (p < ctx + C) return p - ctx;
  return C - 1 - (p - ctx - C);
}

// 4 SNP bytes (values 0..3) -> 8 bits
__device__ __forceinline__ uint32_t squeeze4(uint32_t d) {
  const uint32_t x = d & 0x03030303u;
  const uint32_t y = x | (x >> 6);
  return (y | (y >> 12)) & 0xffu;
}

constexpr int TP = 16;  // trees walked concurrently per lane
constexpr int LB = 16;  // 16-byte loads in flight per lane while staging
constexpr int NPF = 4;  // prefetch batches per window (LB loads each): at most NPF*LB new words per haplotype row group

// Node words as the walks read them from LDS.  The loader's word is (position << 4) | left-mask; while a window's trees are
// copied into LDS every node word is rewritten FOR THAT WINDOW AND BLOCK SHAPE as
//     (byte offset of the word's ring row) << 15 | (2 * field) << 4 | right-mask        right-mask = ~left-mask & 15
// so that a level costs 7 VALU operations: node address, row offset (shift), + lane column, three bit-field extracts
// (field shift, SNP value, direction), j = 2j + direction.
__device__ __forceinline__ uint32_t node_for_window(uint32_t nd, uint32_t g0, uint32_t mask, int T) {
  const uint32_t pos = nd >> 4;
  const uint32_t slot = ((pos >> 4) + g0) & mask;
  return ((slot * (uint32_t)T * 4u) << 15) | ((2u * (pos & 15u)) << 4) | (~nd & 15u);
}

__device__ __forceinline__ uint32_t step(uint32_t j, uint32_t nd, uint32_t xv) {
  const uint32_t v = __builtin_amdgcn_ubfe(xv, __builtin_amdgcn_ubfe(nd, 4, 5), 2);  // the 2-bit SNP value
  return 2 * j + __builtin_amdgcn_ubfe(nd, v, 1);                                     // + 1 iff value v goes right
}

// the tile as the walks see it: this lane's column of the ring (byte address), node words carry the row offset
struct Tile {
  const uint8_t* xcol;  // (const uint8_t*)(xw + haplotype)
  __device__ __forceinline__ uint32_t word(uint32_t nd) const { return *reinterpret_cast<const uint32_t*>(xcol + (nd >> 15)); }
};

template <int D, int NT>
__device__ __forceinline__ void walk(const uint8_t* tb, int tree_bytes, const Tile& tile, float* leaf) {
  uint32_t j[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) j[k] = 1;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    uint32_t nd[NT], xv[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) nd[k] = reinterpret_cast<const uint32_t*>(tb + k * tree_bytes)[j[k]];
#pragma unroll
    for (int k = 0; k < NT; ++k) xv[k] = tile.word(nd[k]);
#pragma unroll
    for (int k = 0; k < NT; ++k) j[k] = step(j[k], nd[k], xv[k]);
  }
#pragma unroll
  for (int k = 0; k < NT; ++k) leaf[k] = reinterpret_cast<const float*>(tb + k * tree_bytes)[j[k]];  // leaves follow the 2^D nodes
}

// ---- staging ------------------------------------------------------------------------------------------------------
// global word g = padded SNPs [16g, 16g+16) of a haplotype row; a lane stages (haplotype hb*8 + lane>>3 of its wave, words
// g_first + 8u + (lane&7)): 8 lanes along a row cover 128 contiguous bytes.
struct Stager {
  const int8_t* X;      // copies of the launch fields (a reference to the kernel argument would force it into scratch)
  int64_t N, ldx, C, ctx, cmax, blk0;
  int T, wsub, hsub, wv, hb0, hbn;  // T haplotypes per block; this wave stages groups hb0 .. hb0+hbn-1 (8 haplotypes each) of its 64
  __device__ __forceinline__ Stager(const int8_t* X_, int64_t N_, int64_t ldx_, int64_t C_, int64_t ctx_, int T_, int halves)
      : X(X_), N(N_), ldx(ldx_), C(C_), ctx(ctx_), T(T_) {
    const int lane = threadIdx.x & 63;
    const int hap = (int)threadIdx.x % T_, half = (int)threadIdx.x / T_;
    wsub = lane & 7; hsub = lane >> 3; wv = hap >> 6;
    hbn = 8 / halves; hb0 = half * hbn;
    cmax = C - 16;  // the model loader guarantees C >= 16
    blk0 = (int64_t)blockIdx.x * T;
  }
  __device__ __forceinline__ const int8_t* row(int hb) const {
    int64_t n = blk0 + wv * 64 + hb * 8 + hsub;
    n = n < N ? n : N - 1;
    return X + n * ldx;
  }
  __device__ __forceinline__ v4u load(const int8_t* r, int64_t g) const {  // unconditional, clamped address
    int64_t c0 = 16 * g - ctx;
    c0 = c0 < 0 ? 0 : (c0 > cmax ? cmax : c0);
    v4u v;
    __builtin_memcpy(&v, r + c0, 16);
    return v;
  }
  __device__ __forceinline__ v4u load_inner(const int8_t* r, int64_t g) const {  // word known to lie inside the row: no clamp
    v4u v;
    __builtin_memcpy(&v, r + (16 * g - ctx), 16);
    return v;
  }
  __device__ __forceinline__ uint32_t squeeze(const v4u& v, const int8_t* r, int64_t g) const {
    const int64_t p0 = 16 * g;
    uint32_t q = squeeze4(v.x) | (squeeze4(v.y) << 8) | (squeeze4(v.z) << 16) | (squeeze4(v.w) << 24);
    if (p0 < ctx || p0 + 16 > ctx + C) {  // rare: the word touches the reflect padding / the row's end
      q = 0;
      for (int b = 0; b < 16; ++b) {
        const int64_t p = p0 + b;
        if (p < C + 2 * ctx) q |= ((uint32_t)(uint8_t)r[pad_src(p, C, ctx)] & 3u) << (2 * b);
      }
    }
    return q;
  }
  // words [ga, gb) of the wave's 64 haplotypes -> ring, synchronously.  Two spellings of one body: as an out-of-line member the
  // object it is called on has to live in memory (104 B of scratch per lane), inlined everywhere the file takes four times as
  // long to compile (35 kernel instances).  The random-forest instances — the default dispatch of that base — inline it; the
  // boosted-tree instances are the fallback behind k_base_forest2 and keep the call.
  __device__ void stage(uint32_t* xw, uint32_t mask, int64_t ga, int64_t gb) const { stage_inl(xw, mask, ga, gb); }
  __device__ __forceinline__ void stage_inl(uint32_t* xw, uint32_t mask, int64_t ga, int64_t gb) const {
    for (int hb = hb0; hb < hb0 + hbn; ++hb) {
      const int8_t* r = row(hb);
      uint32_t* col = xw + wv * 64 + hb * 8 + hsub;
      for (int64_t gw = ga; gw < gb; gw += 8 * LB) {
        v4u v[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) v[u] = load(r, gw + 8 * u + wsub);
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          const int64_t g = gw + 8 * u + wsub;
          if (g < gb) col[(size_t)((uint32_t)g & mask) * T] = squeeze(v[u], r, g);
        }
      }
    }
  }
};

// Register prefetch of the next window's new words [ga, gb) (at most 64, none of them touching the reflect padding: those
// windows are staged synchronously).  Batch Q covers haplotype groups hb = 2Q, 2Q+1 of the wave, 8 words per lane each:
// pf_issue<Q> puts LB loads in flight, pf_consume<Q> squeezes them into sq[Q*LB ..]; every register index is static.
template <int Q>
__device__ __forceinline__ void pf_issue(const Stager& st, int64_t ga, int64_t glast, v4u (&raw)[LB]) {
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int8_t* r = st.row(st.hb0 + 2 * Q + h2);
#pragma unroll
    for (int u = 0; u < LB / 2; ++u) raw[h2 * (LB / 2) + u] = st.load_inner(r, min(ga + 8 * u + st.wsub, glast));
  }
}
template <int Q>
__device__ __forceinline__ void pf_consume(const v4u (&raw)[LB], uint32_t* sq) {
#pragma unroll
  for (int k = 0; k < LB; ++k)
    sq[Q * LB + k] = squeeze4(raw[k].x) | (squeeze4(raw[k].y) << 8) | (squeeze4(raw[k].z) << 16) | (squeeze4(raw[k].w) << 24);
}
// batch q goes out after the previous one was squeezed into its registers (q is block-uniform; q == NB drains the last);
// NB = batches per wave = 4 (the wave stages all 8 haplotype groups of its 64) or 2 (two wave groups share them)
template <int NB>
__device__ __forceinline__ void pf_advance(const Stager& st, int q, int64_t ga, int64_t glast, v4u (&raw)[LB], uint32_t (&sq)[NB * LB]) {
  if (q == 0) pf_issue<0>(st, ga, glast, raw);
  else if (q == 1) { pf_consume<0>(raw, sq); if constexpr (NB > 1) pf_issue<1>(st, ga, glast, raw); }
  else if (q == 2) { if constexpr (NB > 1) pf_consume<1>(raw, sq); if constexpr (NB > 2) pf_issue<2>(st, ga, glast, raw); }
  else if (q == 3) { if constexpr (NB > 2) pf_consume<2>(raw, sq); if constexpr (NB > 3) pf_issue<3>(st, ga, glast, raw); }
  else { if constexpr (NB > 3) pf_consume<3>(raw, sq); }
}

__device__ __forceinline__ void window_words(const ForestLaunch& L, int w, int64_t& g0, int64_t& g1) {
  const int64_t s = (int64_t)w * L.M, width = (w == L.W - 1) ? L.width_last : L.width;
  g0 = s >> 4;
  g1 = ((s + width - 1) >> 4) + 1;
}

// RF = false: boosted trees (XGBBase); RF = true: random forest (RFBase, src/Base/models.py:54-66): same tile and mask nodes
// (mask bit v = "float32(v) <= threshold"); a leaf contributes its class-probability row (float64, expanded per heap slot in
// global memory: 20 trees x 16 leaves x A doubles per window stay in L1/L2), the rows are added in estimator order and divided
// by the tree count, as ForestClassifier.predict_proba does.
// H = wave groups per block: H = 2 puts 8 waves on a 256-haplotype tile — the tile is what fills the LDS, waves are free —
// the two groups stage different haplotype groups, walk different classes of the SAME haplotypes (margins meet in LDS) and group
// 0 writes the window's probabilities; two waves per SIMD overlap each other's LDS latencies and staging instructions.
template <int D, bool RF, int AMAX, int H>
__global__ __launch_bounds__(256 * H) void k_base_forest(ForestLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int NB = NPF / H;      // prefetch batches per wave
  constexpr int TPW = TP / H;      // trees walked side by side per lane (two waves per SIMD need half the chains and registers)
  const int tid = threadIdx.x, NTHR = blockDim.x, T = NTHR / H;
  const int half = tid / T, hap = tid - half * T;
  const int A = L.A, tree_bytes = L.tree_bytes;
  const int Dr = RF ? L.D : D;  // the random-forest walk is depth-generic
  const uint32_t mask = (uint32_t)L.ring - 1u;
  uint32_t* xw = reinterpret_cast<uint32_t*>(lds);                                    // [ring][T]
  uint8_t* tr = lds + (size_t)L.ring * T * 4;                                         // trees of the current window
  float* marg = reinterpret_cast<float*>(tr + (((size_t)L.max_trees * tree_bytes + 15) & ~(size_t)15)) + hap;  // [A][T]

  const int wa = L.w_first + blockIdx.y * L.wrun, wb = min(L.w_first + L.n_windows, wa + L.wrun);
  const Stager st(L.X, L.N, L.ldx, L.C, L.ctx, T, H);
  int64_t g0, g1;
  window_words(L, wa, g0, g1);
  if constexpr (RF) st.stage_inl(xw, mask, g0, g1);
  else st.stage(xw, mask, g0, g1);
  const int64_t n = (int64_t)blockIdx.x * T + hap;

  // a window's trees: uint4 pieces of the loader's records, node words rewritten for the window on their way into LDS
  // pieces per thread held in registers while the previous window is walked (random forest: 2 — its float64 class sums and the
  // word prefetch already fill the 512 registers of a lane; with 8 the kernel spilled 12 VGPRs)
  constexpr int TQ = RF ? 2 : 8 / H;
  const int words_per_tree = tree_bytes / 4, node_words = RF ? tree_bytes / 4 : (1 << D);
  auto put_tree_piece = [&](int e, uint4 v, uint32_t gw0) {
    const int wd = (e * 4) % words_per_tree;  // first word of the piece inside its tree (records are multiples of 16 bytes)
    if (wd < node_words) v.x = node_for_window(v.x, gw0, mask, T);      // per word: a depth-1 record holds 2 node words
    if (wd + 1 < node_words) v.y = node_for_window(v.y, gw0, mask, T);  // and its 2 leaves in ONE piece
    if (wd + 2 < node_words) v.z = node_for_window(v.z, gw0, mask, T);
    if (wd + 3 < node_words) v.w = node_for_window(v.w, gw0, mask, T);
    reinterpret_cast<uint4*>(tr)[e] = v;
  };
  {
    const int t0 = L.win_tree0[wa], np = (L.win_tree0[wa + 1] - t0) * tree_bytes / 16;
    const uint4* src = reinterpret_cast<const uint4*>(L.packed + (size_t)t0 * tree_bytes);
    for (int e = tid; e < np; e += NTHR) put_tree_piece(e, src[e], (uint32_t)g0);
  }

  for (int w = wa; w < wb; ++w) {
    const int t0 = L.win_tree0[w], nt = L.win_tree0[w + 1] - t0;
    __syncthreads();  // tile + trees of window w complete; everybody has read the previous window's margins
    // the next window's trees travel through registers while this one is walked (when they fit TQ pieces per thread)
    uint4 tq[TQ];
    int np_next = 0;
    const uint4* tsrc = nullptr;
    if (w + 1 < wb) {
      const int t1 = L.win_tree0[w + 1];
      np_next = (L.win_tree0[w + 2] - t1) * tree_bytes / 16;
      tsrc = reinterpret_cast<const uint4*>(L.packed + (size_t)t1 * tree_bytes);
    }
    const bool tpre = np_next > 0 && np_next <= TQ * NTHR;
    if (tpre) {
#pragma unroll
      for (int k = 0; k < TQ; ++k) tq[k] = tsrc[min(tid + k * NTHR, np_next - 1)];
    }

    // the next window's new words: prefetched through registers when they fit the batches and none of them touches the
    // reflect padding, else staged after the walks
    v4u raw[LB];
    uint32_t sq[NB * LB];
    int64_t ng0 = 0, ng1 = 0, pf_ga = 0, pf_gb = 0;
    bool pre = false;
    if (w + 1 < wb) {
      window_words(L, w + 1, ng0, ng1);
      pf_ga = ng0 > g1 ? ng0 : g1;
      pf_gb = ng1;
      pre = (pf_gb - pf_ga) <= 8 * (LB / 2) && 16 * pf_ga >= L.ctx && 16 * pf_gb <= L.ctx + L.C && !(L.flags & 2);
    }
    int stage_q = 0;  // batches issued so far
    // called between tree groups: keep batch (progress * NB) in flight
#define GNX_PF_PUMP(done, total)                                                                   \
  if (pre) {                                                                                       \
    const int want_ = min(NB, ((done) * NB) / max((total), 1) + 1);                                \
    while (stage_q < want_) { pf_advance<NB>(st, stage_q, pf_ga, pf_gb - 1, raw, sq); ++stage_q; } \
  }

    const Tile tile{reinterpret_cast<const uint8_t*>(xw + hap)};
    const size_t o = ((size_t)(n < L.N ? n : 0) * L.W + w) * A;
    double acc[AMAX];  // random forest only
    if constexpr (!RF) {
      // ---- walks: class-major packing per window, trees of class c contiguous and in model order; wave group h walks the
      // classes [c_lo, c_hi) of its haplotypes ----------------
      const int32_t* cls0 = L.win_class_tree0 + (size_t)w * (A + 1);  // [A+1] offsets relative to t0
      const int n_groups = (A == 2) ? 1 : A;
      const int c_mid = (H == 1) ? n_groups : (n_groups + 1) / 2;
      const int c_lo = half == 0 ? 0 : c_mid, c_hi = half == 0 ? c_mid : n_groups;
      const int tr_lo = (A == 2) ? 0 : cls0[c_lo], tr_hi = (A == 2) ? (half == 0 ? nt : 0) : cls0[c_hi];
      for (int c = c_lo; c < c_hi; ++c) {
        const int a0 = (A == 2) ? 0 : cls0[c], a1 = (A == 2) ? nt : cls0[c + 1];
        float psum = 0.f;
        int t = a0;
        for (; t + TPW <= a1 && !(L.flags & 1); t += TPW) {
          GNX_PF_PUMP(t - tr_lo, tr_hi - tr_lo)
          float leaf[TPW];
          walk<D, TPW>(tr + (size_t)t * tree_bytes, tree_bytes, tile, leaf);
#pragma unroll
          for (int k = 0; k < TPW; ++k) psum += leaf[k];  // tree order
        }
        for (; t < a1 && !(L.flags & 1); ++t) {
          float leaf[1];
          walk<D, 1>(tr + (size_t)t * tree_bytes, tree_bytes, tile, leaf);
          psum += leaf[0];
        }
        marg[c * T] = psum;
      }
    } else {
#pragma unroll
      for (int a = 0; a < AMAX; ++a) acc[a] = 0.0;
      const size_t leaves = (size_t)1 << Dr;
      for (int t = 0; t < nt; ++t) {
        if ((t & 3) == 0) GNX_PF_PUMP(t, nt)
        const uint32_t* nodes = reinterpret_cast<const uint32_t*>(tr + (size_t)t * tree_bytes);
        uint32_t j = 1;
        for (int d = 0; d < Dr; ++d) {
          const uint32_t nd = nodes[j];
          j = step(j, nd, tile.word(nd));
        }
        const double* v = L.rf_leafval + ((size_t)(t0 + t) * leaves + (j - (uint32_t)leaves)) * A;
#pragma unroll
        for (int a = 0; a < AMAX; ++a)
          if (a < A) acc[a] += v[a];  // estimator order
      }
    }
    // batches the walks did not reach
    if (pre)
      while (stage_q <= NB) { pf_advance<NB>(st, stage_q, pf_ga, pf_gb - 1, raw, sq); ++stage_q; }
    __syncthreads();  // every lane is done with window w's tile and trees; the margins of all classes are in LDS

    // ---- the window's probabilities (wave group 0) ----
    if (half == 0 && n < L.N) {
      if constexpr (!RF) {
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
          for (int a = 1; a < A; ++a) wmax = fmaxf(L.base_score + marg[a * T], wmax);
          double wsum = 0.0;
          float ex[AMAX];
#pragma unroll
          for (int a = 0; a < AMAX; ++a)
            if (a < A) {
              ex[a] = (float)exp((double)((L.base_score + marg[a * T]) - wmax));
              wsum += (double)ex[a];
            }
          const float fs = (float)wsum;
#pragma unroll
          for (int a = 0; a < AMAX; ++a)
            if (a < A) {
              const float p = ex[a] / fs;
              if (L.b32) L.b32[o + a] = p;
              if (L.b64) L.b64[o + a] = (double)p;
            }
        }
      } else {
        const double cnt = (double)nt;
#pragma unroll
        for (int a = 0; a < AMAX; ++a)
          if (a < A) {
            const double p = acc[a] / cnt;
            if (L.b64) L.b64[o + a] = p;
            if (L.b32) L.b32[o + a] = (float)p;
          }
      }
    }
    if (w + 1 >= wb) break;
    // replace the words the next window no longer shares, and the trees
    if (pre) {
#pragma unroll
      for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int u = 0; u < LB / 2; ++u) {
            const int64_t g = pf_ga + 8 * u + st.wsub;
            if (g < pf_gb)
              xw[(size_t)((uint32_t)g & mask) * T + st.wv * 64 + (st.hb0 + 2 * q + h2) * 8 + st.hsub] = sq[q * LB + h2 * (LB / 2) + u];
          }
    } else if (!(L.flags & 4)) {
      if constexpr (RF) st.stage_inl(xw, mask, ng0 > g1 ? ng0 : g1, ng1);
      else st.stage(xw, mask, ng0 > g1 ? ng0 : g1, ng1);
    }
    if (tpre) {
#pragma unroll
      for (int k = 0; k < TQ; ++k)
        if (tid + k * NTHR < np_next) put_tree_piece(tid + k * NTHR, tq[k], (uint32_t)ng0);
    } else {
      for (int e = tid; e < np_next; e += NTHR) put_tree_piece(e, tsrc[e], (uint32_t)ng0);
    }
    g0 = ng0;
    g1 = ng1;
  }
#undef GNX_PF_PUMP
}

template <int D, bool RF, int AMAX, int H>
hipError_t launch_k(const ForestLaunch& L, int haps, size_t lds, hipStream_t s) {
  GNX_LDS_OPTIN((size_t)160 * 1024, k_base_forest<D, RF, AMAX, H>);
  const int n_runs = (L.n_windows + L.wrun - 1) / L.wrun;
  hipLaunchKernelGGL((k_base_forest<D, RF, AMAX, H>), dim3((unsigned)((L.N + haps - 1) / haps), (unsigned)n_runs), dim3(haps * H), lds, s, L);
  return hipGetLastError();
}

template <int D>
hipError_t launch_xgb(const ForestLaunch& L, int haps, size_t lds, bool two, hipStream_t s) {
  if (L.A <= 8) return two ? launch_k<D, false, 8, 2>(L, haps, lds, s) : launch_k<D, false, 8, 1>(L, haps, lds, s);
  return two ? launch_k<D, false, 32, 2>(L, haps, lds, s) : launch_k<D, false, 32, 1>(L, haps, lds, s);
}

#if GNX_FOREST_PART == 0
// windows [w_first, w_first + n_windows), all of padded width `width` (the last window of the chromosome goes alone)
hipError_t launch_range(ForestLaunch L, int w_first, int n_windows, int64_t width, int n_cu, const gnx_tune& tune, hipStream_t s) {
  if (n_windows <= 0) return hipSuccess;
  constexpr size_t kLds = (size_t)160 * 1024;
  L.w_first = w_first;
  L.n_windows = n_windows;
  L.flags = tune.forest_flags;
  // ring: a power of two of word slots that holds any window of this width whatever its start's offset inside a word
  int ring = 16;
  while (ring < (int)((width + 15 + 15) >> 4)) ring <<= 1;
  L.ring = ring;
  // one block fills the LDS (the tile is 4 * ring bytes per haplotype): as many haplotypes per block as fit
  int threads = tune.forest_threads / 64 * 64;
  if (threads < 64 || threads > 256) threads = 256;
  while (threads > 64 && gnx_forest_lds_bytes(L.A, ring, L.max_trees, L.tree_bytes, threads) > kLds) threads -= 64;
  while (threads > 64 && (int64_t)(threads - 64) >= L.N) threads -= 64;
  const size_t lds = gnx_forest_lds_bytes(L.A, ring, L.max_trees, L.tree_bytes, threads);
  if (lds > kLds) return hipErrorInvalidValue;
  // windows per block: long runs re-use the shared half of every window (the first window of a run is staged in full, about
  // two window-steps of work), short runs fill the chip.  Blocks run in rounds of one per CU, so the cost of a run length r
  // is rounds(r) * (r + 2); measured on chr22 / 10 k haplotypes, 140 trees per window: r = 8 -> 2.95 ms, 11 -> 3.1,
  // 12 -> 2.64, 16 -> 2.73, 24 -> 2.73 — the model's order
  const int64_t tiles = (L.N + threads - 1) / threads;
  int64_t wrun = tune.forest_wrun;
  if (wrun <= 0) {
    int64_t best = INT64_MAX;
    for (int64_t r = 1; r <= std::min<int64_t>(16, n_windows); ++r) {
      const int64_t blocks = tiles * ((n_windows + r - 1) / r);
      const int64_t cost = ((blocks + n_cu - 1) / n_cu) * (r + 2);
      if (cost < best) { best = cost; wrun = r; }
    }
  }
  wrun = std::max<int64_t>(1, wrun);
  L.wrun = (int)std::min<int64_t>(wrun, n_windows);
  if (L.rf_leafval) {
    if (L.A <= 8) return launch_k<1, true, 8, 1>(L, threads, lds, s);
    if (L.A <= 16) return launch_k<1, true, 16, 1>(L, threads, lds, s);
    return launch_k<1, true, 32, 1>(L, threads, lds, s);
  }
  // two wave groups per tile when there is more than one class group to split and the block stays within 1024 threads
  const bool two = tune.forest_halves != 1 && L.A > 2 && threads * 2 <= 1024;
  switch (L.D) {
    case 1: return launch_xgb<1>(L, threads, lds, two, s);
    case 2: return launch_xgb<2>(L, threads, lds, two, s);
    case 3: case 4: return gnx_forest_v1_part1(L, threads, lds, two, s);
    case 5: case 6: return gnx_forest_v1_part2(L, threads, lds, two, s);
    case 7: case 8: return gnx_forest_v1_part3(L, threads, lds, two, s);
    default: return hipErrorInvalidValue;
  }
}

#endif
}  // namespace

#if GNX_FOREST_PART == 1
hipError_t gnx_forest_v1_part1(const ForestLaunch& L, int haps, size_t lds, bool two, hipStream_t s) {
  return L.D == 3 ? launch_xgb<3>(L, haps, lds, two, s) : launch_xgb<4>(L, haps, lds, two, s);
}
#elif GNX_FOREST_PART == 2
hipError_t gnx_forest_v1_part2(const ForestLaunch& L, int haps, size_t lds, bool two, hipStream_t s) {
  return L.D == 5 ? launch_xgb<5>(L, haps, lds, two, s) : launch_xgb<6>(L, haps, lds, two, s);
}
#elif GNX_FOREST_PART == 3
hipError_t gnx_forest_v1_part3(const ForestLaunch& L, int haps, size_t lds, bool two, hipStream_t s) {
  return L.D == 7 ? launch_xgb<7>(L, haps, lds, two, s) : launch_xgb<8>(L, haps, lds, two, s);
}
#else
size_t gnx_forest_lds_bytes(int A, int ring_words, int max_trees, int tree_bytes, int threads) {
  return (size_t)ring_words * threads * 4 + (((size_t)max_trees * tree_bytes + 15) & ~(size_t)15) + (size_t)A * threads * 4;
}

// word slots a window of `width` padded SNPs needs (power of two, any alignment of its first SNP inside a word)
int gnx_forest_ring_words(int64_t width) {
  int ring = 16;
  while (ring < (int)((width + 15 + 15) >> 4)) ring <<= 1;
  return ring;
}

hipError_t gnx_launch_base_forest(const ForestLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  // the last window is wider by C mod M (base.py:163-164): its own launch, so that the others get the smaller ring
  hipError_t e = launch_range(L, 0, L.W - 1, L.width, n_cu, tune, s);
  if (e != hipSuccess) return e;
  return launch_range(L, L.W - 1, 1, L.width_last, n_cu, tune, s);
}
#endif

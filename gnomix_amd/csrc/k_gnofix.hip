// k_gnofix.hip — the Gnofix re-phasing loop on gfx950 (round 4): rank strips, read-only; several individuals per CU.
//
// Replaces Gnomix.phase -> gnofix() with its default arguments (reference src/model.py:188-214,
// src/Gnofix/gnofix.py:58-208: check_criterion="disc_smooth", max_center_offset=0, non_lin_s=0, prob_comp="max",
// prior_switch_prob=0.5, padding=True, no naive switch) and track_switch / correct_phase_error
// (src/Gnofix/phasing.py:182-198).
//
// The loop is sequential per individual (an accepted switch changes B from window w to the end), so the parallelism is across
// individuals and inside one smoother evaluation.  Rounds 1-3 (k_gnofix_f32.hip, kept as the fallback) ran ONE 512-thread
// workgroup per CU because the individual's two float32 strips filled the LDS (or lived in a global scratch that every accepted
// switch rewrote): every pipe sat below 20 % while one block waited on its own dependent chains.  What changed:
//  * The smoother only ever asks `p < threshold`, so the strips are 16-bit RANKS (gnx_rank.h; the same quantisation as
//    k_smooth_xgb_rk): k_gnofix_ranks turns B into (2n, W, A) u16 once, fully parallel.
//  * The strips are READ-ONLY.  A switch at w exchanges the two haplotypes from w on; instead of rewriting strips the kernel keeps
//    the per-window switch PARITY it needs anyway (convergence test, final SNP swap): logical haplotype h at window u is physical
//    haplotype h ^ parity(u).  An accepted switch is a flip of parity bits.  Nothing of size W*A lives in LDS or is ever written.
//  * What a phase reads is local to w: a candidate reads S windows of both strips, the re-evaluation 2S+1.  That slice is gathered
//    from L2 into a class-major [hap][class][GP] u16 tile (the tile k_smooth_xgb_rk walks), so a block needs ~25 KB of LDS and five
//    to six 256-thread blocks share a CU: one block's L2 round trips and dependent walks hide behind the others'.
//  * Candidate (4 rows x n_trees walks, trees in L2): the trees are dealt out over the block; a lane walks its few trees on all FOUR
//    rows side by side (the tree's first 16 bytes — root and both children — are fetched once for the four rows; a depth-4 walk is 4
//    dependent L2 round trips and a candidate is ONE batch of them) and parks the leaves in LDS; one lane per (row, class) then adds
//    them in tree order — bit-identical to the sequential predictor.
//  * Re-evaluation after an accepted switch (<= S+1 rows per haplotype): lane = row, consecutive windows in consecutive lanes
//    (rank gathers of lanes on the same node are consecutive halfwords); the block holds THREADS / rows such row sets, each on a
//    class of its own with that class's trees staged in LDS; a lane adds its leaves in tree order.
//  * A candidate's two ORIGINAL rows are rows the smoother has already evaluated: the scope [center-37, center+37] is exactly what row
//    center+1 of slide_window sees (no reflection: center is clipped to [37, W-38]), so their max-probabilities come from a
//    per-row cache (k_gnofix_pmax from the initial smoother pass, kept current through swaps and re-evaluations) and only the two
//    SWITCHED rows are walked: 7 instead of 13 cache-line requests per tree — the candidate phase is bound by the L1's line rate.
//  * Work per individual varies by an order of magnitude (0 .. 13 accepted switches in config 5b) and a launch ends with its slowest
//    individual: the block is 512 threads (two per CU), i.e. the critical path of ONE individual is what is kept short.
//  * The three phases that touch X are kernels of their own, all of the chip on each: k_gnofix_dif ("does this window's SNP block
//    differ between the two haplotypes", the convergence signature's other half) before, k_gnofix_swap (phasing.py:188-198 applied
//    once from the final parity) after.  Inside the per-individual kernel they ran at one CU's 25 GB/s.
//  * The next label change is a find-first-set over a bit mask kept beside the labels (no block barrier per search).
//  * Where accepted switches come thick (a label change at nearly every window: a chaotic smoother on unstructured haplotypes) the
//    re-evaluation is LAZY: a switch marks its rows dirty, the scan brings rows up to date only when it reads them (short batches:
//    one row set per class), the rest at the start of the next sweep.  Same decisions in the same order; see the sweep loop.
#include "gnx_internal.h"
#include "gnx_rank.h"
#include "gnx_exp.h"

namespace {

#define GNX_NOUNROLL _Pragma("clang loop unroll(disable)")
// a value the optimiser cannot see through: what is computed from it stays where it is written (per-thread index arithmetic of
// every phase hoisted out of the sweep / candidate loops cost ~100 VGPRs held for the whole kernel)
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

constexpr int NWR = 8;       // re-evaluation: trees walked side by side per lane (chains of LDS reads)

// ---- 16 SNPs of a haplotype row at any byte address (gfx950 global memory takes unaligned dwordx4; the rows of an individual are
// ldx bytes apart with no alignment promise) ----
struct __attribute__((packed, aligned(1))) snp16 { uint32_t x, y, z, w; };
__device__ __forceinline__ snp16 ld16(const int8_t* p) { snp16 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st16(int8_t* p, const snp16& v) { __builtin_memcpy(p, &v, 16); }
__device__ __forceinline__ bool neq(const snp16& a, const snp16& b) { return ((a.x ^ b.x) | (a.y ^ b.y) | (a.z ^ b.z) | (a.w ^ b.w)) != 0; }

// ---- pre-pass 1: base probabilities -> ranks (float32 cast first: Smooth/utils.py:20) ----
__global__ __launch_bounds__(256) void k_gnofix_ranks(const double* __restrict__ B, int64_t n, const float* __restrict__ U,
                                                      const uint32_t* __restrict__ lut, int K, int steps, uint16_t* __restrict__ R) {
  constexpr int NV = 4;
  const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * NV;
  if (e0 >= n) return;
  float p[NV];
  uint32_t r[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) p[i] = (float)B[min(e0 + i, n - 1)];
  ranks<NV>(U, lut, K, steps, p, r);
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (e0 + i < n) R[e0 + i] = (uint16_t)r[i];
}

// ---- pre-pass 1b: largest probability of every row of the initial smoother pass ----
__global__ __launch_bounds__(256) void k_gnofix_pmax(const float* __restrict__ P, int64_t rows, int A, float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float* p = P + r * A;
  float m = p[0];
  for (int a = 1; a < A; ++a) m = fmaxf(m, p[a]);
  out[r] = m;
}

// ---- pre-pass 1c: longest-first dispatch order.  An individual's work grows with its label changes (every sweep evaluates a
// candidate at each) and a launch ends with its slowest block: individuals are handed to the per-individual kernel in descending
// order of their initial change count (counting sort: histogram, descending prefix, scatter; ties in any order — every individual's
// result is independent of the order). ----
__global__ __launch_bounds__(256) void k_gnofix_count(const int32_t* __restrict__ Y0, int W, int32_t* __restrict__ cnt, int32_t* __restrict__ hist) {
  const int64_t ind = blockIdx.x;
  const int32_t* ym = Y0 + (size_t)2 * ind * W;
  const int32_t* yp = ym + W;
  int c = 0;
  for (int u = 1 + threadIdx.x; u < W; u += 256) c += (ym[u] != ym[u - 1] || yp[u] != yp[u - 1]) ? 1 : 0;
  __shared__ int tot;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  atomicAdd(&tot, c);
  __syncthreads();
  if (threadIdx.x == 0) { cnt[ind] = tot; atomicAdd(&hist[tot], 1); }
}
__global__ __launch_bounds__(256) void k_gnofix_scan(const int32_t* __restrict__ hist, int W, int32_t* __restrict__ start) {
  // descending exclusive prefix over the W + 1 bins by one block: thread t owns a run of bins, the runs' totals are scanned by thread 0
  __shared__ int part[256];
  const int per = (W + 1 + 255) / 256, t = threadIdx.x;
  const int hi = W - t * per, lo = max(hi - per + 1, 0);  // bins hi, hi-1, .. lo
  int sum = 0;
  for (int c = hi; c >= lo; --c) sum += hist[c];
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int k = 0; k < 256; ++k) { const int v = part[k]; part[k] = acc; acc += v; }
  }
  __syncthreads();
  int acc = part[t];
  for (int c = hi; c >= lo; --c) { start[c] = acc; acc += hist[c]; }
}
__global__ __launch_bounds__(256) void k_gnofix_scatter(const int32_t* __restrict__ cnt, int64_t n, int32_t* __restrict__ start, int32_t* __restrict__ order) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) order[atomicAdd(&start[cnt[i]], 1)] = (int32_t)i;
}

// ---- pre-pass 2: per individual and window, "the SNP block differs between the two haplotypes" (gnofix.py:108-113 compares whole
// X_m vectors; with the switch parity, X_m(a) == X_m(b) iff parity_a & dif == parity_b & dif).  Window u covers SNPs
// [u*ws, (u+1)*ws), ws = C // W, the last one up to C (gnofix.py:74, phasing.py:192).  One wave = one 32-bit word of the mask: four
// lanes per window, 64 bytes per step, stopping at the first difference. ----
__global__ __launch_bounds__(256) void k_gnofix_dif(const int8_t* __restrict__ X, int64_t ldx, int64_t C, int W, uint32_t* __restrict__ dif) {
  const int NWD = (W + 31) / 32;
  const int ln = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= NWD) return;  // wave-uniform
  const int64_t ind = blockIdx.y;
  const int8_t* Xm = X + 2 * ind * ldx;
  const int8_t* Xp = Xm + ldx;
  const int64_t ws = C / W;
  uint32_t word = 0;
  for (int hf = 0; hf < 2; ++hf) {
    const int u = q * 32 + hf * 16 + (ln >> 2), sub = ln & 3;
    const bool live = u < W;
    const int64_t j0 = live ? (int64_t)u * ws : 0, j1 = !live ? 0 : (u == W - 1) ? C : j0 + ws;
    bool d = false;
    for (int64_t j = j0;; j += 64) {
      const int64_t a0 = j + sub * 16, a1 = min(a0 + 16, j1);
      if (!d && a0 < j1) {
        if (a1 - a0 == 16) d = neq(ld16(Xm + a0), ld16(Xp + a0));
        else
          for (int64_t i = a0; i < a1; ++i) d |= Xm[i] != Xp[i];
      }
      const unsigned long long bal = __ballot(d);
      d = ((bal >> (ln & ~3)) & 0xfull) != 0;  // the window's four lanes agree
      if (__ballot(!d && j + 64 < j1) == 0) break;
    }
    const unsigned long long bal = __ballot(d && sub == 0);  // window k of this half at bit 4k
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) bits |= (uint32_t)((bal >> (4 * k)) & 1ull) << k;
    word |= bits << (16 * hf);
  }
  if (ln == 0) dif[(size_t)ind * NWD + q] = word;
}

// j / ws without the 32-bit division (two of them per word made k_gnofix_swap_p2 VALU-bound: ~25 instructions each):
// q = umulhi(j, ceil(2^32 / ws)) is j / ws or one more, exact after one comparison for every j < 2^32
struct WsDiv {
  uint32_t ws, inv;
  __device__ __forceinline__ uint32_t operator()(uint32_t j) const {
    if (inv == 0u) return j;  // ws == 1 (one SNP per window): ceil(2^32 / 1) does not fit 32 bits
    uint32_t q = __umulhi(j, inv);
    return q - (q * ws > j ? 1u : 0u);
  }
};
__host__ inline uint32_t gnx_ws_inv(uint32_t ws) { return ws <= 1u ? 0u : (uint32_t)((((uint64_t)1 << 32) + ws - 1) / ws); }  // 0: WsDiv is the identity

// ---- post-pass: correct_phase_error applied once from the final parity (phasing.py:188-198): windows of odd parity exchange their
// SNP blocks.  grid = (16 KB pieces of the chromosome, individuals); a block whose windows are all even leaves without a load. ----
__global__ __launch_bounds__(256) void k_gnofix_swap(int8_t* __restrict__ X, int64_t ldx, int64_t C, int W, const uint32_t* __restrict__ par, uint32_t ws, uint32_t ws_inv) {
  constexpr int PER = 4;  // 16-byte pieces per thread
  const int NWD = (W + 31) / 32;
  const int64_t ind = blockIdx.y;
  const uint32_t* P = par + (size_t)ind * NWD;
  const WsDiv wdiv{ws, ws_inv};
  const int64_t b0 = (int64_t)blockIdx.x * (256 * PER * 16), b1 = min(b0 + 256 * PER * 16, C);
  const int ua = (int)min((uint32_t)(W - 1), wdiv((uint32_t)b0)), ub = (int)min((uint32_t)(W - 1), wdiv((uint32_t)(b1 - 1)));  // C < 2^31 (launcher)
  bool any = false;
  for (int q = ua >> 5; q <= ub >> 5; ++q) {  // block-uniform
    uint32_t m = P[q];
    if (q == ua >> 5) m &= 0xffffffffu << (ua & 31);
    if (q == ub >> 5) m &= 0xffffffffu >> (31 - (ub & 31));
    any |= m != 0;
  }
  if (!any) return;
  int8_t* Xm = X + 2 * ind * ldx;
  int8_t* Xp = Xm + ldx;
  auto odd = [&](int u) { return ((P[u >> 5] >> (u & 31)) & 1u) != 0; };
  snp16 xa[PER], xb[PER];
  int mode[PER];  // 0 nothing, 1 whole piece, 2 byte by byte (a window boundary or the chromosome's end inside the piece)
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t j = b0 + ((int64_t)k * 256 + threadIdx.x) * 16;
    mode[k] = 0;
    if (j >= C) continue;
    const int u0 = (int)min((uint32_t)(W - 1), wdiv((uint32_t)j)), u1 = (int)min((uint32_t)(W - 1), wdiv((uint32_t)min(j + 15, C - 1)));
    if (j + 16 <= C && (u0 == u1 || (u1 == u0 + 1 && odd(u0) == odd(u1)))) mode[k] = odd(u0) ? 1 : 0;
    else mode[k] = 2;
    if (mode[k] == 1) { xa[k] = ld16(Xm + j); xb[k] = ld16(Xp + j); }
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t j = b0 + ((int64_t)k * 256 + threadIdx.x) * 16;
    if (mode[k] == 1) { st16(Xm + j, xb[k]); st16(Xp + j, xa[k]); }
    else if (mode[k] == 2) {
      for (int64_t i = j; i < min(j + 16, C); ++i)
        if (odd((int)min((uint32_t)(W - 1), wdiv((uint32_t)i)))) { const int8_t t = Xm[i]; Xm[i] = Xp[i]; Xp[i] = t; }
    }
  }
}

// Where a walk reads its trees from.  Global copy: buffer loads through a 128-bit descriptor held in SGPRs — a 32-bit word offset
// per chain instead of a 64-bit address pair (flat loads made the candidate's 20 chains cost 235 VGPRs).  LDS stage: plain reads.
struct TreesGlobal {
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ uint32_t u32(uint32_t word) const { return __builtin_amdgcn_raw_buffer_load_b32(rs, word * 4u, 0, 0); }
  __device__ __forceinline__ uint4 u128(uint32_t word) const {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, word * 4u, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
  }
};
struct TreesLds {
  const uint32_t* p;
  __device__ __forceinline__ uint32_t u32(uint32_t word) const { return p[word]; }
  __device__ __forceinline__ uint4 u128(uint32_t word) const { return *reinterpret_cast<const uint4*>(p + word); }
};

// One level of a rank walk in the instructions the hardware has for it (hipcc spends 6-7 VALU operations on the same C++: shift, mask,
// compare, select, add): the rank's LDS address = row + low half of the node word (one SDWA add), the branch = "rank field <= rank"
// straight from the node word's high half (SDWA compare into vcc), the heap index j = 2j + branch (add-with-carry from vcc).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t rank_addr(uint32_t nd, uint32_t row) {
  uint32_t a;
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(a) : "v"(nd), "v"(row));
  return a;
}
__device__ __forceinline__ uint32_t step_j(uint32_t j, uint32_t nd, uint32_t r) {
  asm("v_cmp_le_u32_sdwa vcc, %1, %2 src0_sel:WORD_1 src1_sel:DWORD\n\t"
      "v_addc_co_u32 %0, vcc, %0, %0, vcc"
      : "+v"(j) : "v"(nd), "v"(r) : "vcc");
  return j;
}
// levels 0 and 1 from the tree's first 16 bytes: j = 2 + branch, the level-1 node = branch ? right : left
__device__ __forceinline__ void step_top(uint32_t root, uint32_t lo, uint32_t hi, uint32_t r, uint32_t& j, uint32_t& n1) {
  asm("v_cmp_le_u32_sdwa vcc, %2, %3 src0_sel:WORD_1 src1_sel:DWORD\n\t"
      "v_cndmask_b32 %1, %4, %5, vcc\n\t"
      "v_addc_co_u32_e64 %0, vcc, 1, 1, vcc"
      : "=&v"(j), "=&v"(n1) : "v"(root), "v"(r), "v"(lo), "v"(hi) : "vcc");
}
__device__ __forceinline__ uint32_t lds_rank(uint32_t addr) { return *(const __attribute__((address_space(3))) uint16_t*)(uintptr_t)addr; }
#else   // host pass: never called
__device__ inline uint32_t rank_addr(uint32_t, uint32_t) { return 0; }
__device__ inline uint32_t step_j(uint32_t j, uint32_t, uint32_t) { return j; }
__device__ inline void step_top(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t& j, uint32_t& n1) { j = n1 = 0; }
__device__ inline uint32_t lds_rank(uint32_t) { return 0; }
#endif

// Tree layout of k_gnofix (SmoothXGBDev::gf_packed): 2^D node words in heap order (slot 0 unused; rank field << 16 | byte offset of
// the feature in the tile) followed by 2^D float leaves — the leaf of heap index j is word j.  Words 0..3 = {-, root, node 2, node 3}
// arrive in ONE 16-byte read, so levels 0 and 1 cost one dependent round trip (D >= 2): a depth-4 walk is 4 of them.
// NW trees t0, t0+1, .. side by side on ONE row (T = global memory or the LDS stage, row = LDS address of the row's tile origin).
// With CLAMP the tree index stops at tmax (LDS stage); without, trees past the caller's range are walked and dropped by the caller
// (the global copy is followed by GNX_GF_PAD_TREES zero trees), so every address is base + immediate.
template <int NW, int DT, bool CLAMP, typename Trees>
__device__ __forceinline__ void walk_seq(const Trees T, uint32_t t0, uint32_t tmax, const uint8_t* row, int Drt, float (&out)[NW]) {
  const int D = DT ? DT : Drt;
  const uint32_t TWc = 2u << D;
  const uint32_t rowa = (uint32_t)(uintptr_t)row;  // (the low 32 bits of a generic pointer into the LDS are its LDS address)
  uint32_t j[NW];
  uint32_t tb[NW];
#pragma unroll
  for (int k = 0; k < NW; ++k) tb[k] = (CLAMP ? min(t0 + (uint32_t)k, tmax) : t0 + (uint32_t)k) * TWc;
  auto level = [&]() {
    uint32_t nd[NW], r[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) nd[k] = T.u32(tb[k] + j[k]);
#pragma unroll
    for (int k = 0; k < NW; ++k) r[k] = lds_rank(rank_addr(nd[k], rowa));
#pragma unroll
    for (int k = 0; k < NW; ++k) j[k] = step_j(j[k], nd[k], r[k]);
  };
  if (D >= 2) {
    uint4 top[NW];
    uint32_t r[NW], n1[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) top[k] = T.u128(tb[k]);
#pragma unroll
    for (int k = 0; k < NW; ++k) r[k] = lds_rank(rank_addr(top[k].y, rowa));
#pragma unroll
    for (int k = 0; k < NW; ++k) step_top(top[k].y, top[k].z, top[k].w, r[k], j[k], n1[k]);
#pragma unroll
    for (int k = 0; k < NW; ++k) r[k] = lds_rank(rank_addr(n1[k], rowa));
#pragma unroll
    for (int k = 0; k < NW; ++k) j[k] = step_j(j[k], n1[k], r[k]);
    if constexpr (DT > 0) {
#pragma unroll
      for (int d = 2; d < DT; ++d) level();
    } else {
      for (int d = 2; d < D; ++d) level();
    }
  } else {
#pragma unroll
    for (int k = 0; k < NW; ++k) j[k] = 1;
    for (int d = 0; d < D; ++d) level();
  }
#pragma unroll
  for (int k = 0; k < NW; ++k) out[k] = __uint_as_float(T.u32(tb[k] + j[k]));
}

// Candidate: NT_ trees t0, t0 + STRIDE, .. of the global copy, each on the NR = 2 candidate rows (LDS addresses row + roff[r]); out[k*NR + r].
template <int NT_, int STRIDE, int DT, typename Trees>
__device__ __forceinline__ void walk4(const Trees T, uint32_t t0, const uint8_t* row, const uint32_t (&roff)[2], int Drt,
                                      float* out) {
  const int D = DT ? DT : Drt;
  const uint32_t TWc = 2u << D;
  const uint32_t rowa[2] = {(uint32_t)(uintptr_t)row + roff[0], (uint32_t)(uintptr_t)row + roff[1]};
  uint32_t j[NT_ * 2];
  uint32_t tw[NT_];
#pragma unroll
  for (int k = 0; k < NT_; ++k) tw[k] = (t0 + (uint32_t)(k * STRIDE)) * TWc;
  auto level = [&]() {
    uint32_t nd[NT_ * 2], r[NT_ * 2];
#pragma unroll
    for (int q = 0; q < NT_ * 2; ++q) nd[q] = T.u32(tw[q >> 1] + j[q]);
#pragma unroll
    for (int q = 0; q < NT_ * 2; ++q) r[q] = lds_rank(rank_addr(nd[q], rowa[q & 1]));
#pragma unroll
    for (int q = 0; q < NT_ * 2; ++q) j[q] = step_j(j[q], nd[q], r[q]);
  };
  if (D >= 2) {
    uint4 top[NT_];
#pragma unroll
    for (int k = 0; k < NT_; ++k) top[k] = T.u128(tw[k]);
    uint32_t r[NT_ * 2], n1[NT_ * 2];
#pragma unroll
    for (int q = 0; q < NT_ * 2; ++q) r[q] = lds_rank(rank_addr(top[q >> 1].y, rowa[q & 1]));
#pragma unroll
    for (int q = 0; q < NT_ * 2; ++q) step_top(top[q >> 1].y, top[q >> 1].z, top[q >> 1].w, r[q], j[q], n1[q]);
#pragma unroll
    for (int q = 0; q < NT_ * 2; ++q) r[q] = lds_rank(rank_addr(n1[q], rowa[q & 1]));
#pragma unroll
    for (int q = 0; q < NT_ * 2; ++q) j[q] = step_j(j[q], n1[q], r[q]);
    if constexpr (DT > 0) {
#pragma unroll
      for (int d = 2; d < DT; ++d) level();
    } else {
      for (int d = 2; d < D; ++d) level();
    }
  } else {
#pragma unroll
    for (int q = 0; q < NT_ * 2; ++q) j[q] = 1;
    for (int d = 0; d < D; ++d) level();
  }
#pragma unroll
  for (int q = 0; q < NT_ * 2; ++q) out[q] = __uint_as_float(T.u32(tw[q >> 1] + j[q]));
}

__host__ __device__ inline int gnofix_rows_max(int S, int threads) { return 2 * min(S + 2, threads / 2); }

struct GnofixLds {
  size_t seg, Y, pmax, par, dif, chg, rej, dirty, marg, ex, stage, flags, ct0, total;
};
__host__ __device__ inline GnofixLds gnofix_lds(int W, int A, int S, int GP, int cap, int D, int threads, int n_trees) {
  auto r16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const size_t NWD = (size_t)(W + 31) / 32, nrow = (size_t)gnofix_rows_max(S, threads);
  GnofixLds o{};
  size_t off = 0;
  o.seg = off; off += r16((size_t)2 * A * GP * 2);
  o.Y = off; off += r16((size_t)W * 2 + 16);
  o.pmax = off; off += r16((size_t)2 * W * 4);
  o.par = off; off += r16(NWD * 4);
  o.dif = off; off += r16(NWD * 4);
  o.chg = off; off += r16(NWD * 4 + 4);
  o.rej = off; off += r16(NWD * 4 + 4);
  o.dirty = off; off += r16(NWD * 4 + 4);
  o.marg = off; off += r16(nrow * A * 4);
  o.ex = off; off += r16((size_t)(threads / 64) * 2 * A * 4);
  {  // re-evaluation: `cap` staged trees for each of the block's row sets; candidate: the leaves [2][n_trees] (never both)
    const size_t nset = (size_t)(threads / (int)nrow) > 0 ? (size_t)(threads / (int)nrow) : 1;
    const size_t st = nset * cap * gnx_gf_tree_words(D) * 4, lb = (size_t)2 * n_trees * 4;
    o.stage = off; off += r16(st > lb ? st : lb);
  }
  o.flags = off; off += 256;
  o.ct0 = off; off += r16((size_t)(A + 1) * 4);
  o.total = off;
  return o;
}

// what the per-individual kernel reads of a GnofixLaunch (a kernel argument block of 100 bytes instead of 400: fewer SGPRs held)
struct GnofixK {
  const uint16_t* R;
  const uint32_t* dif;
  uint32_t* par;
  const uint32_t* gf;
  const int32_t* class_tree0;
  const int32_t* Y0;
  const float* P0;   // (2n, W) largest probability per row of the initial smoother pass
  const int32_t* order;  // [n] individual handled by block b
  int32_t* Yout;
  int32_t* n_switches;
  uint32_t* hist;
  int32_t W, A, S, max_it, D, NT, GP, cap;
  float base_score;
};

template <int THREADS, int DT>
__global__ __launch_bounds__(THREADS, (THREADS >= 512 ? 4 : 3)) void k_gnofix(GnofixK L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int PER_T = THREADS >= 1024 ? 2 : THREADS >= 512 ? 3 : 5;  // candidate: most trees one lane walks (on the four rows side by side: chains of L2 loads) per round
  constexpr int PF_MAX = THREADS >= 512 ? 4 : 8;                       // candidate: tile elements per thread that are fetched one candidate ahead
  const int W = L.W, A = L.A, S = L.S, pad = (S + 1) / 2, half = (S - 1) / 2;
  const int D = DT ? DT : L.D, NWD = (W + 31) / 32, GP = L.GP, TW = gnx_gf_tree_words(D), NT = L.NT;
  const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
  const int64_t ind = __builtin_amdgcn_readfirstlane(L.order[blockIdx.x]);   // (scalar: what is derived from it stays out of the vector registers)
  const GnofixLds o = gnofix_lds(W, A, S, GP, L.cap, D, THREADS, NT);
  uint16_t* seg = reinterpret_cast<uint16_t*>(lds + o.seg);   // [2][A][GP] ranks: the tile every walk reads
  uint16_t* Y = reinterpret_cast<uint16_t*>(lds + o.Y);       // labels: maternal | paternal << 8 per window
  float* pmax = reinterpret_cast<float*>(lds + o.pmax);       // [W][2] largest smoother probability of (window, haplotype)'s row
  uint32_t* par = reinterpret_cast<uint32_t*>(lds + o.par);   // switch parity per window
  uint32_t* dif = reinterpret_cast<uint32_t*>(lds + o.dif);   // SNP block differs m vs p
  uint8_t* chg = lds + o.chg;                                 // bit u: the labels of window u differ from window u-1's
  uint32_t* rej = reinterpret_cast<uint32_t*>(lds + o.rej);   // bit u: the candidate at u was evaluated and rejected, and nothing it reads has changed since
  float* marg = reinterpret_cast<float*>(lds + o.marg);       // candidate: [2][A]; re-evaluation: [A][NROW]
  float* ex = reinterpret_cast<float*>(lds + o.ex);           // candidate: exp(margin - row max) [2][A], one copy per wave
  uint32_t* stage = reinterpret_cast<uint32_t*>(lds + o.stage);  // re-evaluation: staged trees of one class
  int* ct0s = reinterpret_cast<int*>(lds + o.ct0);            // class_tree0 (A + 1 entries)
  int* flags = reinterpret_cast<int*>(lds + o.flags);         // [0] accept, [2],[3] rows to re-evaluate [r0, r1), [8..40) past sweeps that differ
  uint32_t* hist = L.hist + (size_t)ind * L.max_it * NWD;
  const uint16_t* __restrict__ R0 = L.R + (size_t)2 * ind * W * A;  // physical haplotype h at + h*W*A
  const size_t WA = (size_t)W * A;
  const uint32_t invA = 0xFFFFFFFFu / (uint32_t)A + 1u;       // n / A = umulhi(n, invA) for n < 65536 (A <= 32)
  const int NROW = gnofix_rows_max(S, THREADS), NSETMAX = max(1, THREADS / NROW);
  int nmax = 0;  // most trees of one class
  for (int c = 0; c < A; ++c) nmax = max(nmax, L.class_tree0[c + 1] - L.class_tree0[c]);
  if (threadIdx.x <= A) ct0s[threadIdx.x] = L.class_tree0[threadIdx.x];  // (published by the barrier of the load phase)
#ifdef GNX_GNOFIX_CLOCKS  // development aid (scripts/dev/gnofix_phases.py): individual i reports phase (i & 7) in n_switches, units of 64 clocks
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
#define TICK(i) { const long long tn = clock64(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define TICK(i)
#endif

  // ---- load: labels, masks ----
  for (int u = tid; u < W; u += THREADS) {
    Y[u] = (uint16_t)(L.Y0[(size_t)2 * ind * W + u] | (L.Y0[(size_t)(2 * ind + 1) * W + u] << 8));
    pmax[2 * u] = L.P0[(size_t)2 * ind * W + u];
    pmax[2 * u + 1] = L.P0[(size_t)(2 * ind + 1) * W + u];
  }
  for (int q = tid; q < NWD; q += THREADS) { par[q] = 0; rej[q] = 0; dif[q] = L.dif[(size_t)ind * NWD + q]; }
  __syncthreads();
  auto parbit = [&](int u) -> int { return (int)((par[u >> 5] >> (u & 31)) & 1u); };
  auto mark_changes = [&](int t_) {  // one byte of the mask (8 windows) per thread
    GNX_NOUNROLL for (int b = t_; b < NWD * 4; b += THREADS) {
      uint32_t m = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int u = b * 8 + k;
        if (u >= 1 && u < W && Y[u] != Y[u - 1]) m |= 1u << k;
      }
      chg[b] = (uint8_t)m;
    }
  };
  mark_changes(tid);
  __syncthreads();
  // check(): "disc_smooth" (gnofix.py:32): the reference walks w = 1 .. W-1 and acts only where a label changes; every wave finds
  // the next such window for itself (same answer in every wave: no barrier).
  // A candidate's decision is a function of the ranks of its scope [lo, lo + S) under the current parity, of the cached
  // probabilities of row center + 1 and of w.  An accepted switch at w' changes none of them for a candidate whose scope lies
  // entirely below w' (untouched) or entirely at / above w' (both strips and both cached rows exchange their roles: the switched
  // pair is the same two rows in the other order, max over the pair unchanged).  So a candidate that was REJECTED stays rejected until a
  // switch within S windows of it is accepted: `rej` remembers it and later sweeps (the reference re-evaluates every label change in
  // every sweep, gnofix.py:93-116) skip the walk — the same decisions, a third to a half of the candidate evaluations.
  auto next_change = [&](int from) -> int {
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(chg);
    const int lq = opaque(ln);  // (keeps the three per-lane mask addresses out of the registers held for the whole kernel)
    for (int q0 = from >> 5; q0 < NWD; q0 += 64) {
      const int q = q0 + lq;
      uint32_t m = q < NWD ? (cw[q] & ~rej[q]) : 0u;
      if (q == from >> 5) m &= 0xffffffffu << (from & 31);
      const unsigned long long bal = __ballot(m != 0);
      if (bal) {
        const int first = __builtin_ctzll(bal);
        const uint32_t mw = (uint32_t)__shfl((int)m, first);
        return (q0 + first) * 32 + __builtin_ctz(mw);
      }
    }
    return W;
  };

  // the candidate's two rows = the switched pair, at tile positions 0..S-1 of the two halves
  const uint32_t roff[2] = {0u, (uint32_t)(A * GP * 2)};
  const uint8_t* crow = reinterpret_cast<const uint8_t*>(seg);
  const uint32_t* __restrict__ GTp = L.gf;
  const TreesGlobal GT{__builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(L.gf), 0, (uint32_t)((NT + GNX_GF_PAD_TREES) * TW * 4), 0x00020000)};  // (reads past the end return 0: the hardware's bounds check)

  int n_switch = 0;
  // Re-evaluation is LAZY where switches come thick (round 4).  The reference calls smoother.predict(B) after every accepted switch
  // (gnofix.py:157-176); what the loop reads of that result before the next accepted switch is only (a) the labels up to the next label
  // change and (b) the cached probabilities of row center + 1 of the next candidate.  A switch within DENSE_GAP windows of the
  // previous one of the same sweep therefore only MARKS its rows [r0, r1) dirty; the scan cleans rows as it needs them, CLEAN_WIN
  // windows at a time — a SHORT batch of the same re-evaluation code: with few rows every class gets a row set of its own (the stage
  // is cut into A slices of cap' trees), so it costs one class's walks instead of A / 3 classes' — and whatever is still dirty is
  // evaluated in full batches at the start of the next sweep / before the outputs.  A row's values depend on the strips and the
  // parity only, so WHEN it is evaluated changes nothing it evaluates to: same decisions, same labels (tests: G5, fuzz, phase_gt2).
  // With a label change at nearly every window (random trees on unstructured haplotypes) a switch costs one short batch instead of
  // 2 (S - 1) rows x n_trees walks.  Isolated switches keep the immediate full batch.
  // (measured on MI355X, worst case / config 5b: 8 / 8 / 3 -> 405 ms / 6.46 ms; 12 / 12 / 2 -> 401 / 6.66; 12 / 24 / 2 -> 452 / 6.69;
  //  36 / 36 / 1 -> 469 / 6.75; 8 / 4 / 4 -> 452 / 6.63; never lazy (round 4's first version) -> 619 / 6.51)
  constexpr int DENSE_GAP = 8, CLEAN_WIN = 8, CLEANS_PER_SWITCH = 3;
  const int gstep = NROW / 2;
  uint32_t* dirty = reinterpret_cast<uint32_t*>(lds + o.dirty);
  for (int q = tid; q < NWD; q += THREADS) dirty[q] = 0;
  __syncthreads();
  auto first_dirty = [&](int from) -> int {  // first dirty window >= from (every wave for itself)
    const int lq = opaque(ln);
    for (int q0 = from >> 5; q0 < NWD; q0 += 64) {
      const int q = q0 + lq;
      uint32_t m = q < NWD ? dirty[q] : 0u;
      if (q == from >> 5) m &= 0xffffffffu << (from & 31);
      const unsigned long long bal = __ballot(m != 0);
      if (bal) {
        const int first = __builtin_ctzll(bal);
        const uint32_t mw = (uint32_t)__shfl((int)m, first);
        return min(W, (q0 + first) * 32 + __builtin_ctz(mw));
      }
    }
    return W;
  };
  auto set_dirty = [&](int lo_, int hi_, bool on, int t_) {  // windows [lo_, hi_)
    GNX_NOUNROLL for (int q = t_; q < NWD; q += THREADS) {
      const int b0 = q * 32;
      if (b0 + 32 > lo_ && b0 < hi_) {
        uint32_t m = 0xffffffffu;
        if (lo_ > b0) m &= 0xffffffffu << (lo_ - b0);
        if (hi_ < b0 + 32) m &= 0xffffffffu >> (b0 + 32 - hi_);
        if (on) dirty[q] |= m; else dirty[q] &= ~m;
      }
    }
  };
  bool any_dirty = false;
  TICK(0)
  for (int it = 0;; ++it) {
    const int ty = opaque(tid);
    bool finishing = it >= L.max_it;
    if (!finishing) {
      // ---- convergence: has this X_m been seen at the start of an earlier sweep? (gnofix.py:108-113) ----
      bool seen = false;
      for (int k0 = 0; k0 < it; k0 += 1024) {  // 1024 past sweeps at a time: one "differs" bit each
        const int nk = min(1024, it - k0);
        if (ty < 32) flags[8 + ty] = 0;
        __syncthreads();
        GNX_NOUNROLL for (int e = ty; e < nk * NWD; e += THREADS) {
          const int k = e / NWD, q = e - k * NWD;
          if (hist[(size_t)(k0 + k) * NWD + q] != (par[q] & dif[q])) atomicOr(&flags[8 + (k >> 5)], 1 << (k & 31));
        }
        __syncthreads();
        for (int k = 0; k < nk; k += 32) {
          uint32_t m = ~(uint32_t)flags[8 + (k >> 5)];
          if (nk - k < 32) m &= (1u << (nk - k)) - 1u;
          seen |= m != 0;
        }
        __syncthreads();
      }
      if (seen) finishing = true;
      else { GNX_NOUNROLL for (int q = ty; q < NWD; q += THREADS) hist[(size_t)it * NWD + q] = par[q] & dif[q]; }
    }
    if (finishing && !any_dirty) break;
    TICK(1)

    // The tile elements of a candidate are per-thread values (e = tid + i * THREADS): when they fit PF_MAX registers, the NEXT
    // candidate's are requested before this one's walks start (a rejected candidate — most are — changes neither the labels nor
    // the parity, so the next change and its ranks are already known) and the L2 round trip hides behind the walks.
    const bool pf_fits = 2 * S * A <= PF_MAX * THREADS;
    uint16_t pfv[PF_MAX];
    bool pf_valid = false;
    auto cand_fetch = [&](int wq, int t_) {
      const int cq = min(max(wq, half), W - 1 - half), loq = cq - half;
#pragma unroll
      for (int i = 0; i < PF_MAX; ++i) {
        const int e = min(t_ + i * THREADS, 2 * S * A - 1);  // clamped: unconditional loads
        const int h = e >= S * A ? 1 : 0, f = e - h * S * A;
        const int sq = (int)__umulhi((uint32_t)f, invA), a = f - sq * A;
        const int u = loq + sq;
        pfv[i] = R0[(size_t)(h ^ parbit(u)) * WA + (size_t)u * A + a];
      }
    };
    int pf_w = -1;
    int scan_from = 1, last_acc = -(1 << 20), cleans_left = 0;
    bool big_pending = false, flushing = any_dirty;
    int bp_lo = 0, bp_hi = 0;
    if (flushing) { bp_lo = __builtin_amdgcn_readfirstlane(first_dirty(0)); bp_hi = min(W, bp_lo + gstep); big_pending = bp_lo < W; flushing = big_pending; any_dirty = big_pending; }
    while (true) {
      if (big_pending) {  // ---- rows [bp_lo, bp_hi) in batches of up to gstep windows x 2 haplotypes, class by class from staged trees ----
        const int tz = opaque(tid);
        __syncthreads();
      const int blo = __builtin_amdgcn_readfirstlane(bp_lo), bhi = __builtin_amdgcn_readfirstlane(bp_hi);  // (block-uniform: scalar registers for what derives from them)
      for (int gb = blo; gb < bhi; gb += gstep) {
        const int nwin = min(gstep, bhi - gb), nrow = 2 * nwin, nj = nwin + S - 1;  // rows read padded windows [gb, gb + nj)
        GNX_NOUNROLL for (int e = tz; e < 2 * nj * A; e += THREADS) {
          const int h = e >= nj * A ? 1 : 0, f = e - h * nj * A;
          const int q = (int)__umulhi((uint32_t)f, invA), a = f - q * A;
          const int u = slide_src(gb + q, W, pad);
          seg[(h * A + a) * GP + q] = R0[(size_t)(h ^ parbit(u)) * WA + (size_t)u * A + a];
        }
        // lane = row (haplotype rh, window gb + rk) of row set `set`; set s walks class c0 + s.  The stage holds NSETMAX * cap trees:
        // a full batch has NSETMAX sets of `cap` trees, a short one up to A sets of capE (a multiple of NWR) — more classes side by
        // side, more staging rounds per class.
        const int nset = min(min(A, THREADS / nrow), max(NSETMAX, NSETMAX * L.cap / NWR));
        const int capE = nset <= NSETMAX ? L.cap : max(NWR, (NSETMAX * L.cap / nset) / NWR * NWR);
        const int set = tz / nrow, rr = tz - set * nrow;
        const bool rlive = set < nset;
        const int rh = rr >= nwin ? 1 : 0, rk = rr - rh * nwin;
        const uint8_t* rrow = reinterpret_cast<const uint8_t*>(seg) + ((size_t)rh * A * GP + rk) * 2;
        const uint32_t* mystage = stage + (size_t)(rlive ? set : 0) * capE * TW;
        GNX_NOUNROLL for (int c0 = 0; c0 < A; c0 += nset) {
          const int c = c0 + set;
          const bool clive = rlive && c < A;
          const int t0 = clive ? ct0s[c] : 0, cn = clive ? ct0s[c + 1] - t0 : 0;
          float ps = 0.f;
          GNX_NOUNROLL for (int k0 = 0; k0 < nmax; k0 += capE) {
            __syncthreads();  // the tile is complete / the previous chunk has been walked
            {  // the chunk [k0, k0 + cap) of the classes c0 .. c0 + nset - 1, one after the other in the stage: all of a thread's
               // 16-byte pieces are requested before the first is stored (one L2 round trip per chunk, not one per piece)
              constexpr int NST = 6;
              const int per_set = capE * (TW / 4), total = nset * per_set;
              const uint4* src = reinterpret_cast<const uint4*>(GTp);
              uint4* dst = reinterpret_cast<uint4*>(stage);
              GNX_NOUNROLL for (int g0 = 0; g0 < total; g0 += NST * THREADS) {
                uint4 v[NST];
                bool ok[NST];
#pragma unroll
                for (int k = 0; k < NST; ++k) {
                  const int g = g0 + tz + k * THREADS;
                  const int sq = g / per_set, wi = g - sq * per_set, cq = c0 + sq;
                  ok[k] = g < total && cq < A;
                  const int ts = ok[k] ? ct0s[cq] + k0 : 0;
                  const int n_s = ok[k] ? max(0, min(capE, ct0s[cq + 1] - ts)) : 0;
                  ok[k] = ok[k] && wi < n_s * (TW / 4);
                  v[k] = src[ok[k] ? (size_t)ts * (TW / 4) + wi : 0];
                }
#pragma unroll
                for (int k = 0; k < NST; ++k)
                  if (ok[k]) dst[g0 + tz + k * THREADS] = v[k];
              }
            }
            __syncthreads();
            const int n_st = max(0, min(capE, cn - k0));
            GNX_NOUNROLL for (int t = 0; t < n_st; t += NWR) {
              float ov[NWR];
              walk_seq<NWR, DT, true>(TreesLds{mystage}, (uint32_t)t, (uint32_t)(n_st - 1), rrow, D, ov);
#pragma unroll
              for (int i = 0; i < NWR; ++i)
                if (t + i < n_st) ps += ov[i];  // tree order
            }
          }
          if (clive) marg[c * NROW + rr] = L.base_score + ps;
        }
        __syncthreads();
        // the rows' softmax: every set takes the exponentials of its classes (three short phases: row maximum, exp in place, sum + arg-max)
        float wmax = 0.f;
        if (rlive) {
          wmax = marg[rr];
          GNX_NOUNROLL for (int a = 1; a < A; ++a) wmax = fmaxf(marg[a * NROW + rr], wmax);
        }
        __syncthreads();
        if (rlive) {
          GNX_NOUNROLL for (int a = set; a < A; a += nset) marg[a * NROW + rr] = gnx_softmax_exp(marg[a * NROW + rr] - wmax);
        }
        __syncthreads();
        if (rlive && set == 0) {  // first maximum wins
          double wsum = 0.0;
          GNX_NOUNROLL for (int a = 0; a < A; ++a) wsum += (double)marg[a * NROW + rr];
          const float fs = (float)wsum;
          int best = 0;
          float bv = marg[rr] / fs;
          GNX_NOUNROLL for (int a = 1; a < A; ++a) { const float v = marg[a * NROW + rr] / fs; if (v > bv) { bv = v; best = a; } }
          reinterpret_cast<uint8_t*>(Y)[2 * (gb + rk) + rh] = (uint8_t)best;
          pmax[2 * (gb + rk) + rh] = bv;
        }
        __syncthreads();
      }
        set_dirty(bp_lo, bp_hi, false, tz);
        mark_changes(tz);
        __syncthreads();
        big_pending = false;
        pf_valid = false;  // (nothing prefetched survives a big batch: its registers are free in there)
#pragma unroll
        for (int i = 0; i < PF_MAX; ++i) pfv[i] = 0;
        if (flushing) {
          bp_lo = __builtin_amdgcn_readfirstlane(first_dirty(0));
          bp_hi = min(W, bp_lo + gstep);
          big_pending = bp_lo < W;
          flushing = big_pending;
          any_dirty = big_pending;
        }
        TICK(7)
        continue;
      }
      if (finishing) break;
      int w = __builtin_amdgcn_readfirstlane(next_change(scan_from));   // (block-uniform values: scalar registers)
      int d = -1;  // >= 0: rows have to be cleaned before the candidate at w can be looked at
      if (any_dirty) {
        // rows that have to be current first: [scan_from - 1, w] (the labels that make w the next change) and row center + 1 (its
        // cached probabilities)
        d = __builtin_amdgcn_readfirstlane(first_dirty(max(scan_from - 1, 0)));
        if (d > min(w, W - 1)) {
          d = -1;
          if (w < W) {
            const int rc = min(max(w, half), W - 1 - half) + 1;
            if (__builtin_amdgcn_readfirstlane((int)((dirty[rc >> 5] >> (rc & 31)) & 1u))) d = rc;
          }
        }
        // clean [d, d + CLEAN_WIN) — a short batch: few rows, so every class gets a row set of its own and the batch costs one class's
        // walks — or a full batch when no change is ahead by the stale labels / the switches have stopped coming (the short batches
        // since the last accepted switch have used up their allowance)
        if (d >= 0) {
          const bool full = w >= W || cleans_left == 0;
          if (!full) --cleans_left;
          bp_lo = d; bp_hi = min(W, d + (full ? gstep : CLEAN_WIN)); big_pending = true;
          continue;
        }
      }
      if (d < 0 && w >= W) break;
      TICK(2)
      const int tz = opaque(tid), lz = tz & 63;
      const int center = min(max(w, half), W - 1 - half);
      const int lo = center - half;  // scope = windows [lo, lo+S)   (gnofix.py:122-130)
      {
        // the switched pair m' = [B0[lo:w], B1[w:hi]], p' = [B1[lo:w], B0[w:hi]] (gnofix.py:144-153); the original pair's
        // probabilities are those of the smoother's row center + 1 (pmax)
        if (pf_fits) {
          if (!(pf_valid && pf_w == w)) cand_fetch(w, tz);
#pragma unroll
          for (int i = 0; i < PF_MAX; ++i) {
            const int e = tz + i * THREADS;
            if (e < 2 * S * A) {
              const int h = e >= S * A ? 1 : 0, f = e - h * S * A;
              const int s = (int)__umulhi((uint32_t)f, invA), a = f - s * A;
              seg[(((lo + s < w) ? h : 1 - h) * A + a) * GP + s] = pfv[i];
            }
          }
        } else {
          GNX_NOUNROLL for (int e = tz; e < 2 * S * A; e += THREADS) {
            const int h = e >= S * A ? 1 : 0, f = e - h * S * A;
            const int s = (int)__umulhi((uint32_t)f, invA), a = f - s * A;
            const int u = lo + s;
            const uint16_t v = R0[(size_t)(h ^ parbit(u)) * WA + (size_t)u * A + a];
            seg[(((u < w) ? h : 1 - h) * A + a) * GP + s] = v;
          }
        }
        const int w_next = __builtin_amdgcn_readfirstlane(next_change(w + 1));  // if this candidate is rejected
        pf_valid = pf_fits && w_next < W;
        pf_w = w_next;
        if (pf_valid) cand_fetch(w_next, opaque(tid));
      }
      __syncthreads();
      TICK(3)
      // 2 rows x n_trees walks: tree t = tz + k * THREADS on both rows side by side, leaves to LDS [row][tree]
      {
        float* leafbuf = reinterpret_cast<float*>(stage);
        for (int tb = 0; tb < NT; tb += THREADS * PER_T) {
          const int per_u = min(PER_T, (NT - tb + THREADS - 1) / THREADS);  // block-uniform
          const uint32_t t0 = (uint32_t)min(tb + tz, NT - 1);              // (GNX_GF_PAD_TREES zero trees follow the last one)
          float lf[PER_T * 2];
          // ONE batch: the lane's per_u trees x 2 rows (D - 1 dependent L2 round trips + the leaves)
          if (per_u <= 1) walk4<1, THREADS, DT>(GT, t0, crow, roff, D, lf);
          else if (PER_T > 2 && per_u == 2) walk4<(PER_T > 2 ? 2 : 1), THREADS, DT>(GT, t0, crow, roff, D, lf);
          else if (PER_T > 3 && per_u == 3) walk4<(PER_T > 3 ? 3 : 1), THREADS, DT>(GT, t0, crow, roff, D, lf);
          else if (PER_T > 4 && per_u == 4) walk4<(PER_T > 4 ? 4 : 1), THREADS, DT>(GT, t0, crow, roff, D, lf);
          else walk4<PER_T, THREADS, DT>(GT, t0, crow, roff, D, lf);
#pragma unroll
          for (int k = 0; k < PER_T; ++k) {
            const int t = tb + tz + k * THREADS;
            if (k < per_u && t < NT) {
#pragma unroll
              for (int r = 0; r < 2; ++r) leafbuf[r * NT + t] = lf[k * 2 + r];
            }
          }
        }
        __syncthreads();
        if (tz < 2 * A) {  // per (row, class): the float32 sum of the class's leaves in tree order (class-major packing)
          const int r = (int)__umulhi((uint32_t)tz, invA), c = tz - r * A;
          const float* lb = leafbuf + r * NT;
          const int t1 = ct0s[c + 1];
          int t = ct0s[c];
          float ps = 0.f;
          GNX_NOUNROLL for (; t + 8 <= t1; t += 8) {  // loads first, then the adds in tree order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = lb[t + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) ps += v[k];
          }
          GNX_NOUNROLL for (; t < t1; ++t) ps += lb[t];
          marg[tz] = L.base_score + ps;
        }
      }
      __syncthreads();
      TICK(4)
      // xgboost's Softmax of the 4 rows and the decision, by EVERY wave for itself (same answer, no second barrier; the LDS
      // operations of one wave execute in order)
      bool accept;
      {
        float* exw = ex + wv * 2 * A;
        GNX_NOUNROLL for (int e = lz; e < 2 * A; e += 64) {
          const int r = (int)__umulhi((uint32_t)e, invA);
          float wmax = marg[r * A];
          GNX_NOUNROLL for (int a = 1; a < A; ++a) wmax = fmaxf(marg[r * A + a], wmax);
          exw[e] = gnx_softmax_exp(marg[e] - wmax);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float mx = 0.f;
        if (lz < 2) {
          double wsum = 0.0;
          GNX_NOUNROLL for (int a = 0; a < A; ++a) wsum += (double)exw[lz * A + a];
          const float fs = (float)wsum;
          mx = exw[lz * A] / fs;
          GNX_NOUNROLL for (int a = 1; a < A; ++a) mx = fmaxf(mx, exw[lz * A + a] / fs);
        }
        const float p_orig = fmaxf(pmax[2 * (center + 1)], pmax[2 * (center + 1) + 1]);  // prob_comp="max" over hap and ancestry
        const float p_sw = fmaxf(__shfl(mx, 0), __shfl(mx, 1));
        accept = p_sw * 0.5f > p_orig * 0.5f;                      // prior_switch_prob = 0.5 (gnofix.py:171)
      }
      TICK(5)
      scan_from = w + 1;
      if (!accept) {
        if (tz == 0) atomicOr(&rej[w >> 5], 1u << (w & 31));
        continue;
      }
      pf_valid = false;

      // ---- accept: flip the parity from w on, relabel ----
      ++n_switch;
      GNX_NOUNROLL for (int q = tz; q < NWD; q += THREADS) {
        const int b0 = q * 32;
        uint32_t m = 0;
        if (w <= b0) m = 0xffffffffu;
        else if (w < b0 + 32) m = 0xffffffffu << (w - b0);
        par[q] ^= m;
      }
      GNX_NOUNROLL for (int q = tz; q < NWD; q += THREADS) {  // candidates within S windows of w are open again
        const int lo_w = max(w - S, 0), hi_w = min(w + S, W - 1), b0 = q * 32;
        if (b0 + 31 >= lo_w && b0 <= hi_w) {
          uint32_t m = 0xffffffffu;
          if (lo_w > b0) m &= 0xffffffffu << (lo_w - b0);
          if (hi_w < b0 + 31) m &= 0xffffffffu >> (b0 + 31 - hi_w);
          rej[q] &= ~m;
        }
      }
      if (tz == 0) { flags[2] = W; flags[3] = 0; }
      __syncthreads();
      // Row w' sees unpadded windows {slide_src(w'+s)}: rows that only see windows >= w exchange their two labels, rows that see
      // both sides are re-evaluated.  (The rows to re-evaluate form one contiguous range [r0, r1): re-evaluating a row of that range
      // that needed nothing, or one that was also swapped, just recomputes its label from the current strips.)
      GNX_NOUNROLL for (int wr = tz; wr < W; wr += THREADS) {
        int mn = W, mx = -1;
        const int j0 = wr, j1 = wr + S - 1;
        const int a0 = max(j0, pad), a1 = min(j1, pad + W - 1);
        if (a0 <= a1) { mn = min(mn, a0 - pad); mx = max(mx, a1 - pad); }
        if (j0 < pad) { const int b1 = min(j1, pad - 1); mn = min(mn, pad - 1 - b1); mx = max(mx, pad - 1 - j0); }
        if (j1 >= pad + W) { const int b0 = max(j0, pad + W); mn = min(mn, W - 1 - (j1 - pad - W)); mx = max(mx, W - 1 - (b0 - pad - W)); }
        if (mn >= w) {
          const uint16_t y = Y[wr];
          Y[wr] = (uint16_t)((y >> 8) | (y << 8));
          const float p0 = pmax[2 * wr];
          pmax[2 * wr] = pmax[2 * wr + 1];
          pmax[2 * wr + 1] = p0;
        } else if (mx >= w) {
          atomicMin(&flags[2], wr);
          atomicMax(&flags[3], wr + 1);
        }
      }
      __syncthreads();
      const int r0 = flags[2], r1 = flags[3];
      TICK(6)
      if (w - last_acc < DENSE_GAP) {  // thick: mark, clean on demand
        set_dirty(r0, r1, true, tz);
        any_dirty = true;
        cleans_left = CLEANS_PER_SWITCH;
        mark_changes(tz);
        __syncthreads();
      } else {
        bp_lo = r0; bp_hi = r1; big_pending = r0 < r1;
        if (!big_pending) { mark_changes(tz); __syncthreads(); }
      }
      last_acc = w;
    }
    if (finishing) break;
    __syncthreads();  // (the history row of this sweep is complete before the next convergence test reads it)
  }

  // ---- outputs: labels, switch count, final parity (k_gnofix_swap applies it to the SNPs) ----
  __syncthreads();
  const int te = opaque(tid);  // (else the load phase's per-thread addresses stay in registers until here)
  for (int u = te; u < W; u += THREADS) {
    L.Yout[(size_t)2 * ind * W + u] = Y[u] & 0xff;
    L.Yout[(size_t)(2 * ind + 1) * W + u] = Y[u] >> 8;
  }
  for (int q = te; q < NWD; q += THREADS) L.par[(size_t)ind * NWD + q] = par[q];
#ifdef GNX_GNOFIX_CLOCKS
  if (te == 0 && L.n_switches) L.n_switches[ind] = (int)(tacc[ind & 7] >> 6);
#else
  if (te == 0 && L.n_switches) L.n_switches[ind] = n_switch;
#endif
}

// ---- the same two passes on 2-BIT rows (the gnx_pack_x layout: SNP j = bits 2 (j % 4).. of byte j / 4; rows 4-byte aligned): a
// 32-bit word = 16 SNPs, a quarter of the bytes of the int8 matrix through HBM ----
__global__ __launch_bounds__(256) void k_gnofix_dif_p2(const uint8_t* __restrict__ P, int64_t ldp, int64_t C, int W, uint32_t* __restrict__ dif) {
  const int NWD = (W + 31) / 32;
  const int ln = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= NWD) return;  // wave-uniform
  const int64_t ind = blockIdx.y;
  const uint32_t* Pm = reinterpret_cast<const uint32_t*>(P + 2 * ind * ldp);
  const uint32_t* Pp = reinterpret_cast<const uint32_t*>(P + (2 * ind + 1) * ldp);
  const int64_t ws = C / W;
  uint32_t word = 0;
  for (int hf = 0; hf < 2; ++hf) {
    const int u = q * 32 + hf * 16 + (ln >> 2), sub = ln & 3;
    const bool live = u < W;
    const int64_t j0 = live ? (int64_t)u * ws : 0, j1 = !live ? 0 : (u == W - 1) ? C : j0 + ws;  // SNPs [j0, j1)
    const int64_t d0 = j0 >> 4, d1 = live ? (j1 - 1) >> 4 : -1;                                  // 32-bit words [d0, d1]
    bool d = false;
    for (int64_t w0 = d0;; w0 += 4) {
      const int64_t wi = w0 + sub;
      if (!d && wi <= d1) {
        uint32_t mask = 0xffffffffu;
        if (wi == d0) mask &= 0xffffffffu << (2 * (int)(j0 & 15));
        if (wi == d1) mask &= 0xffffffffu >> (30 - 2 * (int)((j1 - 1) & 15));
        d = ((Pm[wi] ^ Pp[wi]) & mask) != 0;
      }
      const unsigned long long bal = __ballot(d);
      d = ((bal >> (ln & ~3)) & 0xfull) != 0;  // the window's four lanes agree
      if (__ballot(!d && w0 + 4 <= d1) == 0) break;
    }
    const unsigned long long bal = __ballot(d && sub == 0);  // window k of this half at bit 4k
    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) bits |= (uint32_t)((bal >> (4 * k)) & 1ull) << k;
    word |= bits << (16 * hf);
  }
  if (ln == 0) dif[(size_t)ind * NWD + q] = word;
}

__global__ __launch_bounds__(256) void k_gnofix_swap_p2(uint8_t* __restrict__ P, int64_t ldp, int64_t C, int W, const uint32_t* __restrict__ par, uint32_t ws, uint32_t ws_inv) {
  constexpr int PER = 4;  // 16-byte pieces (64 SNPs) per thread
  const int NWD = (W + 31) / 32;
  const int64_t ind = blockIdx.y;
  const uint32_t* Q = par + (size_t)ind * NWD;
  const WsDiv wdiv{ws, ws_inv};
  const int64_t b0 = (int64_t)blockIdx.x * (256 * PER * 64), b1 = min(b0 + 256 * PER * 64, C);  // SNP range of the block
  if (b0 >= C) return;
  const int ua = (int)min((uint32_t)(W - 1), wdiv((uint32_t)b0)), ub = (int)min((uint32_t)(W - 1), wdiv((uint32_t)(b1 - 1)));  // C < 2^31 (launcher)
  bool any = false;
  for (int q = ua >> 5; q <= ub >> 5; ++q) {  // block-uniform
    uint32_t m = Q[q];
    if (q == ua >> 5) m &= 0xffffffffu << (ua & 31);
    if (q == ub >> 5) m &= 0xffffffffu >> (31 - (ub & 31));
    any |= m != 0;
  }
  if (!any) return;
  uint32_t* Pm = reinterpret_cast<uint32_t*>(P + 2 * ind * ldp);
  uint32_t* Pp = reinterpret_cast<uint32_t*>(P + (2 * ind + 1) * ldp);
  auto odd = [&](int u) { return ((Q[u >> 5] >> (u & 31)) & 1u) != 0; };
  // mask of the 16 SNPs starting at j (a multiple of 16): fields of odd-parity windows
  auto word_mask = [&](int64_t j) -> uint32_t {
    if (j >= C) return 0u;
    const int nf = (int)min((int64_t)16, C - j);
    const int u0 = (int)min((uint32_t)(W - 1), wdiv((uint32_t)j)), u1 = (int)min((uint32_t)(W - 1), wdiv((uint32_t)(j + nf - 1)));
    const uint32_t valid = nf == 16 ? 0xffffffffu : (1u << (2 * nf)) - 1u;
    if (u0 == u1) return odd(u0) ? valid : 0u;
    if (u1 == u0 + 1) {  // one window boundary inside the word (every boundary when windows are >= 16 SNPs): fields below it / from it on.
      // (a per-field loop here ran in EVERY wave — 4 096 SNPs of a wave's pieces always hold a boundary — and made the kernel VALU-bound)
      const uint32_t f0 = (uint32_t)u1 * ws - (uint32_t)j;  // 1..15 fields belong to window u0
      const uint32_t lowm = (1u << (2 * f0)) - 1u;
      return ((odd(u0) ? lowm : 0u) | (odd(u1) ? ~lowm : 0u)) & valid;
    }
    uint32_t mask = 0;
    for (int f = 0; f < nf; ++f)
      if (odd((int)min((uint32_t)(W - 1), wdiv((uint32_t)(j + f))))) mask |= 3u << (2 * f);
    return mask;
  };
  // rows are 64-byte aligned with a pitch that is a multiple of 16 (gnx_gt2_to_p2 / gnx_pack_x callers): 16-byte pieces where the
  // whole piece lies inside the row and inside ONE window (94 % of them at 1000-SNP windows), words otherwise
  const bool wide = ((reinterpret_cast<uintptr_t>(P) | (uintptr_t)ldp) & 15) == 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t j = b0 + ((int64_t)k * 256 + threadIdx.x) * 64;
    if (j >= C) continue;
    const int u0 = (int)min((uint32_t)(W - 1), wdiv((uint32_t)j)), u1 = (int)min((uint32_t)(W - 1), wdiv((uint32_t)min(j + 63, C - 1)));
    if (wide && j + 64 <= C && (u0 == u1 || (u1 == u0 + 1 && odd(u0) == odd(u1)))) {
      if (!odd(u0)) continue;
      uint4* pa = reinterpret_cast<uint4*>(Pm + (j >> 4));
      uint4* pb = reinterpret_cast<uint4*>(Pp + (j >> 4));
      const uint4 a = *pa, b = *pb;
      *pa = b;
      *pb = a;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t mask = word_mask(j + 16 * q);
        if (!mask) continue;
        const uint32_t a = Pm[(j >> 4) + q], b = Pp[(j >> 4) + q];
        Pm[(j >> 4) + q] = (a & ~mask) | (b & mask);
        Pp[(j >> 4) + q] = (b & ~mask) | (a & mask);
      }
    }
  }
}

template <int THREADS>
hipError_t launch_t(const GnofixLaunch& G, int64_t n_ind, hipStream_t s) {
  const size_t lds = gnofix_lds(G.W, G.A, G.S, G.gf_pitch, G.gf_cap, G.d.D, THREADS, G.d.n_trees).total;
  if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;
  const GnofixK L{G.R, G.dif, G.par, G.gf, G.class_tree0, G.Y0, G.pmax0, G.order, G.Yout, G.n_switches, G.hist, G.W, G.A, G.S, G.max_it, G.d.D, G.d.n_trees,
                  G.gf_pitch, G.gf_cap, G.d.base_score};
  if (G.d.D == 4) {
    GNX_LDS_OPTIN(lds, k_gnofix<THREADS, 4>);
    hipLaunchKernelGGL((k_gnofix<THREADS, 4>), dim3((unsigned)n_ind), dim3(THREADS), lds, s, L);
  } else {
    GNX_LDS_OPTIN(lds, k_gnofix<THREADS, 0>);
    hipLaunchKernelGGL((k_gnofix<THREADS, 0>), dim3((unsigned)n_ind), dim3(THREADS), lds, s, L);
  }
  return hipGetLastError();
}

}  // namespace

// staged trees per row set and chunk of the re-evaluation: a whole class where ~40 KB hold one per row set, a multiple of NWR
int gnx_gnofix_cap(int max_class_trees, int D, int S, int threads) {
  const int tb = gnx_gf_tree_words(D) * 4, nset = std::max(1, threads / gnofix_rows_max(S, threads));
  const int cap = std::max(NWR, (40960 / (tb * nset)) / NWR * NWR);
  return std::min(cap, (max_class_trees + NWR - 1) / NWR * NWR);
}

size_t gnx_gnofix_lds_bytes(int W, int A, int S, int pitch, int cap, int D, int threads, int n_trees) {
  return gnofix_lds(W, A, S, pitch, cap, D, threads, n_trees).total;
}

// what depends on the inputs only (ranks of B, "SNP block differs" masks of X): the caller runs it on a side stream beside the
// initial smoother pass (HBM-bound kernels next to an LDS-bound one)
hipError_t gnx_launch_gnofix_prep(const GnofixLaunch& L, int64_t n_ind, hipStream_t s) {
  if (n_ind <= 0) return hipSuccess;
  const int NWD = (L.W + 31) / 32;
  const int64_t n = 2 * n_ind * (int64_t)L.W * L.A;
  hipLaunchKernelGGL(k_gnofix_ranks, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, L.B, n, L.d.rk_thr, L.d.rk_lut, L.d.rk_K,
                     L.d.rk_steps, const_cast<uint16_t*>(L.R));
  for (int64_t i0 = 0; i0 < n_ind; i0 += 65535) {  // the individual is grid.y: at most 65535 per launch
    const unsigned ny = (unsigned)std::min<int64_t>(65535, n_ind - i0);
    uint32_t* dif = const_cast<uint32_t*>(L.dif) + (size_t)i0 * NWD;
    if (L.x_packed)
      hipLaunchKernelGGL(k_gnofix_dif_p2, dim3((unsigned)((NWD + 3) / 4), ny), dim3(256), 0, s, reinterpret_cast<const uint8_t*>(L.X) + 2 * i0 * L.ldx,
                         L.ldx, L.C, L.W, dif);
    else
      hipLaunchKernelGGL(k_gnofix_dif, dim3((unsigned)((NWD + 3) / 4), ny), dim3(256), 0, s, L.X + 2 * i0 * L.ldx, L.ldx, L.C, L.W, dif);
  }
  return hipGetLastError();
}

// after the initial smoother pass (Y0, proba0) and gnx_launch_gnofix_prep
hipError_t gnx_launch_gnofix(const GnofixLaunch& L, int64_t n_ind, int threads, hipStream_t s) {
  if (n_ind <= 0) return hipSuccess;
  {
    hipError_t e0 = hipMemsetAsync(const_cast<int32_t*>(L.order) + n_ind, 0, (size_t)(L.W + 1) * 4, s);  // scratch behind the order: hist[W+1] | cnt[n] | start[W+1]
    if (e0 != hipSuccess) return e0;
    int32_t* cnt = const_cast<int32_t*>(L.order) + n_ind + (L.W + 1);
    int32_t* hist = const_cast<int32_t*>(L.order) + n_ind;
    int32_t* start = cnt + n_ind;
    hipLaunchKernelGGL(k_gnofix_count, dim3((unsigned)n_ind), dim3(256), 0, s, L.Y0, L.W, cnt, hist);
    hipLaunchKernelGGL(k_gnofix_scan, dim3(1), dim3(256), 0, s, hist, L.W, start);
    hipLaunchKernelGGL(k_gnofix_scatter, dim3((unsigned)((n_ind + 255) / 256)), dim3(256), 0, s, cnt, n_ind, start, const_cast<int32_t*>(L.order));
    hipLaunchKernelGGL(k_gnofix_pmax, dim3((unsigned)((2 * n_ind * L.W + 255) / 256)), dim3(256), 0, s, L.proba0, 2 * n_ind * (int64_t)L.W, L.A,
                       const_cast<float*>(L.pmax0));
  }
  hipError_t e;
  if (threads == 256) e = launch_t<256>(L, n_ind, s);
  else if (threads == 1024) e = launch_t<1024>(L, n_ind, s);
  else e = launch_t<512>(L, n_ind, s);
  if (e != hipSuccess) return e;
  const int NWD = (L.W + 31) / 32;
  const uint32_t ws = (uint32_t)(L.C / L.W);
  for (int64_t i0 = 0; i0 < n_ind; i0 += 65535) {  // the individual is grid.y: at most 65535 per launch
    const unsigned ny = (unsigned)std::min<int64_t>(65535, n_ind - i0);
    const uint32_t* par = L.par + (size_t)i0 * NWD;
    if (L.x_packed)
      hipLaunchKernelGGL(k_gnofix_swap_p2, dim3((unsigned)((L.C + 65535) / 65536), ny), dim3(256), 0, s, reinterpret_cast<uint8_t*>(L.X) + 2 * i0 * L.ldx,
                         L.ldx, L.C, L.W, par, ws, gnx_ws_inv(ws));
    else
      hipLaunchKernelGGL(k_gnofix_swap, dim3((unsigned)((L.C + 16383) / 16384), ny), dim3(256), 0, s, L.X + 2 * i0 * L.ldx, L.ldx, L.C, L.W, par, ws, gnx_ws_inv(ws));
  }
  return hipGetLastError();
}

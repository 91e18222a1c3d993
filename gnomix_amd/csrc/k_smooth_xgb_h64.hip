// k_smooth_xgb_h64.hip — the sliding-window tree smoother with LANE = HAPLOTYPE: no LDS bank conflicts by construction.
//
// Same contract and the same arithmetic as k_smooth_xgb_rk.hip (slide_window + XGBClassifier.predict_proba + argmax, reference
// src/Smooth/utils.py:4-29, src/Smooth/smooth.py:40-65, src/Smooth/models.py:8-24; 16-bit ranks instead of float compares,
// leaves summed in tree order: margins bit-identical to the float kernel).  What changes is who shares an LDS bank.
//
// k_smooth_xgb_rk puts 64 consecutive WINDOWS of one haplotype on the lanes of a wave.  A node's feature is a fixed offset into
// the lane's strip, so lanes that took different branches gather from unrelated addresses: 21.7 % of that kernel's LDS cycles are
// bank conflicts (profiles/r02_bench_pmc.json), and its LDS pipe is 85 % busy.  Here a wave holds ONE window of 64 HAPLOTYPES.
// The strip of a block is [padded window][class][64 haplotypes] u16, 128 bytes per (window, class) slot, haplotype h in dword
// h & 31, half h >> 5.  `ds_read_u16` / `ds_read_b32` serve a wave in two groups of 32 lanes and the bank is (address / 4) mod 32
// (MI355X_MICROARCH.md, LDS): inside either group lane l reads bank l & 31 WHATEVER slot its node points at — every rank gather
// is conflict-free however the lanes have diverged.  Node and leaf reads touch at most 2^D consecutive dwords (distinct banks,
// equal addresses broadcast).
//
// The price is the halo: a block needs S - 1 = 74 extra window positions whatever it processes.  It is paid in a cheap currency:
//   * k_smooth_ranks turns B into ranks ONCE (the rank kernel recomputes them for every halo copy), reflect padding of
//     slide_window included, and stores them in the strip layout, [haplotype block][padded window][class][64] — so
//   * staging a block is one contiguous copy (122 positions x A x 128 B = 109 KB at A = 7), ~1 % of the block's walk time:
//     48 windows x 64 haplotypes x 700 trees x 4 levels.
// One block = 64 haplotypes x 48 windows = 16 waves (3 windows per lane, two trees side by side: 6 independent chains), one
// block per CU; trees stream through a double-buffered LDS window as 8-byte nodes {slot offset, rank field}: ds_read_b64 costs
// the LDS array the same 2 cycles as ds_read_b32 and saves the bit-field extraction (rank address = base + offset: 1 VALU).
#include <cstdio>
#include <cstdlib>

#include "gnx_internal.h"

namespace {

constexpr int HB = 64;       // haplotypes per block = lanes
constexpr int RW = 3;        // windows per lane

__device__ __forceinline__ int slide_src(int j, int W, int pad) {  // reflect padding of slide_window (src/Smooth/utils.py:14-17)
  if (j < pad) return pad - 1 - j;
  if (j < pad + W) return j - pad;
  return W - 1 - (j - pad - W);
}

// r = #{U[k] <= p} for NV values side by side (the rank kernel's search); NaN -> 0xFFFF ("never less than a threshold")
template <int NV>
__device__ __forceinline__ void ranks(const float* __restrict__ U, const uint32_t* __restrict__ lut, int K, int steps, const float* p,
                                      uint32_t* r) {
  int lo[NV], hi[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float sc = p[i] * 1024.0f;
    const int b = (int)fminf(fmaxf(sc, 0.0f), 1023.0f);
    const uint32_t e = lut[b];
    lo[i] = (int)(e & 0xffffu);
    hi[i] = (int)(e >> 16);
  }
  for (int s = 0; s < steps; ++s) {
    float u[NV];
    int mid[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      mid[i] = (lo[i] + hi[i]) >> 1;
      u[i] = U[min(mid[i], K - 1)];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool open = lo[i] < hi[i];
      const bool up = open && (u[i] <= p[i]);
      hi[i] = (open && !up) ? mid[i] : hi[i];
      lo[i] = up ? mid[i] + 1 : lo[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) r[i] = (p[i] != p[i]) ? 0xFFFFu : (uint32_t)lo[i];
}

// byte offset of haplotype h inside a 128-byte slot
__device__ __forceinline__ int hap_off(int h) { return (h & 31) * 4 + (h >> 5) * 2; }

// ---- pass 1: B (N, W, A) -> ranks in strip order, Rk[hb][j = 0 .. W + 2 pad - 1][a][64] u16 ---------------------------------
// A block = one haplotype block x 32 windows: the haplotype rows are read along (window, class) (contiguous in B), ranked, parked
// in an LDS tile [row = (window, class)][64 haplotypes] whose dwords are rotated by the row (conflict-free both ways) and written
// out as whole 128-byte slots to the window's padded position and to its mirror image(s) in the reflect padding.
constexpr int RKW = 32;
__global__ __launch_bounds__(256) void k_smooth_ranks(SmoothXGBLaunch L, uint16_t* __restrict__ Rk) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int A = L.A, W = L.W, pad = (L.S + 1) / 2, J = W + 2 * pad;
  const int tid = threadIdx.x;
  const int w0 = blockIdx.x * RKW, nw = min(RKW, W - w0);
  const int64_t hb = blockIdx.y, n0 = hb * HB;
  const int rows = nw * A;            // (window, class) rows of the tile
  const int total = HB * rows;
  constexpr int NV = 4;
  for (int e0 = tid; e0 < total; e0 += NV * 256) {
    float p[NV];
    uint32_t r[NV];
    int dst[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = min(e0 + i * 256, total - 1);  // clamped: loads stay unconditional
      const int h = e / rows, q = e - h * rows;
      const int64_t n = min(n0 + h, L.N - 1);
      const size_t idx = ((size_t)n * W + w0) * A + q;
      p[i] = L.b_is_f64 ? (float)reinterpret_cast<const double*>(L.B)[idx] : reinterpret_cast<const float*>(L.B)[idx];
      dst[i] = q * 128 + ((((h & 31) + q) & 31) << 2) + (h >> 5) * 2;
    }
    ranks<NV>(L.d.rk_thr, L.d.rk_lut, L.d.rk_K, L.d.rk_steps, p, r);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (e0 + i * 256 < total) *reinterpret_cast<uint16_t*>(lds + dst[i]) = (uint16_t)r[i];
  }
  __syncthreads();
  // rows x 32 dwords out; a row goes to 1..3 padded positions
  uint32_t* out = reinterpret_cast<uint32_t*>(Rk) + (size_t)hb * J * A * 32;
  for (int e = tid; e < rows * 32; e += 256) {
    const int q = e >> 5, dw = e & 31;
    const int wl = q / A, a = q - wl * A, w = w0 + wl;
    const uint32_t v = *reinterpret_cast<const uint32_t*>(lds + q * 128 + (((dw + q) & 31) << 2));
    out[((size_t)(w + pad) * A + a) * 32 + dw] = v;
    if (w < pad) out[((size_t)(pad - 1 - w) * A + a) * 32 + dw] = v;                       // left reflection
    if (w >= W - pad) out[((size_t)(pad + W + (W - 1 - w)) * A + a) * 32 + dw] = v;        // right reflection
  }
}

// ---- the walk ---------------------------------------------------------------------------------------------------------------
// one level for three chains: j = 2j + (r >= field)   (v_cmp_le_u32 field, r -> mask; v_addc j, j, j, mask)
__device__ __forceinline__ void step3(uint32_t* j, const uint32_t* f, const uint32_t* r) {
  uint64_t c0, c1, c2;
  asm("v_cmp_le_u32_e64 %[c0], %[f0], %[r0]\n\t"
      "v_cmp_le_u32_e64 %[c1], %[f1], %[r1]\n\t"
      "v_cmp_le_u32_e64 %[c2], %[f2], %[r2]\n\t"
      "v_addc_co_u32 %[j0], %[c0], %[j0], %[j0], %[c0]\n\t"
      "v_addc_co_u32 %[j1], %[c1], %[j1], %[j1], %[c1]\n\t"
      "v_addc_co_u32 %[j2], %[c2], %[j2], %[j2], %[c2]"
      : [j0] "+v"(j[0]), [j1] "+v"(j[1]), [j2] "+v"(j[2]), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)
      : [f0] "v"(f[0]), [r0] "v"(r[0]), [f1] "v"(f[1]), [r1] "v"(r[1]), [f2] "v"(f[2]), [r2] "v"(r[2]));
}
// level 0 of a fresh walk: j = 2 + (r >= root field); the level-1 node {offset, field} picked by the same mask
__device__ __forceinline__ void first3(uint32_t* j, uint32_t field, const uint32_t* r, uint32_t lo_off, uint32_t lo_f, uint32_t hi_off,
                                       uint32_t hi_f, uint32_t* noff, uint32_t* nf) {
  uint64_t c0, c1, c2;
  asm("v_cmp_le_u32_e64 %[c0], %[fd], %[r0]\n\t"
      "v_cmp_le_u32_e64 %[c1], %[fd], %[r1]\n\t"
      "v_cmp_le_u32_e64 %[c2], %[fd], %[r2]\n\t"
      "v_cndmask_b32 %[o0], %[lo], %[ho], %[c0]\n\t"
      "v_cndmask_b32 %[o1], %[lo], %[ho], %[c1]\n\t"
      "v_cndmask_b32 %[o2], %[lo], %[ho], %[c2]\n\t"
      "v_cndmask_b32 %[g0], %[lf], %[hf], %[c0]\n\t"
      "v_cndmask_b32 %[g1], %[lf], %[hf], %[c1]\n\t"
      "v_cndmask_b32 %[g2], %[lf], %[hf], %[c2]\n\t"
      "v_addc_co_u32_e64 %[j0], %[c0], 1, 1, %[c0]\n\t"
      "v_addc_co_u32_e64 %[j1], %[c1], 1, 1, %[c1]\n\t"
      "v_addc_co_u32_e64 %[j2], %[c2], 1, 1, %[c2]"
      : [j0] "=&v"(j[0]), [j1] "=&v"(j[1]), [j2] "=&v"(j[2]), [o0] "=&v"(noff[0]), [o1] "=&v"(noff[1]), [o2] "=&v"(noff[2]),
        [g0] "=&v"(nf[0]), [g1] "=&v"(nf[1]), [g2] "=&v"(nf[2]), [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2)
      : [fd] "v"(field), [r0] "v"(r[0]), [r1] "v"(r[1]), [r2] "v"(r[2]), [lo] "v"(lo_off), [ho] "v"(hi_off), [lf] "v"(lo_f), [hf] "v"(hi_f));
}

__device__ __forceinline__ uint32_t ld_rank(const uint8_t* p) { return *reinterpret_cast<const uint16_t*>(p); }

// TWO trees (tb0, tb1) side by side for the RW windows of a lane, depth D >= 2.  rb[k] = the lane's origin in the strip for
// window k (slot of padded position w_k, class 0, + the haplotype's bytes); a node's offset is (s * A + a) * 128.
template <int D>
__device__ __forceinline__ void walk_pair(const uint8_t* tb0, const uint8_t* tb1, const uint8_t* const* rb, float* psum) {
  uint32_t j[2 * RW], off[2 * RW], f[2 * RW], r[2 * RW];
  const uint2 root0 = *reinterpret_cast<const uint2*>(tb0 + 8), root1 = *reinterpret_cast<const uint2*>(tb1 + 8);
  const uint4 kid0 = *reinterpret_cast<const uint4*>(tb0 + 16), kid1 = *reinterpret_cast<const uint4*>(tb1 + 16);
#pragma unroll
  for (int k = 0; k < RW; ++k) r[k] = ld_rank(rb[k] + root0.x);
#pragma unroll
  for (int k = 0; k < RW; ++k) r[RW + k] = ld_rank(rb[k] + root1.x);
  first3(j, root0.y, r, kid0.x, kid0.y, kid0.z, kid0.w, off, f);
  first3(j + RW, root1.y, r + RW, kid1.x, kid1.y, kid1.z, kid1.w, off + RW, f + RW);
#pragma unroll
  for (int k = 0; k < 2 * RW; ++k) r[k] = ld_rank(rb[k % RW] + off[k]);
  step3(j, f, r);
  step3(j + RW, f + RW, r + RW);
#pragma unroll
  for (int d = 2; d < D; ++d) {
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const uint2 nd = reinterpret_cast<const uint2*>(tb0)[j[k]];
      off[k] = nd.x;
      f[k] = nd.y;
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const uint2 nd = reinterpret_cast<const uint2*>(tb1)[j[RW + k]];
      off[RW + k] = nd.x;
      f[RW + k] = nd.y;
    }
#pragma unroll
    for (int k = 0; k < 2 * RW; ++k) r[k] = ld_rank(rb[k % RW] + off[k]);
    step3(j, f, r);
    step3(j + RW, f + RW, r + RW);
  }
  float l0[RW], l1[RW];
  const float* lf0 = reinterpret_cast<const float*>(tb0 + ((size_t)8 << D)) - (1 << D);  // leaf of heap index j (2^D <= j < 2^(D+1))
  const float* lf1 = reinterpret_cast<const float*>(tb1 + ((size_t)8 << D)) - (1 << D);
#pragma unroll
  for (int k = 0; k < RW; ++k) l0[k] = lf0[j[k]];
#pragma unroll
  for (int k = 0; k < RW; ++k) l1[k] = lf1[j[RW + k]];
#pragma unroll
  for (int k = 0; k < RW; ++k) psum[k] += l0[k];   // tree order: tb0 before tb1
#pragma unroll
  for (int k = 0; k < RW; ++k) psum[k] += l1[k];
}

// one tree, any depth (the tail of an odd group, depth-1 ensembles, run-time depths)
__device__ __forceinline__ float walk_one(const uint8_t* tb, const uint8_t* rb, int D) {
  uint32_t j = 1;
  for (int d = 0; d < D; ++d) {
    const uint2 nd = reinterpret_cast<const uint2*>(tb)[j];
    const uint32_t r = ld_rank(rb + nd.x);
    j = 2 * j + ((r < nd.y) ? 0u : 1u);
  }
  return (reinterpret_cast<const float*>(tb + ((size_t)8 << D)) - (1 << D))[j];
}

// NWAVE waves = NWAVE * RW windows of HB haplotypes per block.  DT = depth (0: run time).
template <int NWAVE, int DT>
__global__ __launch_bounds__(NWAVE * 64) void k_smooth_xgb_h64(SmoothXGBLaunch L, const uint16_t* __restrict__ Rk) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int THREADS = NWAVE * 64, WPB = NWAVE * RW;
  const int A = L.A, W = L.W, S = L.S, pad = (S + 1) / 2, J = W + 2 * pad;
  const int D = DT ? DT : L.d.D;
  const int tree_bytes = L.d.h8_tree_bytes;
  const int P = WPB + S - 1;                                  // padded window positions held
  const int strip_bytes = P * A * 128;
  const int buf_bytes = L.d.h8_max_group * tree_bytes;        // multiple of 16
  uint8_t* strip = lds;
  uint8_t* tbuf0 = lds + strip_bytes;
  uint8_t* tbuf1 = tbuf0 + buf_bytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t hb = blockIdx.y;
  const int w0 = blockIdx.x * WPB;

  // ---- the block's strip: one contiguous run of the rank array (clamped at the chromosome's end) ----
  {
    const uint4* src = reinterpret_cast<const uint4*>(Rk + ((size_t)hb * J + w0) * A * 64);
    const int n16 = strip_bytes / 16;
    const int last = (int)(((size_t)(J - w0) * A * 128) / 16) - 1;  // last piece inside this haplotype block's rows
    for (int e = tid; e < n16; e += THREADS) reinterpret_cast<uint4*>(strip)[e] = src[min(e, last)];
  }

  const int64_t n = hb * HB + lane;
  const uint8_t* rb[RW];
  bool valid[RW];
  const int hoff = hap_off(lane);
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    const int wl = wave * RW + k;
    rb[k] = strip + (size_t)wl * A * 128 + hoff;
    valid[k] = (n < L.N) && (w0 + wl < W);
  }
  // margins parked class-major, [class][haplotype block][window][64 lanes]: whole 256-byte lines
  const size_t cls_stride = (size_t)gridDim.y * W * HB;
  float* mrow = L.marg + ((size_t)hb * W + w0 + wave * RW) * HB + lane;

  // ---- tree groups through the double-buffered LDS window ----
  const int ng = L.d.h8_n_groups;
  constexpr int MAXV = (4096 / 16 + THREADS - 1) / THREADS;  // a group is at most 4 KB (model loader): one piece per thread
  uint4 stg[MAXV];
  const int nv = (buf_bytes / 16 + THREADS - 1) / THREADS;
#define GNX_G_LOAD(g)                                                                               \
  {                                                                                                 \
    const int t0_ = L.d.h8_group_tree0[g], t1_ = L.d.h8_group_tree0[(g) + 1];                       \
    const int last_ = (t1_ - t0_) * tree_bytes / 16 - 1;                                            \
    const uint4* src_ = reinterpret_cast<const uint4*>(L.d.h8_packed + (size_t)t0_ * tree_bytes);   \
    _Pragma("unroll") for (int v = 0; v < MAXV; ++v) if (v < nv) stg[v] = src_[min(v * THREADS + tid, last_)]; \
  }
#define GNX_G_STORE(dst)                                                                            \
  {                                                                                                 \
    _Pragma("unroll") for (int v = 0; v < MAXV; ++v) {                                              \
      const int e_ = v * THREADS + tid;                                                             \
      if (v < nv && e_ * 16 < buf_bytes) *reinterpret_cast<uint4*>((dst) + (size_t)e_ * 16) = stg[v]; \
    }                                                                                               \
  }
  float psum[RW];
#pragma unroll
  for (int k = 0; k < RW; ++k) psum[k] = 0.f;

  GNX_G_LOAD(0);
  GNX_G_STORE(tbuf0);
  __syncthreads();

  int cur_class = L.d.h8_group_class[0];
  for (int g = 0; g < ng; ++g) {
    uint8_t* cur = (g & 1) ? tbuf1 : tbuf0;
    uint8_t* nxt = (g & 1) ? tbuf0 : tbuf1;
    const int gn = min(g + 1, ng - 1);
    GNX_G_LOAD(gn);
    const int cls = L.d.h8_group_class[g];
    if (cls != cur_class) {
#pragma unroll
      for (int k = 0; k < RW; ++k) {
        if (valid[k]) mrow[(size_t)cur_class * cls_stride + (size_t)k * HB] = L.d.base_score + psum[k];
        psum[k] = 0.f;
      }
      cur_class = cls;
    }
    const int nt = L.d.h8_group_tree0[g + 1] - L.d.h8_group_tree0[g];
    int t = 0;
    if constexpr (DT >= 2) {
      for (; t + 1 < nt; t += 2) {
        const uint8_t* tb = cur + (size_t)t * tree_bytes;
        walk_pair<DT>(tb, tb + tree_bytes, rb, psum);
      }
    }
    for (; t < nt; ++t) {
      const uint8_t* tb = cur + (size_t)t * tree_bytes;
#pragma unroll
      for (int k = 0; k < RW; ++k) psum[k] += walk_one(tb, rb[k], D);
    }
    GNX_G_STORE(nxt);
    __syncthreads();
  }
#undef GNX_G_LOAD
#undef GNX_G_STORE
#pragma unroll
  for (int k = 0; k < RW; ++k)
    if (valid[k]) mrow[(size_t)cur_class * cls_stride + (size_t)k * HB] = L.d.base_score + psum[k];

  // ---- softmax (xgboost common/math.h Softmax) + argmax per row, by the lane that parked its margins ----
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    if (!valid[k]) continue;
    const float* mg = mrow + (size_t)k * HB;
    const size_t orow = (size_t)n * W + (w0 + wave * RW + k);
    float* o = L.proba + orow * A;
    float wmax = mg[0];
    for (int a = 1; a < A; ++a) wmax = fmaxf(mg[(size_t)a * cls_stride], wmax);
    double wsum = 0.0;
    for (int a = 0; a < A; ++a) {
      const float e = (float)exp((double)(mg[(size_t)a * cls_stride] - wmax));
      o[a] = e;
      wsum += (double)e;
    }
    const float fs = (float)wsum;
    int best = 0;
    float bv = -1.f;
    for (int a = 0; a < A; ++a) {
      const float p = o[a] / fs;
      o[a] = p;
      if (L.proba64) L.proba64[orow * A + a] = (double)p;
      if (p > bv) { bv = p; best = a; }
    }
    if (L.labels) L.labels[orow] = best;
  }
}

template <int NWAVE>
hipError_t launch_h64(const SmoothXGBLaunch& L, const uint16_t* Rk, size_t lds, hipStream_t s) {
  const dim3 grid((unsigned)((L.W + NWAVE * RW - 1) / (NWAVE * RW)), (unsigned)((L.N + HB - 1) / HB));
  if (L.d.D == 4) {
    GNX_LDS_OPTIN(lds, k_smooth_xgb_h64<NWAVE, 4>);
    hipLaunchKernelGGL((k_smooth_xgb_h64<NWAVE, 4>), grid, dim3(NWAVE * 64), lds, s, L, Rk);
  } else {
    GNX_LDS_OPTIN(lds, k_smooth_xgb_h64<NWAVE, 0>);
    hipLaunchKernelGGL((k_smooth_xgb_h64<NWAVE, 0>), grid, dim3(NWAVE * 64), lds, s, L, Rk);
  }
  return hipGetLastError();
}

size_t lds_need(const SmoothXGBDev& d, int A, int S, int nwave) {
  return (size_t)(nwave * RW + S - 1) * A * 128 + 2 * (size_t)d.h8_max_group * d.h8_tree_bytes;
}

}  // namespace

// waves per block the model's strip allows (16, 8 or 4), 0 when even 4 do not fit the LDS: the caller falls back to k_smooth_xgb_rk
int gnx_smooth_h64_waves(const SmoothXGBDev& d, int A, int S) {
  if (!d.h8_packed) return 0;
  for (int nw : {16, 8, 4})
    if (lds_need(d, A, S, nw) <= (size_t)160 * 1024) return nw;
  return 0;
}

size_t gnx_smooth_h64_rank_bytes(int64_t N, int W, int A, int S) {
  const int pad = (S + 1) / 2;
  return (size_t)((N + HB - 1) / HB) * (size_t)(W + 2 * pad) * A * 128 + 256;
}

hipError_t gnx_launch_smooth_xgb_h64(const SmoothXGBLaunch& L, uint16_t* Rk, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  int nw = gnx_smooth_h64_waves(L.d, L.A, L.S);
  if (tune.sm_nw == 4 || tune.sm_nw == 8) nw = std::min(nw, tune.sm_nw);
  if (nw == 0) return hipErrorInvalidValue;
  {
    const dim3 grid((unsigned)((L.W + RKW - 1) / RKW), (unsigned)((L.N + HB - 1) / HB));
    hipLaunchKernelGGL(k_smooth_ranks, grid, dim3(256), (size_t)RKW * L.A * 128, s, L, Rk);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  const size_t lds = lds_need(L.d, L.A, L.S, nw);
  return nw == 16 ? launch_h64<16>(L, Rk, lds, s) : nw == 8 ? launch_h64<8>(L, Rk, lds, s) : launch_h64<4>(L, Rk, lds, s);
}

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
// Pointer nodes (as k_smooth_xgb_rk's PTR variant): 8 bytes {w0 = rank field << 16 | slot (s * A + a), w1 = LDS address of the left
// child | of the right child << 16}; the last level's children are the leaves.  A level = rank address (v_mad_u32_u16: slot * 128 +
// the lane's strip origin), compare (v_cmp_le_u32_sdwa -> vcc), v_cndmask_b32_sdwa picking a half of w1: 3 VALU, 2 LDS reads.
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(3))) T* lds_at(uint32_t a) {
  return (const __attribute__((address_space(3))) T*)(uintptr_t)a;
}
#else
template <typename T>
__device__ const T* lds_at(uint32_t) { return nullptr; }  // host pass: never called
#endif
__device__ __forceinline__ uint32_t step_ptr(uint32_t w0, uint32_t w1, uint32_t r) {
  uint32_t p;
  asm("v_cmp_le_u32_sdwa vcc, %[n], %[r] src0_sel:WORD_1 src1_sel:DWORD\n\t"
      "v_cndmask_b32_sdwa %[p], %[w], %[w], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
      : [p] "=v"(p)
      : [n] "v"(w0), [r] "v"(r), [w] "v"(w1)
      : "vcc");
  return p;
}
__device__ __forceinline__ void first_ptr(uint32_t root, uint32_t r, uint32_t l0, uint32_t r0, uint32_t l1, uint32_t r1, uint32_t& w0,
                                          uint32_t& w1) {
  asm("v_cmp_le_u32_sdwa vcc, %[n], %[r] src0_sel:WORD_1 src1_sel:DWORD\n\t"
      "v_cndmask_b32 %[x], %[l0], %[r0], vcc\n\t"
      "v_cndmask_b32 %[y], %[l1], %[r1], vcc"
      : [x] "=&v"(w0), [y] "=&v"(w1)
      : [n] "v"(root), [r] "v"(r), [l0] "v"(l0), [r0] "v"(r0), [l1] "v"(l1), [r1] "v"(r1)
      : "vcc");
}
// LDS address of the rank a node asks for: slot * 128 + the lane's origin for the window (rb, an LDS address)
__device__ __forceinline__ uint32_t rank_at(uint32_t w0, uint32_t rb) {
  uint32_t a;
  asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(a) : "v"(w0), "s"(128u), "v"(rb));
  return a;
}

// NT trees (tb, tb + tree_bytes, ...) side by side for the RW windows of a lane: NT * RW independent chains.  rb[k] = LDS address
// of the lane's halfword in slot (padded position w_k, class 0).
template <int D, int NT>
__device__ __forceinline__ void walk_n(const uint8_t* tb, int tree_bytes, const uint32_t* rb, float* psum) {
  static_assert(D >= 2, "the 16-byte tree top holds levels 0 and 1");
  constexpr int NC = NT * RW;
  uint32_t w0[NC], w1[NC], r[NC], p[NC];
  uint4 top[NT];  // {w0 of node 2, w0 of node 3, w0 of the root, w1 of node 2}
#pragma unroll
  for (int t = 0; t < NT; ++t) top[t] = *reinterpret_cast<const uint4*>(tb + t * tree_bytes);
  constexpr uint32_t SIB = (D == 2 ? 8u : 16u) * 0x10001u;  // node 3's children sit right behind node 2's
#pragma unroll
  for (int c = 0; c < NC; ++c) r[c] = *lds_at<uint16_t>(rank_at(top[c / RW].z, rb[c % RW]));
#pragma unroll
  for (int c = 0; c < NC; ++c)
    first_ptr(top[c / RW].z, r[c], top[c / RW].x, top[c / RW].y, top[c / RW].w, top[c / RW].w + SIB, w0[c], w1[c]);
#pragma unroll
  for (int c = 0; c < NC; ++c) r[c] = *lds_at<uint16_t>(rank_at(w0[c], rb[c % RW]));
#pragma unroll
  for (int c = 0; c < NC; ++c) p[c] = step_ptr(w0[c], w1[c], r[c]);
#pragma unroll
  for (int d = 2; d < D; ++d) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const uint2 nd = *lds_at<uint2>(p[c]);
      w0[c] = nd.x;
      w1[c] = nd.y;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) r[c] = *lds_at<uint16_t>(rank_at(w0[c], rb[c % RW]));
#pragma unroll
    for (int c = 0; c < NC; ++c) p[c] = step_ptr(w0[c], w1[c], r[c]);
  }
  float lf[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) lf[c] = *lds_at<float>(p[c]);
#pragma unroll
  for (int t = 0; t < NT; ++t)  // tree order
#pragma unroll
    for (int k = 0; k < RW; ++k) psum[k] += lf[t * RW + k];
}

// NWAVE waves = NWAVE * RW windows of HB haplotypes per block.  DT = depth (0: run time).
template <int NWAVE, int DT, int NT>
__global__ __launch_bounds__(NWAVE * 64) void k_smooth_xgb_h64(SmoothXGBLaunch L, const uint16_t* __restrict__ Rk) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int THREADS = NWAVE * 64, WPB = NWAVE * RW;
  const int A = L.A, W = L.W, S = L.S, pad = (S + 1) / 2, J = W + 2 * pad;
  const int tree_bytes = L.d.h8_tree_bytes;
  const int P = WPB + S - 1;                                  // padded window positions held
  const int strip_bytes = P * A * 128;
  const int buf_bytes = L.d.h8_max_group * tree_bytes;        // multiple of 16
  uint8_t* tbuf0 = lds;                                      // the trees first: node addresses must fit 16 bits
  uint8_t* tbuf1 = tbuf0 + buf_bytes;
  uint8_t* strip = tbuf1 + buf_bytes;
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
  uint32_t rb[RW];
  bool valid[RW];
  const int hoff = hap_off(lane);
#pragma unroll
  for (int k = 0; k < RW; ++k) {
    const int wl = wave * RW + k;
    rb[k] = (uint32_t)(uintptr_t)strip + (uint32_t)(wl * A * 128 + hoff);
    valid[k] = (n < L.N) && (w0 + wl < W);
  }
  // margins parked class-major, [class][haplotype block][window][64 lanes]: whole 256-byte lines
  const size_t cls_stride = (size_t)gridDim.y * W * HB;
  float* mrow = L.marg + ((size_t)hb * W + w0 + wave * RW) * HB + lane;

  // ---- tree groups through the double-buffered LDS window ----
  const int ng = L.d.h8_n_groups;
  constexpr int MAXV = (32768 / 16 + THREADS - 1) / THREADS;  // a group is at most 32 KB (model loader)
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
      /* child addresses: relative to the group -> LDS addresses */                                 \
      const int pc_ = (e_ * 16 % tree_bytes) >> 4;                                                  \
      const uint32_t add_ = pc_ < (1 << (DT - 1)) ? (uint32_t)(uintptr_t)(dst) * 0x10001u : 0u;     \
      stg[v].w += add_;                                                                             \
      stg[v].y += pc_ ? add_ : 0u;                                                                  \
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
    for (; t + NT <= nt; t += NT) walk_n<DT, NT>(cur + (size_t)t * tree_bytes, tree_bytes, rb, psum);
    for (; t < nt; ++t) walk_n<DT, 1>(cur + (size_t)t * tree_bytes, tree_bytes, rb, psum);
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

template <int NWAVE, int DT>
hipError_t launch_h64_d(const SmoothXGBLaunch& L, const uint16_t* Rk, size_t lds, int nt, hipStream_t s) {
  const dim3 grid((unsigned)((L.W + NWAVE * RW - 1) / (NWAVE * RW)), (unsigned)((L.N + HB - 1) / HB));
  if (nt == 2) {
    GNX_LDS_OPTIN(lds, k_smooth_xgb_h64<NWAVE, DT, 2>);
    hipLaunchKernelGGL((k_smooth_xgb_h64<NWAVE, DT, 2>), grid, dim3(NWAVE * 64), lds, s, L, Rk);
  } else {
    GNX_LDS_OPTIN(lds, k_smooth_xgb_h64<NWAVE, DT, 4>);
    hipLaunchKernelGGL((k_smooth_xgb_h64<NWAVE, DT, 4>), grid, dim3(NWAVE * 64), lds, s, L, Rk);
  }
  return hipGetLastError();
}
template <int NWAVE>
hipError_t launch_h64(const SmoothXGBLaunch& L, const uint16_t* Rk, size_t lds, int nt, hipStream_t s) {
  switch (L.d.D) {
    case 2: return launch_h64_d<NWAVE, 2>(L, Rk, lds, nt, s);
    case 3: return launch_h64_d<NWAVE, 3>(L, Rk, lds, nt, s);
    case 4: return launch_h64_d<NWAVE, 4>(L, Rk, lds, nt, s);
    case 5: return launch_h64_d<NWAVE, 5>(L, Rk, lds, nt, s);
    case 6: return launch_h64_d<NWAVE, 6>(L, Rk, lds, nt, s);
  }
  return hipErrorInvalidValue;
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
  const int nt = tune.sm_pair == 2 ? 2 : 4;  // trees side by side per lane (GNX_SM_PAIR=2: two)
  return nw == 16 ? launch_h64<16>(L, Rk, lds, nt, s) : nw == 8 ? launch_h64<8>(L, Rk, lds, nt, s) : launch_h64<4>(L, Rk, lds, nt, s);
}

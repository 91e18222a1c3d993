// k_smooth_xgb_bs.hip — the sliding-window tree smoother WITHOUT tree walks: sorted prefixes + bit-sliced node evaluation.
//
// Same contract and the same arithmetic as k_smooth_xgb_rk.hip / k_smooth_xgb.hip (slide_window + XGBClassifier.predict_proba +
// argmax: reference src/Smooth/utils.py:4-29, src/Smooth/smooth.py:40-65, src/Smooth/models.py:8-24); margins are float32 sums in
// tree order, so outputs are BIT-identical to the walking kernels.
//
// A walk costs a lane ~17 VALU and ~10 LDS operations per (window, tree) — four data-dependent levels — and three rounds of
// variants of it sit at 0.33 of the LDS gather ceiling.  This kernel uses what the walk ignores: feature (s, a) of window w is the
// base probability of class a at padded window w + s, so the answers of node (a, threshold t, offset s) over ALL windows of a
// haplotype are ONE bitmap over the padded windows — "p[w'][a] >= t" — read at a shift of s.
//   k_bs_ranks (pre-pass, one thread per four probabilities): p -> counter index = first counter of its class + per-class rank
//                #{thresholds of class a <= p} (the rank kernel's quantisation with per-class threshold lists; NaN -> the class's
//                last counter), 16 bits, parked in HBM: the dependent table look-ups run at full occupancy instead of inside a block.
//   k_smooth_xgb_bs, per chunk of WC = 128 windows of one haplotype (Wp = WC + S - 1 padded windows):
//   A  sort:     byte histogram of the class's counter indices; an exclusive scan turns it into cnt[a][k] = #{w' : rank < k}, i.e.
//                for every node the number n of padded windows that go LEFT at it; a counting sort gives each class's order pi_a.
//      rows:     G[a][n] = bitmap of the windows that are NOT among the first n of pi_a, for n = 0..Wp (Wp + 1 rows of Wp bits: a
//                prefix OR, built in segments).  Node (a, k, s) is row cnt[a][k] shifted by s: "goes right" for 32 windows per word.
//                (all of A per class by ONE wave, no block barrier: a class's counters, order and rows are its own)
//   then, after the one barrier that publishes every class's rows, every wave on its own again: wave = one class, groups of 32 of its trees:
//   B  planes:   lane = (tree, 64 windows): 15 x (counter byte, three row words, two v_alignbit) and 11 v_bfi select, level by
//                level, the bit of the node each window actually visits: four words = the four bits of 32 leaf indices, interleaved
//                into eight 4-bit indices per word in registers (v_perm_b32 as a four-entry table: spread8).
//   C  leaves:   lane = window (and window + 64): one word of eight leaf indices per tree (shared by eight lanes), v_bfe, one LDS
//                gather of the leaf, summed in tree order.
// (First version: planes as wave masks in SGPRs — v_cndmask + 3 v_addc_co per leaf index — through a per-block scratch line in L2
//  and s_load ... glc: correct, 5.8 ms at config 2, every four trees waited a full L2 round trip.  Second: indices through the LDS,
//  B and C as block-wide phases between barriers, ranks inside the block: 2.6 ms, every phase latency-bound.  LAB_NOTES round 5.)
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gnx_internal.h"
#include "gnx_exp.h"
#include "gnx_rank.h"

namespace {

// reads / writes through an LDS ADDRESS (the low 32 bits of a generic pointer into the LDS): `ds_read vdst, a` with nothing added
#if defined(__HIP_DEVICE_COMPILE__)
template <typename T>
__device__ __forceinline__ __attribute__((address_space(3))) T* lds_at(uint32_t a) {
  return (__attribute__((address_space(3))) T*)(uintptr_t)a;
}
#else
template <typename T>
__device__ T* lds_at(uint32_t) { return nullptr; }  // host pass: never called
#endif

// LDS traffic between the lanes of ONE wave: the wave's LDS operations complete in order, the compiler must not move them across
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t bfi(uint32_t sel, uint32_t a, uint32_t b) {  // sel ? a : b, one instruction (hipcc splits the C form)
  uint32_t d;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(sel), "v"(a), "v"(b));
  return d;
}
// byte K of g with its eight bits spread to every fourth position, in registers: the byte's four 2-bit fields become the four
// selector bytes of a v_perm_b32 over the pool {0x00, 0x01, 0x10, 0x11} (field f -> byte with bit 0 of f at bit 0, bit 1 at bit 4).
// 5 VALU and no LDS gather (a 256-entry table in the LDS costs ~7 cycles of the LDS pipe per look-up: random bytes, 32 banks, and
// the LDS pipe is what bounds phase B).
template <int K>
__device__ __forceinline__ uint32_t spread8(uint32_t g) {
  const uint32_t b = (g >> (8 * K)) & 255u;            // v_bfe_u32
  uint32_t t = (b << 6) | b;                            // v_lshl_or_b32
  t = (t << 12) | t;                                    // v_lshl_or_b32: b | b << 6 | b << 12 | b << 18
  return __builtin_amdgcn_perm(0u, 0x11100100u, t & 0x03030303u);  // selector bytes 0..3 pick bytes of the second operand
}
// byte K of the four go-right planes -> eight 4-bit leaf indices (plane 0 = most significant bit)
template <int K>
__device__ __forceinline__ uint32_t nib8(uint32_t g0, uint32_t g1, uint32_t g2, uint32_t g3) {
  const uint32_t t0 = spread8<K>(g0), t1 = spread8<K>(g1), t2 = spread8<K>(g2), t3 = spread8<K>(g3);
  return (((t0 << 1 | t1) << 1 | t2) << 1) | t3;
}
// 15 go-right words of one block of 32 windows -> the four planes of the leaf index
__device__ __forceinline__ void mux15(const uint32_t* G, uint32_t& g0, uint32_t& g1, uint32_t& g2, uint32_t& g3) {
  g0 = G[1];
  g1 = bfi(g0, G[3], G[2]);
  g2 = bfi(g0, bfi(g1, G[7], G[6]), bfi(g1, G[5], G[4]));
  const uint32_t d0 = bfi(g1, bfi(g2, G[11], G[10]), bfi(g2, G[9], G[8]));
  const uint32_t d1 = bfi(g1, bfi(g2, G[15], G[14]), bfi(g2, G[13], G[12]));
  g3 = bfi(g0, d1, d0);
}

constexpr int QMAX = 16;  // histogram words per thread in the scan
constexpr int WAVE_AREA = 3072;  // per wave: 32 trees x 4 blocks x 16 B of leaf indices + 16 trees x 64 B of leaves

// ---- pre-pass: probabilities -> counter indices ------------------------------------------------------------------------------
struct BsRankArgs {
  const void* B;
  uint16_t* bins;
  const float* thr;
  const uint32_t* lut;
  const int32_t* uoff;
  const int32_t* binoff;
  int64_t total;
  int32_t A, b_is_f64, steps, nthr;
};

// persistent blocks: the bucket table (A x 1024 words) and the threshold lists sit in the LDS, so the dependent look-ups of the
// bisection cost LDS latency; four probabilities per thread and step
__global__ __launch_bounds__(1024) void k_bs_ranks(BsRankArgs Q) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint32_t* const lut = reinterpret_cast<uint32_t*>(lds);
  float* const thr = reinterpret_cast<float*>(lds) + Q.A * 1024;
  int32_t* const uoff = reinterpret_cast<int32_t*>(thr + Q.nthr);
  int32_t* const binoff = uoff + Q.A + 1;
  for (int i = threadIdx.x; i < Q.A * 1024; i += 1024) lut[i] = Q.lut[i];
  for (int i = threadIdx.x; i < Q.nthr; i += 1024) thr[i] = Q.thr[i];
  if ((int)threadIdx.x <= Q.A) { uoff[threadIdx.x] = Q.uoff[threadIdx.x]; binoff[threadIdx.x] = Q.binoff[threadIdx.x]; }
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 4096;
  for (int64_t i0 = ((int64_t)blockIdx.x * 1024 + threadIdx.x) * 4; i0 < Q.total; i0 += stride) {
    float p[4];
    int lo[4], hi[4], cc[4], K[4], uo[4];
    const int c0 = (int)(i0 % Q.A);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t e = min(i0 + i, Q.total - 1);
      int c = c0 + i;
      c -= (c >= Q.A) ? Q.A : 0;
      c -= (c >= Q.A) ? Q.A : 0;  // A >= 2: c0 + 3 < 3 A
      cc[i] = c;
      p[i] = Q.b_is_f64 ? (float)reinterpret_cast<const double*>(Q.B)[e] : reinterpret_cast<const float*>(Q.B)[e];
    }
    if (!Q.b_is_f64 && i0 + 3 < Q.total && (reinterpret_cast<uintptr_t>(Q.B) & 15) == 0) {
      const float4 v = reinterpret_cast<const float4*>(Q.B)[i0 >> 2];
      p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uo[i] = uoff[cc[i]];
      K[i] = uoff[cc[i] + 1] - uo[i];
      const float sc = p[i] * 1024.0f;
      const int b = (int)fminf(fmaxf(sc, 0.0f), 1023.0f);
      const uint32_t en = lut[cc[i] * 1024 + b];
      lo[i] = (int)(en & 0xffffu);
      hi[i] = (int)(en >> 16);
    }
    for (int s = 0; s < Q.steps; ++s) {
      float u[4];
      int mid[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mid[i] = (lo[i] + hi[i]) >> 1;
        u[i] = thr[uo[i] + min(mid[i], K[i] - 1)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool open = lo[i] < hi[i];
        const bool up = open && (u[i] <= p[i]);
        hi[i] = (open && !up) ? mid[i] : hi[i];
        lo[i] = up ? mid[i] + 1 : lo[i];
      }
    }
    uint32_t r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (uint32_t)(binoff[cc[i]] + ((p[i] != p[i]) ? K[i] + 1 : lo[i]));
    if (i0 + 3 < Q.total) {
      *reinterpret_cast<uint2*>(Q.bins + i0) = make_uint2(r[0] | (r[1] << 16), r[2] | (r[3] << 16));
    } else {
      for (int i = 0; i < 4 && i0 + i < Q.total; ++i) Q.bins[i0 + i] = (uint16_t)r[i];
    }
  }
}

// ---- main kernel -----------------------------------------------------------------------------------------------------------
// what the kernel reads, and nothing else: the full launch record (every copy of the ensemble for every smoother kernel) costs more
// scalar registers than a wave has
struct BsArgs {
  const uint16_t* bins;  // (N, W, A) counter indices from the pre-pass
  float* proba;
  double* proba64;
  int32_t* labels;
  const uint32_t* nodes;
  const float* leaves;
  const int32_t* binoff;
  const int32_t* ct0;
  int64_t items;
  int32_t W, A, S, nch, flags;
  uint32_t invA;  // ceil(2^32 / A): e / A = umulhi(e, invA) for the few thousand e of a chunk
  float base_score;
  GnxBsLayout Y;
};

template <int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_smooth_xgb_bs(BsArgs Q) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int WC = 128, NH = 2, TG = 32;
  const GnxBsLayout Y = Q.Y;
  const int A = Q.A, W = Q.W, S = Q.S, pad = (S + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((uint32_t)(uintptr_t)lds != 0u) __builtin_trap();  // node words carry absolute LDS addresses

  uint8_t* const cnt8 = lds + Y.off_cnt;
  uint32_t* const cnt32 = reinterpret_cast<uint32_t*>(lds + Y.off_cnt);
  uint32_t* const hist32 = reinterpret_cast<uint32_t*>(lds + Y.off_hist);
  uint32_t* const P32 = reinterpret_cast<uint32_t*>(lds + Y.off_P);
  uint16_t* const bin16 = reinterpret_cast<uint16_t*>(lds + Y.off_bin);
  uint8_t* const pi8 = lds + Y.off_pi;
  // where the sort's tie counters were: the row build's segment totals, then per wave {leaf indices, leaves}, then margins
  uint32_t* const seg32 = reinterpret_cast<uint32_t*>(lds + Y.off_seg);
  uint32_t* const nibw = hist32 + wave * (WAVE_AREA / 4);          // [TG][4 blocks][4] words: eight 4-bit leaf indices each
  float* const lvw = reinterpret_cast<float*>(nibw + TG * 16);     // [16][16]: the leaves of half a group
  float* const mg = reinterpret_cast<float*>(hist32);              // [WC][A]
  float* const ev_tmp = mg + WC * A;                               // [WC][A]
  const int flags = Q.flags;

  const size_t NWA = (size_t)W * A;

  for (int64_t item = blockIdx.x; item < Q.items; item += gridDim.x) {
    const int64_t n = item / Q.nch;
    const int ch = (int)(item - n * Q.nch);
    const int w0 = ch * WC;

    // ---- A: sort and rows, wave = class, no block barrier: everything below touches only class c's counters, order and rows ----
    if (wave < A && !(flags & 4)) {
      const int c = wave;
      const int b0 = Q.binoff[c], b1 = Q.binoff[c + 1];          // multiples of 16
      const int cw0 = b0 >> 2, cwn = (b1 - b0) >> 2;               // the class's counter words
      uint32_t* const segc = seg32 + c * Y.rw * Y.nseg;
      uint16_t* const binc = bin16 + c * Y.wp;
      uint8_t* const pic = pi8 + c * 256;
      // A0: clear the class's histogram and segment totals
      for (int i = lane; i < cwn; i += 64) hist32[cw0 + i] = 0u;
      for (int i = lane; i < Y.rw * Y.nseg; i += 64) segc[i] = 0u;
      wave_sync();
      // A1: counter indices of the chunk's padded windows (reflected at the chromosome's ends), byte histogram
      for (int wq = lane; wq < Y.wp; wq += 64) {
        const int j = w0 + wq;
        int bin;
        if (j <= W + S - 2) bin = Q.bins[(size_t)n * NWA + (size_t)slide_src(j, W, pad) * A + c];
        else bin = b1 - 1;  // beyond the last padded window any real window reads: never less than a threshold
        binc[wq] = (uint16_t)bin;
        atomicAdd(&hist32[bin >> 2], 1u << ((bin & 3) * 8));
      }
      wave_sync();
      // A2: exclusive scan of the class's byte histogram -> cnt[k] = #{windows of rank < k}; clear hist for A3
      {
        const int q = (cwn + 63) >> 6;  // <= QMAX (launcher)
        uint32_t x[QMAX];
        uint32_t sum = 0;
        const int wi0 = lane * q;
#pragma unroll
        for (int i = 0; i < QMAX; ++i) {
          x[i] = 0u;
          if (i < q) {
            x[i] = (wi0 + i < cwn) ? hist32[cw0 + wi0 + i] : 0u;
            sum = __builtin_amdgcn_sad_u8(x[i], 0u, sum);
          }
        }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t t = __shfl_up(inc, d);
          if (lane >= d) inc += t;
        }
        uint32_t run = inc - sum;
#pragma unroll
        for (int i = 0; i < QMAX; ++i) {
          if (i < q && wi0 + i < cwn) {
            uint32_t out = (run & 255u) * 0x01010101u;
            if (x[i]) {
              const uint32_t y0 = x[i] & 255u, y1 = (x[i] >> 8) & 255u, y2 = (x[i] >> 16) & 255u, y3 = x[i] >> 24;
              const uint32_t p1 = run + y0, p2 = p1 + y1, p3 = p2 + y2;
              out = (run & 255u) | ((p1 & 255u) << 8) | ((p2 & 255u) << 16) | (p3 << 24);
              run = p3 + y3;
              hist32[cw0 + wi0 + i] = 0u;
            }
            cnt32[cw0 + wi0 + i] = out;
          }
        }
      }
      wave_sync();
      // A3: counting sort: position of every padded window in the class's order (ties in arrival order: only the prefix lengths
      //     cnt[k] are ever used as row numbers, and those fall between tie groups); the row build's segment totals on the way
      for (int wq = lane; wq < Y.wp; wq += 64) {
        const int bin = binc[wq];
        const uint32_t old = atomicAdd(&hist32[bin >> 2], 1u << ((bin & 3) * 8));
        const uint32_t pos = (uint32_t)cnt8[bin] + ((old >> ((bin & 3) * 8)) & 255u);
        pic[pos] = (uint8_t)wq;
        atomicOr(&segc[(wq >> 5) * Y.nseg + (int)__umulhi(pos, Y.inv_sl)], 1u << (wq & 31));
      }
      wave_sync();
      // A4: rows G[c][n] = ~(windows among the first n of the order), lane = (word jw of the row, segment sg of n)
      if (lane < Y.rw * Y.nseg && !(flags & 8)) {
        const int sl = Y.sl;  // multiple of 4
        const int sg = lane / Y.rw, jw = lane - sg * Y.rw;
        const int n0 = sg * sl, n1 = min(n0 + sl, Y.wp);
        const uint32_t* pw = reinterpret_cast<const uint32_t*>(pic);
        uint32_t cur = 0;
        for (int s2 = 0; s2 < sg; ++s2) cur |= segc[jw * Y.nseg + s2];
        uint32_t* row = P32 + (size_t)c * Y.nr * Y.rw + jw;
        for (int nn = n0; nn < n1; nn += 4) {
          const uint32_t w4 = pw[nn >> 2];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (nn + k < n1) {
              row[(size_t)(nn + k) * Y.rw] = ~cur;
              const uint32_t w = (w4 >> (8 * k)) & 255u;
              cur |= ((int)(w >> 5) == jw) ? (1u << (w & 31u)) : 0u;
            }
          }
        }
        if (n1 == Y.wp && n0 < n1) row[(size_t)Y.wp * Y.rw] = ~cur;
      }
    }
    __syncthreads();  // every class's counters and rows are in place

    // ---- trees: every wave on its own (wave = class), groups of TG trees: B (planes -> leaf indices) then C (leaves) ----
    float acc[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) acc[h] = 0.f;
    if (wave < A) {
      const int c = wave;
      const int t0 = Q.ct0[c], nt = Q.ct0[c + 1] - t0;
      const int rl = lane >> 1;                       // B: lane = (tree rl of the group, half hf: blocks 2 hf, 2 hf + 1)
      const uint32_t hf8 = (uint32_t)(lane & 1) * 8u;
      const uint32_t sh = (uint32_t)(lane & 7) * 4u;  // C: lane = window (and window + 64)
      const uint32_t* nb = nibw + (lane >> 3);        // + tree * 16 (+ 8 for window + 64)
      const uint32_t rb = (uint32_t)Y.rb;
      const uint32_t nib_addr = (uint32_t)(uintptr_t)nibw + (uint32_t)lane * 32u, lv_addr = (uint32_t)(uintptr_t)lvw + (uint32_t)lane * 16u;

      uint4 cur[8];
      uint4 lcur[2];
      {
        const int tree = t0 + min(rl, nt - 1);
        const uint4* np = reinterpret_cast<const uint4*>(Q.nodes + (size_t)tree * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = np[i];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) lcur[hh] = reinterpret_cast<const uint4*>(Q.leaves + (size_t)(t0 + min(hh * 16 + (lane >> 2), nt - 1)) * 16)[lane & 3];
      }
      for (int g0r = 0; g0r < ((flags & 1) ? 0 : nt); g0r += TG) {
        const int nval = min(TG, nt - g0r);
        // B: 15 counters -> row addresses -> three row words each -> two blocks of go-right bits -> planes -> leaf indices
        if (rl < nval) {
          uint32_t nw[32];
#pragma unroll
          for (int i = 0; i < 8; ++i) { nw[4 * i] = cur[i].x; nw[4 * i + 1] = cur[i].y; nw[4 * i + 2] = cur[i].z; nw[4 * i + 3] = cur[i].w; }
          uint32_t raw[16], row[16];
#pragma unroll
          for (int j = 1; j < 16; ++j) raw[j] = *lds_at<uint8_t>(nw[2 * j] >> 16);
          __builtin_amdgcn_sched_barrier(0);  // all fifteen counter reads in flight before the first is used (hipcc sinks each to its use)
#pragma unroll
          for (int j = 1; j < 16; ++j) row[j] = __umul24(raw[j], rb) + (nw[2 * j + 1] + hf8);
          uint32_t Ga[16], Gb[16];
          {
            uint32_t x0[16], x1[16], x2[16];
#pragma unroll
            for (int j = 1; j < 16; ++j) { x0[j] = lds_at<uint32_t>(row[j])[0]; x1[j] = lds_at<uint32_t>(row[j])[1]; x2[j] = lds_at<uint32_t>(row[j])[2]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 1; j < 16; ++j) {
              Ga[j] = __builtin_amdgcn_alignbit(x1[j], x0[j], nw[2 * j]);
              Gb[j] = __builtin_amdgcn_alignbit(x2[j], x1[j], nw[2 * j]);
            }
          }
          uint32_t g0, g1, g2, g3;
          mux15(Ga, g0, g1, g2, g3);
          *lds_at<uint4>(nib_addr) = make_uint4(nib8<0>(g0, g1, g2, g3), nib8<1>(g0, g1, g2, g3), nib8<2>(g0, g1, g2, g3), nib8<3>(g0, g1, g2, g3));
          mux15(Gb, g0, g1, g2, g3);
          *lds_at<uint4>(nib_addr + 16u) = make_uint4(nib8<0>(g0, g1, g2, g3), nib8<1>(g0, g1, g2, g3), nib8<2>(g0, g1, g2, g3), nib8<3>(g0, g1, g2, g3));
        }
        *lds_at<uint4>(lv_addr) = lcur[0];  // leaves of the group's first 16 trees
        const uint4 lsecond = lcur[1];
        // the next group's nodes and leaves arrive while C runs (clamped: the last group re-reads its own last tree)
        {
          const int tree = t0 + min(g0r + TG + rl, nt - 1);
          const uint4* np = reinterpret_cast<const uint4*>(Q.nodes + (size_t)tree * 32);
#pragma unroll
          for (int i = 0; i < 8; ++i) cur[i] = np[i];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
            lcur[hh] = reinterpret_cast<const uint4*>(Q.leaves + (size_t)(t0 + min(g0r + TG + hh * 16 + (lane >> 2), nt - 1)) * 16)[lane & 3];
        }
        // C: two rounds of 16 trees (the leaves of a round fill the wave's 1 KB)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int nv = min(16, nval - hh * 16);
          if (!(flags & 2)) {
            int r2 = 0;
            for (; r2 + 8 <= nv; r2 += 8) {
              uint32_t wv[8][NH];
              float lf[8][NH];
#pragma unroll
              for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int h = 0; h < NH; ++h) wv[i][h] = nb[(hh * 16 + r2 + i) * 16 + h * 8];
#pragma unroll
              for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int h = 0; h < NH; ++h) lf[i][h] = lvw[(r2 + i) * 16 + ((wv[i][h] >> sh) & 15u)];
#pragma unroll
              for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int h = 0; h < NH; ++h) acc[h] += lf[i][h];
            }
            for (; r2 < nv; ++r2)
#pragma unroll
              for (int h = 0; h < NH; ++h) acc[h] += lvw[r2 * 16 + ((nb[(hh * 16 + r2) * 16 + h * 8] >> sh) & 15u)];
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (hh == 0) *lds_at<uint4>(lv_addr) = lsecond;
        }
      }
    }
    __syncthreads();  // every wave is done with its indices and leaves: the margins go where they were

    // ---- margins -> LDS (window-major), softmax + argmax per window (xgboost common/math.h Softmax) ----
    if (wave < A) {
#pragma unroll
      for (int h = 0; h < NH; ++h) mg[(h * 64 + lane) * A + wave] = Q.base_score + acc[h];
    }
    __syncthreads();
    // e = exp(margin - max of the window) by one thread per (window, class), in place; then one thread per window
    for (int e = tid; e < WC * A; e += NT) {
      const int wl = (int)__umulhi((uint32_t)e, Q.invA);
      const float* m = mg + wl * A;
      float wmax = m[0];
      for (int a = 1; a < A; ++a) wmax = fmaxf(m[a], wmax);
      ev_tmp[e] = gnx_softmax_exp(mg[e] - wmax);
    }
    __syncthreads();
    for (int wl = tid; wl < WC; wl += NT) {
      const int w = w0 + wl;
      if (w >= W) continue;
      const float* ev = ev_tmp + wl * A;
      const size_t orow = (size_t)n * W + w;
      float* o = Q.proba + orow * A;
      double wsum = 0.0;
      for (int a = 0; a < A; ++a) wsum += (double)ev[a];
      const float fs = (float)wsum;
      int best = 0;
      float bv = -1.f;
      for (int a = 0; a < A; ++a) {
        const float p = ev[a] / fs;
        o[a] = p;
        if (Q.proba64) Q.proba64[orow * A + a] = (double)p;
        if (p > bv) { bv = p; best = a; }
      }
      if (Q.labels) Q.labels[orow] = best;
    }
    __syncthreads();  // the next item reuses every table
  }
}

int bs_threads(int A) { return A <= 8 ? 512 : 1024; }  // one wave per class

// LDS of k_bs_ranks: the bucket table, the threshold lists, the two per-class offset tables
size_t bs_rank_lds_bytes(const SmoothXGBDev& d, int A) { return ((size_t)A * 1024 + (size_t)d.bs_nthr + 2 * ((size_t)A + 1)) * 4; }

}  // namespace

GnxBsLayout gnx_bs_layout(int A, int S, int wc, int nbins) {
  const int threads = bs_threads(A), nwave = threads / 64;
  GnxBsLayout y{};
  y.wp = wc + S - 1; y.nr = y.wp + 1;
  y.rw = wc / 32 + ((S - 1) >> 5) + 1;  // last word a shifted read of the last block touches, + 1 (covers every padded window)
  y.rb = y.rw * 4;
  y.nbins = (nbins + 15) & ~15;
  y.nseg = std::max(1, std::min(16, 64 / y.rw));  // a wave builds its class's rows: (word, segment) per lane
  y.sl = (((y.wp + y.nseg - 1) / y.nseg) + 3) & ~3;
  y.inv_sl = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)y.sl - 1) / (uint64_t)y.sl);
  // the sort's scratch (tie counters, counter indices, order: dead once the rows are built) shares its bytes with the row build's
  // segment totals, then the waves' leaf-index words and leaves, then the margins and their exponentials
  const int bin_bytes = (A * y.wp * 2 + 15) & ~15;
  const int seg_bytes = A * y.rw * y.nseg * 4;
  int shared = y.nbins + bin_bytes + A * 256 + seg_bytes;
  shared = std::max(shared, nwave * WAVE_AREA);
  shared = std::max(shared, 2 * wc * A * 4);
  y.hist_bytes = (shared + 15) & ~15;
  y.off_cnt = 0;
  y.off_hist = y.off_cnt + y.nbins;
  y.off_bin = y.off_hist + y.nbins;     // counter indices and order sit behind the tie counters; the segment totals go behind those
  y.off_pi = y.off_bin + bin_bytes;
  y.off_seg = y.off_pi + A * 256;
  y.off_P = y.off_hist + y.hist_bytes;
  y.off_wtot = y.off_P + A * y.nr * y.rb;
  y.total = y.off_wtot + 256;
  return y;
}

bool gnx_smooth_bs_fits(const SmoothXGBDev& d, int A, int S) {
  if (!d.bs_nodes || d.bs_wc != 128) return false;
  const GnxBsLayout Y = gnx_bs_layout(A, S, d.bs_wc, d.bs_nbins);
  const int nt = bs_threads(A);
  if (bs_rank_lds_bytes(d, A) > (size_t)160 * 1024) return false;  // the pre-pass keeps its tables in the LDS
  return Y.total <= 160 * 1024 && Y.wp <= 255 && Y.rw <= 64 && (d.bs_maxbins / 4 + 63) / 64 <= QMAX && A >= 2 && A <= nt / 64;
}

size_t gnx_smooth_bs_scratch_bytes(int64_t N, int W, int A) { return (size_t)N * W * A * 2 + 64; }

hipError_t gnx_launch_smooth_xgb_bs(const SmoothXGBLaunch& L, uint16_t* bins, int n_cu, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  if (!gnx_smooth_bs_fits(L.d, L.A, L.S)) return hipErrorInvalidValue;
  {
    BsRankArgs R{};
    R.B = L.B; R.bins = bins; R.thr = L.d.bs_thr; R.lut = L.d.bs_lut; R.uoff = L.d.bs_uoff; R.binoff = L.d.bs_binoff;
    R.total = L.N * L.W * L.A; R.A = L.A; R.b_is_f64 = L.b_is_f64; R.steps = L.d.bs_steps; R.nthr = L.d.bs_nthr;
    const size_t rl = bs_rank_lds_bytes(L.d, L.A);
    GNX_LDS_OPTIN(rl, k_bs_ranks);
    const unsigned g = (unsigned)std::min<int64_t>((R.total + 4095) / 4096, (int64_t)n_cu * (rl <= (size_t)80 * 1024 ? 2 : 1));
    hipLaunchKernelGGL(k_bs_ranks, dim3(g), dim3(1024), rl, s, R);
  }
  BsArgs Q{};
  Q.bins = bins; Q.proba = L.proba; Q.proba64 = L.proba64; Q.labels = L.labels;
  Q.nodes = L.d.bs_nodes; Q.leaves = L.d.bs_leaves; Q.binoff = L.d.bs_binoff; Q.ct0 = L.d.bs_class_tree0;
  Q.W = L.W; Q.A = L.A; Q.S = L.S;
  Q.invA = (uint32_t)((((uint64_t)1 << 32) + (uint64_t)L.A - 1) / (uint64_t)L.A);
  Q.base_score = L.d.base_score;
  Q.Y = gnx_bs_layout(L.A, L.S, L.d.bs_wc, L.d.bs_nbins);
  static const int env_flags = std::getenv("GNX_BS_FLAGS") ? std::atoi(std::getenv("GNX_BS_FLAGS")) : 0;  // ablation (timing only)
  Q.flags = env_flags;
  Q.nch = (L.W + L.d.bs_wc - 1) / L.d.bs_wc;
  Q.items = L.N * Q.nch;
  const size_t lds = (size_t)Q.Y.total;
  const int nt = bs_threads(L.A);
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)(1024 / nt), (size_t)160 * 1024 / lds));
  const unsigned grid = (unsigned)std::min<int64_t>(Q.items, (int64_t)n_cu * per_cu);
  if (tune.debug)
    fprintf(stderr, "k_smooth_xgb_bs<%d>: lds=%zu blocks/CU=%d grid=%u items=%lld nbins=%d nseg=%d\n", nt, lds, per_cu, grid,
            (long long)Q.items, L.d.bs_nbins, Q.Y.nseg);
  if (nt == 512) {
    GNX_LDS_OPTIN(lds, k_smooth_xgb_bs<512>);
    hipLaunchKernelGGL((k_smooth_xgb_bs<512>), dim3(grid), dim3(512), lds, s, Q);
  } else {
    GNX_LDS_OPTIN(lds, k_smooth_xgb_bs<1024>);
    hipLaunchKernelGGL((k_smooth_xgb_bs<1024>), dim3(grid), dim3(1024), lds, s, Q);
  }
  return hipGetLastError();
}

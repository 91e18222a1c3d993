// k_base_logistic_p2.hip — the exact int8 logistic pass on 2-BIT haplotype rows (gfx950).
//
// Same contract, arithmetic (7 balanced base-256 limbs on v_mfma_i32_16x16x64_i8, one float64 rounding), column slots, pieces,
// flush epilogue and XCD-aware grid as k_base_logistic_i8.hip / k_base_logistic_i8_dl.hip (reference src/Base/base.py:146-180,
// src/Base/models.py:12-21; X values per src/utils.py:153) — B is BIT-IDENTICAL to theirs.  What differs is where X lives:
// the int8 kernels are bound by the L1's miss queue (5.05 GB of L1 fill per launch, 3.45 GB of it X); here
//   * X stays packed in HBM, four SNPs per byte (the gnx_pack_x layout: SNP j = bits 2(j%4).. of byte j/4): a quarter of the fill;
//   * a lane fetches 16 packed bytes = 64 SNPs of ONE haplotype row per load, the four lanes of a row (lane>>4 = 0..3) cover 64
//     contiguous bytes = one RUN of 256 SNPs, 16 rows per wave instruction — no X tile in LDS at all: a wave owns its rows, and
//     which SNP sits at which k position of the MFMA is the weight planes' business (they are laid out to match at model load);
//   * a 32-bit word (16 SNPs) becomes the 16 int8 bytes of the MFMA A operand in 7 VALU operations:
//         reg d = (word >> 2d) & 0x03030303        (d = 0..3; byte b of reg d = field 4b + d)
//     — the VALU was 22 % busy, the matrix pipe 15 %;
//   * pieces start on a byte boundary: a piece [b0, b1) is walked from SNP b0 & ~3 in runs of 256 SNPs (4 MFMA entries), the up to
//     three SNPs before b0 and everything from b1 on meet zero weights; windows are flushed after the last run of their piece;
//   * the digit planes go L2 -> LDS directly (global_load_lds_dwordx4) through an NBUF-deep ring of half-runs (2 entries), ONE block
//     barrier per half-run, exactly as in k_base_logistic_i8_dl.hip; X is two register stages (run r in use, run r+1 in flight; the
//     loads of run r+2 are issued the moment the last word of run r has been unpacked).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gnx_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
struct __attribute__((packed, aligned(1))) xbytes16 { v4i v; };

constexpr int LIMBS = 7;

__device__ __forceinline__ v4i load_x16(const uint8_t* p) {  // unaligned, unconditional global_load_dwordx4
  xbytes16 r;
  __builtin_memcpy(&r, p, 16);
  return r.v;
}

__device__ __forceinline__ v4i unpack16(int w) {  // 16 two-bit fields -> 16 int8 bytes, field 4b + d at byte b of reg d
  const unsigned u = (unsigned)w;
  v4i r;
  r[0] = (int)(u & 0x03030303u);
  r[1] = (int)((u >> 2) & 0x03030303u);
  r[2] = (int)((u >> 4) & 0x03030303u);
  r[3] = (int)((u >> 6) & 0x03030303u);
  return r;
}

__device__ __forceinline__ double combine(const v4i (&acc)[LIMBS], int reg, double scale) {
  long long lo = (long long)acc[0][reg] + ((long long)acc[1][reg] << 8) + ((long long)acc[2][reg] << 16);
  long long hi = (long long)acc[3][reg] + ((long long)acc[4][reg] << 8) + ((long long)acc[5][reg] << 16) +
                 ((long long)acc[6][reg] << 24);
  return ((double)hi * 16777216.0 + (double)lo) * scale;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MT 16-row tiles per wave, NT column tiles, WAVES waves per block, NBUF = 3 ring slots of one half-run (2 entries) each.
template <int MT, int NT, int WAVES, int NBUF>
__global__ __launch_bounds__(WAVES * 64) void k_base_logistic_p2(BaseLRLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int EPS = 2;                          // MFMA entries per plane step (half a run)
  constexpr int ENTRY_BYTES = NT * LIMBS * 1024;  // digit planes of one entry (64 k positions)
  constexpr int STEP_BYTES = EPS * ENTRY_BYTES;
  constexpr int THREADS = WAVES * 64;
  constexpr int ROWS = WAVES * MT * 16;           // haplotypes per block
  constexpr int NKB = STEP_BYTES / 1024;          // 1 KB plane blocks per step
  constexpr int PLD = (NKB + WAVES - 1) / WAVES;  // plane loads per wave per step
  constexpr int D = NBUF - 1;                     // plane steps in flight beyond the one being multiplied
  static_assert(D == 2, "the vmcnt arithmetic below assumes two plane steps (one run) in flight");
  // vector-memory instructions younger than the planes of step s when a wave waits for them: the planes of step s+1 and the X loads
  // of one run (see the issue order in GNX_P2_RUN)
  constexpr int WAITN = PLD + MT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int A = L.A, W = L.W, R = L.d.R;
  uint8_t* vbuf = lds;                             // [NBUF][STEP_BYTES]
  double* zb = reinterpret_cast<double*>(vbuf + (size_t)NBUF * STEP_BYTES) + (size_t)wave * (MT * 16) * A;
  double* tab_ic = reinterpret_cast<double*>(vbuf + (size_t)NBUF * STEP_BYTES) + (size_t)ROWS * A;  // [max_wins][A] intercepts
  double* tab_sc = tab_ic + (size_t)L.max_wins * A;                                                  // [max_wins] 2^-f_w
  int* tab_rb = reinterpret_cast<int*>(tab_sc + L.max_wins);
  int* tab_nfl = tab_rb + L.max_chunks;
  int* tab_fl0 = tab_nfl + L.max_chunks;

  // XCD-aware decomposition: all blocks of one window range on ONE XCD (its L2 serves the range's digit planes)
  int wrange, htile;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    wrange = xcd + 8 * (j / L.n_htiles);
    htile = j % L.n_htiles;
  }
  const int wa = wrange * L.wch;
  if (wa >= W) return;  // whole block exits before any barrier
  const int wb = min(W, wa + L.wch);
  const int r_begin = L.d.win_run0[wa];
  const int r_end = L.d.win_run1[wb - 1];
  const int n_runs = r_end - r_begin;
  const int n_steps = 2 * n_runs;
  const int64_t n0b = (int64_t)htile * ROWS;       // first haplotype of the block
  const int64_t n0 = n0b + (int64_t)wave * (MT * 16);

  for (int e = tid; e < n_runs; e += THREADS) {
    tab_rb[e] = L.d.run_byte[r_begin + e];
    tab_nfl[e] = L.d.run_nflush[r_begin + e];
    tab_fl0[e] = L.d.run_flush0[r_begin + e];
  }
  const int wt0 = max(0, wa - R - 1);
  for (int e = tid; e < L.max_wins; e += THREADS) {
    const int w = min(wt0 + e, W - 1);
    tab_sc[e] = L.d.wscale[w];
    for (int a = 0; a < A; ++a) tab_ic[e * A + a] = L.d.icpt[w * A + a];
  }
  __syncthreads();

  // the lane's rows: tile mt, row i16; its 16 bytes of a run = packed bytes [16 kq, 16 kq + 16)
  const uint8_t* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int64_t n = n0 + mt * 16 + i16;  // rows >= N-1 read the zero-padded copy of the last row (rows past N are never written)
    xrow[mt] = (n >= L.N - 1 ? reinterpret_cast<const uint8_t*>(L.last_row) : reinterpret_cast<const uint8_t*>(L.X) + n * L.ldx) + 16 * kq;
  }
  const int8_t* vsrc = L.d.V2 + (size_t)r_begin * (2 * STEP_BYTES) + (size_t)lane * 16;

  // every load is unconditional and clamped (tail steps re-fetch the last one into a slot nobody reads): the number of
  // vector-memory instructions per step is the constant the vmcnt arithmetic relies on
  auto issue_planes = [&](int step) {
    const int st = min(step, n_steps - 1);
    const int8_t* src = vsrc + (size_t)st * STEP_BYTES;
    uint8_t* vdst = vbuf + (size_t)(step % NBUF) * STEP_BYTES;
#pragma unroll
    for (int it = 0; it < PLD; ++it) {
      const int kb = min(wave + it * WAVES, NKB - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kb * 1024), (lptr_t)(vdst + (size_t)kb * 1024), 16, 0, 0);
    }
  };

  v4i acc[MT][NT][LIMBS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) acc[mt][nt][l] = v4i{0, 0, 0, 0};

  auto mfma_entry = [&](const uint8_t* pb, const v4i (&xa)[MT]) {
    const v4i* vb = reinterpret_cast<const v4i*>(pb) + lane;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        const v4i b = vb[(nt * LIMBS + l) * 64];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[mt], b, acc[mt][nt][l], 0, 0, 0);
      }
  };

  // ---- piece end: the windows that finished with run rl (block-uniform) ----
  auto flush = [&](int rl) {
    const int nfl = tab_nfl[rl];
    if (nfl <= 0) return;
    const int w0 = tab_fl0[rl];
    for (int w = w0; w < w0 + nfl; ++w) {
      const int cbase = (w % R) * A;
      const double scale = tab_sc[w - wt0];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = nt * 16 + i16 - cbase;
          const bool mine = (col >= 0) && (col < A);
          if (mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r)  // int32 16x16 C/D layout: column = lane&15, row = 4*(lane>>4) + reg
              zb[(mt * 16 + 4 * kq + r) * A + col] = combine(acc[mt][nt], r, scale);
          }
#pragma unroll
          for (int l = 0; l < LIMBS; ++l)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][nt][l][r] = mine ? 0 : acc[mt][nt][l][r];
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // zb is wave-private: LDS ops of one wave complete in order
      if (w >= wa && w < wb) {
        // sigmoid, normaliser and division for the wave's MT*16 rows x A classes, spread over all 64 lanes; per element the
        // arithmetic and the class order of the row sum are those of k_base_logistic_i8 (bit-identical B)
        const int ne = MT * 16 * A;
        const double* ic = tab_ic + (w - wt0) * A;
        for (int e = lane; e < ne; e += 64) {
          const int a = e % A;
          zb[e] = 1.0 / (1.0 + exp(-(zb[e] + ic[a])));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int e = lane; e < ne; e += 64) {
          const int rl_ = e / A, a = e - rl_ * A;
          const double* z = zb + rl_ * A;
          double sum = 0.0;
          for (int c = 0; c < A; ++c) sum += z[c];
          const double v = z[a] / sum;
          const int64_t n = n0 + rl_;
          if (n < L.N) {
            const size_t o = ((size_t)n * W + w) * A + a;
            if (L.b64) L.b64[o] = v;
            if (L.b32) L.b32[o] = (float)v;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };

  v4i xsA[MT], xsB[MT];  // two register stages of X: even runs in A, odd runs in B

#define GNX_P2_XLOAD(XS, RUN)                                                          \
  {                                                                                    \
    const int rb_ = tab_rb[min((RUN), n_runs - 1)];                                    \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) XS[mt] = load_x16(xrow[mt] + rb_); \
  }

#define GNX_P2_STEP_SYNC(STEP)                                                                                                  \
  {                                                                                                                             \
    wait_vm<WAITN>(); /* the wave's own plane loads of this step (and everything older: its X run) have landed */               \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                          \
    __builtin_amdgcn_s_barrier(); /* all shares of the step are in LDS; every wave is done with the step before it */           \
    asm volatile("" ::: "memory");                                                                                              \
    issue_planes((STEP) + D);     /* into the slot the previous step just left */                                               \
  }

// one run = two plane steps; issue order per run: planes(2r+2) | planes(2r+3) | X(r+2)
#define GNX_P2_RUN(XS, RUN)                                                                      \
  {                                                                                              \
    const int r_ = (RUN);                                                                        \
    const bool live_ = r_ < n_runs;                                                              \
    v4i xa_[MT];                                                                                 \
    GNX_P2_STEP_SYNC(2 * r_);                                                                    \
    if (live_) {                                                                                 \
      const uint8_t* sb_ = vbuf + (size_t)((2 * r_) % NBUF) * STEP_BYTES;                        \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) xa_[mt] = unpack16(XS[mt][0]);           \
      mfma_entry(sb_, xa_);                                                                      \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) xa_[mt] = unpack16(XS[mt][1]);           \
      mfma_entry(sb_ + ENTRY_BYTES, xa_);                                                        \
    }                                                                                            \
    GNX_P2_STEP_SYNC(2 * r_ + 1);                                                                \
    const uint8_t* sc_ = vbuf + (size_t)((2 * r_ + 1) % NBUF) * STEP_BYTES;                      \
    if (live_) {                                                                                 \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) xa_[mt] = unpack16(XS[mt][2]);           \
      mfma_entry(sc_, xa_);                                                                      \
    }                                                                                            \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) xa_[mt] = unpack16(XS[mt][3]);             \
    GNX_P2_XLOAD(XS, r_ + 2); /* the stage is free: every word of run r has been unpacked */     \
    if (live_) {                                                                                 \
      mfma_entry(sc_ + ENTRY_BYTES, xa_);                                                        \
      flush(r_);                                                                                 \
    }                                                                                            \
  }

  // ---- prologue, in the steady state's issue order (the vmcnt arithmetic counts on it): X(0) | planes(0) | planes(1) | X(1) ----
  GNX_P2_XLOAD(xsA, 0);
  issue_planes(0);
  issue_planes(1);
  GNX_P2_XLOAD(xsB, 1);

  for (int r = 0; r < n_runs; r += 2) {
    GNX_P2_RUN(xsA, r);
    GNX_P2_RUN(xsB, r + 1);
  }
  wait_vm<0>();  // nothing of this block may still be writing LDS when it retires
#undef GNX_P2_RUN
#undef GNX_P2_STEP_SYNC
#undef GNX_P2_XLOAD
}

template <int MT, int NT, int WAVES, int NBUF>
size_t lds_need(int A, int max_runs, int max_wins) {
  return (size_t)NBUF * (2 * NT * LIMBS * 1024) + (size_t)WAVES * MT * 16 * A * sizeof(double) + (size_t)3 * max_runs * sizeof(int) +
         (size_t)max_wins * (A + 1) * sizeof(double);
}

template <int MT, int NT, int WAVES, int NBUF>
hipError_t launch(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  BaseLRLaunch P = L;
  const int haps_per_block = WAVES * MT * 16;
  const int64_t gx = (L.N + haps_per_block - 1) / haps_per_block;
  // window ranges: a multiple of 8 (one XCD each); every range re-walks the runs of its first windows' lead-in, so fewer, longer
  // ranges move fewer bytes and more, shorter ones balance the tail
  const int bpc = tune.lr_bpc > 0 ? tune.lr_bpc : 4;
  int64_t want = ((int64_t)bpc * n_cu + gx - 1) / gx;
  want = std::max<int64_t>(8, ((want + 7) / 8) * 8);
  if (tune.lr_want > 0) want = tune.lr_want;
  int wch = 0, n_ranges = 0;
  size_t lds = 0;
  for (;; want += 8) {
    wch = (int)((L.W + want - 1) / want);
    if (wch < 4) wch = 4;
    n_ranges = (L.W + wch - 1) / wch;
    int max_runs = 0;
    for (int r = 0; r < n_ranges; ++r) {
      const int wa = r * wch, wb = std::min(L.W, wa + wch);
      max_runs = std::max(max_runs, L.h_win_chunk1[(size_t)wb - 1] - L.h_win_chunk0[(size_t)wa]);
    }
    P.max_chunks = max_runs + 8;
    P.max_wins = wch + 2 * L.d.R + 4;
    lds = lds_need<MT, NT, WAVES, NBUF>(L.A, P.max_chunks, P.max_wins);
    if (lds <= (size_t)160 * 1024 || wch == 4) break;
  }
  if (lds > (size_t)160 * 1024) return hipErrorNotSupported;
  lds = std::min(lds + (size_t)std::max(tune.lr_lds_pad, 0), (size_t)160 * 1024);
  const int n_ranges8 = ((n_ranges + 7) / 8) * 8;
  P.wch = wch;
  P.n_htiles = (int)gx;
  GNX_LDS_OPTIN(lds, k_base_logistic_p2<MT, NT, WAVES, NBUF>);
  hipLaunchKernelGGL((k_base_logistic_p2<MT, NT, WAVES, NBUF>), dim3((unsigned)(gx * n_ranges8)), dim3(WAVES * 64), lds, s, P);
  return hipGetLastError();
}

}  // namespace

// returns hipErrorNotSupported when no instantiation fits (the caller widens X to int8 and runs the int8 kernels)
hipError_t gnx_launch_base_logistic_p2(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  if (!L.d.V2 || !L.h_win_chunk0 || !L.h_win_chunk1) return hipErrorNotSupported;
  const bool small = L.N <= 64 * 8;
  const int tm = tune.lr_mt, tw = tune.lr_waves;
  switch (L.d.NT) {
    case 1:
      if (tm == 1 && tw == 4) return launch<1, 1, 4, 3>(L, n_cu, tune, s);
      if (tm == 2 && tw == 8) return launch<2, 1, 8, 3>(L, n_cu, tune, s);
      if (tm == 2 && tw == 16) return launch<2, 1, 16, 3>(L, n_cu, tune, s);
      if (tm == 4 && tw == 4) return launch<4, 1, 4, 3>(L, n_cu, tune, s);
      if (tm == 4 && tw == 8) return launch<4, 1, 8, 3>(L, n_cu, tune, s);
      if (small) return launch<1, 1, 4, 3>(L, n_cu, tune, s);
      return launch<4, 1, 8, 3>(L, n_cu, tune, s);
    case 2:
      if (tm == 1 && tw == 4) return launch<1, 2, 4, 3>(L, n_cu, tune, s);
      if (tm == 1 && tw == 16) return launch<1, 2, 16, 3>(L, n_cu, tune, s);
      if (tm == 2 && tw == 8) return launch<2, 2, 8, 3>(L, n_cu, tune, s);
      if (small) return launch<1, 2, 4, 3>(L, n_cu, tune, s);
      return launch<2, 2, 8, 3>(L, n_cu, tune, s);
    default: return hipErrorNotSupported;
  }
}

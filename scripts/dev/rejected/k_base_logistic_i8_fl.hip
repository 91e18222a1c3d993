// k_base_logistic_i8_fl.hip — k_base_logistic_i8_fl with FLAT column tiles (VERDICT r2 item 4a / DESIGN.md 8.1a).
//
// Same contract, tables, exact int8-limb arithmetic, LDS-direct ring and epilogue as k_base_logistic_i8_fl.hip (reference
// src/Base/base.py:146-180, src/Base/models.py:12-21).  There a 16-column MFMA tile holds 16 class slots of ONE limb, so R*A class
// slots cost ceil(R*A / 16) * 7 tiles: 14 for A = 12 (24 slots, a quarter of the columns padding), 7 for A = 3 (6 slots of 16).
// Here the R*A*7 (slot, limb) columns are laid out flat, q = slot * 7 + limb, and cut into ceil(R*A*7 / 16) tiles: 11 instead of 14
// at A = 12 (-21 % digit-plane bytes, LDS operand reads and MFMAs), 3 instead of 7 at A = 3.  The price is paid at the flush: the
// seven limbs of a finished slot sit in different lanes, so they are exchanged through the wave's epilogue rows in LDS one limb at
// a time (every lane owns (row, class) pairs and adds limb l into its exact int64 halves) before the unchanged recombination
// Z = (double(hi) * 2^24 + double(lo)) * 2^-f_w — the same function of the same seven integers, so B is bit-identical.
// MEASURED (A = 12, chr22, 16 384 haplotypes, scripts/dev/bench_a12.py): 3.21 ms against 2.27 ms for k_base_logistic_i8_dl — the
// flush, already ~0.5 ms of that kernel (twelve float64 exps and divisions per haplotype and window), grows by seven LDS round
// trips per window and wave while every wave of the block waits in it (windows end at the same step for all of them), which costs
// more than the 21 % fewer tiles save.  Not a default anywhere: GNX_LR_FLAT=1 selects it where the model carries flat planes
// (8 .. 11 tiles, A <= 16); kept as the measured answer to "tile the flat column list" (DESIGN.md 5.2).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gnx_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int LIMBS = 7;

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

// MT 16-row tiles per wave, NF flat column tiles, WAVES waves per block, NBUF ring slots; one step = 2 chunks = 128 SNPs.
template <int MT, int NF, int WAVES, int NBUF>
__global__ __launch_bounds__(WAVES * 64) void k_base_logistic_i8_fl(BaseLRLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int CPS = 2;
  constexpr int CHUNK_BYTES = NF * 1024;  // digit planes of one 64-SNP chunk
  constexpr int STEP_BYTES = CPS * CHUNK_BYTES;
  constexpr int THREADS = WAVES * 64;
  constexpr int ROWS = WAVES * MT * 16;           // haplotypes per block
  constexpr int XT_BYTES = ROWS * 128;            // X tile of one step, row-major, 8 swizzled 16-byte pieces per row
  constexpr int XLD = MT * 2;                     // X loads (1 KB = 8 rows each) per wave per step
  constexpr int NKB = STEP_BYTES / 1024;          // 1 KB plane blocks per step
  constexpr int PLD = (NKB + WAVES - 1) / WAVES;  // plane loads per wave per step
  constexpr int G = XLD + PLD;                    // vector-memory instructions per wave per step (constant: clamped, never skipped)
  constexpr int D = NBUF - 1;                     // steps in flight beyond the one being multiplied
  constexpr int KE = 4;                           // (row, class) pairs per lane at the flush: 16 * A / 64 <= 4 (A <= 16, checked by the launcher)
  static_assert(D >= 1 && (D - 1) * G < 64, "ring depth");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int A = L.A, W = L.W, R = L.d.R;
  uint8_t* xt = lds;                               // [NBUF][ROWS][128]
  uint8_t* vbuf = lds + (size_t)NBUF * XT_BYTES;   // [NBUF][STEP_BYTES]
  double* zb = reinterpret_cast<double*>(vbuf + (size_t)NBUF * STEP_BYTES) + (size_t)wave * (MT * 16) * A;
  double* tab_ic = reinterpret_cast<double*>(vbuf + (size_t)NBUF * STEP_BYTES) + (size_t)ROWS * A;  // [max_wins][A] intercepts
  double* tab_sc = tab_ic + (size_t)L.max_wins * A;                                                  // [max_wins] 2^-f_w
  int* tab_j0 = reinterpret_cast<int*>(tab_sc + L.max_wins);
  int* tab_nfl = tab_j0 + L.max_chunks;
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
  const int c_begin = L.d.win_chunk0[wa];
  const int c_end = L.d.win_chunk1[wb - 1];
  const int n_chunks = c_end - c_begin;
  const int n_steps = (n_chunks + CPS - 1) / CPS;
  const int64_t n0b = (int64_t)htile * ROWS;       // first haplotype of the block
  const int64_t n0 = n0b + (int64_t)wave * (MT * 16);

  for (int e = tid; e < n_chunks; e += THREADS) {
    tab_j0[e] = L.d.chunk_j0[c_begin + e];
    tab_nfl[e] = L.d.chunk_nflush[c_begin + e];
    tab_fl0[e] = L.d.chunk_flush0[c_begin + e];
  }
  const int wt0 = max(0, wa - R - 1);
  for (int e = tid; e < L.max_wins; e += THREADS) {
    const int w = min(wt0 + e, W - 1);
    tab_sc[e] = L.d.wscale[w];
    for (int a = 0; a < A; ++a) tab_ic[e * A + a] = L.d.icpt[w * A + a];
  }
  __syncthreads();

  // this lane's part in the X loads: load q of the wave covers rows (wave*XLD + q)*8 .. +8 of the block; the lane fetches
  // row lane>>3, logical piece (lane&7) ^ (lane>>3) (source-side swizzle), i.e. chunk lp>>2 of the step, SNP block lp&3
  const int lp = (lane & 7) ^ (lane >> 3);
  const int8_t* xrow[XLD];
#pragma unroll
  for (int q = 0; q < XLD; ++q) {
    const int64_t n = n0b + (wave * XLD + q) * 8 + (lane >> 3);  // rows >= N-1 read the zero-padded copy of the last row
    xrow[q] = (n >= L.N - 1 ? L.last_row : L.X + n * L.ldx) + 16 * (lp & 3);
  }
  const int8_t* vsrc = L.d.V8F + (size_t)c_begin * CHUNK_BYTES + (size_t)lane * 16;

  // every load is unconditional and clamped (tail steps re-fetch the last step into a slot nobody reads): the number of
  // vector-memory instructions per step is the constant G the vmcnt arithmetic below relies on
  auto issue = [&](int step) {
    const int st = min(step, n_steps - 1);
    const int slot = step % NBUF;
    const int cx = min(st * CPS + (lp >> 2), n_chunks - 1);
    const int j0 = tab_j0[cx];
    uint8_t* xdst = xt + (size_t)slot * XT_BYTES + (size_t)(wave * XLD) * 1024;
#pragma unroll
    for (int q = 0; q < XLD; ++q)
      __builtin_amdgcn_global_load_lds((gptr_t)(xrow[q] + j0), (lptr_t)(xdst + q * 1024), 16, 0, 0);
    const int last_kb = min(CPS, n_chunks - st * CPS) * (CHUNK_BYTES / 1024) - 1;
    const int8_t* src = vsrc + (size_t)st * STEP_BYTES;
    uint8_t* vdst = vbuf + (size_t)slot * STEP_BYTES;
#pragma unroll
    for (int it = 0; it < PLD; ++it) {
      const int kb = min(wave + it * WAVES, last_kb);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kb * 1024), (lptr_t)(vdst + (size_t)kb * 1024), 16, 0, 0);
    }
  };

  v4i acc[MT][NF];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) acc[mt][ft] = v4i{0, 0, 0, 0};

  auto compute_step = [&](int s) {
    const int slot = s % NBUF;
    const uint8_t* sb = vbuf + (size_t)slot * STEP_BYTES;
    const uint8_t* xs = xt + (size_t)slot * XT_BYTES;
#pragma unroll
    for (int k = 0; k < CPS; ++k) {
      const int cl = s * CPS + k;  // chunk index local to the block
      if (cl >= n_chunks) break;
      v4i xa[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int r = wave * (MT * 16) + mt * 16 + i16;
        const int pc = 4 * k + kq;
        xa[mt] = *reinterpret_cast<const v4i*>(xs + r * 128 + ((pc ^ (r & 7)) << 4));
      }
      const v4i* vb = reinterpret_cast<const v4i*>(sb + (size_t)k * CHUNK_BYTES) + lane;
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) {
        const v4i b = vb[ft * 64];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][ft] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[mt], b, acc[mt][ft], 0, 0, 0);
      }

      // ---- piece end: windows that finished here (block-uniform); pieces hold an even number of chunks, so only the second
      // chunk of a step can end one ----
      const int nfl = (k == CPS - 1) ? tab_nfl[cl] : 0;
      if (nfl > 0) {
        const int w0 = tab_fl0[cl];
        for (int w = w0; w < w0 + nfl; ++w) {
          const int cbase = (w % R) * A;
          const double scale = tab_sc[w - wt0];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            // limb by limb through the wave's epilogue rows: lanes that hold limb l of a finished slot publish it as int32
            // zi[row][class], then every lane adds the limb of ITS (row, class) pairs (e = lane + 64 k) into int64 halves
            int* zi = reinterpret_cast<int*>(zb + (size_t)mt * 16 * A);
            long long lo[KE], hi[KE];
#pragma unroll
            for (int k2 = 0; k2 < KE; ++k2) lo[k2] = hi[k2] = 0;
#pragma unroll 1
            for (int l = 0; l < LIMBS; ++l) {  // (a run-time loop: unrolled seven times the exchange's temporaries spill)
#pragma unroll
              for (int ft = 0; ft < NF; ++ft) {
                const int q = ft * 16 + i16, sq = (q * 9363) >> 16, lq = q - 7 * sq;  // slot = q / 7 (exact for q < 2^13), limb
                const int col = sq - cbase;
                if (lq == l && col >= 0 && col < A) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) zi[(4 * kq + r) * A + col] = acc[mt][ft][r];  // C/D layout: row = 4*(lane>>4) + reg
                }
              }
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // zb is wave-private: LDS ops of one wave complete in order
#pragma unroll
              for (int k2 = 0; k2 < KE; ++k2) {
                const int e = lane + 64 * k2;
                const long long v = e < 16 * A ? (long long)zi[e] : 0;
                lo[k2] += l < 3 ? v << (8 * l) : 0;
                hi[k2] += l < 3 ? 0 : v << (8 * (l - 3));
              }
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int k2 = 0; k2 < KE; ++k2) {
              const int e = lane + 64 * k2;
              if (e < 16 * A) zb[(size_t)mt * 16 * A + e] = ((double)hi[k2] * 16777216.0 + (double)lo[k2]) * scale;
            }
#pragma unroll
            for (int ft = 0; ft < NF; ++ft) {
              const int q = ft * 16 + i16, col = ((q * 9363) >> 16) - cbase;
              const bool mine = (col >= 0) && (col < A);
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[mt][ft][r] = mine ? 0 : acc[mt][ft][r];
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // zb is wave-private: LDS ops of one wave complete in order
          if (w >= wa && w < wb) {
            // sigmoid, normaliser and division for the wave's MT*16 rows x A classes, spread over ALL 64 lanes (one lane per
            // row left half the wave idle through 7 double-precision exps and divisions); per element the arithmetic and
            // the class order of the row sum are unchanged
            const int ne = MT * 16 * A;
            const double* ic = tab_ic + (w - wt0) * A;
            for (int e = lane; e < ne; e += 64) {
              const int a = e % A;
              zb[e] = 1.0 / (1.0 + exp(-(zb[e] + ic[a])));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int e = lane; e < ne; e += 64) {
              const int rl = e / A, a = e - rl * A;
              const double* z = zb + rl * A;
              double sum = 0.0;
              for (int c = 0; c < A; ++c) sum += z[c];
              const double v = z[a] / sum;
              const int64_t n = n0 + rl;
              if (n < L.N) {
                const size_t o = ((size_t)n * W + w) * A + a;
                if (L.b64) L.b64[o] = v;
                if (L.b32) L.b32[o] = (float)v;
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    }
  };

  // ---- prologue: steps 0 .. D-1 in flight ----
#pragma unroll
  for (int p = 0; p < D; ++p) issue(p);

  for (int s = 0; s < n_steps; ++s) {
    // the wave's own loads of step s have landed when at most the (D-1)*G younger ones are outstanding (loads retire in
    // order; the epilogue's stores can only make the count conservative)
    wait_vm<(D - 1) * G>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // all shares of step s are in LDS; every wave is done with step s-1
    asm volatile("" ::: "memory");
    issue(s + D);                  // into the slot step s-1 just left
    compute_step(s);
  }
  wait_vm<0>();  // nothing of this block may still be writing LDS when it retires
}

template <int MT, int NF, int WAVES, int NBUF>
size_t lds_need(int A, int max_chunks, int max_wins) {
  return (size_t)NBUF * (WAVES * MT * 16 * 128 + 2 * NF * 1024) + (size_t)WAVES * MT * 16 * A * sizeof(double) +
         (size_t)3 * max_chunks * sizeof(int) + (size_t)max_wins * (A + 1) * sizeof(double);
}

template <int MT, int NF, int WAVES, int NBUF>
hipError_t launch(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  BaseLRLaunch P = L;
  const int haps_per_block = WAVES * MT * 16;
  const int64_t gx = (L.N + haps_per_block - 1) / haps_per_block;
  // window ranges: a multiple of 8 (one XCD each), ~4 blocks per CU in total; more (shorter) ranges if the per-block tables
  // would not fit the LDS next to the ring
  // blocks per CU in total: 4 with one column tile; 2 with more (every range re-reads the digit planes of its first R windows
  // and the planes are the larger share of the traffic there: A = 12, chr22, 16 k haplotypes: 2 -> 2.25 ms, 4 -> 2.39, 8 -> 2.40)
  const int bpc = tune.lr_bpc > 0 ? tune.lr_bpc : (NF > 8 ? 2 : 4);
  int64_t want = ((int64_t)bpc * n_cu + gx - 1) / gx;
  want = std::max<int64_t>(8, ((want + 7) / 8) * 8);
  if (tune.lr_want > 0) want = tune.lr_want;
  int wch = 0, n_ranges = 0;
  size_t lds = 0;
  for (;; want += 8) {
    wch = (int)((L.W + want - 1) / want);
    if (wch < 4) wch = 4;
    n_ranges = (L.W + wch - 1) / wch;
    int max_chunks = 0;
    for (int r = 0; r < n_ranges; ++r) {
      const int wa = r * wch, wb = std::min(L.W, wa + wch);
      max_chunks = std::max(max_chunks, L.h_win_chunk1[(size_t)wb - 1] - L.h_win_chunk0[(size_t)wa]);
    }
    P.max_chunks = max_chunks + 8;
    P.max_wins = wch + 2 * L.d.R + 4;
    lds = lds_need<MT, NF, WAVES, NBUF>(L.A, P.max_chunks, P.max_wins);
    if (lds <= (size_t)160 * 1024 || wch == 4) break;
  }
  if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;
  const int n_ranges8 = ((n_ranges + 7) / 8) * 8;
  P.wch = wch;
  P.n_htiles = (int)gx;
  GNX_LDS_OPTIN(lds, k_base_logistic_i8_fl<MT, NF, WAVES, NBUF>);
  hipLaunchKernelGGL((k_base_logistic_i8_fl<MT, NF, WAVES, NBUF>), dim3((unsigned)(gx * n_ranges8)), dim3(WAVES * 64), lds, s, P);
  return hipGetLastError();
}

}  // namespace

// returns hipErrorNotSupported when no instantiation fits (the caller goes on to k_base_logistic_i8_dl / k_base_logistic_i8)
hipError_t gnx_launch_base_logistic_i8_fl(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  if (!L.h_win_chunk0 || !L.h_win_chunk1 || !L.d.V8F || L.A > 16) return hipErrorNotSupported;
  const bool small = L.N <= 64 * 8;
  const size_t cap = (size_t)160 * 1024 - 6 * 1024;
#define GNX_FL_TRY(MT_, NF_, WV_, NB_) \
  if (lds_need<MT_, NF_, WV_, NB_>(L.A, 0, 0) <= cap) return launch<MT_, NF_, WV_, NB_>(L, n_cu, tune, s);
#define GNX_FL_CASE(NF_)                                                                   \
  case NF_:                                                                                \
    if (small) { GNX_FL_TRY(1, NF_, 4, 2) return hipErrorNotSupported; }                   \
    if (tune.lr_mt != 1) GNX_FL_TRY(2, NF_, 8, 2)                                          \
    GNX_FL_TRY(1, NF_, 8, 2)                                                               \
    return hipErrorNotSupported;
  switch (L.d.NF) {
    GNX_FL_CASE(8) GNX_FL_CASE(9) GNX_FL_CASE(10) GNX_FL_CASE(11)
    default: return hipErrorNotSupported;
  }
#undef GNX_FL_CASE
#undef GNX_FL_TRY
}

// k_base_logistic_i8_dl.hip — the exact int8 logistic pass with LDS-direct loads (gfx950 `global_load_lds_dwordx4`).
//
// Same contract, arithmetic, tables, column slots, flush epilogue and XCD-aware grid as k_base_logistic_i8.hip (reference
// src/Base/base.py:146-180, src/Base/models.py:12-21); only the data movement differs.  The cycle breakdown of that kernel
//  put 29 % of a step into issuing loads, 10 % into publishing register stages to LDS and 26 % into waiting
// at two block barriers per step, at 184 VGPRs of which 48 only park bytes on their way to LDS.  Here
//   * X and the digit planes go HBM/L2 -> LDS directly: no staging VGPRs, no ds_write pass;
//   * an NBUF-deep LDS ring (X tile + planes of one 128-SNP step per slot) replaces the two register stages: steps
//     s+1 .. s+NBUF-1 are in flight while step s is multiplied;
//   * ONE block barrier per step: after `s_waitcnt vmcnt` for the wave's own share of step s, the barrier tells every wave
//     that (a) all shares of step s have landed and (b) everybody is done with step s-1, whose slot is refilled next.
// LDS-direct loads write lane-linear (wave-uniform base + lane*16 B).  The MFMA A operand wants 16 rows x the same 16-byte
// piece per instruction, which on a row-major [row][128 B] tile is a 16-way bank conflict; the cure is the usual XOR
// swizzle, applied on the SOURCE side: lane i of a load fetches row i>>3, logical piece (i&7) ^ (i>>3), so the 8 lanes
// of a row still cover the same 128 contiguous bytes (whole cache lines), and the piece p of row r sits at physical slot
// p ^ (r&7), which the operand read un-swizzles.  Global addresses need no alignment (checked on the hardware for every
// byte offset and odd row strides: scripts/dev/glds_align_probe.hip).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gnx_internal.h"
#include "gnx_exp.h"

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

// MT 16-row tiles per wave, NT column tiles, WAVES waves per block, NBUF ring slots; one step = 2 chunks = 128 SNPs.
template <int MT, int NT, int WAVES, int NBUF>
__global__ __launch_bounds__(WAVES * 64) void k_base_logistic_i8_dl(BaseLRLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int CPS = 2;
  constexpr int CHUNK_BYTES = NT * LIMBS * 1024;  // digit planes of one 64-SNP chunk
  constexpr int STEP_BYTES = CPS * CHUNK_BYTES;
  constexpr int THREADS = WAVES * 64;
  constexpr int ROWS = WAVES * MT * 16;           // haplotypes per block
  constexpr int XT_BYTES = ROWS * 128;            // X tile of one step, row-major, 8 swizzled 16-byte pieces per row
  constexpr int XLD = MT * 2;                     // X loads (1 KB = 8 rows each) per wave per step
  constexpr int NKB = STEP_BYTES / 1024;          // 1 KB plane blocks per step
  constexpr int PLD = (NKB + WAVES - 1) / WAVES;  // plane loads per wave per step
  constexpr int G = XLD + PLD;                    // vector-memory instructions per wave per step (constant: clamped, never skipped)
  constexpr int D = NBUF - 1;                     // steps in flight beyond the one being multiplied
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
  const int8_t* vsrc = L.d.V8 + (size_t)c_begin * CHUNK_BYTES + (size_t)lane * 16;

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

  v4i acc[MT][NT][LIMBS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) acc[mt][nt][l] = v4i{0, 0, 0, 0};

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
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int l = 0; l < LIMBS; ++l) {
          const v4i b = vb[(nt * LIMBS + l) * 64];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt][nt][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[mt], b, acc[mt][nt][l], 0, 0, 0);
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
          if (w >= wa && w < wb && (L.flags & 8)) {
            // ABLATION (GNX_LR_FLAGS=8, timing only): raw logits (z + intercept) instead of probabilities, i.e. the flush without
            // its float64 exps and divisions.  A = 12, chr22: 2.24 -> 2.05 ms — the epilogue's arithmetic is 8 % of this kernel;
            // moving it into the next kernel's pre-pass would move ~1 ms of VALU work per 25 000 chr1 haplotypes with it
            const int ne = MT * 16 * A;
            const double* ic = tab_ic + (w - wt0) * A;
            for (int e = lane; e < ne; e += 64) {
              const int rl = e / A, a = e - rl * A;
              const int64_t n = n0 + rl;
              if (n < L.N) {
                const size_t o = ((size_t)n * W + w) * A + a;
                if (L.b64) L.b64[o] = zb[e] + ic[a];
                if (L.b32) L.b32[o] = (float)(zb[e] + ic[a]);
              }
            }
          } else if (w >= wa && w < wb) {
            // sigmoid, normaliser and division for the wave's MT*16 rows x A classes, spread over ALL 64 lanes (one lane per
            // row left half the wave idle through 7 double-precision exps and divisions); per element the arithmetic and
            // the class order of the row sum are unchanged
            const int ne = MT * 16 * A;
            const double* ic = tab_ic + (w - wt0) * A;
            for (int e = lane; e < ne; e += 64) {
              const int a = e % A;
              zb[e] = gnx_sigmoid(zb[e] + ic[a]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int e = lane; e < ne; e += 64) {
              const int rl = e / A, a = e - rl * A;
              const double* z = zb + rl * A;
              double sum = 0.0;
              for (int c = 0; c < A; ++c) sum += z[c];
              const double v = z[a] * gnx_rcp_nr(sum);
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

template <int MT, int NT, int WAVES, int NBUF>
size_t lds_need(int A, int max_chunks, int max_wins) {
  return (size_t)NBUF * (WAVES * MT * 16 * 128 + 2 * NT * LIMBS * 1024) + (size_t)WAVES * MT * 16 * A * sizeof(double) +
         (size_t)3 * max_chunks * sizeof(int) + (size_t)max_wins * (A + 1) * sizeof(double);
}

template <int MT, int NT, int WAVES, int NBUF>
hipError_t launch(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  BaseLRLaunch P = L;
  P.flags = L.flags | (tune.lr_flags & 8);  // bit 3: ablation, raw logits instead of probabilities (GNX_LR_FLAGS=8, timing only)
  const int haps_per_block = WAVES * MT * 16;
  const int64_t gx = (L.N + haps_per_block - 1) / haps_per_block;
  // window ranges: a multiple of 8 (one XCD each), ~4 blocks per CU in total; more (shorter) ranges if the per-block tables
  // would not fit the LDS next to the ring
  // blocks per CU in total: 4 with one column tile; 2 with more (every range re-reads the digit planes of its first R windows
  // and the planes are the larger share of the traffic there: A = 12, chr22, 16 k haplotypes: 2 -> 2.25 ms, 4 -> 2.39, 8 -> 2.40)
  const int bpc = tune.lr_bpc > 0 ? tune.lr_bpc : (NT >= 2 ? 2 : 4);
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
    lds = lds_need<MT, NT, WAVES, NBUF>(L.A, P.max_chunks, P.max_wins);
    if (lds <= (size_t)160 * 1024 || wch == 4) break;
  }
  if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;
  const int n_ranges8 = ((n_ranges + 7) / 8) * 8;
  P.wch = wch;
  P.n_htiles = (int)gx;
  GNX_LDS_OPTIN(lds, k_base_logistic_i8_dl<MT, NT, WAVES, NBUF>);
  hipLaunchKernelGGL((k_base_logistic_i8_dl<MT, NT, WAVES, NBUF>), dim3((unsigned)(gx * n_ranges8)), dim3(WAVES * 64), lds, s, P);
  return hipGetLastError();
}

}  // namespace

// returns hipErrorNotSupported when no instantiation fits (the caller falls back to k_base_logistic_i8)
hipError_t gnx_launch_base_logistic_i8_dl(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  if (!L.h_win_chunk0 || !L.h_win_chunk1) return hipErrorNotSupported;
  const bool small = L.N <= 64 * 8;
  const int nbuf = tune.lr_nbuf;  // 0 = built-in choice
  // the ring, the per-wave epilogue rows (ROWS x A doubles) and ~6 KB of per-block tables share the 160 KB
  const size_t cap = (size_t)160 * 1024 - 6 * 1024;
#define GNX_DL_TRY(MT_, NT_, WV_, NB_) \
  if (lds_need<MT_, NT_, WV_, NB_>(L.A, 0, 0) <= cap) return launch<MT_, NT_, WV_, NB_>(L, n_cu, tune, s);
  switch (L.d.NT) {
    case 1:
      if (small) { GNX_DL_TRY(1, 1, 4, 3) GNX_DL_TRY(1, 1, 4, 2) return hipErrorNotSupported; }
      if (tune.lr_waves == 16) { if (nbuf != 2) GNX_DL_TRY(1, 1, 16, 3) GNX_DL_TRY(1, 1, 16, 2) }
      if (tune.lr_waves == 12) { GNX_DL_TRY(2, 1, 12, 2) }   // 384 rows per block: 2/3 of the plane bytes per X byte, ring depth 2
      if (nbuf != 2) GNX_DL_TRY(2, 1, 8, 3)
      GNX_DL_TRY(2, 1, 8, 2)
      GNX_DL_TRY(1, 1, 8, 3)
      return hipErrorNotSupported;
    case 2:
      if (small) { GNX_DL_TRY(1, 2, 4, 2) return hipErrorNotSupported; }
      // 16 waves x 16 rows: 56 accumulator registers per wave, 4 waves per SIMD hide the operand reads (2.39 vs 2.56 ms for
      // 8 waves x 32 rows at A = 12)
      if (tune.lr_waves != 8 && tune.lr_mt != 2) GNX_DL_TRY(1, 2, 16, 2)
      if (tune.lr_mt != 1) GNX_DL_TRY(2, 2, 8, 2)
      if (nbuf != 2) GNX_DL_TRY(1, 2, 8, 3)
      GNX_DL_TRY(1, 2, 8, 2)
      return hipErrorNotSupported;
    case 3:  // 33 .. 48 class columns per SNP (e.g. A = 24 at the default context)
      GNX_DL_TRY(1, 3, 8, 2)
      GNX_DL_TRY(1, 3, 4, 2)
      return hipErrorNotSupported;
    case 4:  // 49 .. 64 class columns (A = 32): 112 accumulator registers per wave, no spill without the staging registers
      GNX_DL_TRY(1, 4, 4, 2)
      return hipErrorNotSupported;
    default: return hipErrorNotSupported;
  }
#undef GNX_DL_TRY
}

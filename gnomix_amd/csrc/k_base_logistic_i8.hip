// k_base_logistic_i8.hip — the logistic base pass on the int8 matrix pipe, EXACT arithmetic (gfx950).
//
// Same contract, tiling and piece/chunk tables as k_base_logistic.hip (reference src/Base/base.py:146-180,
// src/Base/models.py:12-21), different arithmetic:
//  * X holds {0,1,2}: the 16 int8 bytes a lane loads from HBM ARE the A operand of
//    v_mfma_i32_16x16x64_i8 (row = haplotype lane&15, k = the 16 SNPs of k-block lane>>4) — no conversion;
//  * each float64 weight is fixed-point: q = round(c * 2^f_w) (f_w per window so |q| < 2^54), split into
//    LIMBS = 7 balanced base-256 digits d_l in [-128,127]; digit plane l of the 14 (= R*A) class columns is
//    one B operand.  Products <= 2*128 and K <= a few thousand keep every int32 accumulator exact, so the
//    logit  sum_k x_k q_k  is computed EXACTLY and order-independently;
//  * recombination: hi = sum_{l>=3} acc_l 2^{8(l-3)}, lo = sum_{l<3} acc_l 2^{8l} in int64 (both < 2^53), then
//    Z = (double(hi) * 2^24 + double(lo)) * 2^-f_w : ONE float64 rounding.  Versus the reference's float64
//    dot product the only difference is the 2^-56 relative weight quantisation — below BLAS reordering noise;
//  * per 64-SNP chunk: 7 MFMAs (~16 cycles each) instead of 16 f64 MFMAs (64 cycles each): the matrix pipe
//    drops from the binding resource to ~1/3 busy and the pass becomes HBM-bound (reads X once);
//  * the digit planes of a chunk (7 KB) are shared by all waves of the block through a double-buffered LDS
//    window; X streams HBM -> VGPR two chunks ahead; the epilogue needs no LDS (every lane owns whole
//    logits; only the A-way normaliser crosses lanes, by a 16-lane butterfly).
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gnx_internal.h"
#include "gnx_exp.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) xbytes16 { v4i v; };

constexpr int LIMBS = 7;

__device__ __forceinline__ v4i load_x16(const int8_t* p) {
  // one unconditional (unaligned) global_load_dwordx4: no branch, so consecutive loads stay in flight together.
  // Reads may run up to 63 bytes past the end of a row: rows other than the last read their successor, the last
  // row is served from a zero-padded copy (BaseLRLaunch::last_row), bytes past C only ever meet zero weights.
  xbytes16 r;
  __builtin_memcpy(&r, p, 16);
  return r.v;
}

__device__ __forceinline__ double combine(const v4i (&acc)[LIMBS], int reg, double scale) {
  long long lo = (long long)acc[0][reg] + ((long long)acc[1][reg] << 8) + ((long long)acc[2][reg] << 16);
  long long hi = (long long)acc[3][reg] + ((long long)acc[4][reg] << 8) + ((long long)acc[5][reg] << 16) +
                 ((long long)acc[6][reg] << 24);
  return ((double)hi * 16777216.0 + (double)lo) * scale;
}

// MT 16-row tiles per wave, NT column tiles, WAVES waves per block; one pipeline step = 128 SNPs (2 chunks).
//
// X path: every thread fetches 16-byte pieces such that 8 consecutive lanes cover 128 contiguous bytes of ONE
// haplotype row (whole cache lines instead of 16 rows x 64 B per wave instruction), parks them in VGPRs for
// TWO compute phases (two register stages: enough bytes in flight to cover HBM latency), then writes them into
// an LDS tile laid out
//     slot(piece p, row r) = p*ROWS + (r ^ p)          (16 bytes per slot)
// which is conflict-free both for the ds_write_b128 of the fetch layout (8 lanes = 8 pieces of one row) and for
// the ds_read_b128 of the MFMA A-operand layout (lane&15 = row, lane>>4 = piece within the chunk).
// workgroup barrier that orders LDS traffic only: __syncthreads() also carries a fence that makes hipcc drain the
// outstanding global loads (s_waitcnt vmcnt(0)), which would serialise the HBM prefetch with every barrier
#define GNX_LDS_BARRIER()                                   \
  {                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("" ::: "memory");                          \
  }

#define GNX_STAGE_LOAD(XS, VS, STEP)                                                                  \
  {                                                                                                   \
    const int st_ = min((STEP), n_steps - 1); /* clamped: tail iterations re-fetch the last step */ \
    int cx_ = st_ * CPS + (fp >> 2);                                                                  \
    if (cx_ > n_chunks - 1) cx_ = n_chunks - 1;                                                       \
    const int j0_ = tab_j0[cx_];                                                                      \
    _Pragma("unroll") for (int q = 0; q < XPT; ++q)                                                   \
        XS[q] = load_x16(frow[q] + j0_);                 \
    const int c0_ = c_begin + st_ * CPS;                                                              \
    const int last_ = min(CPS, c_end - c0_) * (CHUNK_BYTES / 16) - 1;                                 \
    const v4i* src_ = reinterpret_cast<const v4i*>(L.d.V8 + (size_t)c0_ * CHUNK_BYTES);           \
    _Pragma("unroll") for (int v = 0; v < VPT; ++v) VS[v] = src_[min(v * THREADS + tid, last_)];      \
  }

#define GNX_STAGE_STORE(XS, VS, STEP)                                                                 \
  {                                                                                                   \
    _Pragma("unroll") for (int q = 0; q < XPT; ++q) *reinterpret_cast<v4i*>(xt + fslot[q]) = XS[q];   \
    v4i* dst_ = reinterpret_cast<v4i*>(vbuf + (size_t)((STEP) & 1) * STEP_BYTES);                 \
    _Pragma("unroll") for (int v = 0; v < VPT; ++v) {                                                 \
      const int e_ = v * THREADS + tid;                                                               \
      if (e_ < STEP_BYTES / 16) dst_[e_] = VS[v];                                                     \
    }                                                                                                 \
  }

template <int MT, int NT, int WAVES, int CPS>
__global__ __launch_bounds__(WAVES * 64) void k_base_logistic_i8(BaseLRLaunch L) {
  static_assert(CPS == 2, "a step is 2 chunks = 8 pieces of 16 SNPs");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int CHUNK_BYTES = NT * LIMBS * 1024;  // digit planes of one 64-SNP chunk
  constexpr int STEP_BYTES = CPS * CHUNK_BYTES;
  constexpr int THREADS = WAVES * 64;
  constexpr int ROWS = WAVES * MT * 16;           // haplotypes per block
  constexpr int XT_BYTES = ROWS * 128;            // X tile of one step
  constexpr int XPT = ROWS * 8 / THREADS;         // 16-byte X pieces per thread per step (= MT*2)
  constexpr int VPT = (STEP_BYTES / 16 + THREADS - 1) / THREADS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int A = L.A, W = L.W, R = L.d.R;
  uint8_t* xt = lds;                               // [8 pieces][ROWS] x 16 B, single buffer
  uint8_t* vbuf = lds + XT_BYTES;                  // [2][STEP_BYTES]
  double* zb = reinterpret_cast<double*>(vbuf + 2 * STEP_BYTES) + (size_t)wave * (MT * 16) * A;
  double* tab_ic = reinterpret_cast<double*>(vbuf + 2 * STEP_BYTES) + (size_t)ROWS * A;  // [max_wins][A] intercepts
  double* tab_sc = tab_ic + (size_t)L.max_wins * A;                                       // [max_wins] 2^-f_w
  int* tab_j0 = reinterpret_cast<int*>(tab_sc + L.max_wins);
  int* tab_nfl = tab_j0 + L.max_chunks;
  int* tab_fl0 = tab_nfl + L.max_chunks;

  // XCD-aware decomposition: consecutive block ids go to consecutive XCDs, so give all blocks of one
  // window range to ONE XCD (its L2 then serves the range's weights to every haplotype tile)
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

  // the block's slice of the chunk tables lives in LDS (no dependent global loads in the pipeline)
  for (int e = tid; e < n_chunks; e += THREADS) {
    tab_j0[e] = L.d.chunk_j0[c_begin + e];
    tab_nfl[e] = L.d.chunk_nflush[c_begin + e];
    tab_fl0[e] = L.d.chunk_flush0[c_begin + e];
  }
  // windows that can be flushed while walking [c_begin, c_end): from the first flush on, wch + R + 1 of them at most
  const int wt0 = max(0, wa - R - 1);
  for (int e = tid; e < L.max_wins; e += THREADS) {
    const int w = min(wt0 + e, W - 1);
    tab_sc[e] = L.d.wscale[w];
    for (int a = 0; a < A; ++a) tab_ic[e * A + a] = L.d.icpt[w * A + a];
  }
  __syncthreads();

  // fetch layout: piece = tid & 7 (16 SNPs), rows tid>>3, +THREADS/8, ...
  const int fp = tid & 7;
  const int8_t* frow[XPT];
  int fslot[XPT];
#pragma unroll
  for (int q = 0; q < XPT; ++q) {
    const int r = (tid >> 3) + q * (THREADS / 8);
    const int64_t n = n0b + r;  // rows >= N-1 read the padded copy of the last row (rows past N are never written)
    frow[q] = (n >= L.N - 1 ? L.last_row : L.X + n * L.ldx) + 16 * (fp & 3);
    fslot[q] = (fp * ROWS + (r ^ fp)) * 16;
  }

  v4i acc[MT][NT][LIMBS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) acc[mt][nt][l] = v4i{0, 0, 0, 0};

  // two register stages: stage A holds even steps, stage B odd steps
  v4i xsA[XPT], xsB[XPT];
  v4i vsA[VPT], vsB[VPT];

  auto compute_step = [&](int s) {
    const uint8_t* sb = vbuf + (size_t)(s & 1) * STEP_BYTES;
#pragma unroll
    for (int k = 0; k < CPS; ++k) {
      const int cl = s * CPS + k;  // chunk index local to the block
      if (cl >= n_chunks) break;
      v4i xa[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int r = wave * (MT * 16) + mt * 16 + i16;
        const int pc = 4 * k + kq;
        xa[mt] = *reinterpret_cast<const v4i*>(xt + (pc * ROWS + (r ^ pc)) * 16);
      }
      const v4i* vb = reinterpret_cast<const v4i*>(sb + (size_t)k * CHUNK_BYTES) + lane;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int l = 0; l < LIMBS; ++l) {
          const v4i b = vb[(nt * LIMBS + l) * 64];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            acc[mt][nt][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[mt], b, acc[mt][nt][l], 0, 0, 0);
          }
        }

      // ---- piece end: windows that finished here (block-uniform); pieces hold an even number of chunks (model load
      // pads them), so only the second chunk of a step can end one ----
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
          if (w >= wa && w < wb && lane < MT * 16) {
            const int64_t n = n0 + lane;
            double* z = zb + lane * A;
            double sum = 0.0;
            for (int a = 0; a < A; ++a) {
              const double p = gnx_sigmoid(z[a] + tab_ic[(w - wt0) * A + a]);
              z[a] = p;
              sum += p;
            }
            if (n < L.N) {
              const size_t o = ((size_t)n * W + w) * A;
              const double rs = gnx_rcp_nr(sum);
              for (int a = 0; a < A; ++a) {
                const double v = z[a] * rs;
                if (L.b64) L.b64[o + a] = v;
                if (L.b32) L.b32[o + a] = (float)v;
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    }
  };

  // ---- prologue: step 0 into LDS, step 1 in flight in stage B ----
  GNX_STAGE_LOAD(xsA, vsA, 0);
  GNX_STAGE_STORE(xsA, vsA, 0);
  GNX_STAGE_LOAD(xsB, vsB, 1);
  __syncthreads();

  for (int s = 0; s < n_steps; s += 2) {
    // straight-line body (no branches around the memory operations: hipcc's waitcnt insertion stays counted)
    // even step s: request s+2 into stage A (free: step s is already in LDS), compute, publish s+1 from stage B
    GNX_STAGE_LOAD(xsA, vsA, s + 2);
    compute_step(s);
    GNX_LDS_BARRIER();  // every wave is done with the X tile and with plane buffer (s+1)&1's previous contents
    GNX_STAGE_STORE(xsB, vsB, s + 1);
    GNX_LDS_BARRIER();
    // odd step s+1: request s+3 into stage B, compute, publish s+2 from stage A
    GNX_STAGE_LOAD(xsB, vsB, s + 3);
    compute_step(s + 1);  // no-op past the last chunk
    GNX_LDS_BARRIER();
    GNX_STAGE_STORE(xsA, vsA, s + 2);
    GNX_LDS_BARRIER();
  }
}
#undef GNX_LDS_BARRIER
#undef GNX_STAGE_LOAD
#undef GNX_STAGE_STORE

template <int MT, int NT, int WAVES, int CPS>
hipError_t launch(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  BaseLRLaunch P = L;
  const int haps_per_block = WAVES * MT * 16;
  const int64_t gx = (L.N + haps_per_block - 1) / haps_per_block;
  // window ranges: a multiple of 8 (one XCD each), ~4 blocks per CU in total; more (shorter) ranges if the per-block
  // chunk / window tables would not fit the LDS next to the tiles (long chromosomes with few haplotype tiles)
  const int bpc = tune.lr_bpc > 0 ? tune.lr_bpc : 4;  // measured on chr22 / 10 k haplotypes: 32 ranges 1.17 ms, 24 ranges 1.21, 16 ranges 1.31
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
    if (L.h_win_chunk0 && L.h_win_chunk1) {  // exact: the longest chunk span any range walks
      for (int r = 0; r < n_ranges; ++r) {
        const int wa = r * wch, wb = std::min(L.W, wa + wch);
        max_chunks = std::max(max_chunks, L.h_win_chunk1[(size_t)wb - 1] - L.h_win_chunk0[(size_t)wa]);
      }
      max_chunks += 8;
    } else {
      max_chunks = (wch + L.d.R + 2) * L.d.max_piece_chunks + 8;
    }
    P.max_chunks = max_chunks;
    P.max_wins = wch + 2 * L.d.R + 4;
    lds = (size_t)WAVES * MT * 16 * 128 + (size_t)2 * CPS * NT * LIMBS * 1024 + (size_t)WAVES * MT * 16 * L.A * sizeof(double) +
          (size_t)3 * P.max_chunks * sizeof(int) + (size_t)P.max_wins * (L.A + 1) * sizeof(double);
    if (lds <= (size_t)160 * 1024 || wch == 4) break;
  }
  if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;
  lds = std::min(lds + (size_t)std::max(tune.lr_lds_pad, 0), (size_t)160 * 1024);
  const int n_ranges8 = ((n_ranges + 7) / 8) * 8;
  P.wch = wch;
  P.n_htiles = (int)gx;
  GNX_LDS_OPTIN(lds, k_base_logistic_i8<MT, NT, WAVES, CPS>);
  hipLaunchKernelGGL((k_base_logistic_i8<MT, NT, WAVES, CPS>), dim3((unsigned)(gx * n_ranges8)), dim3(WAVES * 64), lds, s, P);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess && tune.debug) std::fprintf(stderr, "k_base_logistic_i8<%d,%d,%d>: lds=%zu grid=%lld wch=%d max_chunks=%d max_wins=%d\n", MT, NT, WAVES, lds, (long long)(gx * n_ranges8), wch, P.max_chunks, P.max_wins);
  return e;
}

}  // namespace

hipError_t gnx_launch_base_logistic_i8(const BaseLRLaunch& L0, int n_cu, const gnx_tune& tune, hipStream_t s) {
  if (L0.N <= 0) return hipSuccess;
  BaseLRLaunch L = L0;
  const bool small = L.N <= 64 * 8;
  // tuning / ablation knobs (development only): GNX_LR_TUNE="mt,waves,cps", GNX_LR_FLAGS=bitmask
  const int tm = tune.lr_mt, tw = tune.lr_waves;
  L.flags = tune.lr_flags;
  if (L.d.NT == 1 && tm) {
    if (tm == 1 && tw == 4) return launch<1, 1, 4, 2>(L, n_cu, tune, s);
    if (tm == 1 && tw == 8) return launch<1, 1, 8, 2>(L, n_cu, tune, s);
    if (tm == 1 && tw == 16) return launch<1, 1, 16, 2>(L, n_cu, tune, s);
    if (tm == 2 && tw == 4) return launch<2, 1, 4, 2>(L, n_cu, tune, s);
    if (tm == 2 && tw == 8) return launch<2, 1, 8, 2>(L, n_cu, tune, s);
    if (tm == 4 && tw == 4) return launch<4, 1, 4, 2>(L, n_cu, tune, s);
    if (tm == 4 && tw == 8) return launch<4, 1, 8, 2>(L, n_cu, tune, s);
    return hipErrorInvalidValue;
  }
  if (L.d.NT == 2 && tm) {
    if (tm == 1 && tw == 4) return launch<1, 2, 4, 2>(L, n_cu, tune, s);
    if (tm == 1 && tw == 8) return launch<1, 2, 8, 2>(L, n_cu, tune, s);
    if (tm == 1 && tw == 16) return launch<1, 2, 16, 2>(L, n_cu, tune, s);
    if (tm == 2 && tw == 4) return launch<2, 2, 4, 2>(L, n_cu, tune, s);
    if (tm == 2 && tw == 8) return launch<2, 2, 8, 2>(L, n_cu, tune, s);
    return hipErrorInvalidValue;
  }
  switch (L.d.NT) {
    case 1: return small ? launch<1, 1, 4, 2>(L, n_cu, tune, s) : launch<2, 1, 8, 2>(L, n_cu, tune, s);
    case 2: return small ? launch<1, 2, 4, 2>(L, n_cu, tune, s) : launch<2, 2, 8, 2>(L, n_cu, tune, s);  // A=12: 2.0 TB/s measured
    case 3: return launch<1, 3, 4, 2>(L, n_cu, tune, s);
    case 4: return launch<1, 4, 4, 2>(L, n_cu, tune, s);
    default: return hipErrorInvalidValue;
  }
}

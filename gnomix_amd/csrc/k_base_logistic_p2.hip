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
//   * pieces start on a 32-bit boundary of the packed row: a piece [b0, b1) is walked from SNP b0 & ~15 in runs of 256 SNPs (4 MFMA
//     entries), the up to 15 SNPs before b0 and everything from b1 on meet zero weights (M = 1000, context 500: 500 + 12 SNPs = exactly
//     two runs); windows are flushed after the last run of their piece.  Loads that are not dword-aligned are split by the texture
//     addresser (~4.5 L1 tag accesses per lane instead of ~0.5: the first version, byte-aligned, ran at the L1's tag rate);
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

constexpr int LIMBS = 7;

__device__ __forceinline__ v4i unpack16(int w) {  // 16 two-bit fields -> 16 int8 bytes, field 4b + d at byte b of reg d
  const unsigned u = (unsigned)w;
  v4i r;
  r[0] = (int)(u & 0x03030303u);
  r[1] = (int)((u >> 2) & 0x03030303u);
  r[2] = (int)((u >> 4) & 0x03030303u);
  r[3] = (int)((u >> 6) & 0x03030303u);
  return r;
}

// Z = (hi 2^24 + lo) 2^-f_w with hi = sum_{l>=3} acc_l 2^{8(l-3)}, lo = sum_{l<3} acc_l 2^{8l}: the same exact integers
// k_base_logistic_i8's combine() forms in int64 (|acc_l| < 2^20 for K <= 2500 SNPs, so |lo| < 2^37, |hi| < 2^45: every fma below is
// exact), here on the float64 pipe: 7 conversions + 5 fmas instead of ~60 instructions of 64-bit integer arithmetic and two
// int64 -> double conversions per accumulator register.  One rounding (the last addition), as there: bit-identical Z.
__device__ __forceinline__ double combine(const v4i (&acc)[LIMBS], int reg, double scale) {
  const double lo = __builtin_fma(__builtin_fma((double)acc[2][reg], 256.0, (double)acc[1][reg]), 256.0, (double)acc[0][reg]);
  const double hi = __builtin_fma(__builtin_fma(__builtin_fma((double)acc[6][reg], 256.0, (double)acc[5][reg]), 256.0, (double)acc[4][reg]),
                                  256.0, (double)acc[3][reg]);
  return (hi * 16777216.0 + lo) * scale;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MT 16-row tiles per wave, NT column tiles, WAVES waves per block, XSN LDS stages of X runs in flight, ZT = 1: the flush goes
// through a 16-row scratch one tile at a time (a quarter of the epilogue rows in LDS), 0: all MT tiles at once.
// Plane ring: 3 slots of one half-run (2 entries) each.
template <int MT, int NT, int WAVES, int XSN, int ZT, int SPLIT>
__global__ __launch_bounds__(WAVES * 64) void k_base_logistic_p2(BaseLRLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int NBUF = 3;
  constexpr int EPS = 2;                          // MFMA entries per plane step (half a run)
  constexpr int ENTRY_BYTES = NT * LIMBS * 1024;  // digit planes of one entry (64 k positions)
  constexpr int STEP_BYTES = EPS * ENTRY_BYTES;
  constexpr int THREADS = WAVES * 64;
  constexpr int NKB = STEP_BYTES / 1024;          // 1 KB plane blocks per step
  constexpr int PLD = (NKB + WAVES - 1) / WAVES;  // plane loads per wave per step
  constexpr int D = NBUF - 1;                     // plane steps in flight beyond the one being multiplied
  constexpr int ZROWS = ZT ? 16 : MT * 16;        // rows of the wave's epilogue scratch
  // Issue order per run r (every wave, unconditional, clamped):  X(r+XSN) | planes(2r+2) | planes(2r+3).  When a wave waits for the
  // planes of the even step 2r, the only younger vector-memory instructions are the planes of 2r+1; for the odd step 2r+1 they are
  // X(r+XSN) and the planes of 2r+2.  X(r) is older than the planes of step 2r (XSN >= 1), so the even wait covers it too.
  constexpr int WAIT_EVEN = PLD, WAIT_ODD = PLD + MT;
  static_assert(XSN >= 1 && WAIT_ODD < 64, "vmcnt is a 6-bit counter");
  // SPLIT = 1: the two streams are issued by DIFFERENT waves, because vmcnt retires in order: a wave that waits for its planes of
  // step s also waits for every X load it issued before them, so with both streams in one wave X never gets more than ~one run of
  // lead however many stages it has (measured: 1, 2 and 3 stages run the same 0.53 ms skeleton).  Even waves issue all plane loads,
  // odd waves issue the X runs of themselves and of their even neighbour; the block barrier of every step publishes both.
  //   even wave, any step s:   planes(s) were issued two steps ago, only the planes of s+1 are younger     -> vmcnt(PLDS)
  //   odd wave, even step 2r:  X(r+XSN) is issued at step 2r+1, so X(r+1) .. X(r+XSN-1) are younger than X(r) -> vmcnt((XSN-1) 2 MT)
  // X(r+XSN) goes into the stage of X(r) one barrier after every wave has read it: 2 XSN - 1 steps of lead.
  constexpr int PLDS = (NKB + WAVES / 2 - 1) / (WAVES / 2);
  constexpr int WAIT_XS = (XSN - 1) * 2 * MT;
  static_assert(!SPLIT || (WAIT_XS < 64 && WAVES % 2 == 0), "vmcnt is a 6-bit counter");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool xwave = SPLIT && (wave & 1);
  const int i16 = lane & 15, kq = lane >> 4;
  const int A = L.A, W = L.W, R = L.d.R;
  uint8_t* vbuf = lds;                                                           // [NBUF][STEP_BYTES]
  uint8_t* xl = vbuf + (size_t)NBUF * STEP_BYTES + (size_t)wave * (XSN * MT * 1024);  // [WAVES][XSN][MT][64 lanes][16 B], wave-private
  double* zb0 = reinterpret_cast<double*>(vbuf + (size_t)NBUF * STEP_BYTES + (size_t)WAVES * XSN * MT * 1024);
  double* zb = zb0 + (size_t)wave * ZROWS * A;
  double* tab_ic = zb0 + (size_t)WAVES * ZROWS * A;  // [max_wins][A] intercepts
  double* tab_sc = tab_ic + (size_t)L.max_wins * A;   // [max_wins] 2^-f_w
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
  const int64_t n0 = (int64_t)htile * (WAVES * MT * 16) + (int64_t)wave * (MT * 16);  // first haplotype of the wave

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

  // the lane's rows: tile mt, row i16; its 16 bytes of a run = packed bytes [16 kq, 16 kq + 16) of the run
  constexpr int XT = SPLIT ? 2 * MT : MT;  // tiles a loading wave fetches: SPLIT: those of waves wave-1 and wave
  const uint8_t* xrow[XT];
#pragma unroll
  for (int mt = 0; mt < XT; ++mt) {
    const int64_t n = n0 + (SPLIT ? mt - MT : mt) * 16 + i16;  // rows >= N-1 read the zero-padded copy of the last row (rows past N are never written)
    xrow[mt] = (n >= L.N - 1 ? reinterpret_cast<const uint8_t*>(L.last_row) : reinterpret_cast<const uint8_t*>(L.X) + n * L.ldx) + 16 * kq;
  }
  const int8_t* vsrc = L.d.V2 + (size_t)r_begin * (2 * STEP_BYTES) + (size_t)lane * 16;

  // every load is unconditional and clamped (tail steps re-fetch the last one into a slot nobody reads): the number of
  // vector-memory instructions per step is the constant the vmcnt arithmetic relies on.  X goes HBM -> LDS directly too
  // (lane-linear: the lane reads back exactly the 16 bytes it fetched), so that no register waits on a load the compiler tracks:
  // hipcc's waitcnt insertion gives up on VGPR loads in flight across the flush's loops and stores (s_waitcnt vmcnt(0) at first use)
  auto issue_x = [&](int run) {
    if (L.flags & 8) return;
    const int rb = tab_rb[min(run, n_runs - 1)];
    if (SPLIT) {  // tiles 0..MT-1 belong to wave - 1, whose stages precede this wave's
#pragma unroll
      for (int mt = 0; mt < XT; ++mt) {
        uint8_t* dst = xl + (mt < MT ? -(XSN * MT * 1024) : 0) + (size_t)(run % XSN) * (MT * 1024) + (mt % MT) * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(xrow[mt] + rb), (lptr_t)dst, 16, 0, 0);
      }
    } else {
      uint8_t* dst = xl + (size_t)(run % XSN) * (MT * 1024);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) __builtin_amdgcn_global_load_lds((gptr_t)(xrow[mt] + rb), (lptr_t)(dst + mt * 1024), 16, 0, 0);
    }
  };
  auto issue_planes = [&](int step) {
    if (L.flags & 16) return;
    const int st = min(step, n_steps - 1);
    const int8_t* src = vsrc + (size_t)st * STEP_BYTES;
    uint8_t* vdst = vbuf + (size_t)(step % NBUF) * STEP_BYTES;
#pragma unroll
    for (int it = 0; it < (SPLIT ? PLDS : PLD); ++it) {
      const int kb = SPLIT ? min((wave >> 1) + it * (WAVES / 2), NKB - 1) : min(wave + it * WAVES, NKB - 1);
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

  const int abl = L.flags;  // development ablations (GNX_LR_FLAGS, timing only): 1 raw logits, 2 no MFMA, 4 no flush
  auto mfma_entry = [&](const uint8_t* pb, const v4i (&xc)[MT], int k) {
    if (abl & 2) return;
    v4i xa[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xa[mt] = unpack16(xc[mt][k]);
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

  // sigmoid, normaliser and division for the ZROWS rows x A classes parked in zb.  Per element the arithmetic and the class order
  // of the row sum are those of k_base_logistic_i8 (bit-identical B); what differs is who does what:
  //   phase 1  LPR = 64 / ZROWS lanes per row: lane (row, sub) turns classes sub, sub + LPR, .. into p = 1 / (1 + exp(-(z + icpt)))
  //   phase 2  every lane sums its row's A values in class order (LPR times redundantly: A additions against A/LPR exps), then
  //            divides its own classes
  //   phase 3  the wave walks the ZROWS x A block linearly, 64 consecutive elements per store (a row's A values are contiguous in
  //            B): (row, class) of element lane + 64 it advance by (64 / A, 64 % A) — no integer division in the loop
  constexpr int LPR = 64 / ZROWS;
  const int frow = lane % ZROWS, fsub = lane / ZROWS;
  const int e_r0 = lane / A, e_a0 = lane - e_r0 * A, e_dr = 64 / A, e_da = 64 - e_dr * A;
  auto emit = [&](int w, int64_t nrow0) {
    double* zr = zb + frow * A;
    if (!(abl & 1)) {
      const double* ic = tab_ic + (w - wt0) * A;
      for (int a = fsub; a < A; a += LPR) zr[a] = 1.0 / (1.0 + exp(-(zr[a] + ic[a])));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double sum = 0.0;
      for (int c = 0; c < A; ++c) sum += zr[c];
      for (int a = fsub; a < A; a += LPR) zr[a] = zr[a] / sum;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    int rl = e_r0, a = e_a0;
    const size_t ow = (size_t)w * A;
    for (int e = lane; e < ZROWS * A; e += 64) {
      const int64_t n = nrow0 + rl;
      if (n < L.N) {
        const size_t o = (size_t)n * W * A + ow + a;
        const double v = zb[e];
        if (L.b64) L.b64[o] = v;
        if (L.b32) L.b32[o] = (float)v;
      }
      a += e_da; rl += e_dr;
      if (a >= A) { a -= A; ++rl; }
    }
  };

  // ---- piece end: the windows that finished with run rl (block-uniform) ----
  auto flush = [&](int rl) {
    const int nfl = tab_nfl[rl];
    if (nfl <= 0 || (abl & 4)) return;
    const int w0 = tab_fl0[rl];
    for (int w = w0; w < w0 + nfl; ++w) {
      const int cbase = (w % R) * A;
      const double scale = tab_sc[w - wt0];
      const bool out = w >= wa && w < wb;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int col = nt * 16 + i16 - cbase;
          const bool mine = (col >= 0) && (col < A);
          if (mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r)  // int32 16x16 C/D layout: column = lane&15, row = 4*(lane>>4) + reg
              zb[((ZT ? 0 : mt * 16) + 4 * kq + r) * A + col] = combine(acc[mt][nt], r, scale);
          }
#pragma unroll
          for (int l = 0; l < LIMBS; ++l)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][nt][l][r] = mine ? 0 : acc[mt][nt][l][r];
        }
        if (ZT) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // zb is wave-private: LDS ops of one wave complete in order
          if (out) emit(w, n0 + mt * 16);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
      if (!ZT) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (out) emit(w, n0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  };

  // ---- prologue, in the steady state's issue order: X(0) .. X(XSN-1) | planes(0) | planes(1) ----
  if (!SPLIT || xwave) {
#pragma unroll
    for (int p = 0; p < XSN; ++p) issue_x(p);
  }
  if (!SPLIT || !xwave) {
    issue_planes(0);
    issue_planes(1);
  }

  for (int r = 0; r < n_runs; ++r) {
    // ---- even step 2r: the planes of the step and X(r) have landed; every wave is done with step 2r-1 ----
    if (SPLIT) {
      if (xwave) wait_vm<WAIT_XS>();
      else wait_vm<PLDS>();
    } else wait_vm<WAIT_EVEN>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(L.flags & 32)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    v4i xc[MT];
    {
      const uint8_t* xs = xl + (size_t)(r % XSN) * (MT * 1024) + lane * 16;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xc[mt] = *reinterpret_cast<const v4i*>(xs + mt * 1024);
    }
    if (SPLIT) {
      if (!xwave) issue_planes(2 * r + D);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the run sits in registers: its stage is free
      issue_x(r + XSN);
      issue_planes(2 * r + D);
    }
    {
      const uint8_t* sb = vbuf + (size_t)((2 * r) % NBUF) * STEP_BYTES;
      mfma_entry(sb, xc, 0);
      mfma_entry(sb + ENTRY_BYTES, xc, 1);
    }
    // ---- odd step 2r+1 ----
    if (SPLIT) {
      if (!xwave) wait_vm<PLDS>();
    } else wait_vm<WAIT_ODD>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(L.flags & 32)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (SPLIT) {
      if (xwave) issue_x(r + XSN);  // every wave has read X(r) out of this stage (before the barrier above)
      else issue_planes(2 * r + 1 + D);
    } else issue_planes(2 * r + 1 + D);
    {
      const uint8_t* sb = vbuf + (size_t)((2 * r + 1) % NBUF) * STEP_BYTES;
      mfma_entry(sb, xc, 2);
      mfma_entry(sb + ENTRY_BYTES, xc, 3);
    }
    flush(r);
  }
  wait_vm<0>();  // nothing of this block may still be writing LDS when it retires
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_base_logistic_p2w: the same pass with DEDICATED loader waves.  A wave that issues vector-memory instructions stalls at the issue
// while the L1's miss queue is full, so in the kernel above the time the memory system needs for X (HBM-bound: ~0.3 ms of the 0.75
// at config 2) was added to the issuing waves' MFMA and flush time instead of hiding behind it (ablations: loads only 0.38 ms, + MFMA
// 0.15, + flush 0.2 = the full kernel).  Here CW compute waves never touch global memory except for the flush's stores; wave CW
// issues every plane load, wave CW + 1 every X load (the rows of all compute waves), and the block barrier of each step publishes
// what has landed:
//   plane loader, step s:   vmcnt((D - 1) NKB), barrier, planes(s + D) into the slot step s - 1 left
//   X loader, run r:        vmcnt((XSN - 1) CW MT), barrier(2r), barrier(2r + 1), X(r + XSN) into the stage every wave read at step 2r
//   compute wave, run r:    barrier(2r), X(r) LDS -> registers, entries 0, 1, barrier(2r + 1), entries 2, 3, flush
template <int MT, int NT, int CW, int XSN, int NBUF>
__global__ __launch_bounds__((CW + 2) * 64) void k_base_logistic_p2w(BaseLRLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int EPS = 2;
  constexpr int ENTRY_BYTES = NT * LIMBS * 1024;
  constexpr int STEP_BYTES = EPS * ENTRY_BYTES;
  constexpr int THREADS = (CW + 2) * 64;
  constexpr int NKB = STEP_BYTES / 1024;
  constexpr int D = NBUF - 1;
  constexpr int ZROWS = MT * 16;
  constexpr int XTILES = CW * MT;
  static_assert((D - 1) * NKB < 64 && (XSN - 1) * XTILES < 64 && XSN >= 1 && D >= 1, "vmcnt is a 6-bit counter");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;
  const int A = L.A, W = L.W, R = L.d.R;
  uint8_t* vbuf = lds;                                               // [NBUF][STEP_BYTES]
  uint8_t* xl0 = vbuf + (size_t)NBUF * STEP_BYTES;                   // [XSN][CW][MT][64 lanes][16 B]
  double* zb0 = reinterpret_cast<double*>(xl0 + (size_t)XSN * XTILES * 1024);
  double* tab_ic = zb0 + (size_t)CW * ZROWS * A;   // [max_wins][A] intercepts
  double* tab_sc = tab_ic + (size_t)L.max_wins * A;  // [max_wins] 2^-f_w
  int* tab_rb = reinterpret_cast<int*>(tab_sc + L.max_wins);
  int* tab_nfl = tab_rb + L.max_chunks;
  int* tab_fl0 = tab_nfl + L.max_chunks;

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
  const int64_t n0b = (int64_t)htile * (CW * MT * 16);  // first haplotype of the block

  for (int e = tid; e < n_runs; e += THREADS) {
    tab_rb[e] = L.d.run_byte[r_begin + e];
    tab_nfl[e] = L.d.run_nflush[r_begin + e];
    tab_fl0[e] = L.d.run_flush0[r_begin + e];
  }
  const int wt0 = max(0, wa - R - 1);
  for (int e = tid; e < L.max_wins * A; e += THREADS) {
    const int ew = e / A, a = e - ew * A;
    tab_ic[e] = L.d.icpt[min(wt0 + ew, W - 1) * A + a];
  }
  for (int e = tid; e < L.max_wins; e += THREADS) tab_sc[e] = L.d.wscale[min(wt0 + e, W - 1)];
  __syncthreads();
  const int abl = L.flags;  // development ablations (GNX_LR_FLAGS, timing only): 1 raw logits, 2 no MFMA, 4 no flush, 8 no X, 16 no planes

  if (wave == CW) {
    // ---- plane loader ----
    const int8_t* vsrc = L.d.V2 + (size_t)r_begin * (2 * STEP_BYTES) + (size_t)lane * 16;
    auto issue_planes = [&](int step) {
      if (abl & 16) return;
      const int8_t* src = vsrc + (size_t)min(step, n_steps - 1) * STEP_BYTES;
      uint8_t* vdst = vbuf + (size_t)(step % NBUF) * STEP_BYTES;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kb * 1024), (lptr_t)(vdst + (size_t)kb * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < D; ++p) issue_planes(p);
    for (int s = 0; s < n_steps; ++s) {
      wait_vm<(D - 1) * NKB>();  // the planes of step s have landed (only those of s+1 .. s+D-1 are younger)
      __builtin_amdgcn_s_barrier();
      issue_planes(s + D);        // every compute wave is done with step s-1, whose slot this is
    }
    wait_vm<0>();
    return;
  }
  if (wave == CW + 1) {
    // ---- X loader: tile t = (compute wave t / MT, its tile t % MT); lane (row i16, 16 packed bytes kq) as in the compute waves ----
    const uint8_t* xrow[XTILES];
#pragma unroll
    for (int t = 0; t < XTILES; ++t) {
      const int64_t n = n0b + t * 16 + i16;  // rows >= N-1 read the zero-padded copy of the last row (rows past N are never written)
      xrow[t] = (n >= L.N - 1 ? reinterpret_cast<const uint8_t*>(L.last_row) : reinterpret_cast<const uint8_t*>(L.X) + n * L.ldx) + 16 * kq;
    }
    auto issue_x = [&](int run) {
      if (abl & 8) return;
      const int rb = tab_rb[min(run, n_runs - 1)];
      uint8_t* dst = xl0 + (size_t)(run % XSN) * (XTILES * 1024);
#pragma unroll
      for (int t = 0; t < XTILES; ++t) __builtin_amdgcn_global_load_lds((gptr_t)(xrow[t] + rb), (lptr_t)(dst + t * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < XSN; ++p) issue_x(p);
    for (int r = 0; r < n_runs; ++r) {
      wait_vm<(XSN - 1) * XTILES>();  // X(r) has landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // step 2r
      __builtin_amdgcn_s_barrier();   // step 2r+1: every compute wave has X(r) in registers
      issue_x(r + XSN);
    }
    wait_vm<0>();
    return;
  }

  // ---- compute waves ----
  const int64_t n0 = n0b + (int64_t)wave * (MT * 16);
  double* zb = zb0 + (size_t)wave * ZROWS * A;
  v4i acc[MT][NT][LIMBS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) acc[mt][nt][l] = v4i{0, 0, 0, 0};

  auto mfma_entry = [&](const uint8_t* pb, const v4i (&xc)[MT], int k) {
    if (abl & 2) return;
    v4i xa[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xa[mt] = unpack16(xc[mt][k]);
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

  // the epilogue of k_base_logistic_p2 (same three phases, same arithmetic: bit-identical B)
  constexpr int LPR = 64 / ZROWS;
  const int frow = lane % ZROWS, fsub = lane / ZROWS;
  const int e_r0 = lane / A, e_a0 = lane - e_r0 * A, e_dr = 64 / A, e_da = 64 - e_dr * A;
  auto emit = [&](int w) {
    double* zr = zb + frow * A;
    if (!(abl & 1)) {
      const double* ic = tab_ic + (w - wt0) * A;
      for (int a = fsub; a < A; a += LPR) zr[a] = 1.0 / (1.0 + exp(-(zr[a] + ic[a])));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      double sum = 0.0;
      for (int c = 0; c < A; ++c) sum += zr[c];
      for (int a = fsub; a < A; a += LPR) zr[a] = zr[a] / sum;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    int rl = e_r0, a = e_a0;
    const size_t ow = (size_t)w * A;
    for (int e = lane; e < ZROWS * A; e += 64) {
      const int64_t n = n0 + rl;
      if (n < L.N) {
        const size_t o = (size_t)n * W * A + ow + a;
        const double v = zb[e];
        if (L.b64) L.b64[o] = v;
        if (L.b32) L.b32[o] = (float)v;
      }
      a += e_da; rl += e_dr;
      if (a >= A) { a -= A; ++rl; }
    }
  };
  auto flush = [&](int rl) {
    const int nfl = tab_nfl[rl];
    if (nfl <= 0 || (abl & 4)) return;
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
      if (w >= wa && w < wb) emit(w);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };

  for (int r = 0; r < n_runs; ++r) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // step 2r: its planes and X(r) are in LDS
    asm volatile("" ::: "memory");
    v4i xc[MT];
    {
      const uint8_t* xs = xl0 + (size_t)(r % XSN) * (XTILES * 1024) + (size_t)wave * (MT * 1024) + lane * 16;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xc[mt] = *reinterpret_cast<const v4i*>(xs + mt * 1024);
    }
    {
      const uint8_t* sb = vbuf + (size_t)((2 * r) % NBUF) * STEP_BYTES;
      mfma_entry(sb, xc, 0);
      mfma_entry(sb + ENTRY_BYTES, xc, 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // step 2r+1
    asm volatile("" ::: "memory");
    {
      const uint8_t* sb = vbuf + (size_t)((2 * r + 1) % NBUF) * STEP_BYTES;
      mfma_entry(sb, xc, 2);
      mfma_entry(sb + ENTRY_BYTES, xc, 3);
    }
    flush(r);
  }
}

template <int MT, int NT, int CW, int XSN, int NBUF>
hipError_t launch_w(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  BaseLRLaunch P = L;
  P.flags = tune.lr_flags;
  const int haps_per_block = CW * MT * 16;
  const int64_t gx = (L.N + haps_per_block - 1) / haps_per_block;
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
    lds = (size_t)NBUF * (2 * NT * LIMBS * 1024) + (size_t)XSN * CW * MT * 1024 + (size_t)CW * MT * 16 * L.A * sizeof(double) +
          (size_t)3 * P.max_chunks * sizeof(int) + (size_t)P.max_wins * (L.A + 1) * sizeof(double);
    if (lds <= (size_t)160 * 1024 || wch == 4) break;
  }
  if (lds > (size_t)160 * 1024) return hipErrorNotSupported;
  const int n_ranges8 = ((n_ranges + 7) / 8) * 8;
  P.wch = wch;
  P.n_htiles = (int)gx;
  if (tune.debug) std::fprintf(stderr, "k_base_logistic_p2w<%d,%d,%d,%d,%d>: lds=%zu grid=%lld wch=%d\n", MT, NT, CW, XSN, NBUF, lds, (long long)(gx * n_ranges8), wch);
  GNX_LDS_OPTIN(lds, k_base_logistic_p2w<MT, NT, CW, XSN, NBUF>);
  hipLaunchKernelGGL((k_base_logistic_p2w<MT, NT, CW, XSN, NBUF>), dim3((unsigned)(gx * n_ranges8)), dim3((CW + 2) * 64), lds, s, P);
  return hipGetLastError();
}

template <int MT, int NT, int WAVES, int XSN, int ZT>
size_t lds_need(int A, int max_runs, int max_wins) {
  return (size_t)3 * (2 * NT * LIMBS * 1024) + (size_t)WAVES * XSN * MT * 1024 + (size_t)WAVES * (ZT ? 16 : MT * 16) * A * sizeof(double) +
         (size_t)3 * max_runs * sizeof(int) + (size_t)max_wins * (A + 1) * sizeof(double);
}

template <int MT, int NT, int WAVES, int XSN, int ZT, int SPLIT = 1>
hipError_t launch(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  BaseLRLaunch P = L;
  P.flags = tune.lr_flags;
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
    lds = lds_need<MT, NT, WAVES, XSN, ZT>(L.A, P.max_chunks, P.max_wins);
    if (lds <= (size_t)160 * 1024 || wch == 4) break;
  }
  if (lds > (size_t)160 * 1024) return hipErrorNotSupported;
  lds = std::min(lds + (size_t)std::max(tune.lr_lds_pad, 0), (size_t)160 * 1024);
  const int n_ranges8 = ((n_ranges + 7) / 8) * 8;
  P.wch = wch;
  P.n_htiles = (int)gx;
  if (tune.debug) std::fprintf(stderr, "k_base_logistic_p2<%d,%d,%d,%d,%d,%d>: lds=%zu grid=%lld wch=%d\n", MT, NT, WAVES, XSN, ZT, SPLIT, lds, (long long)(gx * n_ranges8), wch);
  GNX_LDS_OPTIN(lds, k_base_logistic_p2<MT, NT, WAVES, XSN, ZT, SPLIT>);
  hipLaunchKernelGGL((k_base_logistic_p2<MT, NT, WAVES, XSN, ZT, SPLIT>), dim3((unsigned)(gx * n_ranges8)), dim3(WAVES * 64), lds, s, P);
  return hipGetLastError();
}

}  // namespace

// returns hipErrorNotSupported when no instantiation fits (the caller widens X to int8 and runs the int8 kernels)
hipError_t gnx_launch_base_logistic_p2(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  if (!L.d.V2 || !L.h_win_chunk0 || !L.h_win_chunk1) return hipErrorNotSupported;
  const bool small = L.N <= 64 * 8;
  const int tm = tune.p2_mt, tw = tune.p2_cw, tx = tune.p2_xsn, tb = tune.p2_nbuf ? tune.p2_nbuf : 3;
#define GNX_P2W(MT_, NT_, CW_, XS_, NB_) \
  if (tm == MT_ && tw == CW_ && tx == XS_ && tb == NB_) return launch_w<MT_, NT_, CW_, XS_, NB_>(L, n_cu, tune, s);
#define GNX_P2O(MT_, NT_, WV_, XS_, ZT_, SP_) \
  if (tm == MT_ && tw == WV_ && tx == XS_ && tb == 10 * ZT_ + SP_) return launch<MT_, NT_, WV_, XS_, ZT_, SP_>(L, n_cu, tune, s);
  switch (L.d.NT) {
    case 1:
      if (tune.p2_old) {  // k_base_logistic_p2: every wave loads and multiplies (nbuf field = 10 * ZT + SPLIT)
        GNX_P2O(4, 1, 8, 2, 0, 1) GNX_P2O(4, 1, 8, 2, 0, 0) GNX_P2O(2, 1, 8, 3, 0, 1) GNX_P2O(2, 1, 8, 2, 0, 0)
        return hipErrorInvalidValue;
      }
      if (tm) {
        GNX_P2W(2, 1, 8, 2, 3) GNX_P2W(2, 1, 8, 3, 3) GNX_P2W(2, 1, 8, 4, 3) GNX_P2W(2, 1, 8, 4, 4) GNX_P2W(2, 1, 8, 3, 4)
        GNX_P2W(2, 1, 10, 3, 3) GNX_P2W(2, 1, 12, 3, 3) GNX_P2W(2, 1, 14, 3, 3) GNX_P2W(2, 1, 14, 2, 3) GNX_P2W(2, 1, 12, 2, 4)
        GNX_P2W(4, 1, 6, 2, 3) GNX_P2W(4, 1, 6, 3, 3) GNX_P2W(1, 1, 14, 4, 3)
        return hipErrorInvalidValue;
      }
      if (small) return launch_w<1, 1, 4, 2, 3>(L, n_cu, tune, s);
      return launch_w<2, 1, 8, 4, 4>(L, n_cu, tune, s);
    case 2:
      if (tune.p2_old) {
        GNX_P2O(2, 2, 8, 2, 0, 1) GNX_P2O(2, 2, 8, 2, 0, 0) GNX_P2O(1, 2, 16, 2, 0, 1)
        return hipErrorInvalidValue;
      }
      if (tm) {
        GNX_P2W(2, 2, 6, 2, 3) GNX_P2W(2, 2, 6, 3, 3) GNX_P2W(1, 2, 8, 4, 3) GNX_P2W(1, 2, 14, 4, 3) GNX_P2W(1, 2, 14, 3, 3) GNX_P2W(1, 2, 12, 4, 3)
        return hipErrorInvalidValue;
      }
      if (small) return launch_w<1, 2, 4, 2, 3>(L, n_cu, tune, s);
      return launch_w<2, 2, 6, 3, 3>(L, n_cu, tune, s);
    default: return hipErrorNotSupported;
  }
#undef GNX_P2W
#undef GNX_P2O
}

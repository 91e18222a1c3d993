// k_base_logistic_p2.hip — the exact int8 logistic pass on 2-BIT haplotype rows, wave-specialised (gfx950).
//
// Same contract, arithmetic (7 balanced base-256 limbs on v_mfma_i32_16x16x64_i8, one float64 rounding), pieces, window flush and
// XCD-aware grid as k_base_logistic_i8.hip / k_base_logistic_i8_dl.hip (reference src/Base/base.py:146-180,
// src/Base/models.py:12-21; X values per src/utils.py:153) — B is BIT-IDENTICAL to theirs.  What differs:
//
//  * X stays packed in HBM, four SNPs per byte (the gnx_pack_x layout: SNP j = bits 2(j%4).. of byte j/4).  The int8 kernels are
//    bound by the L1's miss queue (5.05 GB of L1 fill per launch at config 2, 3.45 GB of it X); this is a quarter of the X bytes.
//  * A lane fetches 16 packed bytes = 64 SNPs of ONE haplotype row, the four lanes of a row (lane>>4) cover 64 contiguous bytes = one
//    RUN of 256 SNPs, 16 rows per wave instruction, HBM -> LDS directly (global_load_lds_dwordx4, lane-linear: a compute lane reads
//    back the 16 bytes "its" loader lane fetched).  A 32-bit word (16 SNPs) becomes the 16 int8 bytes of the MFMA A operand in 7 VALU
//    operations:  reg d = (word >> 2d) & 0x03030303  (byte b of reg d = field 4b + d) — which SNP sits at which k position of the
//    MFMA is the weight planes' business: they are laid out to match at model load (gnx_build_lr, V2).
//  * Pieces start on a 32-bit boundary of the packed row: a piece [b0, b1) is walked from SNP b0 & ~15 in runs of 256 SNPs (4 MFMA
//    entries = 2 plane steps); the up to 15 SNPs before b0 and everything from b1 on meet zero weights (M = 1000, context 500:
//    500 + 12 SNPs = exactly two runs).  Loads that are not dword-aligned are split by the texture addresser (~4.5 L1 tag accesses
//    per lane instead of ~0.5: the first, byte-aligned version ran at the L1's tag rate).
//  * ROLES.  A wave that issues vector-memory instructions stalls at the issue while the L1's miss queue is full, and vmcnt retires
//    in order; with every wave loading and multiplying, the memory time (~0.38 ms of loads at config 2), the MFMA time (0.15) and the
//    epilogue's float64 VALU time (0.2) simply added up (0.75 ms; ablations in DESIGN.md).  Here a block is
//        CW compute waves     LDS -> registers, unpack, MFMA; at a window's end: accumulators -> exact logits z in LDS, nothing else
//        EW epilogue waves    z -> sigmoid, row normaliser, division, coalesced stores to B — float64 VALU work that now runs on the
//                             SIMD's vector pipe WHILE the compute waves keep the matrix pipe busy, spread over the steps until the
//                             next window ends (two z buffers)
//        1 plane loader       the digit planes of step s+D, L2 -> LDS ring of NBUF half-runs
//        1 X loader           the runs of ALL compute waves' rows, XSN stages ahead
//    and ONE block barrier per step publishes whatever has landed / been parked:
//        plane loader, step s:  vmcnt((D-1) NKB), barrier, planes(s+D) into the slot step s-1 left
//        X loader, run r:       vmcnt((XSN-1) CW MT), barrier(2r), barrier(2r+1), X(r+XSN) into the stage every wave read at step 2r
//        compute wave, run r:   barrier(2r), X(r) LDS -> registers, entries 0, 1, barrier(2r+1), entries 2, 3, park finished windows
//        epilogue wave:         barrier, its share of the parked window, barrier, ...
//  * COLUMN TILES.  Up to 16 class columns per SNP (R A <= 16, e.g. A = 7 at the default context): one tile, column = slot A + class,
//    as in the int8 kernels.  More: one tile PER SLOT (column = class) and one PASS per tile — blocks of pass t multiply only tile t
//    and finish only the windows w with w % R == t, so every pass is the one-tile kernel (256 rows per block) and X is read R times,
//    at a quarter byte per SNP (the two-tile int8 kernels need 112 accumulator registers per 32 rows).  EXACTLY 24 columns (A = 12 at
//    the default context, BASELINE config 5) run k_base_logistic_p2f below instead: flat column tiles, both slots per wave, X once.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "gnx_internal.h"
#include "gnx_exp.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
// development switches of the 2-bit kernels (compile time; scripts/dev/p2f_defs.sh): prefetch distance of the plane-read pipelines
// (tiles), accumulator registers gathered at a time at a window's end, classes per sigmoid unit and store parts of the epilogue waves
#ifndef GNX_P2F_PD
#define GNX_P2F_PD 2
#endif
#ifndef GNX_P2F_FLUSH_GROUP
#define GNX_P2F_FLUSH_GROUP 2
#endif
#ifndef GNX_P2F_PBR
#define GNX_P2F_PBR 4
#endif
#ifndef GNX_P2F_NSP
#define GNX_P2F_NSP 2
#endif
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

__device__ __forceinline__ void lds_barrier(bool skip = false) {  // skip: development ablation (GNX_LR_FLAGS & 32, timing only)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (!skip) __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// MT 16-row tiles per compute wave, CW compute waves, EW epilogue waves (0: the compute waves finish their windows themselves),
// XSN stages of X runs, NBUF plane slots of one step (2 entries, 14 KB) each, EPR entries per run: 4 (256 SNPs = 64 packed bytes per
// row visit, four lanes per row) or 8 (512 SNPs = 128 bytes = whole cache lines per row visit, eight lanes per row).
template <int MT, int CW, int EW, int XSN, int NBUF, int EPR>
__global__ __launch_bounds__((CW + EW + 2) * 64) void k_base_logistic_p2(BaseLRLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int ENTRY_BYTES = LIMBS * 1024;       // digit planes of one entry (64 k positions) of ONE column tile
  constexpr int STEP_BYTES = 2 * ENTRY_BYTES;
  constexpr int THREADS = (CW + EW + 2) * 64;
  constexpr int NKB = STEP_BYTES / 1024;
  constexpr int D = NBUF - 1;
  constexpr int ZROWS = MT * 16;                  // rows a compute wave parks per window
  constexpr int XTILES = CW * MT;
  constexpr int SPR = EPR / 2;                    // plane steps (block barriers) per run
  constexpr int XLD = EPR / 4;                    // X loads (1 KB each) per 16-row tile and run
  static_assert(EPR == 4 || EPR == 8, "entries per run");
  constexpr int RWS = EW ? CW * ZROWS / EW : ZROWS;  // rows one finishing wave handles per window
  constexpr int LPR = 64 / RWS;                      // lanes per row in the sigmoid phase
  static_assert((D - 1) * NKB < 64 && (XSN - 1) * XTILES * XLD < 64 && XSN >= 1 && D >= 1, "vmcnt is a 6-bit counter");
  static_assert(RWS <= 64 && 64 % RWS == 0 && (!EW || CW % EW == 0), "an epilogue wave takes whole compute waves, at most 64 rows");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;
  const int A = L.A, W = L.W, R = L.d.R, NT2 = L.d.NT2;
  const bool slot_tiles = NT2 > 1;                 // one column tile per slot, one pass per tile
  uint8_t* vbuf = lds;                                               // [NBUF][STEP_BYTES]
  uint8_t* xl0 = vbuf + (size_t)NBUF * STEP_BYTES;                   // [XSN][CW][MT][XLD][64 lanes][16 B]
  double* zq = reinterpret_cast<double*>(xl0 + (size_t)XSN * XTILES * XLD * 1024);  // [2][CW][ZROWS][A] parked logits
  double* tab_ic = zq + (size_t)2 * CW * ZROWS * A;  // [max_wins][A] intercepts
  double* tab_sc = tab_ic + (size_t)L.max_wins * A;   // [max_wins] 2^-f_w
  int* tab_rb = reinterpret_cast<int*>(tab_sc + L.max_wins);
  int* tab_nfl = tab_rb + L.max_chunks;   // windows of THIS pass that end with the run
  int* tab_fl0 = tab_nfl + L.max_chunks;  // the first of them
  int* tab_gap = tab_fl0 + L.max_chunks;  // runs until the next run that ends a window of this pass (0: none follows)

  // XCD-aware decomposition: all blocks of one window range (and pass) on ONE XCD (its L2 serves the range's digit planes)
  int wrange, htile, pass;
  {
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int g = j / L.n_htiles;
    htile = j - g * L.n_htiles;
    pass = g / L.n_rg8;
    wrange = xcd + 8 * (g - pass * L.n_rg8);
  }
  const int wa = wrange * L.wch;
  if (wa >= W) return;  // whole block exits before any barrier
  const int wb = min(W, wa + L.wch);
  const int r_begin = L.d.win_run0[wa];
  const int r_end = L.d.win_run1[wb - 1];
  const int n_runs = r_end - r_begin;
  const int n_steps = SPR * n_runs;
  const int64_t n0b = (int64_t)htile * (CW * MT * 16);  // first haplotype of the block

  for (int e = tid; e < n_runs; e += THREADS) {
    tab_rb[e] = L.d.run_byte[r_begin + e];
    int nf = L.d.run_nflush[r_begin + e], f0 = L.d.run_flush0[r_begin + e];
    if (slot_tiles && nf > 0) {  // windows f0 .. f0+nf-1 end here; this pass owns those with w % R == pass (at most one: nf <= R)
      const int w = f0 + ((pass - f0) % R + R) % R;
      nf = w < f0 + nf ? 1 : 0;
      f0 = w;
    }
    tab_nfl[e] = nf;
    tab_fl0[e] = f0;
  }
  const int wt0 = max(0, wa - R - 1);
  for (int e = tid; e < L.max_wins * A; e += THREADS) {
    const int ew = e / A, a = e - ew * A;
    tab_ic[e] = L.d.icpt[min(wt0 + ew, W - 1) * A + a];
  }
  for (int e = tid; e < L.max_wins; e += THREADS) tab_sc[e] = L.d.wscale[min(wt0 + e, W - 1)];
  __syncthreads();
  if (tid == 0) {
    int next = -1;
    for (int r = n_runs - 1; r >= 0; --r) {
      tab_gap[r] = next < 0 ? 0 : next - r;
      if (tab_nfl[r] > 0) next = r;
    }
  }
  __syncthreads();
  const int abl = L.flags;  // development ablations (GNX_LR_FLAGS, timing only): 1 raw logits, 2 no MFMA, 4 no flush, 8 no X, 16 no planes, 32 no barriers, 64 no combine
  const bool nobar = (abl & 32) != 0;

  if (wave == CW + EW) {
    // ================================================== plane loader ==================================================
    // entry e of the block's run r_begin + e / 4: tile `pass` of [run][4 entries][NT2 tiles][7 limbs][64 lanes][16 B]
    const int8_t* vsrc = L.d.V2 + ((size_t)r_begin * EPR * NT2 + pass) * ENTRY_BYTES + (size_t)lane * 16;
    auto issue_planes = [&](int step) {
      if (abl & 16) return;
      const int8_t* src = vsrc + (size_t)min(step, n_steps - 1) * 2 * NT2 * ENTRY_BYTES;
      uint8_t* vdst = vbuf + (size_t)(step % NBUF) * STEP_BYTES;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const size_t so = (size_t)(kb / LIMBS) * NT2 * ENTRY_BYTES + (size_t)(kb % LIMBS) * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + so), (lptr_t)(vdst + (size_t)kb * 1024), 16, 0, 0);
      }
    };
#pragma unroll
    for (int p = 0; p < D; ++p) issue_planes(p);
    for (int s = 0; s < n_steps; ++s) {
      wait_vm<(D - 1) * NKB>();  // the planes of step s have landed (only those of s+1 .. s+D-1 are younger)
      if (!nobar) __builtin_amdgcn_s_barrier();
      issue_planes(s + D);        // every compute wave is done with step s-1, whose slot this is
    }
    wait_vm<0>();                 // nothing of this wave may still be writing LDS when the block retires
    if (EW && !nobar) __builtin_amdgcn_s_barrier();  // the trailing barrier of the compute / epilogue waves
    return;
  }
  if (wave == CW + EW + 1) {
    // ================================================== X loader ==================================================
    // tile t = (compute wave t / MT, its tile t % MT).  EPR = 4: load 0 of a tile = 16 rows x 64 bytes, lane (row i16, 16 packed bytes
    // kq) exactly as the compute lane that reads them back.  EPR = 8: load q of a tile = rows 8q .. 8q+7 x 128 bytes — eight lanes
    // cover one row's whole run — lane i fetching row 8q + (i >> 3), logical 16-byte piece (i & 7) ^ (i >> 3) (source-side swizzle:
    // LDS-direct loads land lane-linear, and the compute lanes' ds_read_b128 of piece p of 16 rows at a 128-byte pitch would
    // otherwise hit one bank group 8 ways)
    const uint8_t* xrow[XTILES * XLD];
#pragma unroll
    for (int t = 0; t < XTILES * XLD; ++t) {
      const int tile = t / XLD, q = t % XLD;
      const int row = EPR == 8 ? 8 * q + (lane >> 3) : i16;
      const int piece = EPR == 8 ? ((lane & 7) ^ (lane >> 3)) : kq;
      const int64_t n = n0b + tile * 16 + row;  // rows >= N-1 read the zero-padded copy of the last row (rows past N are never written)
      xrow[t] = (n >= L.N - 1 ? reinterpret_cast<const uint8_t*>(L.last_row) : reinterpret_cast<const uint8_t*>(L.X) + n * L.ldx) + 16 * piece;
    }
    auto issue_x = [&](int run) {
      if (abl & 8) return;
      const int rb = tab_rb[min(run, n_runs - 1)];
      uint8_t* dst = xl0 + (size_t)(run % XSN) * (XTILES * XLD * 1024);
#pragma unroll
      for (int t = 0; t < XTILES * XLD; ++t) __builtin_amdgcn_global_load_lds((gptr_t)(xrow[t] + rb), (lptr_t)(dst + t * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < XSN; ++p) issue_x(p);
    for (int r = 0; r < n_runs; ++r) {
      wait_vm<(XSN - 1) * XTILES * XLD>();  // X(r) has landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!nobar) __builtin_amdgcn_s_barrier();   // first step of the run
      if (!nobar) __builtin_amdgcn_s_barrier();   // second step: every compute wave has X(r) in registers
      issue_x(r + XSN);
#pragma unroll
      for (int h = 2; h < SPR; ++h)
        if (!nobar) __builtin_amdgcn_s_barrier();
    }
    wait_vm<0>();
    if (EW && !nobar) __builtin_amdgcn_s_barrier();
    return;
  }

  // ---- what the compute waves and the epilogue waves share: finishing RWS rows x A classes parked at zr0 ----
  // Per element the arithmetic and the class order of the row sum are those of k_base_logistic_i8 (bit-identical B):
  //   phase 1  LPR = 64 / RWS lanes per row: lane (row, sub) turns classes sub, sub + LPR, .. into p = 1 / (1 + exp(-(z + icpt)))
  //   phase 2  every lane sums its row's A values in class order (LPR times redundantly), then divides its own classes
  //   phase 3  the wave walks the RWS x A block linearly, 64 consecutive elements per store (a row's A values are contiguous in
  //            B): (row, class) of element lane + 64 it advance by (64 / A, 64 % A) — no integer division in the loop
  // (`rows` is a compile-time constant at both call sites: RWS in the epilogue waves, ZROWS where a compute wave finishes its own)
  const int e_r0 = lane / A, e_a0 = lane - e_r0 * A, e_dr = 64 / A, e_da = 64 - e_dr * A;
  auto phase1 = [&](double* zr0, int w, int it0, int it1, int rows) {
    if (abl & 1) return;
    const int lpr = 64 / rows, frow = lane % rows, fsub = lane / rows;
    double* zr = zr0 + frow * A;
    const double* ic = tab_ic + (w - wt0) * A;
    for (int it = it0; it < it1; ++it) {
      const int a = fsub + it * lpr;
      if (a < A) zr[a] = gnx_sigmoid(zr[a] + ic[a]);
    }
  };
  auto finish = [&](double* zr0, int w, int64_t nrow0, int rows) {
    if (!(abl & 1)) {
      const int lpr = 64 / rows, frow = lane % rows, fsub = lane / rows;
      double* zr = zr0 + frow * A;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own phase-1 writes (LDS ops of one wave complete in order)
      double sum = 0.0;
      for (int c = 0; c < A; ++c) sum += zr[c];
      const double rs = gnx_rcp_nr(sum);
      for (int a = fsub; a < A; a += lpr) zr[a] = zr[a] * rs;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    int rl = e_r0, a = e_a0;
    const size_t ow = (size_t)w * A;
    for (int e = lane; e < rows * A; e += 64) {
      const int64_t n = nrow0 + rl;
      if (n < L.N && !(abl & 32)) {
        const size_t o = (size_t)n * W * A + ow + a;
        const double v = zr0[e];
        if (L.b64) L.b64[o] = v;
        if (L.b32) L.b32[o] = (float)v;
      }
      a += e_da; rl += e_dr;
      if (a >= A) { a -= A; ++rl; }
    }
  };

  if (EW && wave >= CW) {
    // ================================================== epilogue waves ==================================================
    // Replays the block-uniform window schedule of the compute waves: a run that ends exactly one window of this pass inside
    // [wa, wb) parks it in buffer (parked count & 1); the job then has 2 * gap steps — until the next window of this pass ends —
    // to get through phase 1 (spread evenly over all steps but the last) and the finish (last step).
    const int ew = wave - CW;
    const int64_t nrow0 = n0b + (int64_t)ew * RWS;
    const int n_it = (A + LPR - 1) / LPR;  // phase-1 iterations of a lane
    int parked = 0;
    bool job = false;
    int job_w = 0, job_step = 0, job_nst = 0;
    double* job_z = nullptr;
    auto job_work = [&]() {
      if (!job) return;
      if (job_step < job_nst - 1) {
        phase1(job_z, job_w, n_it * job_step / (job_nst - 1), n_it * (job_step + 1) / (job_nst - 1), RWS);
      } else {
        finish(job_z, job_w, nrow0, RWS);
        job = false;
      }
      ++job_step;
    };
    for (int r = 0; r < n_runs; ++r) {
#pragma unroll
      for (int h = 0; h < SPR; ++h) {
        lds_barrier(nobar);  // step SPR r + h
        job_work();
      }
      // what the compute waves park at the end of this run (visible after the next barrier)
      const int nfl = (abl & 4) ? 0 : tab_nfl[r];
      if (nfl == 1) {
        const int w = tab_fl0[r];
        if (w >= wa && w < wb) {
          job = true;
          job_w = w;
          job_step = 0;
          job_nst = SPR * (tab_gap[r] > 0 ? tab_gap[r] : n_runs - 1 - r);  // 0 (last run): after the trailing barrier
          job_z = zq + (size_t)(parked & 1) * (CW * ZROWS * A) + (size_t)ew * RWS * A;
          ++parked;
        }
      }
    }
    lds_barrier(nobar);  // trailing barrier: the last run's windows are parked
    if (job) {      // nothing follows: all of it at once
      phase1(job_z, job_w, 0, n_it, RWS);
      finish(job_z, job_w, nrow0, RWS);
    }
    return;
  }

  // ================================================== compute waves ==================================================
  const int64_t n0 = n0b + (int64_t)wave * (MT * 16);
  v4i acc[MT][LIMBS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int l = 0; l < LIMBS; ++l) acc[mt][l] = v4i{0, 0, 0, 0};

  auto mfma_entry = [&](const uint8_t* pb, const v4i (&xc)[MT][XLD], int k) {  // k: compile-time after unrolling
    if (abl & 2) return;
    v4i xa[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xa[mt] = unpack16(xc[mt][k >> 2][k & 3]);
    const v4i* vb = reinterpret_cast<const v4i*>(pb) + lane;
#pragma unroll
    for (int l = 0; l < LIMBS; ++l) {
      const v4i b = vb[l * 64];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[mt], b, acc[mt][l], 0, 0, 0);
    }
  };
  // (the hand-written read pipeline of k_base_logistic_p2f's mfma_step, tried here too: 0.580 ms against 0.575 at config 2 — with seven
  // tiles per entry and a third of the accumulators this kernel is not waiting for its LDS reads)

  // ---- piece end: the windows of this pass that finished with run rl (block-uniform) ----
  int parked = 0;
  auto flush = [&](int rl) {
    const int nfl = tab_nfl[rl];
    if (nfl <= 0 || (abl & 4)) return;
    const int w0 = tab_fl0[rl];
    for (int w = w0; w < w0 + nfl; ++w) {
      const int cbase = slot_tiles ? 0 : (w % R) * A;
      const double scale = tab_sc[w - wt0];
      const bool out = w >= wa && w < wb;
      double* zw = zq + (size_t)(parked & 1) * (CW * ZROWS * A) + (size_t)wave * ZROWS * A;
      const int col = i16 - cbase;
      const bool mine = (col >= 0) && (col < A);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (mine && out && !(abl & 64)) {
#pragma unroll
          for (int r = 0; r < 4; ++r)  // int32 16x16 C/D layout: column = lane&15, row = 4*(lane>>4) + reg
            zw[(mt * 16 + 4 * kq + r) * A + col] = combine(acc[mt], r, scale);
        }
#pragma unroll
        for (int l = 0; l < LIMBS; ++l)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mt][l][r] = mine ? 0 : acc[mt][l][r];
      }
      if (!out) continue;
      if (EW && nfl == 1) {
        ++parked;  // an epilogue wave takes it from here (after the next barrier)
      } else {     // several windows end at once (chromosome ends, wide contexts), or no epilogue waves: finish it here
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        phase1(zw, w, 0, (A + 64 / ZROWS - 1) / (64 / ZROWS), ZROWS);
        finish(zw, w, n0, ZROWS);
      }
    }
  };

  for (int r = 0; r < n_runs; ++r) {
    v4i xc[MT][XLD];
#pragma unroll
    for (int h = 0; h < SPR; ++h) {
      lds_barrier(nobar);  // step SPR r + h: its planes (and, at h = 0, X(r)) are in LDS
      if (h == 0) {
        const uint8_t* xs = xl0 + (size_t)(r % XSN) * (XTILES * XLD * 1024) + (size_t)wave * (MT * XLD * 1024);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (EPR == 8) {  // row i16 of the tile sits in load i16 >> 3 at row slot i16 & 7; pieces 2 kq, 2 kq + 1, un-swizzled
            const int rr = i16 & 7;
            const uint8_t* rowp = xs + mt * 2048 + (i16 >> 3) * 1024 + rr * 128;
            xc[mt][0] = *reinterpret_cast<const v4i*>(rowp + (((2 * kq) ^ rr) << 4));
            xc[mt][XLD - 1] = *reinterpret_cast<const v4i*>(rowp + (((2 * kq + 1) ^ rr) << 4));
          } else {
            xc[mt][0] = *reinterpret_cast<const v4i*>(xs + mt * 1024 + lane * 16);
          }
        }
      }
      const uint8_t* sb = vbuf + (size_t)((SPR * r + h) % NBUF) * STEP_BYTES;
      mfma_entry(sb, xc, 2 * h);
      mfma_entry(sb + ENTRY_BYTES, xc, 2 * h + 1);
    }
    flush(r);
  }
  if (EW) lds_barrier(nobar);  // trailing barrier: the last run's parked windows become visible to the epilogue waves
}

template <int MT, int CW, int EW, int XSN, int NBUF, int EPR>
hipError_t launch(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  BaseLRLaunch P = L;
  P.flags = tune.lr_flags;
  const int haps_per_block = CW * MT * 16;
  const int64_t gx = (L.N + haps_per_block - 1) / haps_per_block;
  // window ranges: a multiple of 8 (one XCD each); every range re-walks the runs of its first windows' lead-in, so fewer, longer
  // ranges move fewer bytes and more, shorter ones balance the tail
  const int bpc = tune.lr_bpc > 0 ? tune.lr_bpc : 4;
  int64_t want = ((int64_t)bpc * n_cu + gx * L.d.NT2 - 1) / (gx * L.d.NT2);
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
    lds = (size_t)NBUF * (2 * LIMBS * 1024) + (size_t)XSN * CW * MT * (EPR / 4) * 1024 + (size_t)2 * CW * MT * 16 * L.A * sizeof(double) +
          (size_t)4 * P.max_chunks * sizeof(int) + (size_t)P.max_wins * (L.A + 1) * sizeof(double);
    if (lds <= (size_t)160 * 1024 || wch == 4) break;
  }
  if (lds > (size_t)160 * 1024) return hipErrorNotSupported;
  const int n_ranges8 = ((n_ranges + 7) / 8) * 8;
  P.wch = wch;
  P.n_htiles = (int)gx;
  P.n_rg8 = n_ranges8 / 8;
  if (tune.debug) std::fprintf(stderr, "k_base_logistic_p2<%d,%d,%d,%d,%d,%d>: lds=%zu grid=%lld wch=%d passes=%d\n", MT, CW, EW, XSN, NBUF, EPR, lds, (long long)(gx * n_ranges8 * L.d.NT2), wch, L.d.NT2);
  GNX_LDS_OPTIN(lds, k_base_logistic_p2<MT, CW, EW, XSN, NBUF, EPR>);
  hipLaunchKernelGGL((k_base_logistic_p2<MT, CW, EW, XSN, NBUF, EPR>), dim3((unsigned)(gx * n_ranges8 * L.d.NT2)), dim3((CW + EW + 2) * 64), lds, s, P);
  return hipGetLastError();
}


// =====================================================================================================================================
// FLAT column tiles: R * A == 24 class columns per SNP (A = 12 ancestries at the default context, BASELINE config 5).
//
// One column tile per slot and one pass per tile (above) multiplies 2 x 7 = 14 tiles per 64 SNPs, a quarter of their columns empty,
// and reads X twice; every kernel of this family sits on the same ~5-6 TB/s of L1 / LDS-DMA fill traffic, which at
// config 5a was 30.7 GB of digit planes + 17.9 GB of X per launch.  Here the 24 class columns x 7 limbs are laid side by side as
// 168 FLAT columns, q = 24 limb + column, in ceil(168 / 16) = 11 tiles: 21 % fewer plane bytes and MFMAs, and 11 x 4 = 44 accumulator
// registers per 16 rows hold BOTH slots, so a wave of 32 rows carries 88 — the block keeps 256 rows, reads X ONCE, and runs 14
// waves of 128 registers (8 compute, 4 epilogue, 2 loaders).  Fill per launch: 24.1 + 8.9 GB.
//
// Where limb l of column c lives (tile, lane column) = ((24 l + c) >> 4, (24 l + c) & 15) repeats every two limbs (48 columns = 3 tiles):
//     tile 3k      lanes 0-15: limb 2k   of columns 0-15
//     tile 3k + 1  lanes 0-7 : limb 2k   of columns 16-23;  lanes 8-15: limb 2k+1 of columns 0-7
//     tile 3k + 2  lanes 0-15: limb 2k+1 of columns 8-23                                    (k = 0, 1, 2)
//     tile 9       limb 6 of columns 0-15;   tile 10 lanes 0-7: limb 6 of columns 16-23
// so the HOME lane of column c (lane c for c < 16, lane c - 16 above) finds its even limbs in its own lane and its odd limbs in the
// lane 8 away (one row_ror:8 DPP move each; WHICH tile is chosen in the source lane, a lane predicate): three DPP moves and seven
// selects per accumulator register at a window's end.  Limbs then go in pairs, a_2k + 256 a_2k+1 in int32 (the launcher declines a
// model whose windows are wide enough to overflow: the int8 kernels take it), four conversions and three fmas to Z — one rounding,
// the one combine() makes: Z, and with it B, stay BIT-IDENTICAL to the int8 kernels'.  The eight registers' code is straight-line
// (lanes that are home to no column park their value in the spare float64 behind a row's classes: no exec mask).  A window's end zeroes only its slot's columns (the other slot's window is in
// mid-flight): the lanes to zero repeat every three tiles, three exec-masked regions of 64-bit moves.
// =====================================================================================================================================
constexpr int NFT = GNX_LR_FLAT_TILES;   // 11
constexpr int NCF = GNX_LR_FLAT_COLS;    // 24

__device__ __forceinline__ int ror8(int v) {  // the value of the lane 8 away within the 16-lane row
  return __builtin_amdgcn_update_dpp(0, v, 0x128 /* row_ror:8 */, 0xf, 0xf, true);  // (every lane has a source: bound_ctrl spares the move that would preset the result)
}

template <int MT, int CW, int EW, int XSN, int NBUF, bool DBG>
__global__ __launch_bounds__((CW + EW + 2) * 64) void k_base_logistic_p2f(BaseLRLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int ENTRY_BYTES = NFT * 1024;         // digit planes of one entry (64 k positions): 11 flat tiles
  constexpr int STEP_BYTES = 2 * ENTRY_BYTES;
  constexpr int THREADS = (CW + EW + 2) * 64;
  constexpr int NKB = STEP_BYTES / 1024;          // 22 one-KB loads per step
  constexpr int D = NBUF - 1;
  constexpr int ZROWS = MT * 16;
  constexpr int XTILES = CW * MT;
  constexpr int SPR = 2;                          // steps per run (256 SNPs = 4 entries)
  constexpr int RWS = EW ? CW * ZROWS / EW : ZROWS;  // rows one finishing wave handles per window
  constexpr int CHR = RWS < 64 ? RWS : 64;           // ... in chunks of CHR rows
  constexpr int NCH = RWS / CHR;
  static_assert((D - 1) * NKB < 64 && (XSN - 1) * XTILES < 64 && XSN >= 2 && D >= 1, "vmcnt is a 6-bit counter; two X stages at least");
  static_assert(64 % CHR == 0 && RWS % CHR == 0 && (!EW || (CW * ZROWS) % EW == 0), "an epilogue wave takes whole chunks of rows");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;
  const int A = L.A, W = L.W, R = L.d.R;
  // parked rows are ZA float64 apart, ZA odd: the finishing waves work one row per lane, and rows an even number of 8-byte words
  // apart (A = 12: 24 dwords, four distinct banks for 64 lanes) made every one of their LDS accesses an 8-way bank conflict
  const int ZA = (A & 1) ? A + 2 : A + 1;   // (... and at least one spare float64 behind a row's classes: see the window's end)
  uint8_t* vbuf = lds;                                               // [NBUF][STEP_BYTES]
  uint8_t* xl0 = vbuf + (size_t)NBUF * STEP_BYTES;                   // [XSN][CW][MT][64 lanes][16 B]
  double* zq = reinterpret_cast<double*>(xl0 + (size_t)XSN * XTILES * 1024);  // [2][CW * ZROWS][A] parked logits
  double* tab_ic = zq + (size_t)2 * CW * ZROWS * ZA;   // [max_wins][A] intercepts
  double* tab_sc = tab_ic + (size_t)L.max_wins * A;    // [max_wins] 2^-f_w
  int* tab_rb = reinterpret_cast<int*>(tab_sc + L.max_wins);
  int* tab_nfl = tab_rb + L.max_chunks;
  int* tab_fl0 = tab_nfl + L.max_chunks;
  int* tab_gap = tab_fl0 + L.max_chunks;  // runs until the next run that ends a window (0: none follows)

  int wrange, htile;
  {  // all blocks of one window range on ONE XCD (its L2 serves the range's digit planes)
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int g = j / L.n_htiles;
    htile = j - g * L.n_htiles;
    wrange = xcd + 8 * g;
  }
  const int wa = wrange * L.wch;
  if (wa >= W) return;  // whole block exits before any barrier
  const int wb = min(W, wa + L.wch);
  const int r_begin = L.d.win_run0[wa];
  const int r_end = L.d.win_run1[wb - 1];
  const int n_runs = r_end - r_begin;
  const int n_steps = SPR * n_runs;
  const int64_t n0b = (int64_t)htile * (CW * MT * 16);

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
  if (tid == 0) {
    int next = -1;
    for (int r = n_runs - 1; r >= 0; --r) {
      tab_gap[r] = next < 0 ? 0 : next - r;
      if (tab_nfl[r] > 0) next = r;
    }
  }
  __syncthreads();
  const int abl = L.flags;  // GNX_LR_FLAGS (timing only): 1 raw logits, 2 no MFMA, 4 no flush, 8 no X, 16 no planes, 32 no barriers / stores, 64 no combine,
                            // 128 no epilogue priority, 1024 no stores to B (config 5a, one box: 7.46 ms; 1024: 7.03; 4: 5.38)
  const bool nobar = (abl & 32) != 0;
  const bool dbg_on = DBG && L.dbg != nullptr;   // development instantiation: cycle counters of wave 0 of every role, 16 per block
  unsigned long long* dbg = L.dbg + (size_t)blockIdx.x * 16;
  // ... and of block 8 a trace: [step][16] = when each of the 14 waves reached the step's barrier, [14] = when wave 0 left it, [15] = flush cycles
  constexpr int TRACE_STEPS = 256;
  unsigned long long* trace = dbg_on && blockIdx.x == 8 ? L.dbg + (size_t)gridDim.x * 16 : nullptr;
  auto tr = [&](int step, int slot, unsigned long long v) {
    if (DBG && trace && lane == 0 && step < TRACE_STEPS) trace[(size_t)step * 16 + slot] = v;
  };

  if (wave == CW + EW) {
    // ================================================== plane loader ==================================================
    const int8_t* vsrc = L.d.V2F + (size_t)r_begin * 4 * ENTRY_BYTES + (size_t)lane * 16;
    auto issue_planes = [&](int step) {
      if (abl & 16) return;
      const int8_t* src = vsrc + (size_t)min(step, n_steps - 1) * STEP_BYTES;   // a step's 22 KB are contiguous
      uint8_t* vdst = vbuf + (size_t)(step % NBUF) * STEP_BYTES;
      // (pacing the 22 loads with s_sleep so that the epilogue's stores to B do not queue behind a burst: measured 4 % SLOWER)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kb * 1024), (lptr_t)(vdst + (size_t)kb * 1024), 16, 0, 0);
      }
    };
#pragma unroll
    for (int p = 0; p < D; ++p) issue_planes(p);
    unsigned long long c_wait = 0, c_bar = 0, c_issue = 0;
    for (int s = 0; s < n_steps; ++s) {
      const unsigned long long t0 = dbg_on ? __builtin_readcyclecounter() : 0;
      wait_vm<(D - 1) * NKB>();
      const unsigned long long t1 = dbg_on ? __builtin_readcyclecounter() : 0;
      tr(s, wave, t1);
      if (!nobar) __builtin_amdgcn_s_barrier();
      const unsigned long long t2 = dbg_on ? __builtin_readcyclecounter() : 0;
      issue_planes(s + D);
      if (dbg_on) { c_wait += t1 - t0; c_bar += t2 - t1; c_issue += __builtin_readcyclecounter() - t2; }
    }
    wait_vm<0>();
    if (EW && !nobar) __builtin_amdgcn_s_barrier();
    if (dbg_on && lane == 0) { dbg[8] = c_wait; dbg[9] = c_bar; dbg[10] = c_issue; }
    return;
  }
  if (wave == CW + EW + 1) {
    // ================================================== X loader ==================================================
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
      for (int t = 0; t < XTILES; ++t) {
        __builtin_amdgcn_global_load_lds((gptr_t)(xrow[t] + rb), (lptr_t)(dst + t * 1024), 16, 0, 0);
      }
    };
    // the compute waves read HALF of a run's words per step (8 bytes per lane and row tile: four accumulator-free registers), so
    // stage r % XSN is busy through step 2r + 1 and takes X(r + XSN) only behind the first barrier of run r + 1
#pragma unroll
    for (int p = 0; p + 1 < XSN; ++p) issue_x(p);
    for (int r = 0; r < n_runs; ++r) {
      wait_vm<(XSN >= 2 ? XSN - 2 : 0) * XTILES>();   // X(r) has landed (only X(r+1) .. X(r+XSN-2) are younger)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (DBG) tr(2 * r, wave, __builtin_readcyclecounter());
      if (!nobar) __builtin_amdgcn_s_barrier();   // first step of the run: every wave is done with run r - 1
      if (XSN >= 2) issue_x(r + XSN - 1);
      if (DBG) tr(2 * r + 1, wave, __builtin_readcyclecounter());
      if (!nobar) __builtin_amdgcn_s_barrier();   // second step
    }
    wait_vm<0>();
    if (EW && !nobar) __builtin_amdgcn_s_barrier();
    return;
  }

  // ---- finishing `rows` (<= 64) rows x A classes parked at zr0: arithmetic and class order of k_base_logistic_i8 (bit-identical B).
  // A finishing wave is ONE in-order instruction stream: a sigmoid is a chain of ~40 dependent float64 operations, so a lane works
  // on PB classes at once (PB independent chains interleave: gnx_exp_scN) — per element the same operations in the same order ----
  constexpr int PB = EW ? 4 : 3;   // (self-service, 32 rows per wave: two lanes per row, six classes each at A = 12 = two rounds of three)
  const int e_r0 = lane / A, e_a0 = lane - e_r0 * A, e_dr = 64 / A, e_da = 64 - e_dr * A;
  // rows x A finished values -> B, 64 consecutive elements per store; iterations [i0, i1) of the ceil(rows A / 64)
  // (a store to B waits ~800 cycles at issue behind the loaders' traffic — the L1's queue is full by design: with an even number of
  // classes a lane stores TWO consecutive values of a row, 1 KB per float64 instruction, and every iteration counts double)
  const bool wide_st = (A & 1) == 0 && ((reinterpret_cast<uintptr_t>(L.b64) & 15) | (reinterpret_cast<uintptr_t>(L.b32) & 7)) == 0;
  // ... and FOUR float32 values (16 bytes a lane) when the classes come in fours and only the float32 output is asked for: 1 KB per
  // instruction there too, half the instructions of the pairs
  const bool quad_st = (A & 3) == 0 && !L.b64 && L.b32 && (reinterpret_cast<uintptr_t>(L.b32) & 15) == 0 && !(abl & 2048);
  const int spi = quad_st ? 256 : wide_st ? 128 : 64;   // values per store instruction: the iteration ranges below count in these
  auto store_rows = [&](const double* zr0, int w, int64_t nrow0, int rows, int i0 = 0, int i1 = 1 << 20) {
    if (quad_st) {
      typedef float v4f __attribute__((ext_vector_type(4)));
      const int d_r = 256 / A, d_a = 256 - d_r * A;
      int rl = (4 * lane + 256 * i0) / A, a = 4 * lane + 256 * i0 - rl * A;
      const size_t ow = (size_t)w * A;
      const int e_end = min(rows * A, 256 * i1);
      for (int e = 4 * lane + 256 * i0; e < e_end; e += 256) {
        const int64_t n = nrow0 + rl;
        if (n < L.N && !(abl & (32 | 1024))) {
          const double* zp = zr0 + rl * ZA + a;
          *reinterpret_cast<v4f*>(L.b32 + (size_t)n * W * A + ow + a) = v4f{(float)zp[0], (float)zp[1], (float)zp[2], (float)zp[3]};
        }
        a += d_a; rl += d_r;
        if (a >= A) { a -= A; ++rl; }
      }
      return;
    }
    if (wide_st) {
      typedef double v2d __attribute__((ext_vector_type(2)));
      typedef float v2f __attribute__((ext_vector_type(2)));
      const int d_r = 128 / A, d_a = 128 - d_r * A;
      int rl = (2 * lane + 128 * i0) / A, a = 2 * lane + 128 * i0 - rl * A;
      const size_t ow = (size_t)w * A;
      const int e_end = min(rows * A, 128 * i1);
      for (int e = 2 * lane + 128 * i0; e < e_end; e += 128) {
        const int64_t n = nrow0 + rl;
        if (n < L.N && !(abl & (32 | 1024))) {
          const size_t o = (size_t)n * W * A + ow + a;
          const double* zp = zr0 + rl * ZA + a;
          const v2d v = v2d{zp[0], zp[1]};
          if (L.b64) *reinterpret_cast<v2d*>(L.b64 + o) = v;
          if (L.b32) *reinterpret_cast<v2f*>(L.b32 + o) = v2f{(float)v[0], (float)v[1]};
        }
        a += d_a; rl += d_r;
        if (a >= A) { a -= A; ++rl; }
      }
      return;
    }
    int rl = e_r0, a = e_a0;
    if (i0 > 0) {
      rl = (lane + 64 * i0) / A;
      a = lane + 64 * i0 - rl * A;
    }
    const size_t ow = (size_t)w * A;
    const int e_end = min(rows * A, 64 * i1);
    for (int e = lane + 64 * i0; e < e_end; e += 64) {
      const int64_t n = nrow0 + rl;
      if (n < L.N && !(abl & (32 | 1024))) {
        const size_t o = (size_t)n * W * A + ow + a;
        const double v = zr0[rl * ZA + a];
        if (L.b64) L.b64[o] = v;
        if (L.b32) L.b32[o] = (float)v;
      }
      a += e_da; rl += e_dr;
      if (a >= A) { a -= A; ++rl; }
    }
  };
  auto phase1_n = [&](auto nb, double* zr0, int w, int it0, int it1, int rows) {   // iteration `it` = classes fsub + (NB it + i) lpr, i < NB
    constexpr int NB = decltype(nb)::value;
    if (abl & 1) return;
    const int lpr = 64 / rows, frow = lane % rows, fsub = lane / rows;
    double* zr = zr0 + frow * ZA;
    const double* ic = tab_ic + (w - wt0) * A;
    for (int it = it0; it < it1; ++it) {
      double v[NB];
      int a[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        a[i] = fsub + (NB * it + i) * lpr;
        const int ac = min(a[i], A - 1);
        v[i] = zr[ac] + ic[ac];
      }
      gnx_sigmoidN<NB>(v);
      // (the stores below are conditional, and the compiler moved each chain's last dozen instructions — ldexp, 1 + e, the reciprocal
      // and its Newton steps — INTO its store's branch: four chains one after the other.  Every result is due here, chains interleaved.)
#pragma unroll
      for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(v[i]));
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (a[i] < A) zr[a[i]] = v[i];
    }
  };
  auto phase1 = [&](double* zr0, int w, int it0, int it1, int rows) { phase1_n(std::integral_constant<int, PB>{}, zr0, w, it0, it1, rows); };
  auto normalise = [&](double* zr0, int rows) {
    if (abl & 1) return;
    const int lpr = 64 / rows, frow = lane % rows, fsub = lane / rows;
    double* zr = zr0 + frow * ZA;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lpr == 1) {
      // one lane per row (the epilogue waves): the row's values four at a time — A / 4 LDS round trips for the sum and as many for
      // the scaling instead of one per class each; the additions in class order, as everywhere.  (Scalars, not an array: sixteen
      // values in an array went to scratch whenever the inliner changed its mind about the code around it.)
      double sum = 0.0;
#pragma unroll
      for (int g = 0; g < 16; g += 4) {
        if (g < A) {
          const double a0 = zr[g], a1 = g + 1 < A ? zr[g + 1] : 0.0, a2 = g + 2 < A ? zr[g + 2] : 0.0, a3 = g + 3 < A ? zr[g + 3] : 0.0;
          sum += a0;
          if (g + 1 < A) sum += a1;
          if (g + 2 < A) sum += a2;
          if (g + 3 < A) sum += a3;
        }
      }
      const double rs = gnx_rcp_nr(sum);
#pragma unroll
      for (int g = 0; g < 16; g += 4) {
        if (g < A) {
          const double a0 = zr[g], a1 = g + 1 < A ? zr[g + 1] : 0.0, a2 = g + 2 < A ? zr[g + 2] : 0.0, a3 = g + 3 < A ? zr[g + 3] : 0.0;
          zr[g] = a0 * rs;
          if (g + 1 < A) zr[g + 1] = a1 * rs;
          if (g + 2 < A) zr[g + 2] = a2 * rs;
          if (g + 3 < A) zr[g + 3] = a3 * rs;
        }
      }
    } else {
      double sum = 0.0;
      for (int c = 0; c < A; ++c) sum += zr[c];
      const double rs = gnx_rcp_nr(sum);
      for (int a = fsub; a < A; a += lpr) zr[a] = zr[a] * rs;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  if (EW && wave >= CW) {
    // ================================================== epilogue waves ==================================================
    // A window parked at the end of run r has 2 * gap steps until the next window ends; its RWS rows are NCH chunks of CHR rows,
    // a chunk is n_it phase-1 iterations + the row normalisation + the stores in NSP parts: NCH (n_it + 1 + NSP) units, dealt evenly over the steps.
    const int ew = wave - CW;
    const int64_t nrow0 = n0b + (int64_t)ew * RWS;
    // the epilogue waves are the youngest waves of their SIMDs and lose the VALU arbitration (priority, then age) to the compute waves
    // beside them: their float64 chains would get the left-over issue slots and every wave of the block would wait for them at the
    // step's barrier.  Their work is short; let it go first.
    if (!(abl & 128)) __builtin_amdgcn_s_setprio(2);   // (GNX_LR_FLAGS & 128: without, for A/B timing)
    // (A finishing wave is one latency-bound instruction stream — it takes ~1 650 cycles per step with the MFMAs switched off and
    // ~1 800 with them — and how its work is cut does not move the kernel: 1, 2, 3, 4 or 6 classes per unit, 2 or 4 store parts, units
    // under a per-step time budget all measure within 1-2 % of each other.  GNX_LR_FLAGS bits 16-22 keep the knobs for A/B runs.)
    const int pbr = ((abl >> 16) & 7) ? ((abl >> 16) & 7) : GNX_P2F_PBR;   // classes of a lane per phase-1 unit (GNX_LR_FLAGS bits 16-18: A/B timing)
    const int n_it = (A + pbr * (64 / CHR) - 1) / (pbr * (64 / CHR));   // phase-1 iterations of a lane: pbr classes each
    const int NSP = ((abl >> 20) & 7) ? ((abl >> 20) & 7) : GNX_P2F_NSP;  // the stores of a chunk in NSP parts (bits 20-22)
    const int n_sti = (CHR * A + spi - 1) / spi, sti_part = (n_sti + NSP - 1) / NSP;
    const int upc = n_it + 1 + NSP;                                    // units per chunk
    const int n_units = NCH * upc;
    int parked = 0;
    bool job = false;
    int job_w = 0, job_step = 0, job_nst = 0, job_unit = 0;
    double* job_z = nullptr;
    auto run_units = [&](int u_end) {
      for (; job_unit < u_end; ++job_unit) {
        const int ch = job_unit / upc, it = job_unit - ch * upc;
        double* z = job_z + (size_t)ch * CHR * ZA;
        if (it < n_it) {
          switch (pbr) {
            case 1: phase1_n(std::integral_constant<int, 1>{}, z, job_w, it, it + 1, CHR); break;
            case 2: phase1_n(std::integral_constant<int, 2>{}, z, job_w, it, it + 1, CHR); break;
            case 3: phase1_n(std::integral_constant<int, 3>{}, z, job_w, it, it + 1, CHR); break;
            case 6: phase1_n(std::integral_constant<int, 6>{}, z, job_w, it, it + 1, CHR); break;
            default: phase1_n(std::integral_constant<int, 4>{}, z, job_w, it, it + 1, CHR); break;
          }
        } else if (it == n_it) normalise(z, CHR);
        else store_rows(z, job_w, nrow0 + ch * CHR, CHR, (it - n_it - 1) * sti_part, (it - n_it) * sti_part);
      }
    };
    auto job_work = [&]() {
      if (!job) return;
      ++job_step;   // (32-bit arithmetic: the 64-bit quotient this once was is a ~100-instruction routine, run every step)
      run_units(job_step >= job_nst ? n_units : (n_units * job_step) / job_nst);
      if (job_step >= job_nst) job = false;
    };
    unsigned long long c_bar = 0, c_work = 0, c_max = 0;
    for (int r = 0; r < n_runs; ++r) {
#pragma unroll
      for (int h = 0; h < SPR; ++h) {
        const unsigned long long t0 = dbg_on ? __builtin_readcyclecounter() : 0;
        tr(SPR * r + h, wave, t0);
        lds_barrier(nobar);
        const unsigned long long t1 = dbg_on ? __builtin_readcyclecounter() : 0;
        job_work();
        if (dbg_on) {
          const unsigned long long t2 = __builtin_readcyclecounter();
          c_bar += t1 - t0; c_work += t2 - t1; c_max = t2 - t1 > c_max ? t2 - t1 : c_max;
        }
      }
      const int nfl = (abl & 4) ? 0 : tab_nfl[r];
      if (nfl == 1) {
        const int w = tab_fl0[r];
        if (w >= wa && w < wb) {
          job = true;
          job_w = w;
          job_step = 0;
          job_unit = 0;
          job_nst = SPR * (tab_gap[r] > 0 ? tab_gap[r] : n_runs - 1 - r);  // 0 (last run): after the trailing barrier
          job_z = zq + (size_t)(parked & 1) * (CW * ZROWS * ZA) + (size_t)ew * RWS * ZA;
          ++parked;
        }
      }
    }
    lds_barrier(nobar);  // trailing barrier: the last run's windows are parked
    if (job) run_units(n_units);
    if (dbg_on && lane == 0 && ew == 0) { dbg[4] = c_bar; dbg[5] = c_work; dbg[6] = c_max; dbg[7] = (unsigned long long)n_steps; }
    return;
  }

  // ================================================== compute waves ==================================================
  const int64_t n0 = n0b + (int64_t)wave * (MT * 16);
  v4i acc[MT][NFT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < NFT; ++t) acc[mt][t] = v4i{0, 0, 0, 0};

  // one step = 2 entries x 11 flat tiles x MT row tiles, ONE software pipeline over its 22 KB of digit planes: the planes of tiles
  // t + 1 .. t + PD are on their way from LDS while tile t multiplies.  hipcc does not build it: in this kernel every wait it places
  // is `s_waitcnt lgkmcnt(0)` — "all LDS reads back" (a micro-kernel with the same loop gets counted waits; here none of the 126
  // is) — so a read issued ahead was waited for together with the one that was needed, and behind a first attempt at pinning the
  // order with sched_group_barrier the X words' read took the list's first slot and every read came AFTER its tile's MFMAs: the wave
  // sat through an LDS round trip per tile, 2 370 cycles per step for 44 MFMAs.  The reads and their waits are therefore written out:
  // ds_read_b128 + s_waitcnt lgkmcnt(PD) as ONE inline-asm statement whose output is the register set of tile t + PD (the compiler
  // does not know it is pending; nothing touches it before the MFMAs of tile t + PD, PD statements later), a sched_barrier either side
  // of a tile's MFMAs keeps the stages in order.  LDS returns in order, so the count is exact; a compiler-issued read older than the
  // pipeline (the run's table entries) only makes a wait stricter.  1 450 cycles per step for a SIMD's first wave, 2 100 for its second.
  constexpr int PD = GNX_P2F_PD;   // prefetch distance in tiles
  auto mfma_step = [&](const uint8_t* pb, const int (&xw0)[MT], const int (&xw1)[MT]) {
    if (abl & 2) return;
    constexpr int NT = 2 * NFT;
    const unsigned la = (unsigned)(uintptr_t)(lptr_t)(pb) + (unsigned)lane * 16u;
    v4i buf[PD + 1];
#pragma unroll
    for (int t = 0; t < PD; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(buf[t]) : "v"(la), "n"(t * 1024));
    v4i xa[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xa[mt] = unpack16(xw0[mt]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      // read (tile t + PD) and wait for tile t in ONE asm statement whose only output is the NEW register set: a separate wait tied
      // to the set the MFMAs are about to read made it look freshly written, and the hazard recogniser put an s_nop before them
      if (t + PD < NT)
        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(%3)" : "=v"(buf[(t + PD) % (PD + 1)]) : "v"(la), "n"((t + PD) * 1024), "n"(PD) : "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(NT - 1 - t) : "memory");
      if (t == NFT) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xa[mt] = unpack16(xw1[mt]);
      }
      __builtin_amdgcn_sched_barrier(0);   // the MFMAs stay behind the wait ...
      const int tt = t < NFT ? t : t - NFT;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[mt], buf[t % (PD + 1)], acc[mt][tt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);   // ... and ahead of the next read, which overwrites the set two tiles on
    }
  };

  // ---- EW == 0: SELF-SERVICE epilogue.  Dedicated epilogue waves meet the compute waves at every step's barrier with a different
  // amount of work each step, are the youngest waves of their SIMDs (they lose the VALU arbitration) and, two or four per block,
  // run their float64 chains on one or two instruction streams: every compute wave waited 700-2200 cycles per step for them.
  // (Measured again with the hand-written read pipeline: 7.20 against 6.57 ms for four dedicated waves — kept as a tune option.)
  // Here a compute wave finishes its OWN 32 rows: the window parked at the end of run r is worked off in units (sigmoid rounds, the
  // row normalisation, the stores in parts) over the steps until its next window ends — float64 VALU work of one wave beside the
  // MFMAs of the wave it shares the SIMD with: the waves of a SIMD (w, w + 4) take their unit at opposite ends of the step.
  const int s_nit = (A + PB * (64 / ZROWS) - 1) / (PB * (64 / ZROWS));
  constexpr int S_NSP = 2;
  const int s_nsti = (ZROWS * A + spi - 1) / spi, s_part = (s_nsti + S_NSP - 1) / S_NSP;
  const int s_units = s_nit + 1 + S_NSP;
  bool job = false;
  int job_w = 0, job_step = 0, job_nst = 0, job_unit = 0;
  double* job_z = nullptr;
  auto run_units = [&](int u_end) {
    for (; job_unit < u_end; ++job_unit) {
      if (job_unit < s_nit) phase1(job_z, job_w, job_unit, job_unit + 1, ZROWS);
      else if (job_unit == s_nit) normalise(job_z, ZROWS);
      else store_rows(job_z, job_w, n0, ZROWS, (job_unit - s_nit - 1) * s_part, (job_unit - s_nit) * s_part);
    }
  };
  auto job_work = [&]() {
    if (EW || !job) return;
    ++job_step;
    run_units(job_step >= job_nst ? s_units : (int)((int64_t)s_units * job_step / job_nst));
    if (job_step >= job_nst) job = false;
  };
  const bool late = ((wave >> 2) & 1) != 0;   // the second wave of its SIMD (waves go to SIMDs round robin): its unit follows its MFMAs

  int parked = 0;
  unsigned long long c_f1 = 0, c_f2 = 0, c_f3 = 0, c_fn = 0;   // (instrumented build) a window's end: set-up, gather + combine + park, zeroing; how many
  auto flush = [&](int rl, int nfl, int w0) {   // (nfl, w0 = tab_nfl[rl], tab_fl0[rl]: read by the caller ahead of the run's MFMAs)
    if (nfl <= 0 || (abl & 4)) return;
    // the lane's column / row group as the optimiser cannot see through: everything derived from them below (predicates, the LDS
    // addresses of the parked logits) would otherwise be hoisted out of the run loop into a dozen registers held beside the
    // 88 accumulators for the whole kernel — recomputing them once per window costs nothing
    int i16 = lane & 15, kq = lane >> 4;
    asm volatile("" : "+v"(i16), "+v"(kq));
    for (int w = w0; w < w0 + nfl; ++w) {
      const unsigned long long tf0 = dbg_on ? __builtin_readcyclecounter() : 0;
      const int c0 = (w % R) * A;                  // the window's slot: flat class columns [c0, c0 + A)
      const double scale = tab_sc[w - wt0];
      const bool out = w >= wa && w < wb;
      double* zw = zq + (size_t)(parked & 1) * (CW * ZROWS * ZA) + (size_t)wave * ZROWS * ZA;
      // this lane as the HOME of one column of the slot: column i16 (< 16) or column i16 + 16 (lanes 0-7); never both (A <= 16)
      const bool in_hi = i16 < 8 && i16 + 16 >= c0 && i16 + 16 < c0 + A;
      const bool in_lo = i16 >= c0 && i16 < c0 + A;
      const int col = in_hi ? i16 + 16 : i16;
      const bool home = in_lo || in_hi;
      const bool lo8 = col < 8;
      // ... and as the SOURCE of the odd limbs of the lane 8 away, whose column decides which tile they come from: choosing here,
      // ahead of the move, is one select + one DPP move per odd limb (choosing there: two moves + one select)
      const bool plo8 = ror8(lo8 ? 1 : 0) != 0;
      // where the lane parks (row 4 kq of the wave's rows, class col - c0) and whether it does, worked out ONCE per window: left to
      // itself the compiler re-derived both for each of the eight accumulator registers (~20 instructions each) rather than hold them
      // (a parked row has a spare float64 behind its classes: the lanes that are home to no column of the slot park their —
      // meaningless — value there, and the stores need no exec mask: the eight registers' code is straight-line)
      int zoff = 4 * kq * ZA + (home ? col - c0 : A);
      asm volatile("" : "+v"(zoff));
      const unsigned long long tf1 = dbg_on ? __builtin_readcyclecounter() : 0;
      // Straight-line: no branch and no exec mask between the eight registers (with a uniform branch and a predicated store per
      // register this part took 1 690 cycles per window's end; like this 960).  A window's end is latency-bound, not issue-bound — it
      // takes a wave as long alone on its SIMD as beside the other compute wave doing the same; how many registers' chains the
      // scheduler may interleave (GNX_P2F_FLUSH_GROUP = 1, 2, 4, 8) measured the same.
      if (out && !(abl & 64)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {  // int32 16x16 C/D layout: column = lane & 15, row = 4 (lane >> 4) + reg
            // limbs in pairs, p_k = a_2k + 256 a_2k+1 in int32 (|a| <= 256 K for a window of K SNPs: the launcher declines models with
            // wider windows), V = p_0 + 2^16 p_1 + 2^32 p_2 + 2^48 a_6 as P + 2^32 Q with P, Q exact in float64: ONE rounding, that of combine()
            int pk[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              // (values first, THEN the selects: a conditional between two accumulator elements is a select of ADDRESSES to the front
              // end, and in this straight-line form ten accumulator tiles then lived in scratch)
              const int t0 = acc[mt][3 * k][r], t1 = acc[mt][3 * k + 1][r], t2 = acc[mt][3 * k + 2][r];
              const int ev = in_hi ? t1 : t0;
              const int od = ror8(plo8 ? t1 : t2);
              pk[k] = (int)(((unsigned)od << 8) + (unsigned)ev);
            }
            const int t9 = acc[mt][9][r], t10 = acc[mt][10][r];
            const int l6 = in_hi ? t10 : t9;
            const double P = __builtin_fma((double)pk[1], 65536.0, (double)pk[0]);
            const double Q = __builtin_fma((double)l6, 65536.0, (double)pk[2]);
            zw[zoff + (mt * 16 + r) * ZA] = __builtin_fma(Q, 4294967296.0, P) * scale;
            if (GNX_P2F_FLUSH_GROUP == 1 || ((mt * 4 + r) % GNX_P2F_FLUSH_GROUP) == GNX_P2F_FLUSH_GROUP - 1)
              __builtin_amdgcn_sched_barrier(0);  // GROUP registers' gathers at a time: the accumulators leave no room for all eight
          }
        }
      }
      const unsigned long long tf2 = dbg_on ? __builtin_readcyclecounter() : 0;
      // zero the slot's columns: lane column i16 of tile t is flat column q = 16 t + i16 = 24 limb + column, so the lanes to zero
      // repeat every three tiles — three predicated REGIONS (exec masks, plain moves) instead of a select per register
      // (flat columns >= 168 of the last tile are padding: their planes are zero, zeroing their accumulators too changes nothing)
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        int c = (16 * kk) % NCF + i16;
        c = c >= NCF ? c - NCF : c;
        if (c >= c0 && c < c0 + A) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = kk; t < NFT; t += 3) acc[mt][t] = v4i{0, 0, 0, 0};
        }
      }
      if (dbg_on) { const unsigned long long tf3 = __builtin_readcyclecounter(); c_f1 += tf1 - tf0; c_f2 += tf2 - tf1; c_f3 += tf3 - tf2; ++c_fn; }
      if (!out) continue;
      if (EW && nfl == 1) {
        ++parked;  // an epilogue wave takes it from here (after the next barrier)
      } else if (!EW && nfl == 1) {
        if (job) run_units(s_units);   // (the previous window's units were dealt over the steps up to here: nothing is left)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        job = true;
        job_w = w;
        job_step = 0;
        job_unit = 0;
        job_nst = max(1, SPR * (tab_gap[rl] > 0 ? tab_gap[rl] : n_runs - 1 - rl));
        job_z = zw;
        ++parked;
      } else {     // several windows end at once (chromosome ends, wide contexts): finish it here
        if (!EW && job) { run_units(s_units); job = false; }
        // (a rare path — never at the default context — kept narrow: one class at a time, no batch temporaries beside the accumulators)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(abl & 1)) {
          int ln = lane;
          asm volatile("" : "+v"(ln));   // (this path's addresses worked out HERE: hoisted, they were a spilled register the common path reloaded at every window's end)
          const int lpr = 64 / ZROWS, frow = ln % ZROWS, fsub = ln / ZROWS;
          double* zr = zw + frow * ZA;
          const double* ic = tab_ic + (w - wt0) * A;
          for (int a = fsub; a < A; a += lpr) zr[a] = gnx_sigmoid(zr[a] + ic[a]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          double sum = 0.0;
          for (int c = 0; c < A; ++c) sum += zr[c];
          const double rs = gnx_rcp_nr(sum);
          for (int a = fsub; a < A; a += lpr) zr[a] = zr[a] * rs;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        store_rows(zw, w, n0, ZROWS);
      }
    }
  };

  unsigned long long c_bar = 0, c_mm = 0, c_fl = 0, t_begin = dbg_on ? __builtin_readcyclecounter() : 0;
  typedef int v2i __attribute__((ext_vector_type(2)));
  for (int r = 0; r < n_runs; ++r) {
    // what ends after this run, asked for now: the answer is back long before the run's MFMAs are through (asked for behind them it
    // cost every wave an LDS round trip per run, windows ending or not)
    const int r_nfl = tab_nfl[r], r_fl0 = tab_fl0[r];
#pragma unroll
    for (int h = 0; h < SPR; ++h) {
      const unsigned long long t0 = dbg_on ? __builtin_readcyclecounter() : 0;
      tr(SPR * r + h, wave, t0);
      lds_barrier(nobar);  // step SPR r + h: its planes (and, at h = 0, X(r)) are in LDS
      const unsigned long long t1 = dbg_on ? __builtin_readcyclecounter() : 0;
      if (wave == 0) tr(SPR * r + h, 14, t1);
      // words 2h, 2h + 1 of the lane's 16 packed bytes of run r: the two entries of this step
      const uint8_t* xs = xl0 + (size_t)(r % XSN) * (XTILES * 1024) + (size_t)wave * (MT * 1024) + lane * 16 + 8 * h;
      int x0[MT], x1[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const v2i q = *reinterpret_cast<const v2i*>(xs + mt * 1024);
        x0[mt] = q[0];
        x1[mt] = q[1];
      }
      const uint8_t* sb = vbuf + (size_t)((SPR * r + h) % NBUF) * STEP_BYTES;
      if (!late) job_work();
      __builtin_amdgcn_sched_barrier(0);   // (the X words' DS read stays out of the pipeline's scheduling region)
      mfma_step(sb, x0, x1);
      __builtin_amdgcn_sched_barrier(0);
      if (late) job_work();
      if (dbg_on) { c_bar += t1 - t0; c_mm += __builtin_readcyclecounter() - t1; }
    }
    const unsigned long long t2 = dbg_on ? __builtin_readcyclecounter() : 0;
    flush(r, r_nfl, r_fl0);
    if (dbg_on) c_fl += __builtin_readcyclecounter() - t2;
    if (DBG && wave == 0) tr(SPR * r + SPR - 1, 15, __builtin_readcyclecounter() - t2);
  }
  if (EW) lds_barrier(nobar);  // trailing barrier: the last run's parked windows become visible to the epilogue waves
  if (!EW && job) run_units(s_units);
  if (dbg_on && lane == 0 && wave == 0) { dbg[0] = c_bar; dbg[1] = c_mm; dbg[2] = c_fl; dbg[3] = __builtin_readcyclecounter() - t_begin; dbg[7] = (unsigned long long)n_steps;
    dbg[11] = c_f1; dbg[12] = c_f2; dbg[13] = c_f3; dbg[14] = c_fn; }
}

template <int MT, int CW, int EW, int XSN, int NBUF>
hipError_t launch_flat(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  constexpr bool NODBG = false;
  BaseLRLaunch P = L;
  P.flags = tune.lr_flags;
  {  // limb pairs in int32 at a window's end: |a_2k + 256 a_2k+1| <= 257 * 256 K for a window of K SNPs — K < 32 640.  A model with
     // wider windows (none of the reference's configurations) is declined: the caller widens the rows and runs the int8 kernels.
     // GNX_LR_FLAGS bit 25: decline always (tests)
    int span = 0;
    for (int64_t w = 0; w < L.W; ++w) span = std::max(span, L.h_win_chunk1[(size_t)w] - L.h_win_chunk0[(size_t)w]);
    if (span > 120 || (tune.lr_flags & (1 << 25))) return hipErrorNotSupported;
  }
  const int haps_per_block = CW * MT * 16;
  const int64_t gx = (L.N + haps_per_block - 1) / haps_per_block;
  // window ranges: one block per CU at a time (150 KB of LDS), so the grid is walked in rounds and the last, partial round costs a
  // whole block: ~18 blocks per CU keep that under 3 % (config 5a: 9.2 rounds of 60-window blocks measured 8.26 ms, 18.4 rounds of
  // 30-window blocks 7.88); a range re-walks only the half window of context before its first window, so short ranges are cheap —
  // down to ~24 windows, where the re-walk reaches 4 %
  const int bpc = tune.lr_bpc > 0 ? tune.lr_bpc : 18;
  int64_t want = ((int64_t)bpc * n_cu + gx - 1) / gx;
  want = std::min<int64_t>(want, std::max<int64_t>(8, L.W / 24));
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
    lds = (size_t)NBUF * (2 * NFT * 1024) + (size_t)XSN * CW * MT * 1024 + (size_t)2 * CW * MT * 16 * ((L.A & 1) ? L.A + 2 : L.A + 1) * sizeof(double) +
          (size_t)4 * P.max_chunks * sizeof(int) + (size_t)P.max_wins * (L.A + 1) * sizeof(double);
    if (lds <= (size_t)160 * 1024 || wch == 4) break;
  }
  if (lds > (size_t)160 * 1024) return hipErrorNotSupported;
  const int n_ranges8 = ((n_ranges + 7) / 8) * 8;
  P.wch = wch;
  P.n_htiles = (int)gx;
  P.n_rg8 = n_ranges8 / 8;
  if (tune.debug) std::fprintf(stderr, "k_base_logistic_p2f<%d,%d,%d,%d,%d>: lds=%zu grid=%lld wch=%d\n", MT, CW, EW, XSN, NBUF, lds, (long long)(gx * n_ranges8), wch);
  const size_t nblk = (size_t)(gx * n_ranges8);
  if (!(tune.debug & 2)) {
    GNX_LDS_OPTIN(lds, k_base_logistic_p2f<MT, CW, EW, XSN, NBUF, NODBG>);
    hipLaunchKernelGGL((k_base_logistic_p2f<MT, CW, EW, XSN, NBUF, NODBG>), dim3((unsigned)nblk), dim3((CW + EW + 2) * 64), lds, s, P);
    return hipGetLastError();
  }
  // GNX_DEBUG & 2 (development): the instrumented instantiation — where the waves' cycles go; synchronous, never on a production path
  GNX_LDS_OPTIN(lds, k_base_logistic_p2f<MT, CW, EW, XSN, NBUF, true>);
  constexpr size_t TRACE = 256 * 16;
  if (hipMalloc(&P.dbg, (nblk * 16 + TRACE) * sizeof(unsigned long long)) != hipSuccess) P.dbg = nullptr;
  if (P.dbg) (void)hipMemsetAsync(P.dbg, 0, (nblk * 16 + TRACE) * sizeof(unsigned long long), s);
  hipLaunchKernelGGL((k_base_logistic_p2f<MT, CW, EW, XSN, NBUF, true>), dim3((unsigned)nblk), dim3((CW + EW + 2) * 64), lds, s, P);
  hipError_t err = hipGetLastError();
  if (P.dbg) {
    std::vector<unsigned long long> h(nblk * 16 + TRACE);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), P.dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(P.dbg);
    double sum[16] = {0};
    size_t live = 0;
    for (size_t b = 0; b < nblk; ++b) {
      if (!h[b * 16 + 3]) continue;
      ++live;
      for (int k = 0; k < 16; ++k) sum[k] += (double)h[b * 16 + (size_t)k];
    }
    const double st = sum[7] > 0 ? sum[7] : 1;
    std::fprintf(stderr, "p2f cycles per step (mean of %zu blocks, %.0f steps each): compute wave0: barrier %.0f mfma %.0f flush %.0f total %.0f | epilogue wave0: barrier %.0f "
                 "work %.0f (longest step %.0f) | plane loader: wait %.0f barrier %.0f issue %.0f\n", live, st / (live ? live : 1), sum[0] / st, sum[1] / st, sum[2] / st, sum[3] / st,
                 sum[4] / st, sum[5] / st, sum[6] / (live ? live : 1), sum[8] / st, sum[9] / st, sum[10] / st);
    if (sum[14] > 0) std::fprintf(stderr, "p2f cycles per window's end (compute wave0, %.0f per block): set-up %.0f, gather + combine + park %.0f, zeroing %.0f\n",
                                  sum[14] / (live ? live : 1), sum[11] / sum[14], sum[12] / sum[14], sum[13] / sum[14]);
    if (tune.debug & 4) {  // block 8's steps: how long after the previous barrier opened each role reached this one, and who came last
      const unsigned long long* t = h.data() + nblk * 16;
      std::fprintf(stderr, "p2f trace of block 8: step | cycles since the previous release: compute waves (max, which) | epilogue waves (max) | plane loader | X loader | release | flush\n");
      for (int st2 = 1; st2 < 80 && t[(size_t)st2 * 16 + 14]; ++st2) {
        const unsigned long long rel0 = t[(size_t)(st2 - 1) * 16 + 14];
        auto at = [&](int slot) { const unsigned long long v = t[(size_t)st2 * 16 + slot]; return v > rel0 ? (long long)(v - rel0) : 0LL; };
        long long cm = 0, em = 0; int cw = 0;
        for (int k = 0; k < CW; ++k) if (at(k) > cm) { cm = at(k); cw = k; }
        for (int k = CW; k < CW + EW; ++k) em = std::max(em, at(k));
        std::fprintf(stderr, "  %3d | %6lld (w%d) | %6lld | %6lld | %6lld | %6lld | %6llu | compute", st2, cm, cw, em, at(CW + EW), at(CW + EW + 1), at(14), t[(size_t)(st2 - 1) * 16 + 15]);
        for (int k = 0; k < CW; ++k) std::fprintf(stderr, " %5lld", at(k));
        std::fprintf(stderr, " | epilogue");
        for (int k = CW; k < CW + EW; ++k) std::fprintf(stderr, " %5lld", at(k));
        std::fprintf(stderr, "\n");
      }
    }
  }
  return err;
}

}  // namespace

// L.X = packed matrix (gnx_pack_x layout), L.ldx = its row stride in bytes, L.last_row = zero-padded copy of packed row N-1,
// L.h_win_chunk0/1 = HOST copies of d.win_run0/1.  Returns hipErrorNotSupported when the model has no 2-bit planes or no
// instantiation fits (the caller widens X to int8 and runs the int8 kernels).
hipError_t gnx_launch_base_logistic_p2(const BaseLRLaunch& L, int n_cu, const gnx_tune& tune, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  if (L.d.V2F && L.d.EPR == 4 && L.d.R * L.A == NCF && L.h_win_chunk0 && L.h_win_chunk1) {
    // flat column tiles (24 class columns): 256 rows per block, X read once; the deepest configuration that fits the LDS
    // (GNX_P2_TUNE = "2,8,2,xsn,nbuf" picks the depth: development)
    const int tb = tune.p2_nbuf;
    if (tune.p2_mt == 2) {   // development: GNX_P2_TUNE = "2,8,ew,2,nbuf" with ew = 0 (self-service), 2 or 4 epilogue waves
      if (tune.p2_ew == 0) return tb == 2 ? launch_flat<2, 8, 0, 2, 2>(L, n_cu, tune, s) : launch_flat<2, 8, 0, 2, 3>(L, n_cu, tune, s);
      if (tune.p2_ew == 2) return launch_flat<2, 8, 2, 2, 3>(L, n_cu, tune, s);
      if (tune.p2_ew == 4) return tb == 2 ? launch_flat<2, 8, 4, 2, 2>(L, n_cu, tune, s) : launch_flat<2, 8, 4, 2, 3>(L, n_cu, tune, s);
      return hipErrorInvalidValue;
    }
    hipError_t e = launch_flat<2, 8, 4, 2, 3>(L, n_cu, tune, s);
    if (e == hipErrorNotSupported) e = launch_flat<2, 8, 4, 2, 2>(L, n_cu, tune, s);
    return e;
  }
  if (!L.d.V2 || L.d.NT2 < 1 || (L.d.EPR != 4 && L.d.EPR != 8) || !L.h_win_chunk0 || !L.h_win_chunk1) return hipErrorNotSupported;
  const bool small = L.N <= 64 * 4;
  const int tm = tune.p2_mt, tw = tune.p2_cw, te = tune.p2_ew, tx = tune.p2_xsn, tb = tune.p2_nbuf ? tune.p2_nbuf : 3;
#define GNX_P2(MT_, CW_, EW_, XS_, NB_)                                                                                          \
  if (tm == MT_ && tw == CW_ && te == EW_ && tx == XS_ && tb == NB_)                                                             \
    return L.d.EPR == 8 ? launch<MT_, CW_, EW_, (XS_ > 2 ? 2 : XS_), NB_, 8>(L, n_cu, tune, s) : launch<MT_, CW_, EW_, XS_, NB_, 4>(L, n_cu, tune, s);
  if (tm) {  // GNX_P2_TUNE="mt,cw,ew,xsn,nbuf" (development; with 512-SNP runs at most two X stages: vmcnt is a 6-bit counter)
    GNX_P2(2, 8, 4, 3, 3) GNX_P2(2, 8, 4, 3, 4) GNX_P2(2, 8, 4, 2, 3) GNX_P2(2, 8, 4, 2, 4) GNX_P2(2, 8, 4, 1, 4)
    GNX_P2(2, 8, 0, 3, 4) GNX_P2(1, 4, 0, 2, 3) GNX_P2(1, 8, 2, 4, 3) GNX_P2(1, 8, 4, 4, 3)
    GNX_P2(2, 4, 2, 2, 3) GNX_P2(2, 4, 2, 1, 3) GNX_P2(2, 4, 2, 1, 4) GNX_P2(4, 4, 4, 1, 3)
    return hipErrorInvalidValue;
  }
#undef GNX_P2
  if (L.d.EPR == 8) {  // runs of 512 SNPs: 128 packed bytes per row visit
    if (small) return launch<1, 4, 0, 2, 3, 8>(L, n_cu, tune, s);
    hipError_t e = launch<2, 8, 4, 2, 4, 8>(L, n_cu, tune, s);
    if (e == hipErrorNotSupported) e = launch<2, 8, 4, 2, 3, 8>(L, n_cu, tune, s);
    if (e == hipErrorNotSupported) e = launch<2, 8, 4, 1, 3, 8>(L, n_cu, tune, s);
    if (e == hipErrorNotSupported) e = launch<1, 8, 4, 2, 3, 8>(L, n_cu, tune, s);
    return e;
  }
  if (small) return launch<1, 4, 0, 2, 3, 4>(L, n_cu, tune, s);
  // 256 rows per block, 4 epilogue waves; the parked logits (2 x 256 rows x A doubles) share the LDS with the plane ring and the X
  // stages: the deepest configuration that fits (A <= 12: 4 plane slots + 3 X stages; more classes: shallower)
  hipError_t e = launch<2, 8, 4, 3, 4, 4>(L, n_cu, tune, s);
  if (e == hipErrorNotSupported) e = launch<2, 8, 4, 3, 3, 4>(L, n_cu, tune, s);
  if (e == hipErrorNotSupported) e = launch<2, 8, 4, 2, 3, 4>(L, n_cu, tune, s);
  if (e == hipErrorNotSupported) e = launch<1, 8, 4, 4, 3, 4>(L, n_cu, tune, s);
  return e;
}

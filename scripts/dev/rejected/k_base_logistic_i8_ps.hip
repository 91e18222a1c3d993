// k_base_logistic_i8_ps.hip — "piece-synchronous" variant of the exact int8 logistic pass (gfx950).
//
// Same contract, tables, fixed-point arithmetic and epilogue as k_base_logistic_i8.hip (reference
// src/Base/base.py:146-180, src/Base/models.py:12-21); different choreography, after the cycle breakdown of that kernel
// (26 % of a step waiting at two block barriers per 128 SNPs, 10 % publishing the X tile to LDS, 1.6 GB of
// digit planes per launch through the L1s):
//  * no X tile: every wave loads ITS OWN 16-row tiles straight into the MFMA A-operand layout (lane&15 = haplotype,
//    lane>>4 = 16-SNP block of the chunk: one unaligned 16-byte load per lane per tile and chunk), PF chunks ahead,
//    and walks its window range without ever waiting for another wave's data;
//  * the digit planes of G consecutive chunks (G x 7 KB) are copied by `global_load_lds_dwordx4` (L2 -> LDS, no VGPRs)
//    into a double-buffered LDS window shared by all waves of the block: ONE block barrier per G chunks;
//  * one 16-wave block per CU (512 haplotypes) halves the plane bytes per X byte.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gnx_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) xbytes16 { v4i v; };
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int LIMBS = 7;

__device__ __forceinline__ v4i load_x16(const int8_t* p) {
  // unconditional unaligned 16-byte load; reads may run up to 63 bytes past a row's end (rows other than the last read
  // their successor, the last row is served from a zero-padded copy, bytes past C only ever meet zero weights)
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

// MT 16-row tiles per wave, WAVES waves per block, at most G chunks per plane group.  A group never crosses a piece end, so
// the window flush sits BETWEEN groups, outside the MFMA loop (inside it, hipcc kept two copies of the accumulators
// around the conditional flush: 225+ VGPRs).  PF = chunks of X every wave keeps in flight (register ring).
template <int MT, int WAVES, int G, int PF>
__global__ __launch_bounds__(WAVES * 64) void k_base_logistic_i8_ps(BaseLRLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int CHUNK_BYTES = LIMBS * 1024;      // digit planes of one 64-SNP chunk (one column tile)
  constexpr int GROUP_BYTES = G * CHUNK_BYTES;
  constexpr int THREADS = WAVES * 64;
  constexpr int ROWS = WAVES * MT * 16;          // haplotypes per block
  constexpr int KB_PER_WAVE = (G * LIMBS + WAVES - 1) / WAVES;  // 1 KB plane blocks each wave copies per group
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int A = L.A, W = L.W, R = L.d.R;
  uint8_t* pbuf = lds;                            // [2][G][LIMBS][64 lanes] x 16 B
  double* zb = reinterpret_cast<double*>(pbuf + 2 * GROUP_BYTES) + (size_t)wave * (MT * 16) * A;
  double* tab_ic = reinterpret_cast<double*>(pbuf + 2 * GROUP_BYTES) + (size_t)ROWS * A;  // [max_wins][A] intercepts
  double* tab_sc = tab_ic + (size_t)L.max_wins * A;                                        // [max_wins] 2^-f_w
  int* tab_j0 = reinterpret_cast<int*>(tab_sc + L.max_wins);
  int* tab_nfl = tab_j0 + L.max_chunks;
  int* tab_fl0 = tab_nfl + L.max_chunks;
  int* grp_c0 = tab_fl0 + L.max_chunks;          // [n_groups + 1] first chunk of every group
  int* grp_cnt = grp_c0 + L.max_chunks + 1;      // [1] number of groups

  // XCD-aware decomposition (as k_base_logistic_i8): all blocks of one window range on ONE XCD
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
  const int64_t n0 = (int64_t)htile * ROWS + (int64_t)wave * (MT * 16);  // first haplotype of the wave

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
  if (tid == 0) {  // groups: runs of <= G chunks that end at a flush chunk (a piece end) or at the group size
    int ng = 0, c = 0;
    while (c < n_chunks) {
      int e = c;
      while (tab_nfl[e] == 0 && e - c + 1 < G && e + 1 < n_chunks) ++e;
      grp_c0[ng++] = c;
      c = e + 1;
    }
    grp_c0[ng] = n_chunks;
    grp_cnt[0] = ng;
  }
  __syncthreads();
  const int n_groups = grp_cnt[0];

  // digit planes of group g -> LDS buffer g&1, 1 KB per wave instruction, clamped block index (no branch around the copy)
  const int8_t* vsrc = L.d.V8 + (size_t)c_begin * CHUNK_BYTES;
  auto dma_group = [&](int g) {
    const int gg = min(g, n_groups - 1);
    const int c0 = grp_c0[gg];
    const int last_kb = (grp_c0[gg + 1] - c0) * LIMBS - 1;
    const int8_t* src = vsrc + (size_t)c0 * CHUNK_BYTES;
    uint8_t* dst = pbuf + (size_t)(g & 1) * GROUP_BYTES;
#pragma unroll
    for (int it = 0; it < KB_PER_WAVE; ++it) {
      const int kb = min(wave + it * WAVES, last_kb);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)kb * 1024 + lane * 16), (lptr_t)(dst + (size_t)kb * 1024), 16, 0, 0);
    }
  };

  // A-operand rows of this wave: rows >= N-1 read the padded copy of the last row (rows past N are never written)
  const int8_t* rowp[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int64_t n = n0 + mt * 16 + i16;
    rowp[mt] = (n >= L.N - 1 ? L.last_row : L.X + n * L.ldx) + 16 * kq;
  }

  v4i acc[MT][LIMBS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int l = 0; l < LIMBS; ++l) acc[mt][l] = v4i{0, 0, 0, 0};

  dma_group(0);
  // X ring: slot i holds the A operands of chunk (next + i), i < PF; a slot is refilled (chunk + PF) right after its
  // registers were handed to the MFMAs, so PF chunks per wave are always in flight, across group ends and flushes
  v4i xs[PF][MT];
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    const int j0 = tab_j0[min(p, n_chunks - 1)];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xs[p][mt] = load_x16(rowp[mt] + j0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // prologue only: group 0 has landed
  __syncthreads();

  // one chunk: hand slot P to the MFMAs, refill it with chunk CL + PF (clamped: the tail re-fetches the last chunk); the
  // digit planes of the NEXT chunk of the group (BN, clamped to the group's last chunk) are read from LDS while the
  // MFMAs of this one (BC) run: without that the 7 ds_read_b128 latencies of a chunk were exposed one by one
  // (3200 cycles per chunk and wave for 224 cycles of MFMA work)
#define GNX_PS_CHUNK(P, CL, KLOC, BC, BN)                                                            \
  {                                                                                                   \
    v4i xa[MT];                                                                                       \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) xa[mt] = xs[P][mt];                             \
    {                                                                                                 \
      const int j0_ = tab_j0[min((CL) + PF, n_chunks - 1)];                                           \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) xs[P][mt] = load_x16(rowp[mt] + j0_);         \
    }                                                                                                 \
    const v4i* vn = reinterpret_cast<const v4i*>(pb + (size_t)min((KLOC) + 1, cn - 1) * CHUNK_BYTES) + lane; \
    _Pragma("unroll") for (int l = 0; l < LIMBS; ++l) BN[l] = vn[l * 64];                             \
    _Pragma("unroll") for (int l = 0; l < LIMBS; ++l) {                                               \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
          acc[mt][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa[mt], BC[l], acc[mt][l], 0, 0, 0);     \
    }                                                                                                 \
  }

  v4i bA[LIMBS], bB[LIMBS];
  for (int g = 0; g < n_groups; ++g) {
    dma_group(g + 1);  // the other buffer: every wave left it at the barrier that ended group g-1
    const uint8_t* pb = pbuf + (size_t)(g & 1) * GROUP_BYTES;
    const int c0 = grp_c0[g], cn = grp_c0[g + 1] - c0;
    {
      const v4i* v0 = reinterpret_cast<const v4i*>(pb) + lane;
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) bA[l] = v0[l * 64];
    }
    int k = 0;
#pragma unroll 1
    for (; k + PF <= cn; k += PF) {
#pragma unroll
      for (int p = 0; p < PF; p += 2) {
        GNX_PS_CHUNK(p, c0 + k + p, k + p, bA, bB);
        GNX_PS_CHUNK(p + 1, c0 + k + p + 1, k + p + 1, bB, bA);
      }
    }
    const int rem = cn - k;  // < PF chunks left (block-uniform): slots 0..rem-1, then rotate the ring back into phase
    if (rem > 0) {
#pragma unroll
      for (int p = 0; p < PF - 1; ++p)
        if (p < rem) {
          GNX_PS_CHUNK(p, c0 + k + p, k + p, bA, bB);
#pragma unroll
          for (int l = 0; l < LIMBS; ++l) bA[l] = bB[l];
        }
      v4i t[PF][MT];
#pragma unroll
      for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) t[p][mt] = xs[p][mt];
#pragma unroll
      for (int r = 1; r < PF; ++r)
        if (rem == r) {
#pragma unroll
          for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xs[p][mt] = t[(p + r) % PF][mt];
        }
    }
    // ---- piece end: windows that finished with this group's last chunk (block-uniform; wave-private state only) ----
    const int cl = c0 + cn - 1;
    const int nfl = tab_nfl[cl];
    if (nfl > 0) {
      const int w0 = tab_fl0[cl];
      for (int w = w0; w < w0 + nfl; ++w) {
        const int cbase = (w % R) * A;
        const double scale = tab_sc[w - wt0];
        const int col = i16 - cbase;
        const bool mine = (col >= 0) && (col < A);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r)  // int32 16x16 C/D layout: column = lane&15, row = 4*(lane>>4) + reg
              zb[(mt * 16 + 4 * kq + r) * A + col] = combine(acc[mt], r, scale);
          }
#pragma unroll
          for (int l = 0; l < LIMBS; ++l)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][l][r] = mine ? 0 : acc[mt][l][r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // zb is wave-private: LDS ops of one wave complete in order
        if (w >= wa && w < wb && lane < MT * 16) {
          const int64_t n = n0 + lane;
          double* z = zb + lane * A;
          double sum = 0.0;
          for (int a = 0; a < A; ++a) {
            const double p = 1.0 / (1.0 + exp(-(z[a] + tab_ic[(w - wt0) * A + a])));
            z[a] = p;
            sum += p;
          }
          if (n < L.N) {
            const size_t o = ((size_t)n * W + w) * A;
            for (int a = 0; a < A; ++a) {
              const double v = z[a] / sum;
              if (L.b64) L.b64[o + a] = v;
              if (L.b32) L.b32[o + a] = (float)v;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    // this wave's share of group g+1 has landed: it was issued before this group's cn x MT ring loads, of which at most
    // PF x MT are still in flight (short groups: wait for everything); then the block agrees that buffer g&1 is free and
    // buffer (g+1)&1 is complete
    if (cn >= PF) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF * MT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
#undef GNX_PS_CHUNK
}

template <int MT, int WAVES, int G, int PF>
hipError_t launch_ps(const BaseLRLaunch& L, int n_cu, hipStream_t s) {
  BaseLRLaunch P = L;
  const int rows = WAVES * MT * 16;
  const int64_t gx = (L.N + rows - 1) / rows;
  int bpc = 2;  // window ranges: ~2 blocks per CU in total (one 16-wave block is resident per CU)
  if (const char* t = std::getenv("GNX_LR_BPC")) bpc = std::max(1, std::atoi(t));
  int64_t want = ((int64_t)bpc * n_cu + gx - 1) / gx;
  want = std::max<int64_t>(8, ((want + 7) / 8) * 8);
  if (const char* t = std::getenv("GNX_LR_WANT")) want = std::max(1, std::atoi(t));
  int wch = 0, n_ranges = 0;
  size_t lds = 0;
  for (;; want += 8) {
    wch = (int)((L.W + want - 1) / want);
    if (wch < 4) wch = 4;
    n_ranges = (L.W + wch - 1) / wch;
    int max_chunks = 0;
    if (L.h_win_chunk0 && L.h_win_chunk1) {
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
    lds = (size_t)2 * G * LIMBS * 1024 + (size_t)rows * L.A * sizeof(double) + ((size_t)4 * P.max_chunks + 2) * sizeof(int) +
          (size_t)P.max_wins * (L.A + 1) * sizeof(double);
    if (lds <= (size_t)160 * 1024 || wch == 4) break;
  }
  if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;
  const int n_ranges8 = ((n_ranges + 7) / 8) * 8;
  P.wch = wch;
  P.n_htiles = (int)gx;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_base_logistic_i8_ps<MT, WAVES, G, PF>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_base_logistic_i8_ps<MT, WAVES, G, PF>), dim3((unsigned)(gx * n_ranges8)), dim3(WAVES * 64), lds, s, P);
  return hipGetLastError();
}

}  // namespace

// single column tile only (R*A <= 16); the caller falls back to k_base_logistic_i8 otherwise
hipError_t gnx_launch_base_logistic_i8_ps(const BaseLRLaunch& L, int n_cu, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  if (L.d.NT != 1) return hipErrorInvalidValue;
  int tw = 16, tg = 8, tp = 4;
  if (const char* t = std::getenv("GNX_PS_TUNE")) std::sscanf(t, "%d,%d,%d", &tw, &tg, &tp);
  if (tw == 16 && tg == 8 && tp == 4) return launch_ps<2, 16, 8, 4>(L, n_cu, s);
  if (tw == 16 && tg == 8 && tp == 2) return launch_ps<2, 16, 8, 2>(L, n_cu, s);
  if (tw == 16 && tg == 8 && tp == 8) return launch_ps<2, 16, 8, 8>(L, n_cu, s);
  if (tw == 12 && tg == 8 && tp == 4) return launch_ps<2, 12, 8, 4>(L, n_cu, s);
  if (tw == 12 && tg == 8 && tp == 2) return launch_ps<2, 12, 8, 2>(L, n_cu, s);
  if (tw == 8 && tg == 8 && tp == 4) return launch_ps<2, 8, 8, 4>(L, n_cu, s);
  if (tw == 8 && tg == 8 && tp == 2) return launch_ps<2, 8, 8, 2>(L, n_cu, s);
  if (tw == 8 && tg == 4 && tp == 4) return launch_ps<2, 8, 4, 4>(L, n_cu, s);
  if (tw == 8 && tg == 4 && tp == 2) return launch_ps<2, 8, 4, 2>(L, n_cu, s);
  if (tw == 16 && tg == 4 && tp == 4) return launch_ps<2, 16, 4, 4>(L, n_cu, s);
  return hipErrorInvalidValue;
}

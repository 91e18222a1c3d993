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

#include "gnx_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) xbytes16 { v4i v; };

constexpr int LIMBS = 7;

__device__ __forceinline__ v4i load_x16(const int8_t* p, const int8_t* x_end) {
  xbytes16 r;
  if (__builtin_expect(p + 16 <= x_end, 1)) {
    __builtin_memcpy(&r, p, 16);
  } else {
    uint32_t w[4] = {0, 0, 0, 0};
    for (int b = 0; b < 16; ++b)
      if (p + b < x_end) w[b >> 2] |= (uint32_t)(uint8_t)p[b] << (8 * (b & 3));
    r.v = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
  }
  return r.v;
}

__device__ __forceinline__ double combine(const v4i (&acc)[LIMBS], int reg, double scale) {
  long long lo = (long long)acc[0][reg] + ((long long)acc[1][reg] << 8) + ((long long)acc[2][reg] << 16);
  long long hi = (long long)acc[3][reg] + ((long long)acc[4][reg] << 8) + ((long long)acc[5][reg] << 16) +
                 ((long long)acc[6][reg] << 24);
  return ((double)hi * 16777216.0 + (double)lo) * scale;
}

template <int MT, int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_base_logistic_i8(BaseLRLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t vlds[];
  constexpr int CHUNK_BYTES = NT * LIMBS * 1024;  // digit planes of one 64-SNP chunk
  constexpr int THREADS = WAVES * 64;
  constexpr int VPT = (CHUNK_BYTES / 16 + THREADS - 1) / THREADS;  // 16-byte pieces per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int A = L.A, W = L.W, R = L.d.R;

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
  const int64_t n0 = ((int64_t)htile * WAVES + wave) * (MT * 16);

  const int8_t* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int64_t n = n0 + mt * 16 + i16;
    if (n > L.N - 1) n = L.N - 1;
    xrow[mt] = L.X + n * L.ldx + 16 * kq;
  }

  v4i acc[MT][NT][LIMBS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) acc[mt][nt][l] = v4i{0, 0, 0, 0};

  // ---- pipeline prologue ----
  uint4 vst[VPT];
  auto v_load = [&](int c) {
    const uint4* src = reinterpret_cast<const uint4*>(L.d.V8 + (size_t)c * CHUNK_BYTES);
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int e = v * THREADS + tid;
      if (e < CHUNK_BYTES / 16) vst[v] = src[e];
    }
  };
  auto v_store = [&](int buf) {
    uint4* dst = reinterpret_cast<uint4*>(vlds + (size_t)buf * CHUNK_BYTES);
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int e = v * THREADS + tid;
      if (e < CHUNK_BYTES / 16) dst[e] = vst[v];
    }
  };
  v4i x0[MT], x1[MT];  // X of chunk c (x0) and c+1 (x1); c+2 is issued while c computes
  {
    const int j0 = L.d.chunk_j0[c_begin];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) x0[mt] = load_x16(xrow[mt] + j0, L.x_end);
    if (c_begin + 1 < c_end) {
      const int j1 = L.d.chunk_j0[c_begin + 1];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) x1[mt] = load_x16(xrow[mt] + j1, L.x_end);
    }
  }
  v_load(c_begin);
  v_store(0);
  __syncthreads();

  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    v4i x2[MT];
    if (c + 2 < c_end) {
      const int j2 = L.d.chunk_j0[c + 2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) x2[mt] = load_x16(xrow[mt] + j2, L.x_end);
    }
    if (c + 1 < c_end) v_load(c + 1);

    const v4i* vb = reinterpret_cast<const v4i*>(vlds + (size_t)buf * CHUNK_BYTES) + lane;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int l = 0; l < LIMBS; ++l) {
        const v4i b = vb[(nt * LIMBS + l) * 64];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt][nt][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(x0[mt], b, acc[mt][nt][l], 0, 0, 0);
      }

    // ---- piece end: windows that finished here ----
    const int nfl = L.d.chunk_nflush[c];
    if (nfl > 0) {
      const int w0 = L.d.chunk_flush0[c];
      for (int w = w0; w < w0 + nfl; ++w) {
        const int cbase = (w % R) * A;
        const bool emit = (w >= wa) && (w < wb);
        const double scale = L.d.wscale[w];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          double p[NT][4];
          double sum[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 16 + i16 - cbase;
            const bool mine = (col >= 0) && (col < A);
            const double ic = mine ? L.d.icpt[w * A + col] : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double z = combine(acc[mt][nt], r, scale) + ic;
              p[nt][r] = (mine && emit) ? 1.0 / (1.0 + exp(-z)) : 0.0;
              sum[r] += p[nt][r];
            }
#pragma unroll
            for (int l = 0; l < LIMBS; ++l)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[mt][nt][l][r] = mine ? 0 : acc[mt][nt][l][r];
          }
          if (emit) {
            // normaliser: sum over the A class lanes of this 16-lane row group (other lanes contribute 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              double s = sum[r];
              s += __shfl_xor(s, 1, 64);
              s += __shfl_xor(s, 2, 64);
              s += __shfl_xor(s, 4, 64);
              s += __shfl_xor(s, 8, 64);
              sum[r] = s;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int col = nt * 16 + i16 - cbase;
              if (col >= 0 && col < A) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  // int32 16x16 C/D layout: column = lane&15, row = 4*(lane>>4) + reg
                  const int64_t n = n0 + mt * 16 + 4 * kq + r;
                  if (n < L.N) {
                    const double v = p[nt][r] / sum[r];
                    const size_t o = ((size_t)n * W + w) * A + col;
                    if (L.b64) L.b64[o] = v;
                    if (L.b32) L.b32[o] = (float)v;
                  }
                }
              }
            }
          }
        }
      }
    }

    if (c + 1 < c_end) v_store(buf ^ 1);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { x0[mt] = x1[mt]; x1[mt] = x2[mt]; }
  }
}

template <int MT, int NT, int WAVES>
hipError_t launch(const BaseLRLaunch& L, int n_cu, hipStream_t s) {
  BaseLRLaunch P = L;
  const int haps_per_block = WAVES * MT * 16;
  const int64_t gx = (L.N + haps_per_block - 1) / haps_per_block;
  // window ranges: a multiple of 8 (one XCD each), ~2-3 blocks per CU in total
  int64_t want = (3LL * n_cu + gx - 1) / gx;
  want = std::max<int64_t>(8, ((want + 7) / 8) * 8);
  int wch = (int)((L.W + want - 1) / want);
  if (wch < 4) wch = 4;
  const int n_ranges = (L.W + wch - 1) / wch;
  const int n_ranges8 = ((n_ranges + 7) / 8) * 8;
  P.wch = wch;
  P.n_htiles = (int)gx;
  const size_t lds = (size_t)2 * NT * LIMBS * 1024;
  hipLaunchKernelGGL((k_base_logistic_i8<MT, NT, WAVES>), dim3((unsigned)(gx * n_ranges8)), dim3(WAVES * 64), lds, s, P);
  return hipGetLastError();
}

}  // namespace

hipError_t gnx_launch_base_logistic_i8(const BaseLRLaunch& L, int n_cu, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  const bool small = L.N <= 64 * 8;
  switch (L.d.NT) {
    case 1: return small ? launch<1, 1, 4>(L, n_cu, s) : launch<4, 1, 4>(L, n_cu, s);
    case 2: return small ? launch<1, 2, 4>(L, n_cu, s) : launch<2, 2, 4>(L, n_cu, s);
    case 3: return launch<1, 3, 4>(L, n_cu, s);
    case 4: return launch<1, 4, 4>(L, n_cu, s);
    default: return hipErrorInvalidValue;
  }
}

// k_base_logistic.hip — per-window logistic base classifiers on gfx950 (CDNA4).
//
// Replaces Base.predict_proba_vectorized + LogisticRegressionBase (reference src/Base/base.py:146-180,
// src/Base/models.py:12-21 -> sklearn _predict_proba_lr):  B[n,w,:] = normalise(expit(Xw . coef_w^T + b_w)).
//
// Design:
//  * X (N, C) int8 is streamed ONCE from HBM, 16 bytes per lane straight into VGPRs (no LDS: every
//    X byte feeds exactly one MFMA A-operand, nothing to share between waves).
//  * With context ctx every SNP belongs to R = ceil((M+2ctx)/M) windows (2 for the default 0.5), so
//    the R*A class scores that a SNP contributes to are ONE 16-wide MFMA column tile:
//    column = (window mod R)*A + class.  The per-SNP weight rows V[snp][16] are precomputed in
//    MFMA-fragment order at model load (reflect padding folded into the edge windows' weights).
//  * accumulation is float64 on v_mfma_f64_16x16x4_f64 — float32(B) feeds tree thresholds downstream,
//    f32/bf16 accumulation flips 78 % of those values by an ulp, f64 flips none.
//  * the chromosome is cut into pieces that end exactly where a window ends; after the last chunk
//    of a piece the finished window's column slot is flushed through the fused epilogue
//    (+intercept, expit, normalise, cast) and zeroed for window w+R.
//  * one wave = MT*16 haplotypes x all columns; a block = 4 waves on the same window range
//    (weights are then L1/L2 hits for 3 of the 4 waves); grid = hap tiles x window ranges.
#include "gnx_internal.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) xbytes16 { uint32_t v[4]; };

constexpr int WAVES = 4;

__device__ __forceinline__ xbytes16 load_x16(const int8_t* p) {
  // unconditional (unaligned) global_load_dwordx4; the last row is served from a zero-padded copy
  xbytes16 r;
  __builtin_memcpy(&r, p, 16);
  return r;
}

__device__ __forceinline__ double x_at(const xbytes16& x, int t) {
  // sign-extended byte t -> f64 (v_bfe_i32 + v_cvt_f64_i32)
  return (double)(int32_t)(int8_t)(x.v[t >> 2] >> (8 * (t & 3)));
}

template <int MT, int NT>
__global__ __launch_bounds__(WAVES * 64) void k_base_logistic(BaseLRLaunch L) {
  extern __shared__ __attribute__((aligned(16))) double zbuf_all[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int A = L.A, W = L.W, R = L.d.R;
  double* zb = zbuf_all + (size_t)wave * (MT * 16) * A;

  const int64_t n0 = ((int64_t)blockIdx.x * WAVES + wave) * (MT * 16);
  const int wa = blockIdx.y * L.wch;
  const int wb = min(W, wa + L.wch);
  const int c_begin = L.d.win_chunk0[wa];
  const int c_end = L.d.win_chunk1[wb - 1];

  const int8_t* xrow[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int64_t n = n0 + mt * 16 + i16;  // rows >= N-1 read the padded copy of the last row
    xrow[mt] = (n >= L.N - 1 ? L.last_row : L.X + n * L.ldx) + 16 * kq;
  }

  d4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = d4{0.0, 0.0, 0.0, 0.0};

  // software pipeline: X of chunk c+1 is in flight while chunk c is on the matrix pipe
  xbytes16 xn[MT];
  {
    const int j0 = L.d.chunk_j0[c_begin];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xn[mt] = load_x16(xrow[mt] + j0);
  }

  for (int c = c_begin; c < c_end; ++c) {
    xbytes16 x[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) x[mt] = xn[mt];
    if (c + 1 < c_end) {
      const int j0n = L.d.chunk_j0[c + 1];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xn[mt] = load_x16(xrow[mt] + j0n);
    }
    const double* vp = L.d.V + ((size_t)c * 16 * NT) * 64 + lane;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      double a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = x_at(x[mt], t);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const double b = vp[(size_t)(t * NT + nt) * 64];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt], b, acc[mt][nt], 0, 0, 0);
      }
    }

    // ---- piece end: flush the windows that finished here (block-uniform control flow) ----
    const int nfl = L.d.chunk_nflush[c];
    if (nfl > 0) {
      const int w0 = L.d.chunk_flush0[c];
      for (int w = w0; w < w0 + nfl; ++w) {
        const int cbase = (w % R) * A;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 16 + i16 - cbase;
            const bool mine = (col >= 0) && (col < A);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              // f64 16x16x4 C/D layout: column = lane&15, row = (lane>>4) + 4*reg
              if (mine) zb[(mt * 16 + kq + 4 * r) * A + col] = acc[mt][nt][r];
              acc[mt][nt][r] = mine ? 0.0 : acc[mt][nt][r];
            }
          }
        __syncthreads();
        if (w >= wa && w < wb && lane < MT * 16) {
          const int64_t n = n0 + lane;
          double* z = zb + lane * A;
          double sum = 0.0;
          for (int a = 0; a < A; ++a) {
            const double p = 1.0 / (1.0 + exp(-(z[a] + L.d.icpt[w * A + a])));
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
        __syncthreads();
      }
    }
  }
}

template <int MT, int NT>
hipError_t launch(const BaseLRLaunch& L, int n_cu, hipStream_t s) {
  BaseLRLaunch P = L;
  const int haps_per_block = WAVES * MT * 16;
  const int64_t gx = (L.N + haps_per_block - 1) / haps_per_block;
  // enough window ranges to give every CU ~2 blocks; each extra range re-reads (R-1) pieces
  int64_t want_y = (2LL * n_cu + gx - 1) / gx;
  if (want_y < 1) want_y = 1;
  int wch = (int)((L.W + want_y - 1) / want_y);
  if (wch < 8) wch = L.W < 8 ? L.W : 8;
  P.wch = wch;
  const int gy = (L.W + wch - 1) / wch;
  const size_t lds = (size_t)WAVES * MT * 16 * L.A * sizeof(double);
  hipLaunchKernelGGL((k_base_logistic<MT, NT>), dim3((unsigned)gx, (unsigned)gy), dim3(WAVES * 64), lds, s, P);
  return hipGetLastError();
}

}  // namespace

hipError_t gnx_launch_base_logistic(const BaseLRLaunch& L, int n_cu, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  const bool small = L.N <= 64 * 8;  // few haplotypes (e.g. one individual in the Gnofix second pass)
  switch (L.d.NT) {
    case 1: return small ? launch<1, 1>(L, n_cu, s) : launch<4, 1>(L, n_cu, s);
    case 2: return small ? launch<1, 2>(L, n_cu, s) : launch<4, 2>(L, n_cu, s);
    case 3: return small ? launch<1, 3>(L, n_cu, s) : launch<2, 3>(L, n_cu, s);
    case 4: return small ? launch<1, 4>(L, n_cu, s) : launch<2, 4>(L, n_cu, s);
    default: return hipErrorInvalidValue;
  }
}

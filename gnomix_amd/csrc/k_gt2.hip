// k_gt2.hip — variant-major 2-bit genotypes (the order VCF text arrives in, include/gnomix_io.h) <-> the haplotype-major
// int8 matrix every base kernel reads.
//
// Reference: vcf_to_npy (src/utils.py:104-159) builds X with numpy on the host: reshape + transpose of calldata/GT,
// `fill[:, fmt_idx] = data[:, vcf_idx]` (SNP intersection), `(mat - 1) * -1` on the columns whose REF differs, everything
// that is not 0 / 1 -> 2, astype(int8) — five passes over an (N, C) int64 matrix.  Here the parsed 2-bit rows cross PCIe
// (a quarter of int8, a 32nd of the int64 intermediate) and ONE kernel does the transpose, the column map, the REF flip and
// the missing rule: HBM traffic = G once + X once.
//
// k_gt2_to_x: a block owns 64 model SNPs x 1024 haplotypes.  Load: the 64 source rows' 256-byte pieces (16 lanes cover one
// piece: whole cache lines), flip / missing applied to the packed words (a 32-bit word = 16 haplotypes: lo' = (lo ^ flip) & ~hi
// keeps 2 and 3 -> 2), into LDS [snp][65 words].  Write: thread (g = SNP group of 16, d = word of 16 haplotypes) reads its 16
// words (bank = 16 g + d: conflict-free for 4 groups x 16 words per wave) and emits, per haplotype, the 16 bytes of X — four
// lanes per haplotype row cover 64 contiguous bytes.
#include "gnx_internal.h"

namespace {

constexpr int T2X_SNPS = 64, T2X_WORDS = 64, T2X_LD = 65;

__device__ __forceinline__ uint32_t gt2_fix(uint32_t w, uint32_t flip) {  // 16 fields: 3 -> 2, flip 0 <-> 1
  const uint32_t hi = (w >> 1) & 0x55555555u;
  const uint32_t lo = ((w & 0x55555555u) ^ (flip ? 0x55555555u : 0u)) & ~hi;
  return lo | (hi << 1);
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_gt2_to_x(const uint8_t* __restrict__ G, int64_t V, int64_t ldg, int64_t n0, int64_t N,
                                                  const int32_t* __restrict__ src, int64_t C, int8_t* __restrict__ X, int64_t ldx,
                                                  int64_t n_ctiles) {
  __shared__ uint32_t tile[T2X_SNPS * T2X_LD];
  const int t = threadIdx.x;
  const int64_t ct = blockIdx.x % n_ctiles, ht = blockIdx.x / n_ctiles;
  const int64_t c0 = ct * T2X_SNPS;
  const int64_t h0 = ht * (T2X_WORDS * 16);             // first haplotype of the tile, relative to n0
  const int64_t byte0 = (n0 + h0) >> 2;                 // n0 % 4 == 0 (launcher)
  // ---- load: 64 rows x 256 bytes, 16 bytes per thread and trip ---------------------------------------------------------
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = it * 256 + t;
    const int row = i >> 4, piece = i & 15;
    const int64_t c = c0 + row;
    const int32_t sv = c < C ? src[c] : -1;
    uint32_t w[4] = {0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu};
    if (sv >= 0) {
      const int64_t v = sv & 0x3FFFFFFF;
      const uint32_t flip = ((uint32_t)sv >> 30) & 1u;
      const int64_t off = byte0 + 16 * piece;
      const uint8_t* p = G + v * ldg + off;
      if (ALIGNED && off + 16 <= ldg) {
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t x = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int64_t o = off + 4 * k + b;
            x |= (o < ldg ? (uint32_t)G[v * ldg + o] : 0u) << (8 * b);
          }
          w[k] = x;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = gt2_fix(w[k], flip);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[row * T2X_LD + piece * 4 + k] = w[k];
  }
  __syncthreads();
  // ---- write: thread = (SNP group g, haplotype word d) ---------------------------------------------------------------------
  const int g = (t >> 4) & 3, d = (t & 15) | ((t >> 6) << 4);
  uint32_t w[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) w[k] = tile[(16 * g + k) * T2X_LD + d];
  const int64_t cg = c0 + 16 * g;
  if (cg >= C) return;
  const int nvalid = (int)((C - cg < 16) ? C - cg : 16);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t h = h0 + 16 * d + j;
    if (h >= N) break;
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      o[q] = ((w[4 * q] >> (2 * j)) & 3u) | (((w[4 * q + 1] >> (2 * j)) & 3u) << 8) | (((w[4 * q + 2] >> (2 * j)) & 3u) << 16) |
             (((w[4 * q + 3] >> (2 * j)) & 3u) << 24);
    int8_t* dst = X + h * ldx + cg;
    if (ALIGNED && nvalid == 16) {
      *reinterpret_cast<uint4*>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
    } else {
      for (int k = 0; k < nvalid; ++k) dst[k] = (int8_t)((o[k >> 2] >> (8 * (k & 3))) & 0xFFu);
    }
  }
}

// k_gt2_to_p2: the same tile and load phase, but the haplotype rows leave as 2-bit fields too (the gnx_pack_x layout that
// k_base_logistic_p2 reads: SNP j of a row = bits 2 (j % 4).. of byte j / 4): per thread a 16 x 16 transpose of 2-bit fields, one
// 32-bit word = 16 SNPs of one haplotype per store, four lanes per haplotype row cover 16 contiguous bytes.  HBM traffic = G once
// + a quarter of what k_gt2_to_x writes.
__global__ __launch_bounds__(256) void k_gt2_to_p2(const uint8_t* __restrict__ G, int64_t V, int64_t ldg, int64_t n0, int64_t N,
                                                   const int32_t* __restrict__ src, int64_t C, uint8_t* __restrict__ P, int64_t ldp,
                                                   int64_t n_ctiles, int aligned) {
  __shared__ uint32_t tile[T2X_SNPS * T2X_LD];
  const int t = threadIdx.x;
  const int64_t ct = blockIdx.x % n_ctiles, ht = blockIdx.x / n_ctiles;
  const int64_t c0 = ct * T2X_SNPS;
  const int64_t h0 = ht * (T2X_WORDS * 16);
  const int64_t byte0 = (n0 + h0) >> 2;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int i = it * 256 + t;
    const int row = i >> 4, piece = i & 15;
    const int64_t c = c0 + row;
    const int32_t sv = c < C ? src[c] : -1;
    uint32_t w[4] = {0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu};  // a model SNP the query lacks: missing (2) everywhere
    if (c >= C) { w[0] = w[1] = w[2] = w[3] = 0u; }                          // past the last SNP: zero fields (canonical row tail)
    if (sv >= 0) {
      const int64_t v = sv & 0x3FFFFFFF;
      const uint32_t flip = ((uint32_t)sv >> 30) & 1u;
      const int64_t off = byte0 + 16 * piece;
      const uint8_t* p = G + v * ldg + off;
      if (aligned && off + 16 <= ldg) {
        const uint4 q = *reinterpret_cast<const uint4*>(p);
        w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t x = 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int64_t o = off + 4 * k + b;
            x |= (o < ldg ? (uint32_t)G[v * ldg + o] : 0u) << (8 * b);
          }
          w[k] = x;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = gt2_fix(w[k], flip);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[row * T2X_LD + piece * 4 + k] = w[k];
  }
  __syncthreads();
  const int g = (t >> 4) & 3, d = (t & 15) | ((t >> 6) << 4);
  uint32_t w[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) w[k] = tile[(16 * g + k) * T2X_LD + d];
  const int64_t cg = c0 + 16 * g;
  if (cg >= C) return;
  const int64_t bo = cg >> 2;  // byte of the row that holds SNP cg (cg % 16 == 0)
  const bool whole = bo + 4 <= ldp;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t h = h0 + 16 * d + j;
    if (h >= N) break;
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) o |= ((w[k] >> (2 * j)) & 3u) << (2 * k);
    uint8_t* dst = P + h * ldp + bo;
    if (whole && aligned) *reinterpret_cast<uint32_t*>(dst) = o;
    else
      for (int b = 0; b < 4 && bo + b < ldp && cg + 4 * b < C; ++b) dst[b] = (uint8_t)(o >> (8 * b));
  }
}

// the way back (phased VCF): one thread = one output word = 16 haplotypes of one emitted variant
__global__ __launch_bounds__(256) void k_x_to_gt2(const int8_t* __restrict__ X, int64_t N, int64_t ldx, int64_t n0,
                                                  const int32_t* __restrict__ cols, int64_t V, uint8_t* __restrict__ G, int64_t ldg,
                                                  int64_t words) {
  const int64_t total = V * words;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / words, d = e - r * words;
    const int64_t c = cols[r];
    uint32_t w = 0;
    const int64_t hb = 16 * d;
    const int nh = (int)((N - hb < 16) ? N - hb : 16);
    for (int j = 0; j < nh; ++j) w |= ((uint32_t)X[(hb + j) * ldx + c] & 3u) << (2 * j);
    uint8_t* dst = G + r * ldg + ((n0 + hb) >> 2);
    const int nb = (nh + 3) >> 2;
    for (int b = 0; b < nb; ++b) dst[b] = (uint8_t)(w >> (8 * b));
  }
}

// the same from 2-bit haplotype rows (gnx_pack_x layout)
__global__ __launch_bounds__(256) void k_p2_to_gt2(const uint8_t* __restrict__ P, int64_t N, int64_t ldp, int64_t n0,
                                                   const int32_t* __restrict__ cols, int64_t V, uint8_t* __restrict__ G, int64_t ldg,
                                                   int64_t words) {
  const int64_t total = V * words;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / words, d = e - r * words;
    const int64_t c = cols[r];
    const int64_t cb = c >> 2;
    const int sh = 2 * (int)(c & 3);
    uint32_t w = 0;
    const int64_t hb = 16 * d;
    const int nh = (int)((N - hb < 16) ? N - hb : 16);
    for (int j = 0; j < nh; ++j) w |= (((uint32_t)P[(hb + j) * ldp + cb] >> sh) & 3u) << (2 * j);
    uint8_t* dst = G + r * ldg + ((n0 + hb) >> 2);
    const int nb = (nh + 3) >> 2;
    for (int b = 0; b < nb; ++b) dst[b] = (uint8_t)(w >> (8 * b));
  }
}

}  // namespace

hipError_t gnx_launch_p2_to_gt2(const uint8_t* P, int64_t N, int64_t ldp, int64_t n0, const int32_t* cols, int64_t V, uint8_t* G,
                                int64_t ldg, hipStream_t s) {
  if (N <= 0 || V <= 0) return hipSuccess;
  const int64_t words = (N + 15) / 16;
  const int64_t total = V * words;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)256 * 64);
  hipLaunchKernelGGL(k_p2_to_gt2, dim3(blocks), dim3(256), 0, s, P, N, ldp, n0, cols, V, G, ldg, words);
  return hipGetLastError();
}

hipError_t gnx_launch_gt2_to_x(const uint8_t* G, int64_t V, int64_t ldg, int64_t n0, int64_t N, const int32_t* src, int64_t C,
                               int8_t* X, int64_t ldx, hipStream_t s) {
  if (N <= 0 || C <= 0) return hipSuccess;
  const int64_t n_ctiles = (C + T2X_SNPS - 1) / T2X_SNPS, n_htiles = (N + T2X_WORDS * 16 - 1) / (T2X_WORDS * 16);
  const bool aligned = ((uintptr_t)G % 4 == 0) && (ldg % 4 == 0) && (n0 % 16 == 0) && ((uintptr_t)X % 16 == 0) && (ldx % 16 == 0);
  const dim3 grid((unsigned)(n_ctiles * n_htiles));
  if (aligned) hipLaunchKernelGGL(k_gt2_to_x<true>, grid, dim3(256), 0, s, G, V, ldg, n0, N, src, C, X, ldx, n_ctiles);
  else hipLaunchKernelGGL(k_gt2_to_x<false>, grid, dim3(256), 0, s, G, V, ldg, n0, N, src, C, X, ldx, n_ctiles);
  return hipGetLastError();
}

hipError_t gnx_launch_gt2_to_p2(const uint8_t* G, int64_t V, int64_t ldg, int64_t n0, int64_t N, const int32_t* src, int64_t C,
                                uint8_t* P, int64_t ldp, hipStream_t s) {
  if (N <= 0 || C <= 0) return hipSuccess;
  const int64_t n_ctiles = (C + T2X_SNPS - 1) / T2X_SNPS, n_htiles = (N + T2X_WORDS * 16 - 1) / (T2X_WORDS * 16);
  const int aligned = ((uintptr_t)G % 4 == 0) && (ldg % 4 == 0) && (n0 % 16 == 0) && ((uintptr_t)P % 4 == 0) && (ldp % 4 == 0);
  hipLaunchKernelGGL(k_gt2_to_p2, dim3((unsigned)(n_ctiles * n_htiles)), dim3(256), 0, s, G, V, ldg, n0, N, src, C, P, ldp, n_ctiles, aligned);
  return hipGetLastError();
}

hipError_t gnx_launch_x_to_gt2(const int8_t* X, int64_t N, int64_t ldx, int64_t n0, const int32_t* cols, int64_t V, uint8_t* G,
                               int64_t ldg, hipStream_t s) {
  if (N <= 0 || V <= 0) return hipSuccess;
  const int64_t words = (N + 15) / 16;
  const int64_t total = V * words;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)256 * 64);
  hipLaunchKernelGGL(k_x_to_gt2, dim3(blocks), dim3(256), 0, s, X, N, ldx, n0, cols, V, G, ldg, words);
  return hipGetLastError();
}

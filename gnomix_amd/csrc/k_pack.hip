// k_pack.hip — 2-bit packed haplotypes (4 SNPs per byte) -> the int8 matrix every base kernel reads.
//
// The reference hands X over as int8 {0,1,2} (src/utils.py:153), 1 byte per SNP.  Whole genome that is 17.7 MB per
// haplotype: over PCIe (63 GB/s spec, ~46 GB/s measured) the host link, not the GPU, bounds the path at ~3.5 k
// haplotypes/s/GPU.  X only has three symbols, so the host-pointer entry point gnx_infer_packed takes it
// as 2-bit fields (SNP j of a row = bits 2*(j%4) .. 2*(j%4)+1 of byte j/4: a quarter of the bytes on the link) and this
// pass widens it back in HBM, where the bandwidth is two orders of magnitude higher.  One thread = 16 SNPs: a 4-byte
// load, 16 fields spread to bytes with a multiply (t * 0x41 puts the two fields of a nibble 8 bits apart), a 16-byte store.
#include "gnx_internal.h"

namespace {

__device__ __forceinline__ uint32_t spread4(uint32_t x) {  // 4 fields of one byte -> 4 bytes
  const uint32_t lo = x & 0xFu, hi = (x >> 4) & 0xFu;
  return ((lo * 0x41u) & 0x0303u) | (((hi * 0x41u) & 0x0303u) << 16);
}

template <bool FAST>
__global__ __launch_bounds__(256) void k_unpack2(const uint8_t* __restrict__ P, int64_t N, int64_t ldp, int64_t C,
                                                 int8_t* __restrict__ X, int64_t ldx, int64_t G) {
  const int64_t total = N * G;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = e / G, g = e - n * G;
    const uint8_t* src = P + n * ldp + 4 * g;
    const int64_t j0 = 16 * g;
    const int nsnp = (int)((C - j0 < 16) ? C - j0 : 16);
    uint32_t d = 0;
    if (FAST && nsnp == 16) d = *reinterpret_cast<const uint32_t*>(src);
    else {
      const int nb = (nsnp + 3) >> 2;  // never read past the row's last packed byte
      for (int b = 0; b < nb; ++b) d |= (uint32_t)src[b] << (8 * b);
    }
    uint4 o;
    o.x = spread4(d & 0xFFu);
    o.y = spread4((d >> 8) & 0xFFu);
    o.z = spread4((d >> 16) & 0xFFu);
    o.w = spread4(d >> 24);
    int8_t* dst = X + n * ldx + j0;
    if (FAST && nsnp == 16) *reinterpret_cast<uint4*>(dst) = o;
    else {
      const uint32_t w[4] = {o.x, o.y, o.z, o.w};
      for (int k = 0; k < nsnp; ++k) dst[k] = (int8_t)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
    }
  }
}

}  // namespace

hipError_t gnx_launch_unpack2(const uint8_t* P, int64_t N, int64_t ldp, int64_t C, int8_t* X, int64_t ldx, hipStream_t s) {
  if (N <= 0 || C <= 0) return hipSuccess;
  const int64_t G = (C + 15) / 16;
  const int64_t total = N * G;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)256 * 64);
  const bool fast = ((uintptr_t)P % 4 == 0) && (ldp % 4 == 0) && ((uintptr_t)X % 16 == 0) && (ldx % 16 == 0);
  if (fast) hipLaunchKernelGGL(k_unpack2<true>, dim3(blocks), dim3(256), 0, s, P, N, ldp, C, X, ldx, G);
  else hipLaunchKernelGGL(k_unpack2<false>, dim3(blocks), dim3(256), 0, s, P, N, ldp, C, X, ldx, G);
  return hipGetLastError();
}

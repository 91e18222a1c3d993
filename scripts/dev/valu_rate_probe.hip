#include <hip/hip_runtime.h>
#include <cstdio>
template <int K>
__global__ void k(uint32_t* o, uint32_t s128) {
  uint32_t a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 9;
  for (int i = 0; i < 4096; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (K == 0) { asm volatile("v_mad_u32_u16 %0, %0, %2, %1\n v_mad_u32_u16 %1, %1, %2, %0\n v_mad_u32_u16 %3, %3, %2, %4\n v_mad_u32_u16 %4, %4, %2, %3" : "+v"(a), "+v"(b) : "s"(s128), "v"(c), "v"(d)); }
      if (K == 1) { asm volatile("v_lshl_add_u32 %0, %0, 7, %1\n v_lshl_add_u32 %1, %1, 7, %0\n v_lshl_add_u32 %3, %3, 7, %4\n v_lshl_add_u32 %4, %4, 7, %3" : "+v"(a), "+v"(b) : "s"(s128), "v"(c), "v"(d)); }
      if (K == 2) { asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_add_u32_sdwa %4, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(a), "+v"(b) : "s"(s128), "v"(c), "v"(d)); }
      if (K == 3) { asm volatile("v_cmp_le_u32_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:DWORD\n v_cndmask_b32_sdwa %0, %1, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_cmp_le_u32_sdwa vcc, %3, %4 src0_sel:WORD_1 src1_sel:DWORD\n v_cndmask_b32_sdwa %3, %4, %4, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(a), "+v"(b) : "s"(s128), "v"(c), "v"(d) : "vcc"); }
      if (K == 4) { asm volatile("v_mad_u32_u24 %0, %0, %2, %1\n v_mad_u32_u24 %1, %1, %2, %0\n v_mad_u32_u24 %3, %3, %2, %4\n v_mad_u32_u24 %4, %4, %2, %3" : "+v"(a), "+v"(b) : "s"(s128), "v"(c), "v"(d)); }
    }
  }
  o[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}
template <int K> void run(const char* name, uint32_t* o) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<K><<<256 * 8, 256>>>(o, 128); hipDeviceSynchronize();
  hipEventRecord(e0); k<K><<<256 * 8, 256>>>(o, 128); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 256*8 blocks * 4 waves / (256 CU * 4 SIMD) = 8 waves per SIMD, each 4096*8*4 instr
  double instr_per_simd = 8.0 * 4096 * 8 * 4;
  printf("%s: %.3f ms -> %.2f cycles per wave-instruction (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() { uint32_t* o; hipMalloc(&o, 256 * 8 * 256 * 4); run<0>("v_mad_u32_u16", o); run<1>("v_lshl_add_u32", o); run<2>("v_add_u32_sdwa", o); run<3>("cmp_sdwa+cndmask_sdwa", o); run<4>("v_mad_u32_u24", o); return 0; }

// LDS instruction throughput per CU on gfx950: ds_read_u16 / b32 / b64 / b128, conflict-free addresses, 8 independent reads in
// flight per wave; blocks of 512 threads, `bpc` blocks per CU (occupancy via the dynamic LDS size).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)
template <int K>
__global__ __launch_bounds__(512) void k(uint32_t* o, int iters) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i * 16;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = (K == 0 ? (lane & 31) * 4 + (lane >> 5) * 2 : K == 1 ? lane * 4 : K == 2 ? lane * 8 : lane * 16) + u * 1024;
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      uint32_t v;
      if (K == 0) asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(a[u]));
      if (K == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a[u]));
      if (K == 2) { uint2 w; asm volatile("ds_read_b64 %0, %1" : "=v"(w) : "v"(a[u])); v = w.x; }
      if (K == 3) { uint4 w; asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(a[u])); v = w.x; }
      asm volatile("s_waitcnt lgkmcnt(7)");
      acc += v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  o[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int K>
int run(const char* name, uint32_t* o, int bpc) {
  const int iters = 4096;
  const size_t lds = (size_t)160 * 1024 / bpc - 1024;
  CK(hipFuncSetAttribute((const void*)k<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<K><<<256 * bpc, 512, lds>>>(o, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k<K><<<256 * bpc, 512, lds>>>(o, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_cu = (double)bpc * 8 * iters * 8;
  printf("%-12s %2d waves/CU: %.3f ms -> %.2f cycles per wave-instruction per CU (2.4 GHz)\n", name, bpc * 8, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
  return 0;
}
int main() {
  uint32_t* o;
  CK(hipMalloc(&o, 256 * 4 * 512 * 4));
  for (int bpc : {1, 2, 4}) { run<0>("ds_read_u16", o, bpc); run<1>("ds_read_b32", o, bpc); run<2>("ds_read_b64", o, bpc); run<3>("ds_read_b128", o, bpc); }
  return 0;
}

// does global_load_lds_dwordx4 accept global addresses that are not 16- / 4-byte aligned on gfx950?
// build: hipcc --offload-arch=gfx950 -O2 -o scripts/dev/glds_align_probe.bin scripts/dev/glds_align_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void probe(const uint8_t* src, int mis, int stride, uint8_t* out) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[1024];
  const int lane = threadIdx.x;
  __builtin_amdgcn_global_load_lds((gptr_t)(src + mis + (size_t)lane * stride), (lptr_t)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int k = 0; k < 16; ++k) out[lane * 16 + k] = lds[lane * 16 + k];
}

int main() {
  std::vector<uint8_t> h(1 << 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)((i * 2654435761u) >> 13);
  uint8_t *d, *o;
  hipMalloc(&d, h.size()); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int stride : {16, 128, 133, 370500 % 1000 + 7}) {
    for (int mis = 0; mis < 16; ++mis) {
      hipMemset(o, 0xEE, 1024);
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mis, stride, o);
      std::vector<uint8_t> r(1024);
      hipError_t e = hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 16; ++k) bad += r[l * 16 + k] != h[mis + (size_t)l * stride + k];
      printf("stride %d mis %d: %s bad=%d\n", stride, mis, hipGetErrorString(e), bad);
      bad_total += bad;
    }
  }
  printf("TOTAL bad=%d\n", bad_total);
  return 0;
}

// Issue cost and dependent latency of the float64 instructions the CRF kernels are made of (gfx950): plain v_fma_f64, the DPP form
// (row_newbcast, the only one float64 has), v_mfma_f64_16x16x4_f64.  One workgroup per CU slot x waves per SIMD; clock64 around loops.
//   hipcc -O3 --offload-arch=gfx950 -o f64_rate_probe.bin f64_rate_probe.hip && ./f64_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int ITER = 2000;

template <int MODE, int NCHAIN>
__global__ void probe(double* out, long long* clk) {
  double x = threadIdx.x * 1e-3 + 1.0, c = 1.0000001;
  double acc[NCHAIN];
  v4d macc[NCHAIN];
#pragma unroll
  for (int k = 0; k < NCHAIN; ++k) { acc[k] = k; macc[k] = v4d{0.0, 0.0, 0.0, 0.0}; }
  const long long t0 = clock64();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int k = 0; k < NCHAIN; ++k) {
      if (MODE == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(x), "v"(c));
      if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[k]) : "v"(x), "v"(c));
      if (MODE == 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(macc[k]) : "v"(x), "v"(c));
      if (MODE == 3) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(((int*)&acc[k])[0]) : "v"(((int*)&x)[0]));
    }
  }
  const long long t1 = clock64();
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NCHAIN; ++k) s += acc[k] + macc[k][0] + macc[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE, int NCHAIN>
void run(const char* name, int waves_per_simd) {
  const int blocks = 256 * 4 * waves_per_simd;   // one wave per block
  double* out; long long* clk;
  hipMalloc(&out, (size_t)blocks * 64 * 8); hipMalloc(&clk, (size_t)blocks * 8);
  hipLaunchKernelGGL((probe<MODE, NCHAIN>), dim3(blocks), dim3(64), 0, 0, out, clk);
  hipLaunchKernelGGL((probe<MODE, NCHAIN>), dim3(blocks), dim3(64), 0, 0, out, clk);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), clk, (size_t)blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= blocks;
  printf("%-28s chains/wave %d  waves/SIMD %d : %7.1f clk per instruction per wave  (%.1f clk of the SIMD per instruction)\n", name, NCHAIN, waves_per_simd,
         mean / (ITER * NCHAIN), mean / (ITER * NCHAIN) / waves_per_simd);
  hipFree(out); hipFree(clk);
}

int main() {
  for (int w : {1, 4}) {
    run<0, 1>("v_fma_f64 dependent", w); run<0, 4>("v_fma_f64 4 chains", w);
    run<1, 1>("v_fmac_f64_dpp dependent", w); run<1, 4>("v_fmac_f64_dpp 4 chains", w);
    run<2, 1>("v_mfma_f64_16x16x4 dependent", w); run<2, 4>("v_mfma_f64_16x16x4 4 chains", w);
    run<3, 4>("v_mov_b32_dpp 4 chains", w);
  }
  return 0;
}

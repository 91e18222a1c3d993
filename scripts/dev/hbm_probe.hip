// hbm_probe.hip — stand-alone fetch probe behind DESIGN.md §5.2: how fast does MI355X deliver an int8 matrix X (N rows,
// ldx bytes apart) when a 512-thread block keeps ROWS rows in lockstep and asks for RUN contiguous bytes of every row per
// visit, DEPTH visits in flight, 16 bytes per lane, 8 lanes per 128 bytes — the access order of the logistic pass —
// versus a plain row-sequential stream.  No compute: loaded words are xor-folded into one dummy store per thread.
//   hipcc --offload-arch=gfx950 -O3 scripts/dev/hbm_probe.hip -o gpurun_out/hbm_probe && gpurun_out/hbm_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) xb16 { v4i v; };

// block = 512 threads; thread t: piece (t % PPR) of row (t / PPR) + q * (512 / PPR), PPR = RUN / 16 pieces per row visit
template <int ROWS, int RUN, int DEPTH>
__global__ __launch_bounds__(512) void k_lockstep(const int8_t* X, int64_t ldx, int64_t C, int n_ranges, int* sink) {
  constexpr int PPR = RUN / 16, RPP = 512 / PPR, Q = ROWS / RPP;
  static_assert(Q >= 1, "rows per block vs run length");
  const int tile = blockIdx.x / n_ranges, range = blockIdx.x % n_ranges;
  const int64_t span = (C / n_ranges) / RUN * RUN;
  const int64_t j0 = range * span;
  const int p = threadIdx.x % PPR, r0 = threadIdx.x / PPR;
  const int8_t* base[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) base[q] = X + ((int64_t)tile * ROWS + r0 + q * RPP) * ldx + j0 + 16 * p;
  v4i st[DEPTH][Q];
  v4i acc = {0, 0, 0, 0};
  const int n_vis = (int)(span / RUN);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int q = 0; q < Q; ++q) { xb16 t; __builtin_memcpy(&t, base[q] + (int64_t)(d < n_vis ? d : n_vis - 1) * RUN, 16); st[d][q] = t.v; }
  for (int v = 0; v < n_vis; v += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int q = 0; q < Q; ++q) acc ^= st[d][q];
      const int nv = v + d + DEPTH < n_vis ? v + d + DEPTH : n_vis - 1;  // clamped: loads stay unconditional
#pragma unroll
      for (int q = 0; q < Q; ++q) { xb16 t; __builtin_memcpy(&t, base[q] + (int64_t)nv * RUN, 16); st[d][q] = t.v; }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x7fffffff) sink[0] = 1;
}

// row-sequential stream: a block walks ONE row range contiguously, 512 threads x 16 B = 8 KB per visit
template <int DEPTH>
__global__ __launch_bounds__(512) void k_stream(const int8_t* X, int64_t total, int* sink) {
  const int64_t per = total / gridDim.x / 8192 * 8192;
  const int8_t* b = X + (int64_t)blockIdx.x * per + 16 * threadIdx.x;
  const int n_vis = (int)(per / 8192);
  v4i st[DEPTH];
  v4i acc = {0, 0, 0, 0};
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) st[d] = *reinterpret_cast<const v4i*>(b + (int64_t)(d < n_vis ? d : n_vis - 1) * 8192);
  for (int v = 0; v < n_vis; v += DEPTH)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc ^= st[d];
      const int nv = v + d + DEPTH < n_vis ? v + d + DEPTH : n_vis - 1;
      st[d] = *reinterpret_cast<const v4i*>(b + (int64_t)nv * 8192);
    }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x7fffffff) sink[0] = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <typename F>
static double time_ms(F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGetLastError());
  return ms / 5;
}

int main() {
  const int64_t N = 10240, C = 370500, ldx = C;
  int8_t* X; int* sink;
  CK(hipMalloc(&X, (size_t)N * ldx + 4096));
  CK(hipMemset(X, 1, (size_t)N * ldx + 4096));
  CK(hipMalloc(&sink, 4));
  const double gb = (double)N * C / 1e9;
  const int n_ranges = 24;
#define LOCK(ROWS, RUN, DEPTH)                                                                                   \
  {                                                                                                              \
    const int grid = (int)(N / ROWS) * n_ranges;                                                                 \
    for (int bpc : {0, 2, 1}) { /* blocks per CU capped through a dummy dynamic-LDS request (0 = uncapped) */    \
      const size_t lds = bpc == 0 ? 0 : (bpc == 2 ? 70 * 1024 : 100 * 1024);                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lockstep<ROWS, RUN, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      const double ms = time_ms([&] { hipLaunchKernelGGL((k_lockstep<ROWS, RUN, DEPTH>), dim3(grid), dim3(512), lds, 0, X, ldx, C, n_ranges, sink); }); \
      std::printf("lockstep rows/block=%3d run=%4d B depth=%d (in flight %3d KB/block) blocks/CU<=%d: %.3f ms  %.2f TB/s\n", ROWS, RUN, DEPTH, \
                  ROWS * RUN * DEPTH / 1024, bpc, ms, gb / ms);                                                   \
    }                                                                                                            \
  }
  LOCK(256, 128, 1) LOCK(256, 128, 2) LOCK(256, 128, 3) LOCK(256, 128, 4)
  LOCK(128, 128, 2) LOCK(128, 128, 4) LOCK(128, 128, 8)
  LOCK(64, 128, 4) LOCK(64, 128, 8)
  LOCK(256, 256, 2) LOCK(128, 512, 2) LOCK(32, 512, 8) LOCK(16, 2048, 4)
  {
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_stream<4>), dim3(2048), dim3(512), 0, 0, X, N * ldx, sink); });
    std::printf("row-sequential stream (2048 blocks x 8 KB visits, depth 4): %.3f ms  %.2f TB/s\n", ms, gb / ms);
    const double ms2 = time_ms([&] { hipLaunchKernelGGL((k_stream<8>), dim3(1024), dim3(512), 0, 0, X, N * ldx, sink); });
    std::printf("row-sequential stream (1024 blocks x 8 KB visits, depth 8): %.3f ms  %.2f TB/s\n", ms2, gb / ms2);
  }
  return 0;
}

// k_smooth_cnn.hip — the convolutional smoother of the reference's "large" mode on gfx950.
//
// Replaces CNN_Smoother.predict_proba (reference src/Smooth/models.py:35-42 -> src/Smooth/cnn.py:37-55, 166-171):
//   B (N, W, A) -> float32 tensor (N, A, W) -> nn.Conv1d(A, A, kernel_size=S, padding=(S-1)//2) -> Softmax over the A
//   output channels -> (N, W, A).
// The reference constructs the layer with padding_mode="reflection", a string torch never implemented: torch <= 1.4
// ignores it (any mode other than "circular" zero-pads), torch >= 1.5 refuses to construct the module.  Wherever the
// reference's CNN runs it therefore ZERO-pads, and so does this kernel.
//
// One wave = 64 consecutive windows of one haplotype, lane = window.  The padded strip of the haplotype's base probabilities
// sits in LDS as [window][class] float32 (what torch.tensor(B, dtype=float) holds): the only per-lane operand of a tap.  A tap's
// output weights are the same for every lane, so they come through the SCALAR cache straight into SGPRs (the loader stores the
// weights as [a_in][tap][a_out padded]: one s_load per tap, no LDS, no VGPR), and the multiply-adds are v_pk_fma_f32 — two output
// channels per instruction with the lane's probability splat over both halves.  Per tap: one ds_read_b32 and AP/2 packed FMAs
// (round 1: three LDS reads and AP scalar FMAs).  Accumulation is float32, taps in (a_in, s) order, each channel's chain unchanged;
// the result is compared with torch's own conv1d within the north star's 1e-5 (the summation order of the backend's GEMM is not defined).
#include "gnx_internal.h"

namespace {

constexpr int WS = 64;
typedef float f2 __attribute__((ext_vector_type(2)));

template <int AP>  // output channels padded to AP = gnx_cnn_ap(A)
__global__ __launch_bounds__(256) void k_smooth_cnn(SmoothCNNLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int A = L.A, W = L.W, S = L.S, pad = (S - 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  const int strip_w = WS + S - 1;
  float* strip = reinterpret_cast<float*>(lds);                  // [nwave][strip_w][A]
  const int64_t n = (int64_t)blockIdx.y * nwave + wave;
  const int w0 = blockIdx.x * WS;
  float* st = strip + (size_t)wave * strip_w * A;
  const int64_t nc = n < L.N ? n : L.N - 1;
  for (int e = lane; e < strip_w * A; e += 64) {
    const int q = e / A, a = e - q * A;
    const int w = w0 + q - pad;
    float v = 0.f;                                               // zero padding outside [0, W)
    if (w >= 0 && w < W) {
      const size_t idx = ((size_t)nc * W + w) * A + a;
      v = L.b_is_f64 ? (float)reinterpret_cast<const double*>(L.B)[idx] : reinterpret_cast<const float*>(L.B)[idx];
    }
    st[e] = v;
  }
  __syncthreads();
  const int w = w0 + lane;
  const float* __restrict__ wt = L.weight;
  f2 acc2[AP / 2];
#pragma unroll
  for (int q = 0; q < AP / 2; ++q) acc2[q] = f2{L.bias[2 * q], L.bias[2 * q + 1]};
  for (int ai = 0; ai < A; ++ai) {
    const float* xs = st + (size_t)lane * A + ai;
    const float* wr = wt + (size_t)ai * S * AP;                  // wave-uniform
#pragma unroll 5
    for (int s = 0; s < S; ++s) {
      const float x = xs[s * A];
      const f2 xx = f2{x, x};
#pragma unroll
      for (int q = 0; q < AP / 2; ++q) {
        const f2 w2 = f2{wr[s * AP + 2 * q], wr[s * AP + 2 * q + 1]};
        acc2[q] = __builtin_elementwise_fma(xx, w2, acc2[q]);     // outputs past A accumulate zeros and are never read
      }
    }
  }
  float acc[AP];
#pragma unroll
  for (int q = 0; q < AP / 2; ++q) { acc[2 * q] = acc2[q].x; acc[2 * q + 1] = acc2[q].y; }
  constexpr int AMAX = AP;
  if (n >= L.N || w >= W) return;
  float mx = acc[0];
#pragma unroll
  for (int y = 1; y < AMAX; ++y)
    if (y < A) mx = fmaxf(mx, acc[y]);
  float sum = 0.f;
#pragma unroll
  for (int y = 0; y < AMAX; ++y)
    if (y < A) { acc[y] = (float)exp((double)(acc[y] - mx)); sum += acc[y]; }
  const size_t o = ((size_t)n * W + w) * A;
  int best = 0;
  float bv = -1.f;
#pragma unroll
  for (int y = 0; y < AMAX; ++y)
    if (y < A) {
      const float p = acc[y] / sum;
      if (L.proba32) L.proba32[o + y] = p;
      if (L.proba64) L.proba64[o + y] = (double)p;
      if (p > bv) { bv = p; best = y; }
    }
  if (L.labels) L.labels[(size_t)n * W + w] = best;
}

}  // namespace

hipError_t gnx_launch_smooth_cnn(const SmoothCNNLaunch& L, hipStream_t s) {
  if (L.N <= 0) return hipSuccess;
  int nwave = 4;
  auto lds_of = [&](int nw) { return (size_t)nw * (WS + L.S - 1) * L.A * sizeof(float); };
  while (nwave > 1 && lds_of(nwave) > (size_t)64 * 1024) nwave >>= 1;
  const size_t lds = lds_of(nwave);
  if (lds > (size_t)160 * 1024) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((L.W + WS - 1) / WS), (unsigned)((L.N + nwave - 1) / nwave));
#define GNX_CNN_LAUNCH(AM)                                                                                              \
  {                                                                                                                      \
    GNX_LDS_OPTIN(lds, k_smooth_cnn<AM>);                                                                              \
    hipLaunchKernelGGL(k_smooth_cnn<AM>, grid, dim3(nwave * 64), lds, s, L);                                             \
  }
  if (L.A <= 8) GNX_CNN_LAUNCH(8) else if (L.A <= 16) GNX_CNN_LAUNCH(16) else GNX_CNN_LAUNCH(32)
#undef GNX_CNN_LAUNCH
  return hipGetLastError();
}

// k_smooth_cnn.hip — the convolutional smoother of the reference's "large" mode on gfx950.
//
// Replaces CNN_Smoother.predict_proba (reference src/Smooth/models.py:35-42 -> src/Smooth/cnn.py:37-55, 166-171):
//   B (N, W, A) -> float32 tensor (N, A, W) -> nn.Conv1d(A, A, kernel_size=S, padding=(S-1)//2) -> Softmax over the A
//   output channels -> (N, W, A).
// The reference constructs the layer with padding_mode="reflection", a string torch never implemented: torch <= 1.4
// ignores it (any mode other than "circular" zero-pads), torch >= 1.5 refuses to construct the module.  Wherever the
// reference's CNN runs it therefore ZERO-pads, and so does this kernel.
//
// One wave = 64 consecutive windows of one haplotype, lane = window; NWAVE haplotypes per block share the weights in LDS
// ([a_in][s][a_out] so that a lane reads the A_out weights of one tap as a contiguous, wave-uniform run).  The padded
// strip of the haplotype's base probabilities sits in LDS as [window][class] float32 (what torch.tensor(B, dtype=float)
// holds).  Accumulation is float32, taps in (a_in, s) order; the result is compared with torch's own conv1d within the
// north star's 1e-5 (the summation order of the backend's GEMM is not defined).
#include "gnx_internal.h"

namespace {

constexpr int WS = 64;

template <int AMAX>
__global__ __launch_bounds__(256) void k_smooth_cnn(SmoothCNNLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int A = L.A, W = L.W, S = L.S, pad = (S - 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
  const int strip_w = WS + S - 1;
  const bool wl = L.w_in_lds != 0;                               // weights that do not fit the LDS stay in global memory
  const int AP = (A + 3) & ~3;                                   // weight rows padded to whole float4s (one tap = AP/4 LDS reads)
  float* wgt = reinterpret_cast<float*>(lds);                   // [A_in][S][AP]
  float* strip = wgt + (wl ? (size_t)A * S * AP : 0);            // [nwave][strip_w][A]
  if (wl)
    for (int e = tid; e < A * S * AP; e += blockDim.x) {
      const int ai = e / (S * AP), r = e - ai * S * AP, s = r / AP, ao = r - s * AP;
      wgt[e] = ao < A ? L.weight[((size_t)ao * A + ai) * S + s] : 0.f;  // torch layout (out, in, k)
    }
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
  float acc[AMAX];
#pragma unroll
  for (int y = 0; y < AMAX; ++y) acc[y] = (y < A) ? L.bias[y] : 0.f;
  for (int ai = 0; ai < A; ++ai)
    for (int s = 0; s < S; ++s) {
      const float x = st[(size_t)(lane + s) * A + ai];
      if (wl) {
        const float4* wr = reinterpret_cast<const float4*>(wgt + ((size_t)ai * S + s) * AP);  // wave-uniform address
#pragma unroll
        for (int q = 0; q < AMAX / 4; ++q)
          if (4 * q < A) {
            const float4 w4 = wr[q];
            acc[4 * q + 0] = fmaf(x, w4.x, acc[4 * q + 0]);  // outputs past A accumulate zeros and are never read
            acc[4 * q + 1] = fmaf(x, w4.y, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(x, w4.z, acc[4 * q + 2]);
            acc[4 * q + 3] = fmaf(x, w4.w, acc[4 * q + 3]);
          }
      } else {
#pragma unroll
        for (int y = 0; y < AMAX; ++y)
          if (y < A) acc[y] = fmaf(x, L.weight[((size_t)y * A + ai) * S + s], acc[y]);
      }
    }
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

hipError_t gnx_launch_smooth_cnn(const SmoothCNNLaunch& L0, hipStream_t s) {
  if (L0.N <= 0) return hipSuccess;
  SmoothCNNLaunch L = L0;
  int nwave = 4;
  const int AP = (L.A + 3) & ~3;
  L.w_in_lds = ((size_t)L.A * L.S * AP * sizeof(float) <= (size_t)96 * 1024) ? 1 : 0;
  auto lds_of = [&](int nw) { return ((L.w_in_lds ? (size_t)L.A * L.S * AP : 0) + (size_t)nw * (WS + L.S - 1) * L.A) * sizeof(float); };
  while (nwave > 1 && lds_of(nwave) > (size_t)128 * 1024) nwave >>= 1;
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

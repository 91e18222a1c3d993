// gnx_rank.h — the rank quantisation shared by k_smooth_xgb_rk.hip and k_gnofix.hip (device code only).
//
// A tree of the smoother only ever asks "p < threshold" (xgboost: fvalue < split_condition -> left).  The model loader sorts the
// ensemble's distinct thresholds U[0..K) once (gnx_model_build.hip: build_xgb_rk); a base probability p is replaced by
// r(p) = #{k : U[k] <= p} (16 bits) and a node's threshold U[k] by k + 1, so that  p < U[k]  <=>  r(p) < k + 1: every walk takes
// the branch the float compare takes.  NaN -> 0xFFFF ("never less than a threshold").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ int slide_src(int j, int W, int pad) {
  // reflect padding of slide_window (src/Smooth/utils.py:14-17)
  if (j < pad) return pad - 1 - j;
  if (j < pad + W) return j - pad;
  return W - 1 - (j - pad - W);
}

// NV independent rank computations side by side: r = #{U[k] <= p}; NaN -> 0xFFFF ("never less than a threshold")
template <int NV>
__device__ __forceinline__ void ranks(const float* __restrict__ U, const uint32_t* __restrict__ lut, int K, int steps,
                                      const float* p, uint32_t* r) {
  int lo[NV], hi[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float sc = p[i] * 1024.0f;
    int b = (int)fminf(fmaxf(sc, 0.0f), 1023.0f);  // NaN -> 0 (fmaxf drops it), fixed up below
    const uint32_t e = lut[b];
    lo[i] = (int)(e & 0xffffu);
    hi[i] = (int)(e >> 16);
  }
  for (int s = 0; s < steps; ++s) {
    float u[NV];
    int mid[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      mid[i] = (lo[i] + hi[i]) >> 1;
      u[i] = U[min(mid[i], K - 1)];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool open = lo[i] < hi[i];
      const bool up = open && (u[i] <= p[i]);
      hi[i] = (open && !up) ? mid[i] : hi[i];
      lo[i] = up ? mid[i] + 1 : lo[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) r[i] = (p[i] != p[i]) ? 0xFFFFu : (uint32_t)lo[i];
}

}  // namespace

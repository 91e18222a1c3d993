// k_calibrate.hip — Calibrator.transform on gfx950 (reference src/Smooth/Calibration.py:57-69 -> sklearn
// IsotonicRegression(out_of_bounds="clip").transform per class, then Calibrator.normalize, :26-41).
//   per class c: x clipped to [X_min, X_max]; linear interpolation on (X_thresholds_, y_thresholds_) exactly as
//   scipy interp1d(kind="linear") evaluates it: k = clip(searchsorted(x_thr, x, side="left"), 1, n-1),
//   slope = (y[k]-y[k-1])/(x[k]-x[k-1]),  y = slope*(x - x[k-1]) + y[k-1];
//   normalize: 2 classes -> p0 = 1 - p1; else p /= sum(p); NaN -> 1/A; (1, 1+1e-5] -> 1.
// Arithmetic type follows sklearn/scipy: maps fitted on float32 probabilities (the xgb smoother) applied to float32
// inputs are evaluated entirely in float32 (clip, slope, interpolant), anything else in float64 with the thresholds
// widened; the result is then stored in a float64 array and normalised in float64.  One thread per row.
#include "gnx_internal.h"

namespace {

__global__ __launch_bounds__(256) void k_calibrate(CalibLaunch L) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= L.R) return;
  const int A = L.A;
  // one class's interpolated value.  Evaluated twice per class (once for the row sum, once for the output) instead of being
  // kept in a per-thread array: a run-time-indexed double p[32] lives in scratch (272 B per lane), and this pass is a few
  // binary searches per row — recomputing is cheaper than the round trip and yields the same bits.
  auto value = [&](int c) -> double {
    double x = L.in_is_f64 ? reinterpret_cast<const double*>(L.in)[r * A + c] : (double)reinterpret_cast<const float*>(L.in)[r * A + c];
    const int o0 = L.off[c], n = L.off[c + 1] - o0;
    const double* xs = L.x + o0;
    const double* ys = L.y + o0;
    if (n == 1) return ys[0];
    x = fmin(fmax(x, xs[0]), xs[n - 1]);
    int lo = 0, hi = n;  // searchsorted(side="left"): first index with xs[idx] >= x
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (xs[mid] < x) lo = mid + 1; else hi = mid;
    }
    const int k = lo < 1 ? 1 : (lo > n - 1 ? n - 1 : lo);
    if (L.thr_f32 && !L.in_is_f64) {  // all operands are float32 values: same operations in float32
      const float xf = (float)x, x0 = (float)xs[k - 1], x1 = (float)xs[k], y0 = (float)ys[k - 1], y1 = (float)ys[k];
      const float slope = (y1 - y0) / (x1 - x0);
      return (double)(slope * (xf - x0) + y0);
    }
    if (L.thr_f32) {                  // float64 input, float32 maps: the slope is a float32 quantity
      const float slope = ((float)ys[k] - (float)ys[k - 1]) / ((float)xs[k] - (float)xs[k - 1]);
      return (double)slope * (x - xs[k - 1]) + ys[k - 1];
    }
    const double slope = (ys[k] - ys[k - 1]) / (xs[k] - xs[k - 1]);
    return slope * (x - xs[k - 1]) + ys[k - 1];
  };
  double sum = 0.0;
  if (A != 2)
    for (int c = 0; c < A; ++c) sum += value(c);
  const double p1 = (A == 2) ? value(1) : 0.0;
  int best = 0;
  double pbest = 0.0;
  for (int c = 0; c < A; ++c) {   // (out64 may alias `in`: every input of the row has been read for the sum by now; class c's own
    double pc = (A == 2) ? (c == 0 ? 1.0 - p1 : p1) : value(c) / sum;   //  value is re-read here before it is overwritten)
    if (pc != pc) pc = 1.0 / A;
    if (pc > 1.0 && pc <= 1.0 + 1e-5) pc = 1.0;
    if (c == 0 || pc > pbest) { best = c; pbest = pc; }   // first maximum wins, as np.argmax
    if (L.out64) L.out64[r * A + c] = pc;
    if (L.out32) L.out32[r * A + c] = (float)pc;
  }
  if (L.labels) L.labels[r] = best;
}

}  // namespace

hipError_t gnx_launch_calibrate(const CalibLaunch& L, hipStream_t s) {
  if (L.R <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_calibrate, dim3((unsigned)((L.R + 255) / 256)), dim3(256), 0, s, L);
  return hipGetLastError();
}

// k_train_cnn.hip — training the convolutional smoother of the reference's "large" mode on gfx950.
//
// Replaces CNN.fit (reference src/Smooth/cnn.py:104-140, called by Smoother.train, src/Smooth/smooth.py:28-38, for
// CNN_Smoother, src/Smooth/models.py:35-42):
//   nn.Conv1d(A, A, S, padding=(S-1)//2)   (zero padding wherever the reference runs: see k_smooth_cnn.hip)
//   loss = NLLLoss()(log(Softmax(dim=1)(conv(B)) + 1e-8), y)          mean over the batch's rows x windows     cnn.py:57-75
//   torch.optim.Adam(lr = 1e-3), DataLoader(batch_size = 128, shuffle = True), max_ep = 250 epochs             cnn.py:32,104-118,172
// Dropout layers are constructed by the reference and never called.  All arithmetic float32, as torch.tensor(B, dtype=float).
//
// Per batch three launches on the context's stream, no host round trip:
//   k_cnn_fwd    one thread per (row of the batch, window), a block = 64 windows of one row whose padded probabilities sit in LDS;
//                a tap's A weights are wave-uniform (scalar loads from a [a_in][tap][a_out] copy); softmax, the row's loss
//                term, and g = dL/dlogit, stored class-major per row (what the weight gradient streams);
//   k_cnn_wgrad  dW[c][a][s] = sum over (row, window) of g[row][c][w] * B[row][a][w + s - pad]: a block owns one (c, a) pair and
//                a slice of 4 batch rows (g rows and padded x rows staged in LDS), thread = tap; partial sums per slice (no
//                atomics: the reduction order is fixed);
//   k_cnn_adam   slice partials summed in slice order, then torch's Adam step for that parameter (exp_avg, exp_avg_sq, bias
//                corrections, denom = sqrt(v) / sqrt(bc2) + eps, p -= lr / bc1 * m / denom); thread 0 also adds up the loss.
// The batch's rows are addressed through an index list (the epoch's permutation, uploaded once), B is held transposed (N, A, W)
// like torch's tensor.  Sizes are tiny next to inference (128 x 317 x 7 x 7 x 75 = 150 M multiply-adds per batch and direction).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gnx_internal.h"

#define HIPCHK(ctx, expr)                                                                          \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return gnx_fail((ctx), GNX_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
  } while (0)

namespace {

constexpr int MAXA_LIMIT = 32;  // most output channels k_cnn_fwd is instantiated for
constexpr int SLICE = 4;     // batch rows per weight-gradient slice

// (N, W, A) float32 / float64 -> (N, A, W) float32
__global__ void k_cnn_transpose(const void* B, int b_is_f64, int64_t N, int W, int A, float* Bt) {
  const int64_t total = N * (int64_t)W * A;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = e / ((int64_t)W * A);
    const int r = (int)(e - n * (int64_t)W * A), a = r / W, w = r - a * W;
    const size_t src = ((size_t)n * W + w) * A + a;
    Bt[e] = b_is_f64 ? (float)reinterpret_cast<const double*>(B)[src] : reinterpret_cast<const float*>(B)[src];
  }
}

// logits, softmax, loss term and dL/dlogit.  Block = one row of the batch x FT consecutive windows; the row's zero-padded
// probabilities of those windows sit in LDS ([a_in][FT + S - 1]: a tap's operand is lane + s, conflict-free), a tap's A output
// weights are wave-uniform and come through the scalar cache from wt = [a_in][tap][AP] (kept in step by k_cnn_adam).
constexpr int FT = 64;
template <int AP>
__global__ __launch_bounds__(FT) void k_cnn_fwd(const float* __restrict__ Bt, const int32_t* __restrict__ y, const int64_t* __restrict__ rows, int nb,
                                                int W, int A, int S, const float* __restrict__ wt, const float* __restrict__ bias, float log_eps,
                                                float* __restrict__ g, float* __restrict__ loss_part) {
  extern __shared__ float xt[];  // [A][FT + S - 1]
  const int pad = (S - 1) / 2, tw = FT + S - 1;
  const int b = blockIdx.y, w0 = blockIdx.x * FT, lane = threadIdx.x;
  const int64_t n = rows[b];
  for (int e = lane; e < A * tw; e += FT) {
    const int a = e / tw, q = e - a * tw, ww = w0 + q - pad;
    xt[e] = (ww >= 0 && ww < W) ? Bt[((size_t)n * A + a) * W + ww] : 0.f;
  }
  __syncthreads();
  const int w = w0 + lane;
  float z[AP];
#pragma unroll
  for (int c = 0; c < AP; ++c) z[c] = c < A ? bias[c] : 0.f;
  for (int a = 0; a < A; ++a) {
    const float* x = xt + a * tw + lane;
    const float* wr = wt + (size_t)a * S * AP;
    for (int s = 0; s < S; ++s) {
      const float xv = x[s];
#pragma unroll
      for (int c = 0; c < AP; ++c) z[c] = fmaf(wr[s * AP + c], xv, z[c]);  // padded channels carry zero weights
    }
  }
  float term = 0.f;
  if (w < W) {
    float zmax = z[0];
#pragma unroll
    for (int c = 1; c < AP; ++c)
      if (c < A) zmax = fmaxf(zmax, z[c]);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < AP; ++c)
      if (c < A) { z[c] = expf(z[c] - zmax); sum += z[c]; }
    const int lab = y[(size_t)n * W + w];
    float py = 0.f;
#pragma unroll
    for (int c = 0; c < AP; ++c)
      if (c < A) { z[c] = z[c] / sum; py = (c == lab) ? z[c] : py; }
    term = -logf(py + log_eps);
    // d(-log(p_y + eps) / (nb W)) / dz_c = -(p_y / (p_y + eps)) (delta_cy - p_c) / (nb W)
    const float k = py / (py + log_eps) / ((float)nb * (float)W);
#pragma unroll
    for (int c = 0; c < AP; ++c)
      if (c < A) g[((size_t)b * A + c) * W + w] = k * (z[c] - (c == lab ? 1.f : 0.f));
  }
  // the block's loss terms in lane order (one wave: shuffles)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) term += __shfl_down(term, o, 64);
  if (lane == 0) loss_part[(size_t)b * gridDim.x + blockIdx.x] = term;
}

// partial weight gradients: block = (c, a) x slice of SLICE batch rows, thread = tap; pw[slice][c][a][s], pb[slice][c].
// The slice's g rows and zero-padded x rows are staged in LDS: a tap's operands are gl[w] (broadcast) and xl[w + s] (consecutive).
__global__ __launch_bounds__(128) void k_cnn_wgrad(const float* __restrict__ Bt, const int64_t* __restrict__ rows, int nb, int W, int A, int S,
                                                   const float* __restrict__ g, float* __restrict__ pw, float* __restrict__ pb) {
  extern __shared__ float sm[];  // gl [SLICE][W4] | xl [SLICE][W4 + S - 1]
  const int c = blockIdx.x / A, a = blockIdx.x - c * A, sl = blockIdx.y;
  const int pad = (S - 1) / 2, W4 = (W + 3) & ~3, xw = W4 + S - 1;
  const int b0 = sl * SLICE, nr = min(nb, b0 + SLICE) - b0;
  float* gl = sm;
  float* xl = sm + SLICE * W4;
  for (int e = threadIdx.x; e < SLICE * W4; e += blockDim.x) {
    const int r = e / W4, w = e - r * W4;
    gl[e] = (r < nr && w < W) ? g[((size_t)(b0 + r) * A + c) * W + w] : 0.f;
  }
  for (int e = threadIdx.x; e < SLICE * xw; e += blockDim.x) {
    const int r = e / xw, q = e - r * xw, ww = q - pad;
    xl[e] = (r < nr && ww >= 0 && ww < W) ? Bt[((size_t)rows[b0 + r] * A + a) * W + ww] : 0.f;
  }
  __syncthreads();
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // four independent chains (window mod 4)
    for (int r = 0; r < SLICE; ++r) {
      const float* gr = gl + r * W4;
      const float* x = xl + r * xw + s;
      for (int w = 0; w < W4; w += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = fmaf(gr[w + u], x[w + u], acc[u]);
      }
    }
    pw[(((size_t)sl * A + c) * A + a) * S + s] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  }
  if (a == 0 && threadIdx.x == 127) {  // (a lane without a tap when S <= 127)
    float acc = 0.f;
    for (int r = 0; r < SLICE; ++r)
      for (int w = 0; w < W; ++w) acc += gl[r * W4 + w];
    pb[(size_t)sl * A + c] = acc;
  }
}

struct AdamState {
  float beta1, beta2, eps;
  float step, bc2_sqrt;  // lr / (1 - beta1^t), sqrt(1 - beta2^t): computed in double on the host, as torch does in Python
};

// one parameter per thread: weight (A*A*S) then bias (A); torch.optim.Adam's single-tensor step
__global__ void k_cnn_adam(int n_slices, int A, int S, int AP, float* __restrict__ wt, const float* __restrict__ pw, const float* __restrict__ pb, float* __restrict__ weight,
                           float* __restrict__ bias, float* __restrict__ m, float* __restrict__ v, AdamState st, const float* __restrict__ loss_part,
                           int n_loss, float inv_count, double* __restrict__ loss_out) {
  const int nw = A * A * S, np_ = nw + A;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < np_) {
    float gsum = 0.f;
    if (i < nw)
      for (int sl = 0; sl < n_slices; ++sl) gsum += pw[(size_t)sl * nw + i];
    else
      for (int sl = 0; sl < n_slices; ++sl) gsum += pb[(size_t)sl * A + (i - nw)];
    float* p = i < nw ? weight + i : bias + (i - nw);
    const float mi = m[i] * st.beta1 + gsum * (1.f - st.beta1);           // exp_avg.mul_(beta1).add_(grad, alpha = 1 - beta1)
    const float vi = v[i] * st.beta2 + (gsum * gsum) * (1.f - st.beta2);  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / st.bc2_sqrt + st.eps;
    const float pn = *p - st.step * (mi / denom);                          // param.addcdiv_(exp_avg, denom, value = -step_size)
    *p = pn;
    if (i < nw) {  // k_cnn_fwd's copy, [a_in][tap][AP]
      const int c = i / (A * S), r = i - c * A * S;
      wt[(size_t)r * AP + c] = pn;
    }
  }
  if (i == 0 && loss_out) {
    double acc = 0.0;
    for (int k = 0; k < n_loss; ++k) acc += (double)loss_part[k];
    *loss_out += acc * (double)inv_count;
  }
}

}  // namespace

// gnx_train_cnn (include/gnomix_hip.h): host arrays in, trained weight / bias out
extern "C" int gnx_train_cnn(gnx_ctx* ctx, const void* B, int32_t b_is_f64, const int32_t* y, int64_t N, int32_t W, int32_t A, int32_t S,
                             const gnx_cnn_params* P, const int64_t* order, float* weight, float* bias, double* loss) {
  if (!ctx) return GNX_EINVAL;
  if (!ctx->usable) return gnx_fail(ctx, GNX_EINVAL, "train_cnn: context is not usable");
  if (!B || !y || !P || !weight || !bias) return gnx_fail(ctx, GNX_EINVAL, "train_cnn: NULL argument");
  if (N <= 0 || W <= 0 || A < 2 || A > MAXA_LIMIT || S < 1 || (S & 1) == 0)
    return gnx_fail(ctx, GNX_EINVAL, "train_cnn: need N, W > 0, 2 <= A <= 32 and an odd kernel size S");
  const int AP = A <= 4 ? 4 : A <= 8 ? 8 : A <= 16 ? 16 : 32;  // channels k_cnn_fwd keeps in registers
  const size_t lds_fwd = (size_t)A * (FT + S - 1) * sizeof(float);
  const size_t lds_wg = (size_t)SLICE * (2 * (size_t)((W + 3) & ~3) + S - 1) * sizeof(float);
  if (lds_fwd > (size_t)64 * 1024 || lds_wg > (size_t)160 * 1024)
    return gnx_fail(ctx, GNX_EUNSUPPORTED, "train_cnn: a slice of rows (4 x (2 W + S) floats) or a window tile (A x (64 + S)) exceeds the LDS");
  if (P->epochs < 0 || P->batch < 1 || !(P->lr > 0) || !(P->beta1 >= 0 && P->beta1 < 1) || !(P->beta2 >= 0 && P->beta2 < 1) || !(P->eps > 0))
    return gnx_fail(ctx, GNX_EINVAL, "train_cnn: bad optimiser parameters");
  for (int64_t i = 0; i < N * W; ++i)
    if (y[i] < 0 || y[i] >= A) return gnx_fail(ctx, GNX_EINVAL, "train_cnn: label outside [0, A)");
  if (order)
    for (int64_t i = 0; i < (int64_t)P->epochs * N; ++i)
      if (order[i] < 0 || order[i] >= N) return gnx_fail(ctx, GNX_EINVAL, "train_cnn: row index outside [0, N) in `order`");
  GNX_BIND_DEVICE(ctx);
  hipStream_t s = ctx->stream;
  const int nbmax = (int)std::min<int64_t>(P->batch, N);
  const int n_slices_max = (nbmax + SLICE - 1) / SLICE;
  const int nw = A * A * S, np_ = nw + A;
  const int tiles = (W + FT - 1) / FT;
  const int fwd_blocks_max = nbmax * tiles;
  // one allocation: Bt | y | rows | g | pw | pb | weight | bias | m | v | loss partials | loss per epoch | wt
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_raw = take((size_t)N * W * A * (b_is_f64 ? 8 : 4));
  const size_t o_bt = take((size_t)N * W * A * 4), o_y = take((size_t)N * W * 4), o_rows = take((size_t)N * 8);
  const size_t o_g = take((size_t)nbmax * A * W * 4), o_pw = take((size_t)n_slices_max * nw * 4), o_pb = take((size_t)n_slices_max * A * 4);
  const size_t o_w = take((size_t)nw * 4), o_b = take((size_t)A * 4), o_m = take((size_t)np_ * 4), o_v = take((size_t)np_ * 4);
  const size_t o_lp = take((size_t)fwd_blocks_max * 4), o_le = take((size_t)std::max(1, P->epochs) * 8), o_wt = take((size_t)A * S * AP * 4);
  int rc = gnx_ws_reserve(ctx, ctx->ws_misc, off);
  if (rc != GNX_OK) return rc;
  uint8_t* base = (uint8_t*)ctx->ws_misc.p;
  float* d_bt = (float*)(base + o_bt);
  int32_t* d_y = (int32_t*)(base + o_y);
  int64_t* d_rows = (int64_t*)(base + o_rows);
  float *d_g = (float*)(base + o_g), *d_pw = (float*)(base + o_pw), *d_pb = (float*)(base + o_pb), *d_w = (float*)(base + o_w), *d_b = (float*)(base + o_b);
  float *d_m = (float*)(base + o_m), *d_v = (float*)(base + o_v), *d_lp = (float*)(base + o_lp);
  double* d_le = (double*)(base + o_le);
  float* d_wt = (float*)(base + o_wt);
  std::vector<float> wt0((size_t)A * S * AP, 0.f);  // [a_in][tap][AP] copy of the (a_out, a_in, tap) weights; padded channels stay zero
  for (int c = 0; c < A; ++c)
    for (int r = 0; r < A * S; ++r) wt0[(size_t)r * AP + c] = weight[(size_t)c * A * S + r];
  HIPCHK(ctx, hipMemcpyAsync(d_wt, wt0.data(), wt0.size() * 4, hipMemcpyHostToDevice, s));
  if (lds_wg > (size_t)64 * 1024)
    HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cnn_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wg));
  HIPCHK(ctx, hipMemcpyAsync(base + o_raw, B, (size_t)N * W * A * (b_is_f64 ? 8 : 4), hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(d_y, y, (size_t)N * W * 4, hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(d_w, weight, (size_t)nw * 4, hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemcpyAsync(d_b, bias, (size_t)A * 4, hipMemcpyHostToDevice, s));
  HIPCHK(ctx, hipMemsetAsync(d_m, 0, (size_t)np_ * 4, s));
  HIPCHK(ctx, hipMemsetAsync(d_v, 0, (size_t)np_ * 4, s));
  HIPCHK(ctx, hipMemsetAsync(d_le, 0, (size_t)std::max(1, P->epochs) * 8, s));
  hipLaunchKernelGGL(k_cnn_transpose, dim3(1024), dim3(256), 0, s, (const void*)(base + o_raw), (int)b_is_f64, N, (int)W, (int)A, d_bt);
  HIPCHK(ctx, hipGetLastError());
  std::vector<int64_t> ident;
  if (!order) {
    ident.resize((size_t)N);
    for (int64_t i = 0; i < N; ++i) ident[(size_t)i] = i;
    HIPCHK(ctx, hipMemcpyAsync(d_rows, ident.data(), (size_t)N * 8, hipMemcpyHostToDevice, s));
  }
  const int64_t n_batches = (N + P->batch - 1) / P->batch;
  double b1t = 1.0, b2t = 1.0;  // beta^t
  for (int ep = 0; ep < P->epochs; ++ep) {
    if (order) {
      HIPCHK(ctx, hipStreamSynchronize(s));  // the previous epoch still reads d_rows
      HIPCHK(ctx, hipMemcpyAsync(d_rows, order + (size_t)ep * N, (size_t)N * 8, hipMemcpyHostToDevice, s));
    }
    for (int64_t bi = 0; bi < n_batches; ++bi) {
      const int nb = (int)std::min<int64_t>(P->batch, N - bi * P->batch);
      const int fb = nb * tiles, ns = (nb + SLICE - 1) / SLICE;
      const int64_t* rows = d_rows + bi * P->batch;
#define GNX_CNN_FWD(AP_)                                                                                                                  \
  hipLaunchKernelGGL(k_cnn_fwd<AP_>, dim3(tiles, nb), dim3(FT), lds_fwd, s, (const float*)d_bt, (const int32_t*)d_y, rows, nb, (int)W, (int)A, \
                     (int)S, (const float*)d_wt, (const float*)d_b, (float)P->log_eps, d_g, d_lp)
      if (AP == 4) GNX_CNN_FWD(4);
      else if (AP == 8) GNX_CNN_FWD(8);
      else if (AP == 16) GNX_CNN_FWD(16);
      else GNX_CNN_FWD(32);
#undef GNX_CNN_FWD
      hipLaunchKernelGGL(k_cnn_wgrad, dim3(A * A, ns), dim3(128), lds_wg, s, (const float*)d_bt, rows, nb, (int)W, (int)A, (int)S, (const float*)d_g, d_pw, d_pb);
      b1t *= P->beta1;
      b2t *= P->beta2;
      AdamState st;
      st.beta1 = (float)P->beta1; st.beta2 = (float)P->beta2; st.eps = (float)P->eps;
      st.step = (float)(P->lr / (1.0 - b1t));
      st.bc2_sqrt = (float)std::sqrt(1.0 - b2t);
      // per-epoch loss = mean over batches of the batch's mean loss (cnn.py:120: running_loss / len(generator))
      const float inv = (float)(1.0 / ((double)nb * W * (double)n_batches));
      hipLaunchKernelGGL(k_cnn_adam, dim3((np_ + 255) / 256), dim3(256), 0, s, ns, (int)A, (int)S, AP, d_wt, (const float*)d_pw, (const float*)d_pb, d_w, d_b, d_m, d_v,
                         st, (const float*)d_lp, fb, inv, d_le + ep);
    }
    HIPCHK(ctx, hipGetLastError());
  }
  HIPCHK(ctx, hipMemcpyAsync(weight, d_w, (size_t)nw * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(ctx, hipMemcpyAsync(bias, d_b, (size_t)A * 4, hipMemcpyDeviceToHost, s));
  if (loss && P->epochs > 0) HIPCHK(ctx, hipMemcpyAsync(loss, d_le, (size_t)P->epochs * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(ctx, hipStreamSynchronize(s));
  return GNX_OK;
}

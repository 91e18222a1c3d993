/*
 * gnx_oracle.c — CPU restatement of the Gnomix inference hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; gnomix_amd/ never does (the product path
 * fails loudly when the HIP library is missing).
 *
 * Every function restates one row of SURVEY.md §8(a) from the formulas there and cites the
 * reference file:line (under /root/reference) whose behaviour it follows.  Nothing here is
 * copied from the reference: the reference is Python that delegates to sklearn / xgboost /
 * libsvm / CRFsuite; this file is plain scalar C99.
 *
 * Pinning status (see DESIGN.md §3 and tests/golden/make_golden.py):
 *   a1-a3 logistic base ............ pinned against the imported reference (golden G1)
 *   a4/a4' CovRSK + SVC probability . pinned against the imported reference (golden G2)
 *   a5 slide_window ................ pinned against the imported reference (golden G3)
 *   a6 xgboost tree walk + softmax .. PARITY UNPINNED: xgboost==1.1.1 is a third-party wheel,
 *                                     absent from /root/reference and from this image; restated
 *                                     from its documented model schema (see gnxo_xgb_* below)
 *   a7 CRFsuite marginals ........... PARITY UNPINNED: sklearn-crfsuite==0.3.6 / CRFsuite absent;
 *                                     restated as the textbook scaled forward-backward
 *   a8 argmax ....................... pinned (numpy first-max-wins)
 *   a9 Gnofix control flow .......... pinned against the imported reference (golden G5)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GNXO_OK 0
#define GNXO_EINVAL (-1)
#define GNXO_ENOMEM (-2)

/* ------------------------------------------------------------------------------------------
 * a1  Base.pad  (src/Base/base.py:41-44)
 *     Xp = [flip(X[:, :ctx]), X, flip(X[:, -ctx:])]   ->  column of X behind padded position p
 * ---------------------------------------------------------------------------------------- */
static inline int64_t pad_src(int64_t p, int64_t C, int64_t ctx) {
  if (p < ctx) return ctx - 1 - p;
  if (p < ctx + C) return p - ctx;
  return C - 1 - (p - ctx - C);
}

int64_t gnxo_pad_src(int64_t p, int64_t C, int64_t ctx) { return pad_src(p, C, ctx); }

/* ------------------------------------------------------------------------------------------
 * a2+a3  Base.predict_proba_vectorized (src/Base/base.py:146-180) with
 *        LogisticRegressionBase (src/Base/models.py:12-21) -> sklearn _predict_proba_lr (OvR):
 *   window i<W-1 = Xp[:, i*M : i*M+M_];  last window = Xp[:, -(M_+rem):]   (base.py:157-164)
 *   Z = Xw . coef^T + intercept   (the missing code 2 is used as the NUMBER 2)
 *   P = 1/(1+exp(-Z));  P /= sum_a P
 * coef is (W, A, ldc) row-major, ldc >= M_+rem, entries beyond a window's width ignored.
 * B is (N, W, A) float64.
 * ---------------------------------------------------------------------------------------- */
/* windows [w0, w1) only: B keeps its (N, W, A) shape and the other windows are left untouched.  This is how the CPU baseline
 * of bench.py spreads the work over the host's cores: a core owns a few WINDOWS (A x M_ doubles of weights that stay in its
 * cache) and runs every haplotype of the sample through them — splitting the haplotypes instead makes every core stream all
 * of the chromosome's weights (41 MB at chr22) from DRAM. */
int gnxo_base_lr_range(const int8_t* X, int64_t N, int64_t ldx, int64_t C, int64_t M, int64_t ctx,
                       int64_t A, const double* coef, int64_t ldc, const double* intercept, int64_t w0, int64_t w1, double* B) {
  if (M <= 0 || C < M || ctx < 0 || ctx > C || A <= 0 || A > 64) return GNXO_EINVAL;
  const int64_t W = C / M, rem = C - M * W, M_ = M + 2 * ctx;
  if (rem == 0) return GNXO_EINVAL; /* base.py:158 relies on C % M != 0 (gnomix.py:124-125) */
  if (ldc < M_ + rem) return GNXO_EINVAL;
  if (w0 < 0 || w1 > W || w0 > w1) return GNXO_EINVAL;
  double p[64];
  /* window-major: one window's coefficients (A x M_ doubles) stay in cache while every haplotype of the call goes through
   * them; per (haplotype, window, class) the arithmetic and its order are what sklearn's decision_function + expit do */
  for (int64_t i = w0; i < w1; ++i) {
    const int64_t start = i * M, len = (i == W - 1) ? M_ + rem : M_;
    for (int64_t n = 0; n < N; ++n) {
      const int8_t* x = X + n * ldx;
      double s = 0.0;
      for (int64_t a = 0; a < A; ++a) {
        const double* c = coef + (i * A + a) * ldc;
        double z = 0.0;
        for (int64_t k = 0; k < len; ++k) z += c[k] * (double)x[pad_src(start + k, C, ctx)];
        z += intercept[i * A + a];
        p[a] = 1.0 / (1.0 + exp(-z));
        s += p[a];
      }
      for (int64_t a = 0; a < A; ++a) B[(n * W + i) * A + a] = p[a] / s;
    }
  }
  return GNXO_OK;
}

int gnxo_base_lr(const int8_t* X, int64_t N, int64_t ldx, int64_t C, int64_t M, int64_t ctx,
                 int64_t A, const double* coef, int64_t ldc, const double* intercept, double* B) {
  if (M <= 0 || C < M) return GNXO_EINVAL;
  return gnxo_base_lr_range(X, N, ldx, C, M, ctx, A, coef, ldc, intercept, 0, C / M, B);
}

/* ------------------------------------------------------------------------------------------
 * a5  slide_window (src/Smooth/utils.py:4-29): reflect-pad by pad=(S+1)//2 windows, row (n,w) =
 *     Bp[n, w:w+S].ravel() cast to float32.  slide_src maps padded window j=w+s to a window of B.
 * ---------------------------------------------------------------------------------------- */
static inline int64_t slide_src(int64_t j, int64_t W, int64_t pad) {
  if (j < pad) return pad - 1 - j;
  if (j < pad + W) return j - pad;
  return W - 1 - (j - pad - W);
}

int64_t gnxo_slide_src(int64_t j, int64_t W, int64_t S) { return slide_src(j, W, (S + 1) / 2); }

/* out: (N*W, S*A) float32.  B may be float64 (is_f64=1) or float32. */
int gnxo_slide_window(const void* B, int is_f64, int64_t N, int64_t W, int64_t A, int64_t S,
                      float* out) {
  const int64_t pad = (S + 1) / 2;
  if (W < pad) return GNXO_EINVAL;
  for (int64_t n = 0; n < N; ++n)
    for (int64_t w = 0; w < W; ++w) {
      float* o = out + (n * W + w) * S * A;
      for (int64_t s = 0; s < S; ++s) {
        const int64_t src = slide_src(w + s, W, pad);
        for (int64_t a = 0; a < A; ++a) {
          const int64_t idx = (n * W + src) * A + a;
          o[s * A + a] = is_f64 ? (float)((const double*)B)[idx] : ((const float*)B)[idx];
        }
      }
    }
  return GNXO_OK;
}

/* ------------------------------------------------------------------------------------------
 * a6  XGB_Smoother model (src/Smooth/models.py:14-20) -> xgboost==1.1.1 (third-party, pinned in
 *     requirements.txt:11; NOT in /root/reference, NOT installed: PARITY UNPINNED).
 *     Call sites: smooth.py:46 (predict_proba), gnofix.py:157.
 *
 *     Restated from xgboost's documented model schema (doc/tutorials/saving_model + the JSON
 *     schema: per tree left_children, right_children, split_indices, split_conditions,
 *     default_left; tree_info = output group of each tree) and its CPU predictor:
 *       leaf walk : at an internal node go LEFT iff fvalue < split_condition, missing (NaN) ->
 *                   default child; a node is a leaf iff left_child == -1 and its value is
 *                   split_condition (RegTree::GetLeafIndex / GetNext)
 *       margin    : for each output group g, psum = 0f; for trees in model order with
 *                   tree_info==g: psum += leaf (float32);  margin_g = base_margin + psum,
 *                   base_margin = base_score (0.5 for multi:softprob) (CPUPredictor::PredValue)
 *       transform : softmax (common/math.h Softmax): wmax=max; e_i = expf(x_i - wmax);
 *                   wsum (double) += e_i;  out_i = e_i / (float)wsum
 * Tree storage here = xgboost's own arrays, concatenated over trees (tree_off = node offsets).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n_trees;
  int32_t n_class;
  const int32_t* tree_off;   /* n_trees+1 node offsets */
  const int32_t* left;       /* child index within the tree, -1 = leaf */
  const int32_t* right;
  const int32_t* feat;       /* split_indices */
  const float* cond;         /* split_conditions; leaf value at leaves */
  const uint8_t* default_left; /* may be NULL (features are never NaN on this path) */
  const int32_t* tree_class; /* tree_info */
  float base_score;
} gnxo_trees;

static inline float tree_leaf(const gnxo_trees* T, int32_t t, const float* f) {
  const int32_t o = T->tree_off[t];
  int32_t nid = 0;
  while (T->left[o + nid] != -1) {
    const float fv = f[T->feat[o + nid]];
    if (fv != fv) nid = (T->default_left && T->default_left[o + nid]) ? T->left[o + nid] : T->right[o + nid];
    else nid = (fv < T->cond[o + nid]) ? T->left[o + nid] : T->right[o + nid];
  }
  return T->cond[o + nid];
}

static void xgb_row(const gnxo_trees* T, const float* f, float* out /* n_class */) {
  const int K = T->n_class;
  for (int g = 0; g < K; ++g) {
    float psum = 0.0f;
    for (int32_t t = 0; t < T->n_trees; ++t)
      if (T->tree_class[t] == g) psum += tree_leaf(T, t, f);
    out[g] = T->base_score + psum;
  }
  float wmax = out[0];
  for (int g = 1; g < K; ++g) wmax = fmaxf(out[g], wmax);
  double wsum = 0.0;
  for (int g = 0; g < K; ++g) { out[g] = expf(out[g] - wmax); wsum += out[g]; }
  for (int g = 0; g < K; ++g) out[g] /= (float)wsum;
}

/* model.predict_proba on an explicit feature matrix (R, F) float32 -> (R, n_class) float32 */
int gnxo_xgb_predict_proba(const gnxo_trees* T, const float* feats, int64_t R, int64_t F, float* proba) {
  for (int64_t r = 0; r < R; ++r) xgb_row(T, feats + r * F, proba + r * T->n_class);
  return GNXO_OK;
}

/* a8 argmax, first max wins (np.argmax: smooth.py:61, gnomix.py:58) */
static inline int32_t argmax_f32(const float* p, int K) {
  int32_t b = 0;
  for (int k = 1; k < K; ++k) if (p[k] > p[b]) b = k;
  return b;
}

/* Smoother.predict_proba (smooth.py:40-56) for XGB_Smoother: slide_window -> model -> reshape,
 * without materialising the (N*W, S*A) matrix.  B float64 or float32 (N,W,A); proba float32
 * (N,W,A); labels int32 (N,W) (may be NULL). */
int gnxo_smooth_xgb(const gnxo_trees* T, const void* B, int is_f64, int64_t N, int64_t W, int64_t A,
                    int64_t S, float* proba, int32_t* labels) {
  const int64_t pad = (S + 1) / 2;
  if (W < pad || T->n_class != A) return GNXO_EINVAL;
  float* f = (float*)malloc(sizeof(float) * (size_t)(S * A));
  if (!f) return GNXO_ENOMEM;
  for (int64_t n = 0; n < N; ++n)
    for (int64_t w = 0; w < W; ++w) {
      for (int64_t s = 0; s < S; ++s) {
        const int64_t src = slide_src(w + s, W, pad);
        for (int64_t a = 0; a < A; ++a) {
          const int64_t idx = (n * W + src) * A + a;
          f[s * A + a] = is_f64 ? (float)((const double*)B)[idx] : ((const float*)B)[idx];
        }
      }
      float* o = proba + (n * W + w) * A;
      xgb_row(T, f, o);
      if (labels) labels[n * W + w] = argmax_f32(o, (int)A);
    }
  free(f);
  return GNXO_OK;
}

/* ------------------------------------------------------------------------------------------
 * a7  CRF_Smoother (src/Smooth/models.py:27-32, src/Smooth/crf.py:17-67) -> sklearn-crfsuite
 *     0.3.6 / CRFsuite (third-party; absent: PARITY UNPINNED).  Features at position t are the
 *     attributes "0".."A-1" with values B[n,t,a] (crf.py:24-27); all_possible_states and
 *     all_possible_transitions (crf.py:12-13) give dense state weights theta[a][y] and transition
 *     weights tau[y'][y].  predict_marginals (crf.py:65) = marginals of the linear chain:
 *        state_t(y) = sum_a theta[a][y] * B[n,t,a];   psi_t(y) = exp(state_t(y))
 *        alpha_0 = psi_0, alpha_t(y) = psi_t(y) * sum_y' alpha_{t-1}(y') exp(tau[y'][y]), each
 *        alpha_t scaled to sum 1 (scale c_t);  beta_{W-1} = c_{W-1},
 *        beta_t(y') = c_t * sum_y exp(tau[y'][y]) psi_{t+1}(y) beta_{t+1}(y)
 *        marginal_t(y) = alpha_t(y) * beta_t(y) / c_t
 *     (CRFsuite crf1d_context.c: crf1dc_alpha_score / crf1dc_beta_score / crf1dc_marginal_point.)
 *     B float64 (N,W,A) -> proba float64 (N,W,A).
 * ---------------------------------------------------------------------------------------- */
int gnxo_smooth_crf(const double* B, int64_t N, int64_t W, int64_t A, const double* state /*A x A [attr][label]*/,
                    const double* trans /*A x A [from][to]*/, double* proba, int32_t* labels) {
  if (A > 64) return GNXO_EINVAL;
  double* psi = (double*)malloc(sizeof(double) * (size_t)(W * A) * 3 + sizeof(double) * (size_t)(W + A * A));
  if (!psi) return GNXO_ENOMEM;
  double* alpha = psi + W * A;
  double* beta = alpha + W * A;
  double* scale = beta + W * A;
  double* et = scale + W;
  for (int64_t i = 0; i < A * A; ++i) et[i] = exp(trans[i]);
  for (int64_t n = 0; n < N; ++n) {
    const double* b = B + n * W * A;
    for (int64_t t = 0; t < W; ++t)
      for (int64_t y = 0; y < A; ++y) {
        double s = 0.0;
        for (int64_t a = 0; a < A; ++a) s += state[a * A + y] * b[t * A + a];
        psi[t * A + y] = exp(s);
      }
    /* forward */
    for (int64_t t = 0; t < W; ++t) {
      double sum = 0.0;
      for (int64_t y = 0; y < A; ++y) {
        double v;
        if (t == 0) v = psi[y];
        else {
          double acc = 0.0;
          for (int64_t yp = 0; yp < A; ++yp) acc += alpha[(t - 1) * A + yp] * et[yp * A + y];
          v = acc * psi[t * A + y];
        }
        alpha[t * A + y] = v;
        sum += v;
      }
      scale[t] = (sum != 0.0) ? 1.0 / sum : 1.0;
      for (int64_t y = 0; y < A; ++y) alpha[t * A + y] *= scale[t];
    }
    /* backward */
    for (int64_t y = 0; y < A; ++y) beta[(W - 1) * A + y] = scale[W - 1];
    for (int64_t t = W - 2; t >= 0; --t)
      for (int64_t yp = 0; yp < A; ++yp) {
        double acc = 0.0;
        for (int64_t y = 0; y < A; ++y) acc += et[yp * A + y] * psi[(t + 1) * A + y] * beta[(t + 1) * A + y];
        beta[t * A + yp] = acc * scale[t];
      }
    for (int64_t t = 0; t < W; ++t) {
      double* o = proba + (n * W + t) * A;
      for (int64_t y = 0; y < A; ++y) o[y] = alpha[t * A + y] * beta[t * A + y] / scale[t];
      if (labels) {
        int32_t bi = 0;
        for (int64_t y = 1; y < A; ++y) if (o[y] > o[bi]) bi = (int32_t)y;
        labels[n * W + t] = bi;
      }
    }
  }
  free(psi);
  return GNXO_OK;
}

/* ------------------------------------------------------------------------------------------
 * a4  CovRSK string kernel (src/Base/string_kernel.py:80-110).
 *     K(x,y) = sum over positions t of cov_tri_t, where inside a run of equal symbols cov_tri
 *     counts the m in Ms with m <= run-so-far (string_kernel.py:91-101), reset to 0 at a mismatch.
 *     Symbols are compared as raw int8 (2==2 is a match).  Ms_ohe[r] = 1 iff r in Ms
 *     (length >= M_+1).  K is (Nq, Nt) int64.
 * ---------------------------------------------------------------------------------------- */
int gnxo_covrsk(const int8_t* Xq, int64_t Nq, int64_t ldq, const int8_t* Xt, int64_t Nt, int64_t ldt,
                int64_t Mw, const uint8_t* Ms_ohe, int64_t* K) {
  for (int64_t q = 0; q < Nq; ++q)
    for (int64_t r = 0; r < Nt; ++r) {
      const int8_t* x = Xq + q * ldq;
      const int8_t* y = Xt + r * ldt;
      int64_t tri = 0, cov = 0, k = 0;
      for (int64_t t = 0; t < Mw; ++t) {
        if (x[t] == y[t]) { tri += 1; cov += Ms_ohe[tri]; k += cov; }
        else { tri = 0; cov = 0; }
      }
      K[q * Nt + r] = k;
    }
  return GNXO_OK;
}

/* plain triangular-number string kernel (string_kernel.py:5-24): K = sum over runs L(L+1)/2 */
int gnxo_string_kernel(const int8_t* Xq, int64_t Nq, int64_t ldq, const int8_t* Xt, int64_t Nt, int64_t ldt,
                       int64_t Mw, int64_t* K) {
  for (int64_t q = 0; q < Nq; ++q)
    for (int64_t r = 0; r < Nt; ++r) {
      const int8_t* x = Xq + q * ldq;
      const int8_t* y = Xt + r * ldt;
      int64_t tri = 0, k = 0;
      for (int64_t t = 0; t < Mw; ++t) {
        if (x[t] == y[t]) { tri += 1; k += tri; } else tri = 0;
      }
      K[q * Nt + r] = k;
    }
  return GNXO_OK;
}

/* ------------------------------------------------------------------------------------------
 * a4'  SVC.predict_proba with a precomputed kernel row (sklearn -> libsvm; libsvm ships inside
 *      the sklearn wheel, third-party: sklearn/svm/src/libsvm/svm.cpp — svm_predict_values,
 *      sigmoid_predict, multiclass_probability, svm_predict_probability).  Restated:
 *   SVs are grouped by class (n_support[c] each, start[c] = prefix sum); for the pair (i<j), p-th
 *   pair:  dec = sum_{sv in i} dual[j-1][sv] K[sv] + sum_{sv in j} dual[i][sv] K[sv] - rho[p]
 *   (sklearn stores _intercept_ = -rho);  r_ij = clip(sigmoid_predict(dec, probA[p], probB[p]),
 *   1e-7, 1-1e-7), r_ji = 1 - r_ij;  Wu-Lin-Weng coupling (multiclass_probability).
 *   Krow = K(query, all training rows); support[sv] indexes training rows.
 * ---------------------------------------------------------------------------------------- */
static double sigmoid_predict(double dec, double A_, double B_) {
  const double fApB = dec * A_ + B_;
  if (fApB >= 0) return exp(-fApB) / (1.0 + exp(-fApB));
  return 1.0 / (1.0 + exp(fApB));
}

static void multiclass_probability(int k, const double* r /* k x k */, double* p) {
  int t, j, iter, max_iter = (k > 100) ? k : 100;
  double* Q = (double*)malloc(sizeof(double) * (size_t)(k * k + k));
  double* Qp = Q + k * k;
  double pQp, eps = 0.005 / k;
  for (t = 0; t < k; t++) {
    p[t] = 1.0 / k;
    Q[t * k + t] = 0;
    for (j = 0; j < t; j++) { Q[t * k + t] += r[j * k + t] * r[j * k + t]; Q[t * k + j] = Q[j * k + t]; }
    for (j = t + 1; j < k; j++) { Q[t * k + t] += r[j * k + t] * r[j * k + t]; Q[t * k + j] = -r[j * k + t] * r[t * k + j]; }
  }
  for (iter = 0; iter < max_iter; iter++) {
    pQp = 0;
    for (t = 0; t < k; t++) {
      Qp[t] = 0;
      for (j = 0; j < k; j++) Qp[t] += Q[t * k + j] * p[j];
      pQp += p[t] * Qp[t];
    }
    double max_error = 0;
    for (t = 0; t < k; t++) { double e = fabs(Qp[t] - pQp); if (e > max_error) max_error = e; }
    if (max_error < eps) break;
    for (t = 0; t < k; t++) {
      double diff = (-Qp[t] + pQp) / Q[t * k + t];
      p[t] += diff;
      pQp = (pQp + diff * (diff * Q[t * k + t] + 2 * Qp[t])) / (1 + diff) / (1 + diff);
      for (j = 0; j < k; j++) { Qp[j] = (Qp[j] + diff * Q[t * k + j]) / (1 + diff); p[j] /= (1 + diff); }
    }
  }
  free(Q);
}

/* Krow: (Nq, Nt) int64 kernel values; support: (nSV,) int32; dual: (k-1, nSV) double;
 * intercept/probA/probB: (k(k-1)/2,) (sklearn _intercept_, _probA, _probB); n_support: (k,)
 * proba out: (Nq, k) double */
int gnxo_svc_predict_proba(const int64_t* Krow, int64_t Nq, int64_t Nt, int k, int64_t nSV,
                           const int32_t* support, const double* dual, const double* intercept,
                           const double* probA, const double* probB, const int32_t* n_support,
                           double* proba) {
  if (k < 2 || k > 64) return GNXO_EINVAL;
  int start[64];
  start[0] = 0;
  for (int i = 1; i < k; ++i) start[i] = start[i - 1] + n_support[i - 1];
  double* kv = (double*)malloc(sizeof(double) * (size_t)nSV + sizeof(double) * (size_t)(k * k));
  if (!kv) return GNXO_ENOMEM;
  double* r = kv + nSV;
  const double min_prob = 1e-7;
  for (int64_t q = 0; q < Nq; ++q) {
    for (int64_t s = 0; s < nSV; ++s) kv[s] = (double)Krow[q * Nt + support[s]];
    int p = 0;
    for (int i = 0; i < k; ++i)
      for (int j = i + 1; j < k; ++j) {
        double sum = 0;
        const int si = start[i], sj = start[j], ci = n_support[i], cj = n_support[j];
        const double* c1 = dual + (int64_t)(j - 1) * nSV;
        const double* c2 = dual + (int64_t)i * nSV;
        for (int t = 0; t < ci; ++t) sum += c1[si + t] * kv[si + t];
        for (int t = 0; t < cj; ++t) sum += c2[sj + t] * kv[sj + t];
        sum += intercept[p]; /* = -rho[p] */
        double v = sigmoid_predict(sum, probA[p], probB[p]);
        if (v < min_prob) v = min_prob;
        if (v > 1 - min_prob) v = 1 - min_prob;
        r[i * k + j] = v;
        r[j * k + i] = 1 - v;
        ++p;
      }
    multiclass_probability(k, r, proba + q * k);
  }
  free(kv);
  return GNXO_OK;
}

/* ------------------------------------------------------------------------------------------
 * Tree-ensemble bases ("forest" bases): XGBBase (src/Base/models.py:24-35): per window
 *   XGBClassifier(n_estimators=20, max_depth=4, learning_rate=0.1, missing=missing_encoding).predict_proba(Xw)
 * -> xgboost==1.1.1 (third-party, absent: PARITY UNPINNED).  Restated from the same documented schema as the
 * smoother (gnxo_xgb_* above) plus the parts this path adds:
 *   - features are the window's SNPs cast to float32; a value equal to `missing` is dropped from the DMatrix, i.e.
 *     the walk takes the node's DEFAULT child (default_left) there;
 *   - A >= 3: objective multi:softprob (margin_c = base_score + sum, softmax);  A == 2: binary:logistic, one tree
 *     per round, margin = logit(base_score) + sum leaves, p1 = 1/(1+expf(-margin)), proba = [1-p1, p1].
 * Trees of all windows are concatenated; win_tree0[w]..win_tree0[w+1] are window w's trees.  B is (N, W, A) float32.
 * ---------------------------------------------------------------------------------------- */
static inline float forest_leaf(const gnxo_trees* T, int32_t t, const int8_t* xw, int missing) {
  const int32_t o = T->tree_off[t];
  int32_t nid = 0;
  while (T->left[o + nid] != -1) {
    const int v = xw[T->feat[o + nid]];
    if (v == missing) nid = (T->default_left && T->default_left[o + nid]) ? T->left[o + nid] : T->right[o + nid];
    else nid = ((float)v < T->cond[o + nid]) ? T->left[o + nid] : T->right[o + nid];
  }
  return T->cond[o + nid];
}

int gnxo_base_forest(const gnxo_trees* T, const int32_t* win_tree0, const int8_t* X, int64_t N, int64_t ldx, int64_t C,
                     int64_t M, int64_t ctx, int64_t A, int missing, float* B) {
  const int64_t W = C / M, rem = C - M * W, M_ = M + 2 * ctx;
  if (rem == 0 || A < 2 || A > 64) return GNXO_EINVAL;
  int8_t* xw = (int8_t*)malloc((size_t)(M_ + rem));
  if (!xw) return GNXO_ENOMEM;
  float out[64];
  for (int64_t n = 0; n < N; ++n)
    for (int64_t i = 0; i < W; ++i) {
      const int64_t len = (i == W - 1) ? M_ + rem : M_;
      for (int64_t k = 0; k < len; ++k) xw[k] = X[n * ldx + pad_src(i * M + k, C, ctx)];
      float* o = B + (n * W + i) * A;
      if (A == 2) {
        float psum = 0.0f;
        for (int32_t t = win_tree0[i]; t < win_tree0[i + 1]; ++t) psum += forest_leaf(T, t, xw, missing);
        const float margin = logf(T->base_score / (1.0f - T->base_score)) + psum;  /* ProbToMargin of binary:logistic */
        const float p1 = 1.0f / (1.0f + expf(-margin));
        o[0] = 1.0f - p1;
        o[1] = p1;
      } else {
        for (int g = 0; g < A; ++g) {
          float psum = 0.0f;
          for (int32_t t = win_tree0[i]; t < win_tree0[i + 1]; ++t)
            if (T->tree_class[t] == g) psum += forest_leaf(T, t, xw, missing);
          out[g] = T->base_score + psum;
        }
        float wmax = out[0];
        for (int g = 1; g < A; ++g) wmax = fmaxf(out[g], wmax);
        double wsum = 0.0;
        for (int g = 0; g < A; ++g) { out[g] = expf(out[g] - wmax); wsum += out[g]; }
        for (int g = 0; g < A; ++g) o[g] = out[g] / (float)wsum;
      }
    }
  free(xw);
  return GNXO_OK;
}

/* ------------------------------------------------------------------------------------------
 * RFBase (src/Base/models.py:54-66): per window sklearn RandomForestClassifier(n_estimators=20, max_depth=4)
 * .predict_proba(Xw).  scikit-learn IS installed here, so this restatement is pinned against the reference's own
 * RFBase objects (tests/golden/G9_rf.npz).  sklearn semantics restated:
 *   - sklearn/tree/_tree.pyx  Tree._apply_dense: X is cast to float32; at an internal node go LEFT iff
 *     X[i, feature] <= threshold (threshold is float64); leaves have children_left == -1;
 *   - DecisionTreeClassifier.predict_proba: the leaf's class-weight row, divided by its sum (the converter stores the rows
 *     already normalised with the same numpy expression, `value` below);
 *   - ForestClassifier.predict_proba (sklearn/ensemble/_forest.py): all_proba += tree proba for every tree (float64; the
 *     reference runs the trees of a window single-threaded, models.py:62-63, so in estimator order), then
 *     all_proba /= n_estimators.
 * Code 2 ("missing") is an ordinary number here.  B is (N, W, A) float64.
 * ---------------------------------------------------------------------------------------- */
int gnxo_base_rforest(const int32_t* win_tree0, const int32_t* tree_off, const int32_t* left, const int32_t* right,
                      const int32_t* feat, const double* thr, const double* value, const int8_t* X, int64_t N, int64_t ldx,
                      int64_t C, int64_t M, int64_t ctx, int64_t A, double* B) {
  const int64_t W = C / M, rem = C - M * W, M_ = M + 2 * ctx;
  if (rem == 0 || A < 2 || A > 64) return GNXO_EINVAL;
  int8_t* xw = (int8_t*)malloc((size_t)(M_ + rem));
  if (!xw) return GNXO_ENOMEM;
  for (int64_t n = 0; n < N; ++n)
    for (int64_t i = 0; i < W; ++i) {
      const int64_t len = (i == W - 1) ? M_ + rem : M_;
      for (int64_t k = 0; k < len; ++k) xw[k] = X[n * ldx + pad_src(i * M + k, C, ctx)];
      double* o = B + (n * W + i) * A;
      for (int64_t a = 0; a < A; ++a) o[a] = 0.0;
      const int32_t t0 = win_tree0[i], t1 = win_tree0[i + 1];
      for (int32_t t = t0; t < t1; ++t) {
        const int32_t off = tree_off[t];
        int32_t nid = 0;
        while (left[off + nid] != -1) nid = ((double)(float)xw[feat[off + nid]] <= thr[off + nid]) ? left[off + nid] : right[off + nid];
        const double* v = value + (size_t)(off + nid) * A;
        for (int64_t a = 0; a < A; ++a) o[a] += v[a];
      }
      const double nt = (double)(t1 - t0);
      for (int64_t a = 0; a < A; ++a) o[a] /= nt;
    }
  free(xw);
  return GNXO_OK;
}

/* ------------------------------------------------------------------------------------------
 * Polynomial string kernel (src/Base/string_kernel.py:40-61, PolynomialStringKernelBase models.py:178-193):
 *   contigs = lengths of the runs of equal SNPs, a run being closed by every mismatch and by the end of the window
 *             (so two adjacent mismatches contribute a 0);  K = int( np.sum(contigs ** p) / p )   (K is an int matrix).
 * `run_value[L]` = L ** p as numpy computed it (the caller passes np.arange(width+1) ** p: data, not code).
 * np.sum over a contiguous float64 vector is NOT a left-to-right sum: numpy/core/src/umath/loops_utils.h.src
 * `DOUBLE_pairwise_sum`, reached through the add.reduce inner loop over the whole vector (initial value = the identity 0):
 *   S = PW(a, n)     (pinned against np.sum itself by tests/test_oracle_golden.py::test_numpy_pairwise_sum_order)
 *   PW(a, n): n < 8   -> 0. + a[0] + a[1] + ...                       (left to right)
 *             n <= 128 -> r[k] = a[k] (k < 8); r[k] += a[i + k] for i = 8, 16, ... < n - n % 8;
 *                         ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), then the n % 8 tail left to right
 *             else     -> n2 = n / 2; n2 -= n2 % 8;  PW(a, n2) + PW(a + n2, n - n2)
 * ---------------------------------------------------------------------------------------- */
static double np_pairwise(const double* a, int64_t n) {
  if (n < 8) {
    double res = 0.;
    for (int64_t i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}

int gnxo_poly_kernel(const int8_t* Xq, int64_t Nq, int64_t ldq, const int8_t* Xt, int64_t Nt, int64_t ldt, int64_t Mw,
                     const double* run_value, double p, int64_t* K) {
  double* c = (double*)malloc((size_t)(Mw + 2) * sizeof(double));
  if (!c) return GNXO_ENOMEM;
  for (int64_t q = 0; q < Nq; ++q)
    for (int64_t r = 0; r < Nt; ++r) {
      const int8_t* x = Xq + q * ldq;
      const int8_t* y = Xt + r * ldt;
      int64_t n = 0, counter = 0;
      for (int64_t j = 0; j < Mw; ++j) {
        if (x[j] == y[j]) ++counter;
        else { c[n++] = run_value[counter]; counter = 0; }
      }
      c[n++] = run_value[counter];
      const double s = np_pairwise(c, n);
      K[q * Nt + r] = (int64_t)(s / p);  /* numpy float -> int assignment truncates toward zero */
    }
  free(c);
  return GNXO_OK;
}

/* np.sum of a contiguous float64 vector, exposed so that the tests can pin the summation order against numpy itself */
double gnxo_np_sum(const double* a, int64_t n) { return n <= 0 ? 0.0 : np_pairwise(a, n); }

/* ------------------------------------------------------------------------------------------
 * CNN smoother of the "large" mode (src/Smooth/models.py:35-42, src/Smooth/cnn.py:37-55, 166-171, 173-184):
 *   B (N, W, A) -> torch.tensor(transpose(B, [0, 2, 1]), dtype=torch.float) -> nn.Conv1d(A, A, kernel_size=S,
 *   padding=(S-1)//2, padding_mode="reflection") -> nn.Softmax(dim=1) -> swapaxes -> (N, W, A) float32.
 * "reflection" is not a padding mode torch implements: torch <= 1.4 falls through to zero padding, torch >= 1.5 refuses to
 * build the layer, so the reference's CNN zero-pads wherever it runs; the golden vector (G11) is produced by the
 * reference's own CNN class under exactly that reading.  float32 arithmetic as in torch; the order in which the backend
 * sums the A*S taps is not defined, so this restatement (taps in (a_in, s) order) agrees with torch to a few float32 ulps
 * and is compared at the north star's 1e-5.
 * weight is torch's (A_out, A_in, S), bias (A_out,).
 * ---------------------------------------------------------------------------------------- */
int gnxo_smooth_cnn(const double* B, int64_t N, int64_t W, int64_t A, int64_t S, const float* weight, const float* bias,
                    float* proba, int64_t* labels) {
  if (S <= 0 || S % 2 == 0 || A < 1 || A > 64) return GNXO_EINVAL;
  const int64_t pad = (S - 1) / 2;
  float out[64];
  for (int64_t n = 0; n < N; ++n)
    for (int64_t w = 0; w < W; ++w) {
      for (int64_t y = 0; y < A; ++y) {
        float acc = bias[y];
        for (int64_t a = 0; a < A; ++a)
          for (int64_t s = 0; s < S; ++s) {
            const int64_t ww = w + s - pad;
            if (ww < 0 || ww >= W) continue;  /* zero padding */
            acc += (float)B[(n * W + ww) * A + a] * weight[(y * A + a) * S + s];
          }
        out[y] = acc;
      }
      float mx = out[0];
      for (int64_t y = 1; y < A; ++y) mx = fmaxf(mx, out[y]);
      float sum = 0.0f;
      for (int64_t y = 0; y < A; ++y) { out[y] = expf(out[y] - mx); sum += out[y]; }
      int64_t best = 0;
      for (int64_t y = 0; y < A; ++y) {
        const float p = out[y] / sum;
        proba[(n * W + w) * A + y] = p;
        if (p > proba[(n * W + w) * A + best]) best = y;
      }
      if (labels) labels[n * W + w] = best;
    }
  return GNXO_OK;
}

/* ------------------------------------------------------------------------------------------
 * f4  Training the tree smoother (src/Smooth/smooth.py:28-38 Smoother.train ->
 *     src/Smooth/models.py:14-20 XGBClassifier(n_estimators=100, max_depth=4, learning_rate=0.1,
 *     reg_lambda=1, reg_alpha=0, objective='multi:softprob', num_class=A).fit(slide_window(B), y))
 *
 *     xgboost itself is absent (third party, see a6), so there is no arithmetic to be bit-equal
 *     to; what is restated here is the ALGORITHM the call asks for — second-order gradient
 *     boosting of A one-vs-rest regression trees per round on the softmax objective
 *     (g = p - 1[y=c], h = max(2p(1-p), 1e-16); split gain GL^2/(HL+l) + GR^2/(HR+l) - G^2/(H+l);
 *     leaf -eta*G/(H+l); min_child_weight on H) — in the histogram form (tree_method="hist":
 *     <= 256 quantile bins per feature) with every sum held in fixed point, so that the HIP
 *     trainer (k_train_gbt.hip) and this port produce IDENTICAL trees, bit for bit.  The spec,
 *     shared with the device code by construction and checked by tests/test_train_gbt.py:
 *       - features of row (n,w): slide_window's f = s*A + a  ->  float32(B)[n, src(w+s), a]
 *       - cuts of class column a: bucket(v) = clamp(int(v*65536), 0, 65535); the k-th cut
 *         (k = 1..max_bin-1) is (u+1)/65536 for the first bucket u whose cumulative count reaches
 *         k*N*W/max_bin; duplicates dropped; bin(v) = #{cuts <= v}; "bin <= j" <=> v < cut[j]
 *       - p = softmax(F) through gnx_det_exp (plain IEEE operations in a fixed order), g and h
 *         rounded to multiples of 2^-30 and summed as int64
 *       - best split of a node: largest gain, ties to the lowest feature, then the lowest bin;
 *         taken if gain > max(gamma, 1e-6) and both children have H >= min_child_weight
 * ------------------------------------------------------------------------------------------ */
static double gnx_det_exp(double x) { /* x <= 0 */
  if (!(x > -745.0)) return 0.0;
  const double LOG2E = 1.4426950408889634, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  const double kf = rint(x * LOG2E);
  const double r = (x - kf * LN2_HI) - kf * LN2_LO;
  double p = 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  return ldexp(p, (int)kf);
}

#define GBT_FIX 1073741824.0 /* 2^30 */
#define GBT_MAXNODES 63      /* heap positions of a tree of depth <= 5 */

typedef struct {
  int32_t n_rounds, max_depth, max_bin, tree_method; /* tree_method: 0 = histogram (max_bin quantile bins), 1 = exact greedy */
  double eta, lambda, gamma, min_child_weight, base_score;
} gnxo_gbt_params;

/* exact greedy (xgboost's tree_method="exact", the default XGBClassifier(...) of src/Smooth/models.py:14-20 gets for data of this
 * size; ColMaker::EnumerateSplit restated from its documentation, NOT pinned to xgboost — absent from this image): per node and
 * feature the node's rows in ascending feature value; between two consecutive DISTINCT values v0 < v1 the split "x < thr" with
 * thr = (v0 + v1) * 0.5f in float32 (v1 itself when that rounds down to v0, so that v0 < thr <= v1 always holds) sends the rows up
 * to v0 left.  Same gain, constraints, fixed-point sums and first-best-wins order (features ascending, then values ascending) as the
 * histogram form. */
typedef struct { float v; int64_t g, h; } gbt_vgh;
static int gbt_vgh_cmp(const void* a, const void* b) {
  const float x = ((const gbt_vgh*)a)->v, y = ((const gbt_vgh*)b)->v;
  return (x > y) - (x < y);
}
static float gbt_mid(float v0, float v1) {
  const float m = (v0 + v1) * 0.5f;
  return m > v0 ? m : v1;
}

/* Outputs (caller-allocated): tree_off[T+1], tree_class[T], and node arrays of capacity T*63: left, right, feat (int32),
 * cond (float32: threshold of an internal node, value of a leaf).  T = n_rounds*A.  Returns the number of nodes or < 0.
 * loss_out[r] (optional) = mean multi-class log loss BEFORE round r's trees (r = 0..n_rounds), from the same p. */
int64_t gnxo_train_gbt(const void* B, int is_f64, const int32_t* y, int64_t N, int64_t W, int64_t A, int64_t S,
                       const gnxo_gbt_params* P, int32_t* tree_off, int32_t* tree_class, int32_t* left, int32_t* right,
                       int32_t* feat, float* cond, double* loss_out) {
  const int64_t pad = (S + 1) / 2, Wp = W + 2 * pad, R = N * W, F = S * A;
  const int D = P->max_depth, MB = P->max_bin;
  if (N <= 0 || W < 2 * S || A < 2 || S < 1 || (S & 1) == 0 || D < 1 || D > 5 || MB < 2 || MB > 256 || P->n_rounds < 1) return GNXO_EINVAL;
  float* Bf = (float*)malloc((size_t)R * A * sizeof(float));
  uint8_t* Bq = (uint8_t*)malloc((size_t)N * Wp * A);
  uint32_t* cnt = (uint32_t*)calloc((size_t)65536, sizeof(uint32_t));
  float* cuts = (float*)malloc((size_t)A * 256 * sizeof(float));
  int32_t* ncut = (int32_t*)calloc((size_t)A, sizeof(int32_t));
  uint8_t* lut = (uint8_t*)malloc((size_t)65536);
  float* Fm = (float*)malloc((size_t)R * A * sizeof(float));
  int64_t* gq = (int64_t*)malloc((size_t)R * A * sizeof(int64_t));
  int64_t* hq = (int64_t*)malloc((size_t)R * A * sizeof(int64_t));
  uint8_t* pos = (uint8_t*)malloc((size_t)R);
  int64_t* hist = (int64_t*)malloc((size_t)32 * F * 256 * 2 * sizeof(int64_t)); /* [node of level][f][bin][g,h] */
  if (!Bf || !Bq || !cnt || !cuts || !ncut || !lut || !Fm || !gq || !hq || !pos || !hist) return GNXO_ENOMEM;
  for (int64_t i = 0; i < R * A; ++i) Bf[i] = is_f64 ? (float)((const double*)B)[i] : ((const float*)B)[i];
  /* cuts and quantised, reflect-padded strips */
  for (int64_t a = 0; a < A; ++a) {
    memset(cnt, 0, 65536 * sizeof(uint32_t));
    for (int64_t i = 0; i < R; ++i) {
      const float v = Bf[i * A + a];
      int b = (v > 0.0f) ? (int)(v * 65536.0f) : 0;
      if (v >= 1.0f) b = 65535; /* also keeps the float -> int conversion in range */
      cnt[b]++;
    }
    int k = 1, nc = 0;
    uint64_t cum = 0;
    for (int u = 0; u < 65536 && k < MB; ++u) {
      cum += cnt[u];
      int hit = 0;
      while (k < MB && cum >= (uint64_t)k * (uint64_t)R / (uint64_t)MB) { ++k; hit = 1; }
      if (hit && u < 65535) cuts[a * 256 + nc++] = (float)(u + 1) / 65536.0f;
    }
    ncut[a] = nc;
    int c = 0;
    for (int u = 0; u < 65536; ++u) { /* bin = #{cuts <= v} = #{cuts*65536 <= bucket} */
      while (c < nc && (int)(cuts[a * 256 + c] * 65536.0f) <= u) ++c;
      lut[u] = (uint8_t)c;
    }
    for (int64_t n = 0; n < N; ++n)
      for (int64_t j = 0; j < Wp; ++j) {
        const float v = Bf[(n * W + slide_src(j, W, pad)) * A + a];
        int b = (v > 0.0f) ? (int)(v * 65536.0f) : 0;
        if (v >= 1.0f) b = 65535;
        Bq[(n * Wp + j) * A + a] = lut[b];
      }
  }
  for (int64_t i = 0; i < R * A; ++i) Fm[i] = (float)P->base_score;
  int64_t nn = 0;
  int32_t t = 0;
  tree_off[0] = 0;
  for (int r = 0; r <= P->n_rounds; ++r) {
    /* gradients of the round (one softmax per row for all A trees) */
    double loss = 0.0;
    for (int64_t i = 0; i < R; ++i) {
      float m = Fm[i * A];
      for (int64_t c = 1; c < A; ++c) m = Fm[i * A + c] > m ? Fm[i * A + c] : m;
      double e[64], sum = 0.0;
      for (int64_t c = 0; c < A; ++c) { e[c] = gnx_det_exp((double)(Fm[i * A + c] - m)); sum += e[c]; }
      for (int64_t c = 0; c < A; ++c) {
        const double p = e[c] / sum;
        const double g = p - (y[i] == c ? 1.0 : 0.0);
        double h = 2.0 * p * (1.0 - p);
        if (h < 1e-16) h = 1e-16;
        gq[c * R + i] = (int64_t)llrint(g * GBT_FIX);
        hq[c * R + i] = (int64_t)llrint(h * GBT_FIX);
      }
      const double py = e[y[i]] / sum;
      loss -= log(py > 1e-300 ? py : 1e-300);
    }
    if (loss_out) loss_out[r] = loss / (double)R;
    if (r == P->n_rounds) break;
    for (int64_t c = 0; c < A; ++c, ++t) {
      int64_t nG[GBT_MAXNODES], nH[GBT_MAXNODES];
      int32_t nF[GBT_MAXNODES], nB[GBT_MAXNODES], st[GBT_MAXNODES]; /* st: 0 unused, 1 open, 2 internal, 3 leaf */
      float nV[GBT_MAXNODES], nT[GBT_MAXNODES]; /* nT: exact mode's thresholds */
      memset(st, 0, sizeof(st));
      const int64_t* g = gq + c * R;
      const int64_t* h = hq + c * R;
      nG[0] = 0; nH[0] = 0;
      for (int64_t i = 0; i < R; ++i) { nG[0] += g[i]; nH[0] += h[i]; pos[i] = 0; }
      st[0] = 1;
      for (int d = 0; d < D; ++d) {
        const int base = (1 << d) - 1, nl = 1 << d;
        int any = 0;
        for (int k = 0; k < nl; ++k) any |= st[base + k] == 1;
        if (!any) break;
        if (P->tree_method == 1) { /* ---- exact greedy ---- */
          gbt_vgh* buf = (gbt_vgh*)malloc((size_t)R * sizeof(gbt_vgh));
          if (!buf) return GNXO_ENOMEM;
          for (int k = 0; k < nl; ++k) {
            const int node = base + k;
            if (st[node] != 1) continue;
            const double Gd = (double)nG[node] / GBT_FIX, Hd = (double)nH[node] / GBT_FIX;
            const double root_term = Gd * Gd / (Hd + P->lambda);
            double best = P->gamma > 1e-6 ? P->gamma : 1e-6;
            int bf = -1;
            float bthr = 0.f;
            int64_t bGL = 0, bHL = 0;
            for (int64_t f = 0; f < F; ++f) {
              const int64_t sft = f / A, a = f % A;
              int64_t m = 0;
              for (int64_t n = 0; n < N; ++n)
                for (int64_t w = 0; w < W; ++w) {
                  const int64_t i = n * W + w;
                  if (pos[i] != node) continue;
                  buf[m].v = Bf[(n * W + slide_src(w + sft, W, pad)) * A + a];
                  buf[m].g = g[i];
                  buf[m].h = h[i];
                  ++m;
                }
              qsort(buf, (size_t)m, sizeof(gbt_vgh), gbt_vgh_cmp);
              int64_t GL = 0, HL = 0;
              for (int64_t q = 0; q + 1 < m; ++q) {
                GL += buf[q].g;
                HL += buf[q].h;
                if (!(buf[q].v < buf[q + 1].v)) continue;
                const double gl = (double)GL / GBT_FIX, hl = (double)HL / GBT_FIX;
                const double gr = (double)(nG[node] - GL) / GBT_FIX, hr = (double)(nH[node] - HL) / GBT_FIX;
                if (hl < P->min_child_weight || hr < P->min_child_weight) continue;
                const double gain = (gl * gl / (hl + P->lambda) + gr * gr / (hr + P->lambda)) - root_term;
                if (gain > best) { best = gain; bf = (int)f; bthr = gbt_mid(buf[q].v, buf[q + 1].v); bGL = GL; bHL = HL; }
              }
            }
            if (bf >= 0) {
              st[node] = 2; nF[node] = bf; nT[node] = bthr; nB[node] = 0;
              const int l = 2 * node + 1, rr = 2 * node + 2;
              st[l] = 1; st[rr] = 1;
              nG[l] = bGL; nH[l] = bHL; nG[rr] = nG[node] - bGL; nH[rr] = nH[node] - bHL;
            } else {
              st[node] = 3;
            }
          }
          free(buf);
          for (int64_t n = 0; n < N; ++n)
            for (int64_t w = 0; w < W; ++w) {
              const int64_t i = n * W + w;
              const int node = pos[i];
              if (node < base || st[node] != 2) continue;
              const float v = Bf[(n * W + slide_src(w + nF[node] / A, W, pad)) * A + nF[node] % A];
              pos[i] = (uint8_t)(v < nT[node] ? 2 * node + 1 : 2 * node + 2);
            }
          continue;
        }
        memset(hist, 0, (size_t)nl * F * 256 * 2 * sizeof(int64_t));
        for (int64_t n = 0; n < N; ++n)
          for (int64_t w = 0; w < W; ++w) {
            const int64_t i = n * W + w;
            const int k = (int)pos[i] - base;
            if (k < 0 || k >= nl || st[pos[i]] != 1) continue;
            const uint8_t* q = Bq + (n * Wp + w) * A; /* feature f = s*A + a is q[f] */
            int64_t* hk = hist + (size_t)k * F * 512;
            for (int64_t f = 0; f < F; ++f) { hk[(f * 256 + q[f]) * 2] += g[i]; hk[(f * 256 + q[f]) * 2 + 1] += h[i]; }
          }
        for (int k = 0; k < nl; ++k) {
          const int node = base + k;
          if (st[node] != 1) continue;
          const double Gd = (double)nG[node] / GBT_FIX, Hd = (double)nH[node] / GBT_FIX;
          const double root_term = Gd * Gd / (Hd + P->lambda);
          double best = P->gamma > 1e-6 ? P->gamma : 1e-6;
          int bf = -1, bb = -1;
          int64_t bGL = 0, bHL = 0;
          for (int64_t f = 0; f < F; ++f) {
            const int nc = ncut[f % A];
            int64_t GL = 0, HL = 0;
            const int64_t* hf = hist + ((size_t)k * F + f) * 512;
            for (int j = 0; j < nc; ++j) { /* bins 0..j left, threshold cut[j] */
              GL += hf[j * 2];
              HL += hf[j * 2 + 1];
              const double gl = (double)GL / GBT_FIX, hl = (double)HL / GBT_FIX;
              const double gr = (double)(nG[node] - GL) / GBT_FIX, hr = (double)(nH[node] - HL) / GBT_FIX;
              if (hl < P->min_child_weight || hr < P->min_child_weight) continue;
              const double gain = (gl * gl / (hl + P->lambda) + gr * gr / (hr + P->lambda)) - root_term;
              if (gain > best) { best = gain; bf = (int)f; bb = j; bGL = GL; bHL = HL; }
            }
          }
          if (bf >= 0) {
            st[node] = 2; nF[node] = bf; nB[node] = bb;
            const int l = 2 * node + 1, rr = 2 * node + 2;
            st[l] = 1; st[rr] = 1;
            nG[l] = bGL; nH[l] = bHL; nG[rr] = nG[node] - bGL; nH[rr] = nH[node] - bHL;
          } else {
            st[node] = 3;
          }
        }
        for (int64_t n = 0; n < N; ++n)
          for (int64_t w = 0; w < W; ++w) {
            const int64_t i = n * W + w;
            const int node = pos[i];
            if (node < base || st[node] != 2) continue;
            const uint8_t q = Bq[(n * Wp + w) * A + nF[node]];
            pos[i] = (uint8_t)(q <= nB[node] ? 2 * node + 1 : 2 * node + 2);
          }
      }
      /* leaves, emitted tree (nodes in heap order), margins */
      int32_t idx[GBT_MAXNODES];
      int32_t cntn = 0;
      for (int node = 0; node < GBT_MAXNODES; ++node) {
        if (st[node] == 1) st[node] = 3;
        if (st[node] == 3) {
          const double Gd = (double)nG[node] / GBT_FIX, Hd = (double)nH[node] / GBT_FIX;
          nV[node] = (float)(P->eta * (-Gd / (Hd + P->lambda)));
        }
        if (st[node] >= 2) idx[node] = cntn++;
      }
      for (int node = 0; node < GBT_MAXNODES; ++node) {
        if (st[node] < 2) continue;
        const int64_t o = nn + idx[node];
        if (st[node] == 2) {
          left[o] = idx[2 * node + 1]; right[o] = idx[2 * node + 2]; feat[o] = nF[node];
          cond[o] = P->tree_method == 1 ? nT[node] : cuts[(nF[node] % A) * 256 + nB[node]];
        } else {
          left[o] = -1; right[o] = -1; feat[o] = 0; cond[o] = nV[node];
        }
      }
      nn += cntn;
      tree_off[t + 1] = (int32_t)nn;
      tree_class[t] = (int32_t)c;
      for (int64_t i = 0; i < R; ++i) Fm[i * A + c] += nV[pos[i]];
    }
  }
  free(Bf); free(Bq); free(cnt); free(cuts); free(ncut); free(lut); free(Fm); free(gq); free(hq); free(pos); free(hist);
  return nn;
}

/*
 * gnomix_hip.h — C ABI of libgnomix_hip.so: the MI355X (gfx950) implementation of the Gnomix
 * inference hot path  X (phased SNPs, int8) -> per-window base classifier -> B -> sliding-window
 * smoother -> per-window ancestry probabilities / labels  (+ the Gnofix re-phasing loop).
 *
 * The reference (AI-sandbox/gnomix, /root/reference) is pure Python and has no FFI of its own; the
 * entry points below are what a binding for this path replaces, one per reference call site:
 *
 *   gnx_model_load        <- pickle.load of src.model.Gnomix            gnomix.py:26-35 (load_model)
 *   gnx_base_predict      <- Base.predict_proba(X)                      src/Base/base.py:129-180, gnomix.py:55
 *   gnx_smooth_predict    <- Smoother.predict_proba(B) / .predict(B)    src/Smooth/smooth.py:40-65, gnomix.py:57-58
 *   gnx_infer             <- Gnomix.predict_proba(X) / .predict(X)      src/model.py:169-179, gnomix.py:72
 *   gnx_smooth_rows       <- smoother.model.predict_proba(rows)         src/Gnofix/gnofix.py:157
 *   gnx_gnofix            <- Gnomix.phase(X, B) -> gnofix() per indiv.  src/model.py:188-214, src/Gnofix/gnofix.py:58-208
 *   gnx_train_logistic    <- Base.train(X, y) of LogisticRegressionBase   src/Base/base.py:104-127, src/model.py:113,155
 *   gnx_train_gbt         <- Smoother.train(B, y) of XGB_Smoother         src/Smooth/smooth.py:28-38, src/model.py:137
 *   gnx_train_crf         <- Smoother.train(B, y) of CRF_Smoother         src/Smooth/crf.py:51-58, src/Smooth/models.py:27-32
 *   gnx_train_cnn         <- Smoother.train(B, y) of CNN_Smoother         src/Smooth/cnn.py:104-118, src/Smooth/models.py:35-42
 *
 * Conventions
 *   - return 0 (GNX_OK) or a negative GNX_E* code; the message is kept per context (gnx_last_error).
 *     No exception or signal crosses this ABI.  (The reference aborts with Python assert/exception:
 *     src/model.py:193-194, src/Smooth/models.py:13, src/Smooth/smooth.py:31 — the Python mirror in
 *     gnomix_amd/ turns the codes back into those exceptions.)
 *   - plain pointers and sizes only; the library never frees or keeps caller memory
 *     (gnx_model_load copies what it needs).
 *   - un-suffixed entry points take HOST pointers, are synchronous and stage through the context's
 *     device workspaces (gnx_infer / gnx_infer_packed in batches, the copy-in of batch i+1, the kernels of
 *     batch i and the copy-out of batch i-1 overlapped on three streams); *_dev entry points take DEVICE pointers (same device as the context), are
 *     asynchronous on the context stream (gnx_set_stream / gnx_synchronize) and never allocate
 *     when the workspace is already large enough.
 *   - one gnx_ctx per device; a context and its models are not thread-safe (the reference is
 *     single-threaded at this level: src/model.py:205); different contexts may live on different
 *     threads or processes (one process per GPU under torch.distributed).
 *   - haplotype rows 2i, 2i+1 of X are the two haplotypes of individual i (src/utils.py:121-123);
 *     X values are {0,1,2=missing} int8, the value 2 enters the logistic model as the number 2
 *     (sklearn sees it as a feature value) and string kernels as a third symbol.
 */
#ifndef GNOMIX_HIP_H
#define GNOMIX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNX_ABI_VERSION 14

typedef struct gnx_ctx gnx_ctx;
typedef struct gnx_model gnx_model;

enum {
  GNX_OK = 0,
  GNX_EINVAL = -1,       /* bad argument / inconsistent model description */
  GNX_ENOMEM = -2,       /* host or device allocation failed */
  GNX_EHIP = -3,         /* HIP runtime error (message has the hipError string) */
  GNX_EUNSUPPORTED = -4, /* valid in the reference but not built here (message says what) */
  GNX_ESTATE = -5,       /* call not valid for this model (e.g. phasing with a CRF smoother) */
  GNX_ESTALE = -6        /* gnx_model_desc.prepared does not belong to this model / library / settings, or is truncated: nothing was
                            loaded; load again without it (and write a new one) */
};

enum { GNX_SVC_KERNEL_SUBSTRINGS = 0, GNX_SVC_KERNEL_POLY = 1 };
enum { GNX_BASE_NONE = 0, GNX_BASE_LOGISTIC = 1, GNX_BASE_COVRSK_SVC = 2, GNX_BASE_FOREST = 3, GNX_BASE_RFOREST = 4 };
enum { GNX_SMOOTH_NONE = 0, GNX_SMOOTH_XGB = 1, GNX_SMOOTH_CRF = 2, GNX_SMOOTH_CNN = 3 };

/* kernel ids for gnx_profile_get */
enum {
  GNX_K_BASE_LOGISTIC = 0,
  GNX_K_SMOOTH_XGB = 1,
  GNX_K_BASE_COVRSK = 2,
  GNX_K_SMOOTH_CRF = 3,
  GNX_K_GNOFIX = 4,
  GNX_K_SMOOTH_ROWS = 5,
  GNX_K_CALIBRATE = 6,
  GNX_K_BASE_FOREST = 7,
  GNX_K_SMOOTH_CNN = 8,
  GNX_K_COUNT = 9
};

/* Per-window SVC of CovRSKBase (src/Base/models.py:195-215 -> sklearn.svm.SVC(kernel=callable,
 * probability=True)); field names follow the fitted sklearn attributes. */
typedef struct gnx_svc_window {
  const int8_t* xfit;       /* (n_fit, width) training rows, row-major (sklearn __Xfit) */
  int32_t n_fit;
  int32_t width;            /* M_ (or M_+rem for the last window) */
  const int32_t* support;   /* (n_sv,) indices into xfit (support_) */
  int32_t n_sv;
  const double* dual_coef;  /* (A-1, n_sv) (_dual_coef_) */
  const double* intercept;  /* (A(A-1)/2,) (_intercept_ = -rho) */
  const double* prob_a;     /* (A(A-1)/2,) (_probA) */
  const double* prob_b;     /* (A(A-1)/2,) (_probB) */
  const int32_t* n_support; /* (A,) (_n_support) */
  const int32_t* ms;        /* CovSample lengths for this width (string_kernel.py:80-89); unused for GNX_SVC_KERNEL_POLY */
  int32_t n_ms;
  int32_t kernel_kind;      /* GNX_SVC_KERNEL_*: 0 = substring counts over the lengths `ms` (CovRSK; every length = the plain
                               string kernel), 1 = polynomial string kernel (string_kernel.py:40-61) */
  double poly_p;            /* POLY: K = int(np.sum(run_value[run lengths]) / poly_p), p = 1.2 in the reference */
  const double* run_value;  /* POLY: (width+1,) value of a run of L equal SNPs = L ** p as numpy computed it */
} gnx_svc_window;

/* Everything a pickled src.model.Gnomix carries for inference (src/model.py:28-88), as flat host
 * arrays.  W = C / M (src/model.py:32); the reference requires C % M != 0 (gnomix.py:124-125). */
typedef struct gnx_model_desc {
  int32_t abi_version; /* GNX_ABI_VERSION */
  int32_t A;           /* ancestries */
  int64_t C;           /* SNPs */
  int64_t M;           /* window size in SNPs */
  int64_t ctx;         /* context SNPs each side = int(M*context_ratio) (src/model.py:47) */
  int32_t S;           /* smoother width in windows, odd (src/Smooth/smooth.py:14) */
  int32_t base_kind;   /* GNX_BASE_* */
  int32_t smooth_kind; /* GNX_SMOOTH_* */
  int32_t reserved0;

  /* GNX_BASE_LOGISTIC: LogisticRegression(solver=liblinear) per window, OvR (models.py:12-21) */
  const double* lr_coef;      /* (W, A, lr_ldc): coef_ of window i in [i][a][0:width_i], rest ignored */
  int64_t lr_ldc;             /* >= M + 2ctx + rem */
  const double* lr_intercept; /* (W, A) */

  /* GNX_BASE_COVRSK_SVC */
  const gnx_svc_window* svc;  /* (W,) */

  /* GNX_SMOOTH_XGB: xgboost model schema, node arrays of all trees concatenated
   * (src/Smooth/models.py:14-20: multi:softprob, tree t belongs to class tree_class[t]) */
  int32_t n_trees;
  int32_t n_nodes;           /* length of left/right/feat/cond (0 = not given: tree_off[n_trees] is trusted) */
  const int32_t* tree_off;   /* (n_trees+1,) node offsets: 0, strictly increasing, tree_off[n_trees] == n_nodes */
  const int32_t* left;       /* child index within the tree, -1 at leaves */
  const int32_t* right;
  const int32_t* feat;       /* split feature = s*A + a of the (S*A)-wide sliding window */
  const float* cond;         /* split condition (go left iff f < cond); leaf value at leaves */
  const int32_t* tree_class; /* (n_trees,) */
  float base_score;          /* 0.5 */
  int32_t reserved2;

  /* GNX_SMOOTH_CRF: linear-chain CRF (src/Smooth/crf.py:9-15) */
  const double* crf_state;   /* (A, A) [attribute a][label y] */
  const double* crf_trans;   /* (A, A) [from y'][to y] */

  /* optional Calibrator (src/Smooth/Calibration.py:19-69): per class c the fitted IsotonicRegression's
   * X_thresholds_ / y_thresholds_ in [calib_off[c], calib_off[c+1]); NULL = no calibrator trained */
  const int32_t* calib_off;  /* (A+1,) */
  const double* calib_x;
  const double* calib_y;
  int32_t calib_is_f32;      /* the isotonic maps were fitted on float32 probabilities (the xgb smoother's output): sklearn then
                                interpolates in float32, and so does the kernel for float32 inputs */
  int32_t reserved3;

  /* GNX_BASE_FOREST: one gradient-boosted tree ensemble per window (XGBBase, src/Base/models.py:24-35:
   * XGBClassifier(n_estimators=20, max_depth=4, missing=missing_encoding)); xgboost model schema, all windows'
   * trees concatenated.  A >= 3: multi:softprob, tree t adds to class fb_tree_class[t]; A == 2: binary:logistic
   * (every tree adds to the one margin, fb_tree_class ignored).  Split features are SNP indices WITHIN the
   * window's padded slice [i*M, i*M + width_i).  A SNP equal to fb_missing follows fb_default_left. */
  int32_t fb_n_trees;
  int32_t fb_missing;               /* missing_encoding, 2 (src/Base/base.py:25) */
  const int32_t* fb_win_tree0;      /* (W+1,) first tree of each window */
  const int32_t* fb_tree_off;       /* (fb_n_trees+1,) node offsets */
  const int32_t* fb_left;           /* child index within the tree, -1 at leaves */
  const int32_t* fb_right;
  const int32_t* fb_feat;
  const float* fb_cond;             /* split condition (left iff x < cond); leaf value at leaves */
  const uint8_t* fb_default_left;   /* per node: 1 = missing goes left */
  const int32_t* fb_tree_class;     /* (fb_n_trees,) */
  float fb_base_score;              /* 0.5 */
  int32_t fb_n_nodes;               /* length of the fb_ node arrays (0 = not given) */

  /* GNX_BASE_RFOREST: one random forest per window (RFBase, src/Base/models.py:54-66:
   * RandomForestClassifier(n_estimators=20, max_depth=4)); sklearn's tree arrays (tree_.children_left/right, feature,
   * threshold) of all trees of all windows concatenated.  Left iff float32(x) <= threshold; a leaf contributes its
   * class-probability row rf_value[node]; the window's output is the mean over its trees (float64). */
  int32_t rf_n_trees;
  int32_t rf_n_nodes;               /* length of the rf_ node arrays (0 = not given) */
  const int32_t* rf_win_tree0;      /* (W+1,) */
  const int32_t* rf_tree_off;       /* (rf_n_trees+1,) node offsets */
  const int32_t* rf_left;           /* -1 at leaves */
  const int32_t* rf_right;
  const int32_t* rf_feat;           /* SNP index within the window's padded slice */
  const double* rf_thr;
  const double* rf_value;           /* (n_nodes, A) what DecisionTreeClassifier.predict_proba returns at that node */

  /* GNX_SMOOTH_CNN: the "large" mode's smoother (src/Smooth/cnn.py:37-55): one Conv1d(A, A, kernel_size=S,
   * padding=(S-1)/2, zero padding — see k_smooth_cnn.hip) over the windows + softmax over the A output channels */
  const float* cnn_weight;          /* (A_out, A_in, S) smoothNet[0].weight */
  const float* cnn_bias;            /* (A_out,) smoothNet[0].bias */

  /* optional: the logistic base's PREPARED digit planes as gnx_model_export_prepared wrote them for this very model (a command line
   * that loads the same model.pkl / .gnx on every start keeps them in a file beside it: preparing the planes is most of what
   * gnx_model_load does).  Checked against a hash of lr_coef, the geometry, the ABI version and the plane settings; anything that
   * does not match -> GNX_ESTALE, nothing loaded.  NULL / 0: prepare from lr_coef. */
  const void* prepared;
  int64_t prepared_bytes;
} gnx_model_desc;

typedef struct gnx_model_info {
  int64_t C, M, ctx, W;
  int32_t A, S, base_kind, smooth_kind;
  int32_t n_trees, tree_depth;
  int64_t device_bytes; /* HBM held by the model */
} gnx_model_info;

int gnx_abi_version(void);
/* how the library was built: bit 0 = `make EXPERIMENTS=1` (the measured-slower kernels parked under scripts/dev/rejected/ are linked in
 * and reachable through their development knobs; the default build does not contain them) */
#define GNX_BUILD_EXPERIMENTS 0x1
int gnx_build_flags(void);
/* GPUs this process sees (HIP_VISIBLE_DEVICES applied); 0 without a usable runtime.  One gnx_ctx per device: gnx_init(d), 0 <= d < count. */
int gnx_device_count(void);

/* context */
int gnx_init(int device, gnx_ctx** out);
void gnx_ctx_free(gnx_ctx* ctx);
const char* gnx_last_error(const gnx_ctx* ctx);
int gnx_set_stream(gnx_ctx* ctx, void* hip_stream); /* borrow a hipStream_t; NULL is HIP's default (null) stream */
int gnx_reset_stream(gnx_ctx* ctx);                 /* back to the context's own non-blocking stream */
int gnx_synchronize(gnx_ctx* ctx);
/* page-locked host memory for the host-pointer entry points: buffers allocated here move over PCIe by DMA at link rate
 * (pageable memory goes through the runtime's staging copies).  The reference hands over numpy arrays (gnomix.py:48-49):
 * a caller that lets its VCF reader fill a buffer from gnx_host_alloc avoids that extra pass. */
int gnx_host_alloc(gnx_ctx* ctx, size_t bytes, void** out);
int gnx_host_free(gnx_ctx* ctx, void* p);
/* allocation flags of a gnx_host_alloc buffer (hipHostGetFlags).  Every buffer is hipHostMallocPortable (bit 0): page-locked for
 * EVERY device of the process, because the one-process multi-GPU file path (gnomix_amd/multi.py) hands the same parsed genotype
 * rows and output arrays to contexts on different devices. */
#define GNX_HOST_PORTABLE 0x1u
int gnx_host_flags(const void* p, unsigned* flags);
/* Device binding.  HIP's "current device" belongs to the calling THREAD (a new thread starts on device 0); every entry point of this
 * library that touches the GPU binds its context's device for the duration of the call and restores the caller's on return, so
 * several contexts may be driven from one thread (torch tensors on several GPUs) as well as one context per thread.
 * Diagnostics for that contract: the device ordinal (hipPointerGetAttributes) of every LIVE device workspace of the context, at most
 * n of them written to out; returns how many workspaces are live (may exceed n), negative on error.  A tripwire for multi-GPU
 * nodes: every entry must equal the device the context was created on (tests/test_gpu_devices.py). */
int gnx_debug_ws_devices(gnx_ctx* ctx, int32_t* out, int32_t n);

/* model */
int gnx_model_load(gnx_ctx* ctx, const gnx_model_desc* desc, gnx_model** out);
void gnx_model_free(gnx_model* model);
int gnx_model_get_info(const gnx_model* model, gnx_model_info* out);
/* smooth.calibrate (gnomix.py:367): when on AND the model carries a calibrator, smoother outputs (probabilities and the
 * labels derived from them) go through Calibrator.transform; without a calibrator the reference prints a notice and
 * returns the original probabilities (smooth.py:48-52) — so does this (GNX_OK, outputs unchanged). */
int gnx_model_set_calibrate(gnx_model* model, int on);
/* the prepared planes of a loaded logistic model as one relocatable blob for gnx_model_desc.prepared: *bytes = its size (0 for models
 * without a logistic base); buf == NULL only reports the size; cap < size -> GNX_EINVAL. */
int gnx_model_export_prepared(gnx_model* model, void* buf, int64_t cap, int64_t* bytes);

/* Base.predict_proba: X (N, ldx>=C) int8 -> B (N, W, A).  Either output may be NULL.
 * b_f32 is what the XGB smoother consumes (src/Smooth/utils.py:20), b_f64 what the reference returns. */
int gnx_base_predict(gnx_model* model, const int8_t* X, int64_t N, int64_t ldx, float* b_f32, double* b_f64);
int gnx_base_predict_dev(gnx_model* model, const int8_t* dX, int64_t N, int64_t ldx, float* d_b_f32, double* d_b_f64);

/* Smoother.predict_proba / predict: B (N, W, A) (float64 if b_is_f64 else float32) ->
 * proba (N, W, A) and labels (N, W) (argmax, first max wins).  Any output may be NULL.
 * XGB computes in float32 (proba_f64 is the widened copy); CRF computes in float64. */
int gnx_smooth_predict(gnx_model* model, const void* B, int b_is_f64, int64_t N, float* proba_f32,
                       double* proba_f64, int32_t* labels);
int gnx_smooth_predict_dev(gnx_model* model, const void* dB, int b_is_f64, int64_t N, float* d_proba_f32,
                           double* d_proba_f64, int32_t* d_labels);

/* Gnomix.predict_proba / predict: base + smoother with B kept on the device. */
int gnx_infer(gnx_model* model, const int8_t* X, int64_t N, int64_t ldx, float* proba_f32, double* proba_f64,
              int32_t* labels);
int gnx_infer_dev(gnx_model* model, const int8_t* dX, int64_t N, int64_t ldx, float* d_proba_f32,
                  double* d_proba_f64, int32_t* d_labels);

/* 2-bit packed haplotypes.  The reference's contract is int8 {0,1,2}, one byte per SNP (src/utils.py:153) and gnx_infer
 * keeps accepting exactly that; whole genome it is 17.7 MB per haplotype, so the 63 GB/s host link bounds the host-pointer
 * path at a few thousand haplotypes/s/GPU.  A caller that can hand over X as 2-bit fields moves a quarter of
 * the bytes: SNP j of a row lives in bits 2*(j%4)..2*(j%4)+1 of byte j/4 (value = the int8 code, 0..3), rows ldp bytes apart
 * (ldp >= ceil(C/4); gnx_packed_row_bytes(C) = the canonical stride, a multiple of 4, its tail zeroed by gnx_pack_x).
 *   gnx_pack_x          host utility: int8 (N, ldx) -> packed (N, ldp) on n_threads host threads (<= 0: all cores, at most
 *                       64); GNX_EINVAL if a value is outside 0..3.  Needs no context and no GPU.
 *   gnx_infer_packed    gnx_infer on packed host input: batches are copied, widened on the device and run through the same
 *                       kernels, H2D / kernels / D2H overlapped on three streams; results are bit-identical to gnx_infer's.
 *   gnx_unpack_x_dev    the widening pass alone, on the context stream (device pointers).
 *   gnx_infer_packed_dev  device-resident packed input.
 *   gnx_base_predict_packed_dev  Base.predict_proba (src/Base/base.py:146-180) on device-resident packed input.
 * With the logistic base (up to 32 class columns per SNP, i.e. A <= 16 at the default context) packed rows are NOT widened:
 * k_base_logistic_p2 reads the 2-bit rows and expands them to the int8 MFMA operand in registers — a quarter of the X bytes
 * through HBM and the L1s; B is bit-identical to the int8 entry points'.  Other bases widen to int8 in device scratch first. */
int64_t gnx_packed_row_bytes(int64_t C);
int gnx_pack_x(const int8_t* X, int64_t N, int64_t ldx, int64_t C, uint8_t* packed, int64_t ldp, int n_threads);
int gnx_unpack_x_dev(gnx_ctx* ctx, const uint8_t* d_packed, int64_t N, int64_t ldp, int64_t C, int8_t* dX, int64_t ldx);
int gnx_infer_packed(gnx_model* model, const uint8_t* packed, int64_t N, int64_t ldp, float* proba_f32, double* proba_f64,
                     int32_t* labels);
int gnx_base_predict_packed_dev(gnx_model* model, const uint8_t* d_packed, int64_t N, int64_t ldp, float* d_b_f32,
                                double* d_b_f64);
int gnx_infer_packed_dev(gnx_model* model, const uint8_t* d_packed, int64_t N, int64_t ldp, float* d_proba_f32,
                         double* d_proba_f64, int32_t* d_labels);

/* smoother.model.predict_proba on explicit rows (R, S*A) float32 -> (R, A) float32 (XGB only). */
int gnx_smooth_rows(gnx_model* model, const float* rows, int64_t R, float* proba);

/* Calibrator.transform on explicit rows (R, A) (float64 if proba_is_f64 else float32) -> (R, A) float64
 * (src/Smooth/Calibration.py:57-69).  GNX_ESTATE when the model carries no calibrator. */
int gnx_calibrate_rows(gnx_model* model, const void* proba, int proba_is_f64, int64_t R, double* out);

/* Gnomix.phase: for each of n_ind individuals (haplotype rows 2i, 2i+1 of X and of B) run the
 * Gnofix loop with the reference's default arguments.  X (2*n_ind, ldx) int8 is re-phased IN
 * PLACE, B (2*n_ind, W, A) float64 is read only, Y (2*n_ind, W) receives Gnofix's labels and
 * n_switches (n_ind,) (may be NULL) the number of accepted switches. */
int gnx_gnofix(gnx_model* model, int8_t* X, int64_t ldx, const double* B, int64_t n_ind, int32_t max_it,
               int32_t* Y, int32_t* n_switches);
int gnx_gnofix_dev(gnx_model* model, int8_t* dX, int64_t ldx, const double* dB, int64_t n_ind, int32_t max_it,
                   int32_t* dY, int32_t* d_n_switches);
/* the same on device-resident 2-bit rows (gnx_pack_x layout; dP 4-byte aligned, ldp a multiple of 4): the SNP blocks of the
 * windows with odd final switch parity are exchanged in the packed rows (a quarter of the bytes of the int8 matrix) */
int gnx_gnofix_packed_dev(gnx_model* model, uint8_t* d_packed, int64_t ldp, const double* dB, int64_t n_ind, int32_t max_it,
                          int32_t* dY, int32_t* d_n_switches);

/* Base.train for the logistic base (src/Base/base.py:104-127 -> per window
 * LogisticRegression(penalty="l2", C=3., solver="liblinear", max_iter=1000).fit(X_w, y_w), src/Base/models.py:12-21; called
 * twice by Gnomix.train, src/model.py:104-167): all W windows x A one-vs-rest problems (A == 2: one problem per window, as
 * sklearn) of  min_w 1/2 w'w + C_reg sum_i log(1 + exp(-y_i w'[x_i, 1]))  minimised at once on the device, float64.
 *   X (N, ldx) int8 {0,1,2}, y (N, W) int32 window labels in [0, A)  (rows = haplotypes)
 *   tol: stop a problem when |grad| <= tol * |grad at w = 0| (liblinear stops at ~1e-4 scaled by the class balance; the
 *        default 1e-9 converges to the optimum that the reference's solver approximates); max_iter bounds Newton steps and is
 *        itself clamped to 200 (a Newton-CG run takes 10-20); GNX_OK is returned even when a problem has not reached tol —
 *        check info.worst_rel_gradient
 *   coef (W, A, ldc) / intercept (W, A): HOST outputs in exactly the layout gnx_model_desc.lr_coef / lr_intercept take
 *        (A == 2: rows (-w, +w), see gnomix_amd.convert.lr_rows_from_sklearn); ldc >= M + 2 ctx + C % M
 * The unsuffixed entry point takes host X / y and stages them; _dev takes device X / y (context's device). */
typedef struct gnx_train_info {
  int32_t newton_iterations, cg_iterations, n_problems, reserved;
  double worst_rel_gradient; /* max over problems of |grad| / |grad at 0| on return */
  double objective_sum;      /* sum over problems of the objective at the returned w */
} gnx_train_info;
int gnx_train_logistic(gnx_ctx* ctx, const int8_t* X, int64_t N, int64_t ldx, const int32_t* y, int64_t C, int64_t M,
                       int64_t ctx_snps, int32_t A, double C_reg, double tol, int32_t max_iter, double* coef, int64_t ldc,
                       double* intercept, gnx_train_info* info);
int gnx_train_logistic_dev(gnx_ctx* ctx, const int8_t* dX, int64_t N, int64_t ldx, const int32_t* dy, int64_t C, int64_t M,
                           int64_t ctx_snps, int32_t A, double C_reg, double tol, int32_t max_iter, double* coef, int64_t ldc,
                           double* intercept, gnx_train_info* info);

/* ---- training the tree smoother: Smoother.train of XGB_Smoother (src/Smooth/smooth.py:28-38, src/Smooth/models.py:14-20:
 *      XGBClassifier(n_estimators=100, max_depth=4, learning_rate=0.1, reg_lambda=1, objective='multi:softprob').fit(slide_window(B), y))
 * Second-order gradient boosting of A regression trees per round on the softmax objective, in the histogram form (<= max_bin
 * quantile bins per class column; gradient sums in fixed point, so the trees do not depend on scheduling and equal the CPU
 * oracle's bit for bit).  xgboost itself is a third-party fitter outside the reference tree: this entry point reproduces the
 * algorithm the call asks for, not xgboost's floating-point trajectory.
 *   B (N, W, A) base probabilities (what Base.predict_proba returned for the smoother's training haplotypes), float32 or float64
 *   y (N, W) int32 labels in [0, A);  W >= 2 S, S odd.  gnx_train_gbt rejects labels outside the range; gnx_train_gbt_dev (arrays
 *     already in HBM) does not read them back to check: a row whose label is outside [0, A) counts as belonging to no class
 *     (labels are only ever compared, never used as an index)
 *   outputs (HOST, caller-allocated): tree_off[T+1], tree_class[T], left / right / feat (int32) and cond (float32) with room for
 *     (2^(max_depth+1) - 1) T nodes (63 T at the limit max_depth = 5), T = n_rounds * A — exactly the arrays gnx_model_desc takes (a leaf has left = right = -1 and its value in cond;
 *     tree t belongs to class t % A); *n_nodes = nodes written; loss[n_rounds + 1] (optional) = mean log loss before each round
 *     and after the last. */
typedef struct gnx_gbt_params {
  int32_t n_rounds;          /* 100  (n_estimators) */
  int32_t max_depth;         /* 4, at most 5 */
  int32_t max_bin;           /* 256, at most 256 */
  int32_t tree_method;       /* 0 = histogram (max_bin quantile bins per class column, xgboost's "hist"); 1 = exact greedy: a candidate
                                between every two distinct feature values of a node's rows (xgboost's "exact"); NaN probabilities sort as +inf */
  double eta;                /* 0.1  (learning_rate) */
  double lambda;             /* 1.0  (reg_lambda) */
  double gamma;              /* 0.0  (min_split_loss) */
  double min_child_weight;   /* 1.0 */
  double base_score;         /* 0.5 */
} gnx_gbt_params;
int gnx_train_gbt(gnx_ctx* ctx, const void* B, int32_t b_is_f64, const int32_t* y, int64_t N, int32_t W, int32_t A, int32_t S,
                  const gnx_gbt_params* params, int32_t* tree_off, int32_t* tree_class, int32_t* left, int32_t* right,
                  int32_t* feat, float* cond, int64_t* n_nodes, double* loss);
int gnx_train_gbt_dev(gnx_ctx* ctx, const void* dB, int32_t b_is_f64, const int32_t* dy, int64_t N, int32_t W, int32_t A, int32_t S,
                      const gnx_gbt_params* params, int32_t* tree_off, int32_t* tree_class, int32_t* left, int32_t* right,
                      int32_t* feat, float* cond, int64_t* n_nodes, double* loss);

/* ---- training the convolutional smoother: CNN.fit (src/Smooth/cnn.py:104-118) as Smoother.train calls it for CNN_Smoother
 *      (src/Smooth/smooth.py:28-38, src/Smooth/models.py:35-42).  nn.Conv1d(A, A, S, padding = (S-1)/2) with zero padding,
 *      loss = NLLLoss(log(softmax + log_eps), y) averaged over the batch's rows x windows (cnn.py:57-75), torch.optim.Adam
 *      (cnn.py:32), mini-batches of `batch` rows in the order `order` gives for each epoch (the DataLoader's shuffle, cnn.py:172;
 *      NULL = 0 .. N-1 every epoch).  All arithmetic float32.
 *      B (N, W, A) float32 / float64 and y (N, W) on the host; weight (A, A, S) and bias (A) hold the initial parameters on
 *      entry (torch's default: uniform(+-1/sqrt(A*S))) and the trained ones on return; loss (epochs) receives each epoch's mean
 *      batch loss (the number cnn.py:120 prints) or may be NULL. */
typedef struct gnx_cnn_params {
  int32_t epochs;            /* 250  (max_ep) */
  int32_t batch;             /* 128  (DataLoader batch_size) */
  double lr;                 /* 1e-3 (Adam) */
  double beta1, beta2, eps;  /* 0.9, 0.999, 1e-8 (Adam defaults) */
  double log_eps;            /* 1e-8 (cnn.py:73) */
} gnx_cnn_params;
int gnx_train_cnn(gnx_ctx* ctx, const void* B, int32_t b_is_f64, const int32_t* y, int64_t N, int32_t W, int32_t A, int32_t S,
                  const gnx_cnn_params* params, const int64_t* order, float* weight, float* bias, double* loss);

/* ---- training the linear-chain CRF smoother: CRF.fit (src/Smooth/crf.py:51-58: sklearn_crfsuite.CRF(algorithm="lbfgs",
 *      max_iterations=10000, all_possible_transitions=True, all_possible_states=True)) as Smoother.train calls it for
 *      CRF_Smoother (src/Smooth/smooth.py:28-38, src/Smooth/models.py:27-32).  Minimises CRFsuite's objective
 *      f = - sum log p(y | x) + c2 |w|^2 over state (attribute, label) and transition (from, to) weights — the arrays
 *      gnx_model_desc.crf_state / crf_trans take — by L-BFGS with every evaluation on the device (k_train_crf.hip).
 *      state / trans (A, A) hold the starting point on entry (CRFsuite starts from zeros) and the fit on return. */
typedef struct gnx_crf_params {
  double c1;                 /* 0.0  (L1: only 0 is built) */
  double c2;                 /* 1.0  (L2, CRFsuite's default) */
  double epsilon;            /* 1e-8: stop when |g| / max(1, |w|) < epsilon (CRFsuite: 1e-5) */
  int32_t max_iterations;    /* 10000 (crf.py:7) */
  int32_t memory;            /* 10: L-BFGS pairs kept (CRFsuite: 6) */
} gnx_crf_params;
typedef struct gnx_crf_info {
  int32_t iterations, evaluations;
  double objective, grad_norm;
  int32_t converged, reserved;
} gnx_crf_info;
int gnx_train_crf(gnx_ctx* ctx, const void* B, int32_t b_is_f64, const int32_t* y, int64_t N, int32_t W, int32_t A,
                  const gnx_crf_params* params, double* state, double* trans, gnx_crf_info* info);

/* ---- one isotonic map of the calibrator: Calibrator.fit (src/Smooth/Calibration.py:43-55) fits, per class i,
 *      sklearn IsotonicRegression(out_of_bounds='clip') on (proba[:, i], y == class i) with float32 probabilities.
 * Host arithmetic (no context, no device): x, y (n,) float32 in any order -> thresholds x_thr / y_thr (caller-allocated, n each),
 * *n_thr of them; they go into gnx_model_desc.calib_x / calib_y (as float64) with calib_is_f32 = 1. */
int gnx_fit_isotonic_f32(const float* x, const float* y, int64_t n, float* x_thr, float* y_thr, int64_t* n_thr);

/* per-kernel device time, measured with hipEvents on the context stream around every launch */
int gnx_profile_enable(gnx_ctx* ctx, int on);
int gnx_profile_reset(gnx_ctx* ctx);
int gnx_profile_get(gnx_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* GNOMIX_HIP_H */

"""Training on the device.

Tree smoother: `Smoother.train` of `XGB_Smoother` (reference src/Smooth/smooth.py:28-38, src/Smooth/models.py:14-20, called
from Gnomix.train, src/model.py:117) — gnx_train_gbt, histogram gradient boosting with fixed-point sums (k_train_gbt.hip).

Logistic base: the device side of `Base.train` for
`LogisticRegressionBase` (reference src/Base/base.py:104-127, src/Base/models.py:12-21, called from Gnomix.train,
src/model.py:113 and :155).  gnx_train_logistic minimises liblinear's L2-regularised logistic objective for all
W windows x A one-vs-rest problems at once (k_train_lr.hip); this module is the ctypes call and the glue that turns the
result into a GnxModelData / a fresh device model."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .model import GnxModelData


def train_logistic_arrays(X, y, M, context, A, C_reg=3.0, tol=1e-9, max_iter=1000, ctx=None, device=0):
    """X (N, C) int8 {0,1,2}, y (N, W) window labels -> (lr_coef (W, A, ldc) f64, lr_intercept (W, A) f64, info dict).
    Defaults are the reference's (C=3., max_iter=1000); tol is OUR stopping rule |grad| <= tol |grad(0)| (the reference's
    liblinear stops near 1e-4: the default here converges to the optimum it approximates)."""
    ctx = ctx or _lib.default_context(device)
    X = np.ascontiguousarray(X, dtype=np.int8)
    N, Cn = X.shape
    W = Cn // int(M)
    y = np.ascontiguousarray(y, dtype=np.int32)
    if y.shape != (N, W):
        raise ValueError(f"y must be (N, W) = ({N}, {W}), got {y.shape}")
    if y.min() < 0 or y.max() >= A:
        raise ValueError("labels must lie in [0, A)")
    ldc = int(M) + 2 * int(context) + (Cn - int(M) * W)
    coef = np.zeros((W, int(A), ldc), np.float64)
    icpt = np.zeros((W, int(A)), np.float64)
    info = _lib.TrainInfo()
    ctx.check(ctx.lib.gnx_train_logistic(ctx.h, X.ctypes.data, N, X.shape[1], y.ctypes.data, Cn, int(M), int(context), int(A),
                                         float(C_reg), float(tol), int(max_iter), coef.ctypes.data, ldc, icpt.ctypes.data, C.byref(info)))
    out = dict(newton_iterations=info.newton_iterations, cg_iterations=info.cg_iterations, n_problems=info.n_problems,
               worst_rel_gradient=info.worst_rel_gradient, objective_sum=info.objective_sum)
    # the library bounds the Newton steps (at most 200 whatever max_iter says: include/gnomix_hip.h) and returns what it has: a
    # window that did not get near liblinear's own stopping level (1e-4) must not pass as a trained model silently
    if info.worst_rel_gradient > max(float(tol), 1e-4):
        import warnings
        warnings.warn("gnx_train_logistic stopped after %d Newton steps with |grad| / |grad(0)| = %.3g on its worst window "
                      "(tol %.3g): the logistic base is not converged" % (info.newton_iterations, info.worst_rel_gradient, tol),
                      RuntimeWarning, stacklevel=2)
    return coef, icpt, out


def train_logistic_base(data: GnxModelData, X, y, **kw) -> dict:
    """fit the logistic base of `data` in place (lr_coef / lr_intercept) -> info"""
    coef, icpt, info = train_logistic_arrays(X, y, data.M, data.context, data.A, **kw)
    data.base_kind, data.lr_coef, data.lr_intercept = "logistic", coef, icpt
    return info


def lr_objective(coef_row, intercept, Xw, ypm, C_reg=3.0):
    """liblinear's primal objective of ONE binary problem: 1/2 (|w|^2 + b^2) + C sum log(1 + exp(-y (w.x + b))) — numpy, for
    tests and for judging a fit against the reference's (the bias is a regularised feature: intercept_scaling = 1)"""
    z = Xw.astype(np.float64) @ coef_row + intercept
    return 0.5 * (float(coef_row @ coef_row) + float(intercept) ** 2) + C_reg * float(np.sum(np.logaddexp(0.0, -ypm * z)))


def train_gbt_arrays(B, y, S, n_rounds=100, max_depth=4, learning_rate=0.1, reg_lambda=1.0, gamma=0.0, min_child_weight=1.0,
                     max_bin=256, base_score=0.5, tree_method="hist", ctx=None, device=0):
    """B (N, W, A) base probabilities of the smoother's training haplotypes (float32 / float64; a CUDA tensor stays on the
    device), y (N, W) labels -> (dict of tree arrays as GnxModelData takes them, losses (n_rounds + 1,)).  Defaults are the
    reference's XGBClassifier arguments (src/Smooth/models.py:14-20).  tree_method: "hist" (max_bin quantile bins per class column) or
    "exact" (xgboost's exact greedy enumeration: a candidate between every two distinct values of a node's rows; slower, and like the
    histogram form not pinned to xgboost's own floating-point trajectory)."""
    if tree_method not in ("hist", "exact"):
        raise ValueError("tree_method is 'hist' or 'exact'")
    ctx = ctx or _lib.default_context(device)
    on_dev = hasattr(B, "is_cuda") and B.is_cuda
    if on_dev:
        import torch
        assert B.is_contiguous() and B.dtype in (torch.float32, torch.float64)
        N, W, A = B.shape
        is64 = B.dtype == torch.float64
        y = y if (hasattr(y, "is_cuda") and y.is_cuda) else torch.as_tensor(np.ascontiguousarray(y, dtype=np.int32), device=B.device)
        assert y.dtype == torch.int32 and y.is_contiguous() and tuple(y.shape) == (N, W)
        b_ptr, y_ptr, fn = B.data_ptr(), y.data_ptr(), ctx.lib.gnx_train_gbt_dev
        ctx.set_stream(torch.cuda.current_stream(ctx.device).cuda_stream)
    else:
        B = np.ascontiguousarray(B)
        if B.dtype != np.float64:
            B = np.ascontiguousarray(B, dtype=np.float32)
        N, W, A = B.shape
        is64 = B.dtype == np.float64
        y = np.ascontiguousarray(y, dtype=np.int32)
        if y.shape != (N, W):
            raise ValueError(f"y must be (N, W) = ({N}, {W}), got {y.shape}")
        b_ptr, y_ptr, fn = B.ctypes.data, y.ctypes.data, ctx.lib.gnx_train_gbt
    T = int(n_rounds) * int(A)
    # a complete tree of that depth (the library rejects depths outside 1..5 itself: include/gnomix_hip.h, gnx_gbt_params)
    per_tree = 2 ** (min(max(int(max_depth), 1), 5) + 1) - 1
    tree_off = np.zeros(T + 1, np.int32); tree_class = np.zeros(T, np.int32)
    left = np.zeros(T * per_tree, np.int32); right = np.zeros(T * per_tree, np.int32); feat = np.zeros(T * per_tree, np.int32)
    cond = np.zeros(T * per_tree, np.float32); loss = np.zeros(int(n_rounds) + 1, np.float64)
    nn = C.c_int64(0)
    P = _lib.GbtParams(int(n_rounds), int(max_depth), int(max_bin), 1 if tree_method == "exact" else 0, float(learning_rate), float(reg_lambda), float(gamma),
                       float(min_child_weight), float(base_score))
    ctx.check(fn(ctx.h, b_ptr, int(is64), y_ptr, int(N), int(W), int(A), int(S), C.byref(P), tree_off.ctypes.data, tree_class.ctypes.data,
                 left.ctypes.data, right.ctypes.data, feat.ctypes.data, cond.ctypes.data, C.addressof(nn), loss.ctypes.data))
    n = nn.value
    trees = dict(tree_off=tree_off, left=left[:n].copy(), right=right[:n].copy(), feat=feat[:n].copy(), cond=cond[:n].copy(),
                 tree_class=tree_class)
    return trees, loss


def train_gbt_smoother(data: GnxModelData, B, y, **kw) -> np.ndarray:
    """fit the tree smoother of `data` in place (smooth_kind "xgb", tree arrays, base_score) -> losses per round"""
    trees, loss = train_gbt_arrays(B, y, data.S, **kw)
    data.smooth_kind = "xgb"
    for k, v in trees.items():
        setattr(data, k, v)
    data.base_score = float(kw.get("base_score", 0.5))
    return loss


def cnn_init(A, S, seed=None):
    """nn.Conv1d(A, A, S)'s default initialisation (torch.nn.modules.conv._ConvNd.reset_parameters): kaiming_uniform_(a = sqrt 5)
    on the weight and uniform(+-1/sqrt(fan_in)) on the bias are both uniform(+-1/sqrt(A * S)); numpy's generator, not torch's"""
    rng = np.random.RandomState(seed)
    bound = 1.0 / np.sqrt(A * S)
    return (rng.uniform(-bound, bound, size=(A, A, S)).astype(np.float32), rng.uniform(-bound, bound, size=A).astype(np.float32))


def train_cnn_arrays(B, y, S, weight=None, bias=None, max_ep=250, batch_size=128, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, log_eps=1e-8,
                     shuffle=True, seed=None, order=None, ctx=None, device=0):
    """CNN.fit (src/Smooth/cnn.py:104-118) on the device: B (N, W, A) base probabilities, y (N, W) labels -> (weight (A, A, S),
    bias (A,), per-epoch mean batch loss).  `weight` / `bias` = initial parameters (default: cnn_init(A, S, seed));
    `order` (max_ep, N) = the rows' order in every epoch (default: a fresh permutation per epoch when `shuffle`, as the
    reference's DataLoader(shuffle=True), drawn from numpy's RandomState(seed))."""
    ctx = ctx or _lib.default_context(device)
    B = np.ascontiguousarray(B)
    if B.dtype != np.float64:
        B = np.ascontiguousarray(B, dtype=np.float32)
    N, W, A = B.shape
    y = np.ascontiguousarray(y, dtype=np.int32)
    if y.shape != (N, W):
        raise ValueError(f"y must be (N, W) = ({N}, {W}), got {y.shape}")
    if weight is None or bias is None:
        weight, bias = cnn_init(A, S, seed)
    weight = np.array(weight, dtype=np.float32, order="C")
    bias = np.array(bias, dtype=np.float32, order="C")
    if weight.shape != (A, A, S) or bias.shape != (A,):
        raise ValueError(f"weight / bias must be ({A}, {A}, {S}) / ({A},), got {weight.shape} / {bias.shape}")
    if order is None and shuffle:
        rng = np.random.RandomState(None if seed is None else seed + 1)
        order = np.stack([rng.permutation(N) for _ in range(int(max_ep))]) if max_ep > 0 else np.zeros((0, N), np.int64)
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.int64)
        if order.shape != (int(max_ep), N):
            raise ValueError(f"order must be (max_ep, N) = ({int(max_ep)}, {N}), got {order.shape}")
    loss = np.zeros(max(int(max_ep), 1), np.float64)
    P = _lib.CnnParams(int(max_ep), int(batch_size), float(lr), float(betas[0]), float(betas[1]), float(eps), float(log_eps))
    ctx.check(ctx.lib.gnx_train_cnn(ctx.h, B.ctypes.data, int(B.dtype == np.float64), y.ctypes.data, int(N), int(W), int(A), int(S), C.byref(P),
                                    order.ctypes.data if order is not None else None, weight.ctypes.data, bias.ctypes.data, loss.ctypes.data))
    return weight, bias, loss[:int(max_ep)]


def train_cnn_smoother(data: GnxModelData, B, y, **kw) -> np.ndarray:
    """fit the convolutional smoother of `data` in place (smooth_kind "cnn", cnn_weight / cnn_bias) -> losses per epoch"""
    w, b, loss = train_cnn_arrays(B, y, data.S if data.S % 2 else data.S - 1, **kw)
    data.smooth_kind, data.cnn_weight, data.cnn_bias = "cnn", w, b
    return loss


def train_crf_arrays(B, y, c2=1.0, epsilon=1e-8, max_iterations=10000, memory=10, state0=None, trans0=None, ctx=None, device=0):
    """CRF.fit (src/Smooth/crf.py:51-58) on the device: B (N, W, A) base probabilities (the attributes' values), y (N, W) labels ->
    (state (A, A) [attribute][label], trans (A, A) [from][to], info dict).  Minimises CRFsuite's L2-regularised negative
    log-likelihood (c1 = 0, c2 = 1 as sklearn_crfsuite.CRF's defaults) from zeros, like CRFsuite, to a tighter tolerance."""
    ctx = ctx or _lib.default_context(device)
    B = np.ascontiguousarray(B)
    if B.dtype != np.float64:
        B = np.ascontiguousarray(B, dtype=np.float32)
    N, W, A = B.shape
    y = np.ascontiguousarray(y, dtype=np.int32)
    if y.shape != (N, W):
        raise ValueError(f"y must be (N, W) = ({N}, {W}), got {y.shape}")
    state = np.zeros((A, A), np.float64) if state0 is None else np.array(state0, dtype=np.float64, order="C")
    trans = np.zeros((A, A), np.float64) if trans0 is None else np.array(trans0, dtype=np.float64, order="C")
    if state.shape != (A, A) or trans.shape != (A, A):
        raise ValueError(f"state0 / trans0 must be ({A}, {A})")
    P = _lib.CrfParams(0.0, float(c2), float(epsilon), int(max_iterations), int(memory))
    info = _lib.CrfInfo()
    ctx.check(ctx.lib.gnx_train_crf(ctx.h, B.ctypes.data, int(B.dtype == np.float64), y.ctypes.data, int(N), int(W), int(A), C.byref(P),
                                    state.ctypes.data, trans.ctypes.data, C.byref(info)))
    return state, trans, {"iterations": info.iterations, "evaluations": info.evaluations, "objective": info.objective,
                          "grad_norm": info.grad_norm, "converged": bool(info.converged)}


def train_crf_smoother(data: GnxModelData, B, y, **kw) -> dict:
    """fit the CRF smoother of `data` in place (smooth_kind "crf", crf_state / crf_trans) -> info"""
    st, tr, info = train_crf_arrays(B, y, **kw)
    data.smooth_kind, data.crf_state, data.crf_trans = "crf", st, tr
    return info

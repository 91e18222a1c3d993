"""Training the per-window logistic base on the device: the device side of `Base.train` for
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
    return coef, icpt, dict(newton_iterations=info.newton_iterations, cg_iterations=info.cg_iterations, n_problems=info.n_problems,
                            worst_rel_gradient=info.worst_rel_gradient, objective_sum=info.objective_sum)


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

"""Fitting the calibrator: `Calibrator.fit` (reference src/Smooth/Calibration.py:43-55) = one
sklearn IsotonicRegression(out_of_bounds="clip") per class on (proba[:, i], y == class i), called from
`Smoother.train_calibrator` (src/Smooth/smooth.py:81-92) on the smoother's probabilities of a 5 % sample of the haplotypes.
The isotonic fit itself is gnx_fit_isotonic_f32 (host arithmetic in the library: sort, merge, pool adjacent violators)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def fit_isotonic(x, y):
    """x, y (n,) -> (X_thresholds_, y_thresholds_) float32, as IsotonicRegression(out_of_bounds="clip").fit on float32 inputs"""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    y = np.ascontiguousarray(y, dtype=np.float32).reshape(-1)
    if x.shape != y.shape or x.size == 0:
        raise ValueError("fit_isotonic: x and y must be non-empty and of one length")
    xt, yt = np.empty_like(x), np.empty_like(x)
    n = C.c_int64(0)
    rc = _lib.load().gnx_fit_isotonic_f32(x.ctypes.data, y.ctypes.data, x.size, xt.ctypes.data, yt.ctypes.data, C.addressof(n))
    if rc != 0:
        raise _lib.GnxError(rc, "gnx_fit_isotonic_f32 failed")
    return xt[:n.value].copy(), yt[:n.value].copy()


def fit_calibrator(proba, y, n_classes):
    """proba (R, A) smoother probabilities (cast to float32 like the xgb smoother's), y (R,) labels -> the calib_* fields of
    GnxModelData.  Column i is fitted against the i-th class of sorted(unique(y)) (OneHotEncoder's order, Calibration.py:51-52);
    every class must occur."""
    proba = np.asarray(proba, dtype=np.float32).reshape(-1, n_classes)
    y = np.asarray(y).reshape(-1)
    classes = np.unique(y)
    if len(classes) != n_classes:
        raise ValueError("calibrator training data does not include all populations")
    off, xs, ys = [0], [], []
    for i in range(n_classes):
        xt, yt = fit_isotonic(proba[:, i], (y == classes[i]).astype(np.float32))
        xs.append(xt.astype(np.float64)); ys.append(yt.astype(np.float64))
        off.append(off[-1] + len(xt))
    return dict(calib_off=np.array(off, np.int32), calib_x=np.concatenate(xs), calib_y=np.concatenate(ys), calib_is_f32=True)

"""HipSmoother — the reference's Smoother plugin interface (src/Smooth/smooth.py:7-65) served by the
HIP kernels.  `.model.predict_proba(rows)` is kept because Gnofix calls it (gnofix.py:157)."""
from __future__ import annotations

from time import time

import numpy as np


class _RowModel:
    """stands where XGBClassifier stands: predict_proba on explicit (R, S*A) rows"""

    def __init__(self, dev):
        self.dev = dev

    def predict_proba(self, rows):
        rows = np.asarray(rows)
        return self.dev.smooth_rows(rows.reshape(rows.shape[0], -1))


class HipSmoother:

    def __init__(self, device_model, calibrate=None, mode_filter=0, n_jobs=None, seed=None, verbose=False):
        d = device_model.data
        self._calibrate = None
        self.dev = device_model
        self.W = d.C // d.M
        self.A = d.A
        self.S = d.S if d.S % 2 else d.S - 1   # smooth.py:14
        self._calibrate = None
        self.calibrator = d.calib_off is not None or None   # truthy when the model carries fitted isotonic maps
        self.calibrate = calibrate
        self.mode_filter = mode_filter
        self.n_jobs = n_jobs
        self.seed = seed
        self.verbose = verbose
        self.gnofix = d.smooth_kind == "xgb"   # only XGB_Smoother sets it (Smooth/models.py:12)
        self.model = _RowModel(device_model) if d.smooth_kind == "xgb" else None
        self.time = {}

    @property
    def dev(self):
        return self._dev

    @dev.setter
    def dev(self, m):
        """every (re)binding of the device model — training builds fresh ones — carries the calibrate switch over: a new
        gnx_model starts with calibration off (the reference keeps calibrating after Gnomix.train's base retrain, model.py:119-167)"""
        self._dev = m
        m.set_calibrate(bool(self._calibrate))

    @property
    def calibrate(self):
        return self._calibrate

    @calibrate.setter
    def calibrate(self, on):  # gnomix.py:367 pokes this attribute after loading the model
        self._calibrate = on
        self.dev.set_calibrate(bool(on))

    def train(self, B, y, **kw):
        """Smoother.train (smooth.py:28-38) for the tree smoother: gradient boosting on the device with the reference's
        XGBClassifier arguments (Smooth/models.py:14-20), then the device model is swapped for the freshly trained one.
        B (N, W, A) base probabilities of the smoother's training haplotypes, y (N, W) labels."""
        from .train import train_gbt_smoother, train_cnn_smoother, train_crf_smoother
        from .model import DeviceModel
        y = np.asarray(y)
        assert len(np.unique(y)) == self.A, "Smoother training data does not include all populations"   # smooth.py:30
        if self.dev.data.smooth_kind == "cnn":   # CNN_Smoother: CNN.fit (Smooth/cnn.py:104-118) on the device
            t = time()
            self.train_loss = train_cnn_smoother(self.dev.data, B, y.reshape(np.asarray(B).shape[0], -1), ctx=self.dev.ctx, **kw)
            self.dev = DeviceModel(self.dev.data, ctx=self.dev.ctx)
            self.time["train"] = time() - t
            return self
        if self.dev.data.smooth_kind == "crf":   # CRF_Smoother: CRFsuite's objective, L-BFGS with device evaluations
            t = time()
            self.train_info = train_crf_smoother(self.dev.data, B, y.reshape(np.asarray(B).shape[0], -1), ctx=self.dev.ctx, **kw)
            if not self.train_info.get("converged", True):   # as the logistic base does: a fit that stopped early must not pass silently
                import warnings
                warnings.warn("gnx_train_crf stopped after %s iterations without reaching its gradient tolerance: the CRF smoother is not "
                              "converged" % self.train_info.get("iterations", "?"), RuntimeWarning, stacklevel=2)
            self.dev = DeviceModel(self.dev.data, ctx=self.dev.ctx)
            self.time["train"] = time() - t
            return self
        t = time()
        self.train_loss = train_gbt_smoother(self.dev.data, B, y.reshape(B.shape[0], -1), ctx=self.dev.ctx, **kw)
        self.dev = DeviceModel(self.dev.data, ctx=self.dev.ctx)   # (a HipGnomix re-binds base and fused path: HipGnomix.train_smoother)
        self.gnofix = True
        self.model = _RowModel(self.dev)
        self.time["train"] = time() - t
        return self

    def train_calibrator(self, B, y, frac=0.05):
        """Smoother.train_calibrator (smooth.py:81-92): uncalibrated probabilities of a random `frac` of the haplotypes (numpy's
        global generator, as the reference), one isotonic map per class (gnomix_amd.calibrate), device model re-loaded with them"""
        from .calibrate import fit_calibrator
        from .model import DeviceModel
        if self.dev.data.smooth_kind != "xgb":
            # Calibrator.fit on a float64 smoother (CRF) fits sklearn's isotonic maps on float64 inputs without the float32
            # tie merging gnx_fit_isotonic_f32 reproduces: not built, and not silently approximated
            raise NotImplementedError("train_calibrator is built for the tree smoother (float32 probabilities)")
        B = np.asarray(B)
        y = np.asarray(y)
        calibrate = self.calibrate
        self.calibrate = False
        idxs = np.random.choice(len(B), int(frac * len(B)), replace=False)
        proba = self.predict_proba(B[idxs]).reshape(-1, self.A)
        for k, v in fit_calibrator(proba, y[idxs].reshape(-1), self.A).items():
            setattr(self.dev.data, k, v)
        self.dev = DeviceModel(self.dev.data, ctx=self.dev.ctx)   # (a HipGnomix re-binds: HipGnomix.train)
        if self.model is not None:
            self.model.dev = self.dev
        self.calibrator = True
        self.calibrate = calibrate
        return self

    def evaluate(self, B=None, y=None, y_pred=None):
        """(accuracy %, balanced accuracy %) rounded to two decimals (smooth.py:67-79)"""
        from .metrics import accuracy_pair
        if B is not None:
            y_pred = self.predict(B)
        elif y_pred is None:
            raise ValueError("Need either Base probabilities or y predictions.")
        return accuracy_pair(y, y_pred)

    def predict_proba(self, B):
        """B (N, W, A) -> (N, W, A): float32 for the xgb smoother, float64 for crf (smooth.py:40-56)."""
        t = time()
        if self.calibrate:
            if self.calibrator is None:
                print("No calibrator found, returning original probabilities.")  # smooth.py:49-50
        proba, _ = self.dev.smooth_predict(B, want_proba=True, want_labels=False)
        self.time["inference"] = time() - t
        return proba.reshape(-1, self.W, self.A)

    def predict(self, B):
        """argmax labels (N, W) int64, first max wins (smooth.py:58-65)."""
        if self.mode_filter:
            raise NotImplementedError("mode_filter != 0 is out of scope (default 0; broken on scipy>=1.11 in the reference)")
        _, lab = self.dev.smooth_predict(B, want_proba=False, want_labels=True)
        return lab.astype(np.int64)

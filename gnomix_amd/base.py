"""HipBase — the reference's Base plugin interface (src/Base/base.py:8-27, 129-180, 214-216) served
by the HIP kernels.  Same attribute names (C, M, W, A, context, n_jobs, vectorize, time) and the
same method contracts, so run_inference()/Gnomix.predict read unchanged (gnomix.py:55)."""
from __future__ import annotations

from time import time

import numpy as np


class HipBase:

    def __init__(self, device_model, n_jobs=None, verbose=False):
        d = device_model.data
        self.dev = device_model
        self.C = d.C
        self.M = d.M
        self.W = d.C // d.M
        self.A = d.A
        self.context = d.context
        self.missing_encoding = 2
        self.n_jobs = n_jobs
        self.verbose = verbose
        self.vectorize = True          # poked by gnomix.py:370; the device path is always "vectorized"
        self.base_multithread = False
        self.log_inference = False
        self.time = {}

    def train(self, X, y, verbose=False):
        """Base.train (base.py:104-127) for the logistic base: fits every window's LogisticRegression on the device and swaps
        the device model for the freshly trained one.  X (N, C) int8, y (N, W) window labels."""
        from .train import train_logistic_base
        from .model import DeviceModel
        if self.dev.data.base_kind not in (None, "logistic"):
            raise NotImplementedError("on-device training is built for the logistic base (LogisticRegressionBase)")
        t = time()
        self.train_info = train_logistic_base(self.dev.data, X, y, ctx=self.dev.ctx)
        self.dev = DeviceModel(self.dev.data, ctx=self.dev.ctx)   # (a HipGnomix re-binds its smoother: HipGnomix.train_base)
        self.time["train"] = time() - t
        return self

    def evaluate(self, X=None, y=None, B=None):
        """(accuracy %, balanced accuracy %) rounded to two decimals, from SNPs or from base probabilities (base.py:214-228)"""
        from .metrics import accuracy_pair
        if X is not None:
            y_pred = self.predict(X)
        elif B is not None:
            y_pred = np.argmax(B, axis=-1)
        else:
            raise ValueError("Need either SNP input or estimated probabilities to evaluate.")
        return accuracy_pair(y, y_pred)

    def predict_proba(self, X):
        """X (N, C) int8-like -> B (N, W, A) float64, as Base.predict_proba (base.py:129-180)."""
        t = time()
        _, B = self.dev.base_predict(X, want_f32=False, want_f64=True)
        self.time["inference"] = time() - t
        return B

    def predict_proba_f32(self, X):
        """float32(B): what the XGB smoother actually consumes (Smooth/utils.py:20)."""
        b32, _ = self.dev.base_predict(X, want_f32=True, want_f64=False)
        return b32

    def predict(self, X):
        return np.argmax(self.predict_proba(X), axis=-1)  # base.py:214-216

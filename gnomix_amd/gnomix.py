"""HipGnomix — the inference surface of the reference's model object (src/model.py:12-214):
predict / predict_proba / phase and the attributes run_inference() and the writers read
(C, M, A, S, W, context, snp_pos, snp_ref, snp_alt, population_order, gen_map_df, base, smooth)."""
from __future__ import annotations

import sys

import numpy as np

from .base import HipBase
from .model import DeviceModel, GnxModelData
from .smooth import HipSmoother


class HipGnomix:

    def __init__(self, data: GnxModelData, device: int = 0, ctx=None, calibrate=False, verbose=False, prepared=None):
        self.data = data
        self.dev = DeviceModel(data, ctx=ctx, device=device, prepared=prepared)
        self.C, self.M, self.A, self.S = data.C, data.M, data.A, data.S
        self.W = self.C // self.M
        self.context = data.context
        self.snp_pos, self.snp_ref, self.snp_alt = data.snp_pos, data.snp_ref, data.snp_alt
        self.population_order = data.population_order
        self.calibrate = calibrate
        self.n_cores = None
        self.verbose = verbose
        self.gen_map_df = {}
        if data.gen_map_pos is not None:
            try:
                import pandas as pd
                self.gen_map_df = pd.DataFrame({"pos": data.gen_map_pos, "pos_cm": data.gen_map_cm})
            except ImportError:
                self.gen_map_df = {"pos": data.gen_map_pos, "pos_cm": data.gen_map_cm}
        self.base = HipBase(self.dev, verbose=verbose)
        self.smooth = HipSmoother(self.dev, calibrate=calibrate, verbose=verbose)
        self.time = {}

    @classmethod
    def load(cls, path, **kw):
        return cls(GnxModelData.load(path), **kw)

    def save(self, path):
        """write the model (base, smoother, calibrator, metadata) as a .gnx archive: what Gnomix.save pickles (src/model.py:100-102)"""
        self.dev.data.save(path)
        return path

    def train_base(self, X, y):
        """the base half of Gnomix.train (src/model.py:113, 155): fit the logistic base on the device, then re-bind base,
        smoother and the fused path to the freshly loaded model"""
        self.base.train(X, y)
        self.dev = self.base.dev
        self.smooth.dev = self.dev
        if self.smooth.model is not None:
            self.smooth.model.dev = self.dev
        return self

    def train_smoother(self, B, y, **kw):
        """the smoother half of Gnomix.train (src/model.py:116-117): fit the tree smoother on the device, re-bind everything"""
        self.smooth.train(B, y, **kw)
        self.dev = self.smooth.dev
        self.base.dev = self.dev
        return self

    def conf_matrix(self, y, y_pred):
        from .metrics import confusion
        return confusion(y, y_pred)

    def _score_splits(self, splits):
        """The bookkeeping half of Gnomix.train (src/model.py:127-151), one split at a time.  The reference scores the base on a
        split's own haplotypes and the smoother on labels predicted from base probabilities: for "train" those are two different
        sets (base: train1; smoother: train2, whose probabilities `train` already holds), for "val" one.  Keys as the reference
        stores them: <base|smooth>_<split>_acc[_bal] in `accuracies`, the split's name in `Confusion_Matrices`."""
        acc, cms = {}, {}
        for name, parts in splits.items():
            if parts is None:
                continue
            (X, y), smoother_part = parts
            B = self.base.predict_proba(X)
            labels = self.smooth.predict(B)
            B_s, y_s = smoother_part if smoother_part is not None else (B, y)
            labels_s = labels if smoother_part is None else self.smooth.predict(B_s)
            acc["base_%s_acc" % name], acc["base_%s_acc_bal" % name] = self.base.evaluate(X=None, y=y, B=B)
            acc["smooth_%s_acc" % name], acc["smooth_%s_acc_bal" % name] = self.smooth.evaluate(B=None, y=y_s, y_pred=labels_s)
            cms[name] = self.conf_matrix(y=y, y_pred=labels)
        return acc, cms

    def train(self, data, retrain_base=True, evaluate=True, verbose=False, **smoother_kw):
        """Gnomix.train (src/model.py:104-167) on the device: base on train1, smoother on the base's probabilities of train2,
        the reference's accuracies / confusion matrices, base again on all the data.
        data = ((X_t1, y_t1), (X_t2, y_t2), (X_v, y_v)) with X_v possibly None."""
        from time import time
        t0 = time()
        (X_t1, y_t1), (X_t2, y_t2), (X_v, y_v) = data
        self.train_base(X_t1, y_t1)
        B_t2 = self.base.predict_proba(X_t2)
        self.train_smoother(B_t2, y_t2, **smoother_kw)
        if self.calibrate:   # src/model.py:119-124: balanced w.r.t. the train1 class distribution
            self.smooth.train_calibrator(self.base.predict_proba(X_t1), y_t1)
            self.dev = self.smooth.dev
            self.base.dev = self.dev
        if evaluate:
            self.accuracies, self.Confusion_Matrices = self._score_splits(
                {"train": ((X_t1, y_t1), (B_t2, y_t2)), "val": None if X_v is None else ((X_v, y_v), None)})
        if retrain_base:
            parts = [(X_t1, y_t1), (X_t2, y_t2)] + ([(X_v, y_v)] if X_v is not None else [])
            self.train_base(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]))
        self.time["training"] = round(time() - t0, 2)
        return self

    def predict(self, X):
        """labels (N, W) — base + smoother fused on the device, B never leaves HBM (model.py:169-173)"""
        _, lab = self.dev.infer(X, want_proba=False, want_labels=True)
        return lab.astype(np.int64)

    def predict_proba(self, X):
        """probabilities (N, W, A) (model.py:175-179)"""
        p, _ = self.dev.infer(X, want_proba=True, want_labels=False)
        return p

    def phase(self, X, B=None, verbose=False):
        """Gnofix re-phasing (model.py:188-214): -> X_phased (N, C) int, Y_phased (N, W) int"""
        assert self.smooth is not None, "Smoother is not trained, returning original haplotypes"
        assert self.smooth.gnofix, "Type of Smoother ({}) does not currently support re-phasing".format(self.smooth)
        X = np.asarray(X)
        if B is None:
            B = self.base.predict_proba(X)
        Xp, Y, _ = self.dev.gnofix(X, B)
        return Xp.astype(int), Y.astype(int)

"""Flat model container (".gnx") + the device model handle.

The reference's model artefact is a pickle of src.model.Gnomix holding sklearn / xgboost objects
(gnomix.py:26-35, 209).  GnxModelData is the same information as plain arrays — everything
Gnomix.predict / predict_proba / phase and the writers consume (src/model.py:28-88) — so it can be
saved without pickling third-party classes and handed to the C ABI (gnx_model_desc) as is.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib

GNX_FILE_VERSION = 1


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


@dataclass
class GnxModelData:
    C: int
    M: int
    A: int
    S: int = 75
    context: int = 0                      # SNPs each side = int(M*context_ratio) (src/model.py:47)
    base_kind: str | None = None          # "logistic" | "covrsk" | "forest" | "rforest"
    smooth_kind: str | None = None        # "xgb" | "crf" | "cnn"
    # logistic base: coef_ / intercept_ of LogisticRegression per window (src/Base/models.py:12-21)
    lr_coef: np.ndarray | None = None     # (W, A, ldc) float64, window i uses [:, :width_i]
    lr_intercept: np.ndarray | None = None  # (W, A)
    # CovRSK base: per-window fitted SVC (src/Base/models.py:195-215)
    svc: list | None = None               # list of dicts: xfit, support, dual_coef, intercept, prob_a, prob_b, n_support, ms
    # forest base: per-window XGBClassifier (src/Base/models.py:24-35), xgboost model schema, all windows concatenated
    fb_win_tree0: np.ndarray | None = None   # (W+1,) first tree of each window
    fb_tree_off: np.ndarray | None = None
    fb_left: np.ndarray | None = None
    fb_right: np.ndarray | None = None
    fb_feat: np.ndarray | None = None        # SNP index within the window's padded slice
    fb_cond: np.ndarray | None = None
    fb_default_left: np.ndarray | None = None
    fb_tree_class: np.ndarray | None = None
    fb_base_score: float = 0.5
    fb_missing: int = 2                      # missing_encoding (src/Base/base.py:25)
    # rforest base: per-window sklearn RandomForestClassifier (src/Base/models.py:54-66), tree_ arrays concatenated
    rf_win_tree0: np.ndarray | None = None   # (W+1,)
    rf_tree_off: np.ndarray | None = None
    rf_left: np.ndarray | None = None
    rf_right: np.ndarray | None = None
    rf_feat: np.ndarray | None = None
    rf_thr: np.ndarray | None = None         # float64
    rf_value: np.ndarray | None = None       # (n_nodes, A) float64: predict_proba row of each node
    # xgb smoother in xgboost's model schema (src/Smooth/models.py:14-20)
    tree_off: np.ndarray | None = None
    left: np.ndarray | None = None
    right: np.ndarray | None = None
    feat: np.ndarray | None = None
    cond: np.ndarray | None = None
    tree_class: np.ndarray | None = None
    base_score: float = 0.5
    # crf smoother (src/Smooth/crf.py)
    crf_state: np.ndarray | None = None   # (A, A) [attribute][label]
    crf_trans: np.ndarray | None = None   # (A, A) [from][to]
    # cnn smoother (src/Smooth/cnn.py): Conv1d(A, A, S) weight (A_out, A_in, S) and bias (A_out,), float32
    cnn_weight: np.ndarray | None = None
    cnn_bias: np.ndarray | None = None
    # optional Calibrator (src/Smooth/Calibration.py): per-class isotonic thresholds, concatenated
    calib_off: np.ndarray | None = None   # (A+1,) int32
    calib_x: np.ndarray | None = None
    calib_y: np.ndarray | None = None
    calib_is_f32: bool = False            # maps fitted on float32 probabilities (sklearn then interpolates in float32)
    # dataset metadata used by the writers (src/model.py:40-44, 88)
    snp_pos: np.ndarray | None = None
    snp_ref: np.ndarray | None = None
    snp_alt: np.ndarray | None = None
    population_order: list | None = None
    gen_map_pos: np.ndarray | None = None
    gen_map_cm: np.ndarray | None = None
    extra: dict = field(default_factory=dict)

    @property
    def W(self):
        return self.C // self.M  # src/model.py:32

    @property
    def rem(self):
        return self.C - self.M * self.W

    @property
    def M_(self):
        return self.M + 2 * self.context

    def window_width(self, i):
        return self.M_ + (self.rem if i == self.W - 1 else 0)  # base.py:163-164

    @property
    def n_trees(self):
        return 0 if self.tree_off is None else len(self.tree_off) - 1

    # ---- persistence -------------------------------------------------------------------------------
    def save(self, path, compress=False):
        """a flat .gnx (npz) archive.  Stored uncompressed by default: logistic weights are float64 noise to a compressor (the
        chr22 model shrinks by a few percent) and inflating them costs the command line 0.6 s of its ~1.4 s; load reads both."""
        d = {"gnx_version": GNX_FILE_VERSION}
        for k, v in self.__dict__.items():
            if v is None or k in ("svc", "extra"):
                continue
            if k == "population_order":
                d[k] = np.array([str(p) for p in v])
            else:
                d[k] = np.asarray(v)
        if self.svc is not None:
            d["svc_n"] = len(self.svc)
            for i, w in enumerate(self.svc):
                for kk, vv in w.items():
                    d[f"svc{i}_{kk}"] = np.asarray(vv)
        with open(path, "wb") as f:
            (np.savez_compressed if compress else np.savez)(f, **d)

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        if int(z["gnx_version"]) != GNX_FILE_VERSION:
            raise ValueError("unsupported .gnx version")
        kw = {}
        for k in cls.__dataclass_fields__:
            if k in z.files:
                v = z[k]
                if k in ("C", "M", "A", "S", "context", "fb_missing"):
                    v = int(v)
                elif k in ("base_score", "fb_base_score"):
                    v = float(v)
                elif k == "calib_is_f32":
                    v = bool(v)
                elif k in ("base_kind", "smooth_kind"):
                    v = str(v)
                elif k == "population_order":
                    v = [str(p) for p in v]
                kw[k] = v
        m = cls(**kw)
        if "svc_n" in z.files:
            m.svc = []
            for i in range(int(z["svc_n"])):
                pre = f"svc{i}_"
                m.svc.append({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
        return m

    # ---- C ABI description ---------------------------------------------------------------------------
    def to_desc(self):
        """-> (gnx_model_desc, keepalive list of the arrays the pointers refer to)"""
        keep = []

        def ptr(a, dt):
            a = _c(a, dt)
            keep.append(a)
            return a.ctypes.data

        d = _lib.ModelDesc()
        d.abi_version = _lib.GNX_ABI_VERSION
        d.A, d.C, d.M, d.ctx, d.S = int(self.A), int(self.C), int(self.M), int(self.context), int(self.S)
        d.base_kind = {None: _lib.BASE_NONE, "logistic": _lib.BASE_LOGISTIC, "covrsk": _lib.BASE_COVRSK_SVC,
                       "forest": _lib.BASE_FOREST, "rforest": _lib.BASE_RFOREST}[self.base_kind]
        d.smooth_kind = {None: _lib.SMOOTH_NONE, "xgb": _lib.SMOOTH_XGB, "crf": _lib.SMOOTH_CRF, "cnn": _lib.SMOOTH_CNN}[self.smooth_kind]
        W, A = self.W, self.A
        if self.base_kind == "logistic":
            coef = _c(self.lr_coef, np.float64)
            if coef.shape[:2] != (W, A):
                raise ValueError(f"lr_coef must be (W={W}, A={A}, ldc), got {coef.shape}")
            icpt = _c(self.lr_intercept, np.float64)
            if icpt.shape != (W, A):
                raise ValueError("lr_intercept must be (W, A)")
            d.lr_coef, d.lr_ldc, d.lr_intercept = ptr(coef, np.float64), coef.shape[2], ptr(icpt, np.float64)
        elif self.base_kind == "covrsk":
            if self.svc is None or len(self.svc) != W:
                raise ValueError("svc must list one fitted SVC per window")
            arr = (_lib.SvcWindow * W)()
            for i, w in enumerate(self.svc):
                xf = _c(w["xfit"], np.int8)
                s = arr[i]
                s.xfit, s.n_fit, s.width = ptr(xf, np.int8), xf.shape[0], xf.shape[1]
                sup = _c(w["support"], np.int32)
                s.support, s.n_sv = ptr(sup, np.int32), len(sup)
                s.dual_coef = ptr(w["dual_coef"], np.float64)
                s.intercept = ptr(w["intercept"], np.float64)
                s.prob_a = ptr(w["prob_a"], np.float64)
                s.prob_b = ptr(w["prob_b"], np.float64)
                s.n_support = ptr(w["n_support"], np.int32)
                if "poly_p" in w and float(w["poly_p"]) > 0:   # polynomial string kernel (string_kernel.py:40-61)
                    rv = _c(w["run_value"], np.float64)
                    if len(rv) < xf.shape[1] + 1:
                        raise ValueError("run_value must hold width+1 values")
                    s.kernel_kind, s.poly_p, s.run_value = 1, float(w["poly_p"]), ptr(rv, np.float64)
                    s.ms, s.n_ms = None, 0
                else:
                    ms = _c(w["ms"], np.int32)
                    s.ms, s.n_ms = ptr(ms, np.int32), len(ms)
            keep.append(arr)
            d.svc = C.addressof(arr)
        elif self.base_kind == "forest":
            wt0 = _c(self.fb_win_tree0, np.int32)
            if wt0.shape != (W + 1,):
                raise ValueError("fb_win_tree0 must be (W+1,)")
            d.fb_n_trees = len(self.fb_tree_off) - 1
            d.fb_n_nodes = len(self.fb_left)
            d.fb_missing = int(self.fb_missing)
            d.fb_win_tree0 = ptr(wt0, np.int32)
            d.fb_tree_off = ptr(self.fb_tree_off, np.int32)
            d.fb_left = ptr(self.fb_left, np.int32)
            d.fb_right = ptr(self.fb_right, np.int32)
            d.fb_feat = ptr(self.fb_feat, np.int32)
            d.fb_cond = ptr(self.fb_cond, np.float32)
            dl = self.fb_default_left if self.fb_default_left is not None else np.zeros(len(self.fb_left), np.uint8)
            d.fb_default_left = ptr(dl, np.uint8)
            tc = self.fb_tree_class if self.fb_tree_class is not None else np.zeros(d.fb_n_trees, np.int32)
            d.fb_tree_class = ptr(tc, np.int32)
            d.fb_base_score = float(self.fb_base_score)
        elif self.base_kind == "rforest":
            wt0 = _c(self.rf_win_tree0, np.int32)
            if wt0.shape != (W + 1,):
                raise ValueError("rf_win_tree0 must be (W+1,)")
            val = _c(self.rf_value, np.float64)
            if val.ndim != 2 or val.shape[1] != A or val.shape[0] != len(self.rf_left):
                raise ValueError("rf_value must be (n_nodes, A)")
            d.rf_n_trees = len(self.rf_tree_off) - 1
            d.rf_n_nodes = len(self.rf_left)
            d.rf_win_tree0 = ptr(wt0, np.int32)
            d.rf_tree_off = ptr(self.rf_tree_off, np.int32)
            d.rf_left = ptr(self.rf_left, np.int32)
            d.rf_right = ptr(self.rf_right, np.int32)
            d.rf_feat = ptr(self.rf_feat, np.int32)
            d.rf_thr = ptr(self.rf_thr, np.float64)
            d.rf_value = ptr(val, np.float64)
        if self.smooth_kind == "xgb":
            d.n_trees = self.n_trees
            d.n_nodes = len(self.left)
            d.tree_off = ptr(self.tree_off, np.int32)
            d.left = ptr(self.left, np.int32)
            d.right = ptr(self.right, np.int32)
            d.feat = ptr(self.feat, np.int32)
            d.cond = ptr(self.cond, np.float32)
            d.tree_class = ptr(self.tree_class, np.int32)
            d.base_score = float(self.base_score)
        elif self.smooth_kind == "crf":
            d.crf_state = ptr(self.crf_state, np.float64)
            d.crf_trans = ptr(self.crf_trans, np.float64)
        elif self.smooth_kind == "cnn":
            wgt = _c(self.cnn_weight, np.float32)
            if wgt.shape != (A, A, int(self.S)):
                raise ValueError(f"cnn_weight must be (A, A, S) = ({A}, {A}, {self.S}), got {wgt.shape}")
            d.cnn_weight = ptr(wgt, np.float32)
            d.cnn_bias = ptr(_c(self.cnn_bias, np.float32).reshape(A), np.float32)
        if self.calib_off is not None:
            d.calib_off = ptr(self.calib_off, np.int32)
            d.calib_x = ptr(self.calib_x, np.float64)
            d.calib_y = ptr(self.calib_y, np.float64)
            d.calib_is_f32 = int(bool(self.calib_is_f32))
        return d, keep


class DeviceModel:
    """gnx_model handle: the model resident in HBM of one device."""

    def __init__(self, data: GnxModelData, ctx: _lib.Context | None = None, device: int = 0, prepared=None):
        """prepared: the logistic base's planes as export_prepared() of the SAME model wrote them (uint8 array / buffer): skips their
        preparation; a blob of another model, library or setting raises GnxError with code GNX_ESTALE and loads nothing"""
        self.ctx = ctx or _lib.default_context(device)
        self.lib = self.ctx.lib
        self.data = data
        desc, keep = data.to_desc()
        if prepared is not None:
            pb = np.ascontiguousarray(np.frombuffer(prepared, dtype=np.uint8) if not isinstance(prepared, np.ndarray) else prepared.view(np.uint8).reshape(-1))
            keep.append(pb)
            desc.prepared, desc.prepared_bytes = pb.ctypes.data, pb.size
        h = C.c_void_p()
        self.ctx.check(self.lib.gnx_model_load(self.ctx.h, C.byref(desc), C.byref(h)))
        del keep
        self.h = h
        self.ctx._models.add(self)
        info = _lib.ModelInfo()
        self.ctx.check(self.lib.gnx_model_get_info(h, C.byref(info)))
        self.info = info
        self.W, self.A, self.S, self.C, self.M = int(info.W), int(info.A), int(info.S), int(info.C), int(info.M)

    def export_prepared(self):
        """the logistic base's prepared planes as a uint8 array for DeviceModel(..., prepared=...) of the same model (empty for other bases)"""
        n = C.c_int64()
        self.ctx.check(self.lib.gnx_model_export_prepared(self.h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.uint8)
        if n.value:
            self.ctx.check(self.lib.gnx_model_export_prepared(self.h, out.ctypes.data, out.size, C.byref(n)))
        return out

    def set_calibrate(self, on):
        self.ctx.check(self.lib.gnx_model_set_calibrate(self.h, int(bool(on))))
        self.calibrated = bool(on) and self.data.calib_off is not None

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.lib.gnx_model_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host (numpy) entry points: synchronous ------------------------------------------------------
    def _x(self, X):
        X = np.asarray(X)
        if X.ndim != 2 or X.shape[1] != self.C:
            raise ValueError(f"X must be (N, C={self.C}), got {X.shape}")
        if X.dtype != np.int8:
            X = X.astype(np.int8)  # vcf_to_npy hands int8 (src/utils.py:153)
        return np.ascontiguousarray(X)

    def base_predict(self, X, want_f32=False, want_f64=True):
        X = self._x(X)
        N = X.shape[0]
        b32 = np.empty((N, self.W, self.A), np.float32) if want_f32 else None
        b64 = np.empty((N, self.W, self.A), np.float64) if want_f64 else None
        self.ctx.check(self.lib.gnx_base_predict(self.h, X.ctypes.data, N, X.shape[1],
                                                 b32.ctypes.data if want_f32 else None,
                                                 b64.ctypes.data if want_f64 else None))
        return b32, b64

    def _b(self, B):
        B = np.asarray(B)
        if B.ndim != 3 or B.shape[1:] != (self.W, self.A):
            raise ValueError(f"B must be (N, W={self.W}, A={self.A}), got {B.shape}")
        if B.dtype not in (np.float32, np.float64):
            B = B.astype(np.float64)
        return np.ascontiguousarray(B)

    def smooth_predict(self, B, want_proba=True, want_labels=True, proba_dtype=None):
        B = self._b(B)
        N = B.shape[0]
        native64 = self.data.smooth_kind == "crf" or getattr(self, "calibrated", False)
        pd = proba_dtype or (np.float64 if native64 else np.float32)
        p = np.empty((N, self.W, self.A), pd) if want_proba else None
        lab = np.empty((N, self.W), np.int32) if want_labels else None
        p32 = p.ctypes.data if (want_proba and pd == np.float32) else None
        p64 = p.ctypes.data if (want_proba and pd == np.float64) else None
        self.ctx.check(self.lib.gnx_smooth_predict(self.h, B.ctypes.data, int(B.dtype == np.float64), N, p32, p64,
                                                   lab.ctypes.data if want_labels else None))
        return p, lab

    def proba_dtype(self):
        """dtype of the probabilities this model's smoother returns: float64 for CRF and for calibrated output, else float32"""
        native64 = self.data.smooth_kind == "crf" or getattr(self, "calibrated", False)
        return np.dtype(np.float64 if native64 else np.float32)

    def _outs(self, N, want_proba, want_labels, proba_dtype, out):
        """output arrays of the host-pointer entry points; `out=(proba, labels)` lets the caller supply them (e.g. page-locked
        arrays from Context.pinned_empty, reused across calls: D2H then runs at link rate instead of through pageable staging)"""
        native64 = self.data.smooth_kind == "crf" or getattr(self, "calibrated", False)
        pd = np.dtype(proba_dtype or (np.float64 if native64 else np.float32))
        p = lab = None
        if out is not None:
            p, lab = out
            if p is not None and (p.shape != (N, self.W, self.A) or p.dtype not in (np.float32, np.float64) or not p.flags.c_contiguous):
                raise ValueError("out[0] must be a C-contiguous float32/float64 (N, W, A) array")
            if lab is not None and (lab.shape != (N, self.W) or lab.dtype != np.int32 or not lab.flags.c_contiguous):
                raise ValueError("out[1] must be a C-contiguous int32 (N, W) array")
            if p is not None:
                pd = p.dtype
        if p is None and want_proba:
            p = np.empty((N, self.W, self.A), pd)
        if lab is None and want_labels:
            lab = np.empty((N, self.W), np.int32)
        return p, lab, pd

    def infer(self, X, want_proba=True, want_labels=True, proba_dtype=None, out=None):
        X = self._x(X)
        N = X.shape[0]
        p, lab, pd = self._outs(N, want_proba, want_labels, proba_dtype, out)
        want_proba, want_labels = p is not None, lab is not None
        p32 = p.ctypes.data if (want_proba and pd == np.float32) else None
        p64 = p.ctypes.data if (want_proba and pd == np.float64) else None
        self.ctx.check(self.lib.gnx_infer(self.h, X.ctypes.data, N, X.shape[1], p32, p64,
                                          lab.ctypes.data if want_labels else None))
        return p, lab

    # ---- 2-bit packed input (gnx_pack_x / gnx_infer_packed): a quarter of the bytes over the host link -------------
    def pack_x(self, X, out=None, n_threads=0):
        """int8 (N, C) {0,1,2} -> packed uint8 (N, gnx_packed_row_bytes(C)); `out` may be a page-locked array from
        Context.pinned_empty.  Raises GnxError(-1) when X holds a value outside 0..3."""
        X = self._x(X)
        N = X.shape[0]
        ldp = int(self.lib.gnx_packed_row_bytes(self.C))
        if out is None:
            out = self.ctx.pinned_empty((N, ldp), np.uint8)
        if out.shape != (N, ldp) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError(f"out must be C-contiguous uint8 {(N, ldp)}")
        rc = self.lib.gnx_pack_x(X.ctypes.data, N, X.shape[1], self.C, out.ctypes.data, ldp, int(n_threads))
        if rc != _lib.GNX_OK:
            raise _lib.GnxError(rc, "gnx_pack_x: a value outside {0, 1, 2, 3} cannot be packed in 2 bits")
        return out

    def infer_packed(self, P, N=None, want_proba=True, want_labels=True, proba_dtype=None, out=None):
        P = np.ascontiguousarray(P, dtype=np.uint8)
        if P.ndim != 2 or P.shape[1] < (self.C + 3) // 4:
            raise ValueError(f"packed X must be (N, >= ceil(C/4) = {(self.C + 3) // 4}) uint8, got {P.shape}")
        N = P.shape[0] if N is None else int(N)
        if N < 0 or N > P.shape[0]:
            raise ValueError(f"infer_packed: N = {N} but the packed matrix has {P.shape[0]} rows")
        p, lab, pd = self._outs(N, want_proba, want_labels, proba_dtype, out)
        want_proba, want_labels = p is not None, lab is not None
        p32 = p.ctypes.data if (want_proba and pd == np.float32) else None
        p64 = p.ctypes.data if (want_proba and pd == np.float64) else None
        self.ctx.check(self.lib.gnx_infer_packed(self.h, P.ctypes.data, N, P.shape[1], p32, p64,
                                                 lab.ctypes.data if want_labels else None))
        return p, lab

    # ---- the file path: parsed VCF rows (variant-major 2-bit, include/gnomix_io.h) straight to the outputs --------------------
    def _gt2_args(self, G, N, src):
        G = np.asarray(G)
        if G.dtype != np.uint8 or G.ndim != 2 or not G.flags.c_contiguous:
            raise ValueError("G must be a C-contiguous uint8 (n_variants, ldg) array of 2-bit genotype rows")
        N = int(N)
        if N < 0 or G.shape[1] < (N + 3) // 4:
            raise ValueError(f"G rows hold {4 * G.shape[1]} haplotypes, fewer than N = {N}")
        src = np.ascontiguousarray(src, dtype=np.int32)
        if src.shape != (self.C,):
            raise ValueError(f"src must be (C={self.C},) int32 (vcfio.column_map), got {src.shape}")
        return G, N, src

    def infer_gt2(self, G, N, src, want_proba=True, want_labels=True, proba_dtype=None, out=None):
        """vcf_to_npy + predict_proba + argmax of gnomix.py:49-58 on the parsed query: G (n_variants, ldg) the reader's 2-bit
        rows (VcfData.gt2), N = 2 * samples, src = vcfio.column_map(...)[0].  The (N, C) matrix exists only in HBM."""
        G, N, src = self._gt2_args(G, N, src)
        p, lab, pd = self._outs(N, want_proba, want_labels, proba_dtype, out)
        want_proba, want_labels = p is not None, lab is not None
        p32 = p.ctypes.data if (want_proba and pd == np.float32) else None
        p64 = p.ctypes.data if (want_proba and pd == np.float64) else None
        self.ctx.check(self.lib.gnx_infer_gt2(self.h, G.ctypes.data, G.shape[0], G.shape[1], N, src.ctypes.data, p32, p64,
                                              lab.ctypes.data if want_labels else None))
        return p, lab

    def phase_gt2(self, G, N, src, out_cols=None, max_it=50, want_proba=True, proba_dtype=None, out=None):
        """gnomix.py:60-72 (phase=True) on the parsed query: Gnofix on every individual, then predict_proba of the re-phased
        haplotypes.  -> (G_phased or None, proba, labels (N, W) int32, n_switches (N/2,) int32); G_phased (len(out_cols), ldg)
        holds X_phased[:, out_cols] as 2-bit rows for the phased VCF."""
        G, N, src = self._gt2_args(G, N, src)
        if N % 2:
            raise ValueError("phase_gt2: N = 2 * individuals")
        p, lab, pd = self._outs(N, want_proba, True, proba_dtype, out)
        p32 = p.ctypes.data if (p is not None and pd == np.float32) else None
        p64 = p.ctypes.data if (p is not None and pd == np.float64) else None
        nsw = np.empty((N // 2,), np.int32)
        Go = cols = None
        if out_cols is not None:
            cols = np.ascontiguousarray(out_cols, dtype=np.int32)
            Go = np.zeros((len(cols), G.shape[1]), np.uint8)
        self.ctx.check(self.lib.gnx_phase_gt2(self.h, G.ctypes.data, G.shape[0], G.shape[1], N, src.ctypes.data, int(max_it),
                                              cols.ctypes.data if cols is not None else None, len(cols) if cols is not None else 0,
                                              Go.ctypes.data if Go is not None else None, G.shape[1], p32, p64, lab.ctypes.data,
                                              nsw.ctypes.data))
        return Go, p, lab, nsw

    def smooth_rows(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        F = self.S * self.A
        if rows.ndim != 2 or rows.shape[1] != F:
            raise ValueError(f"rows must be (R, S*A={F})")
        out = np.empty((rows.shape[0], self.A), np.float32)
        self.ctx.check(self.lib.gnx_smooth_rows(self.h, rows.ctypes.data, rows.shape[0], out.ctypes.data))
        return out

    def calibrate_rows(self, proba):
        """Calibrator.transform on explicit (R, A) rows -> float64"""
        proba = np.ascontiguousarray(proba)
        if proba.dtype not in (np.float32, np.float64):
            proba = proba.astype(np.float64)
        out = np.empty(proba.shape, dtype=np.float64)
        self.ctx.check(self.lib.gnx_calibrate_rows(self.h, proba.ctypes.data, int(proba.dtype == np.float64), proba.shape[0],
                                                   out.ctypes.data))
        return out

    def gnofix(self, X, B, max_it=50, inplace=False, out=None):
        """Re-phase individuals (rows 2i, 2i+1 of X and B).  Returns (X re-phased, labels i32 (2n, W), n_switches i32 (n,)).
        `inplace=True` re-phases the caller's C-contiguous int8 X itself (what the C ABI does; no host copy of X — keep X, B
        and `out=(Y, nsw)` in `ctx.pinned_empty` arrays and batches overlap on three streams); the default works on a copy."""
        if inplace:
            if not (isinstance(X, np.ndarray) and X.dtype == np.int8 and X.flags.c_contiguous and X.flags.writeable):
                raise ValueError("gnofix(inplace=True) needs a writeable C-contiguous int8 array")
        else:
            X = np.array(X, dtype=np.int8, order="C", copy=True)
        B = np.ascontiguousarray(B, dtype=np.float64)
        if X.ndim != 2 or X.shape[1] != self.C or X.shape[0] % 2:
            raise ValueError(f"gnofix: X must be (2n, C={self.C}), got {X.shape}")
        if B.shape != (X.shape[0], self.W, self.A):
            raise ValueError(f"gnofix: B must be (2n={X.shape[0]}, W={self.W}, A={self.A}), got {B.shape}")
        n_ind = X.shape[0] // 2
        if out is not None:
            Y, nsw = out
            if Y.shape != (2 * n_ind, self.W) or Y.dtype != np.int32 or not Y.flags.c_contiguous:
                raise ValueError("gnofix: out[0] must be C-contiguous int32 of shape (2n, W)")
            if nsw.shape != (n_ind,) or nsw.dtype != np.int32 or not nsw.flags.c_contiguous:
                raise ValueError("gnofix: out[1] must be C-contiguous int32 of shape (n,)")
        else:
            Y = np.empty((2 * n_ind, self.W), np.int32)
            nsw = np.empty((n_ind,), np.int32)
        self.ctx.check(self.lib.gnx_gnofix(self.h, X.ctypes.data, X.shape[1], B.ctypes.data, n_ind, int(max_it),
                                           Y.ctypes.data, nsw.ctypes.data))
        return X, Y, nsw

    # ---- device (torch tensor) entry points: asynchronous on torch's current stream ---------------------
    def _bind_torch_stream(self):
        import torch
        self.ctx.set_stream(torch.cuda.current_stream(self.ctx.device).cuda_stream)

    def infer_device(self, X_t, want_proba=True, want_labels=True):
        """X_t: torch int8 CUDA tensor (N, C) resident in HBM -> (proba f32 tensor, labels i32 tensor)."""
        import torch
        assert X_t.is_cuda and X_t.dtype == torch.int8 and X_t.dim() == 2 and X_t.stride(1) == 1
        self._bind_torch_stream()
        N = X_t.shape[0]
        p = torch.empty((N, self.W, self.A), dtype=torch.float32, device=X_t.device) if want_proba else None
        lab = torch.empty((N, self.W), dtype=torch.int32, device=X_t.device) if want_labels else None
        self.ctx.check(self.lib.gnx_infer_dev(self.h, X_t.data_ptr(), N, X_t.stride(0), p.data_ptr() if want_proba else None,
                                              None, lab.data_ptr() if want_labels else None))
        return p, lab

    def base_predict_device(self, X_t, f64=False):
        import torch
        assert X_t.is_cuda and X_t.dtype == torch.int8 and X_t.dim() == 2 and X_t.stride(1) == 1
        self._bind_torch_stream()
        N = X_t.shape[0]
        B = torch.empty((N, self.W, self.A), dtype=torch.float64 if f64 else torch.float32, device=X_t.device)
        self.ctx.check(self.lib.gnx_base_predict_dev(self.h, X_t.data_ptr(), N, X_t.stride(0),
                                                     None if f64 else B.data_ptr(), B.data_ptr() if f64 else None))
        return B

    def pack_device(self, X_t):
        """int8 (N, C) CUDA tensor -> the gnx_pack_x layout as a uint8 CUDA tensor (N, gnx_packed_row_bytes(C)): SNP j of a row =
        bits 2 (j % 4).. of byte j // 4 (torch ops; a test / bench utility, the product packs on the host or in k_gt2)"""
        import torch
        assert X_t.is_cuda and X_t.dtype == torch.int8 and X_t.dim() == 2
        N, C = X_t.shape
        ldp = int(self.lib.gnx_packed_row_bytes(self.C))
        P = torch.zeros((N, ldp), dtype=torch.uint8, device=X_t.device)
        step = max(1, (1 << 28) // max(C, 1))
        for n0 in range(0, N, step):
            x = X_t[n0:n0 + step].to(torch.uint8) & 3
            pad = (-C) % 4
            if pad:
                x = torch.nn.functional.pad(x, (0, pad))
            x = x.view(x.shape[0], -1, 4)
            P[n0:n0 + step, :x.shape[1]] = x[:, :, 0] | (x[:, :, 1] << 2) | (x[:, :, 2] << 4) | (x[:, :, 3] << 6)
        return P

    def base_predict_packed_device(self, P_t, f64=False):
        """Base.predict_proba on device-resident 2-bit rows (gnx_base_predict_packed_dev)"""
        import torch
        assert P_t.is_cuda and P_t.dtype == torch.uint8 and P_t.dim() == 2 and P_t.stride(1) == 1
        self._bind_torch_stream()
        N = P_t.shape[0]
        B = torch.empty((N, self.W, self.A), dtype=torch.float64 if f64 else torch.float32, device=P_t.device)
        self.ctx.check(self.lib.gnx_base_predict_packed_dev(self.h, P_t.data_ptr(), N, P_t.stride(0),
                                                            None if f64 else B.data_ptr(), B.data_ptr() if f64 else None))
        return B

    def infer_packed_device(self, P_t, want_proba=True, want_labels=True):
        """P_t: torch uint8 CUDA tensor (N, >= ceil(C/4)) of 2-bit rows resident in HBM -> (proba f32 tensor, labels i32 tensor)."""
        import torch
        assert P_t.is_cuda and P_t.dtype == torch.uint8 and P_t.dim() == 2 and P_t.stride(1) == 1
        self._bind_torch_stream()
        N = P_t.shape[0]
        p = torch.empty((N, self.W, self.A), dtype=torch.float32, device=P_t.device) if want_proba else None
        lab = torch.empty((N, self.W), dtype=torch.int32, device=P_t.device) if want_labels else None
        self.ctx.check(self.lib.gnx_infer_packed_dev(self.h, P_t.data_ptr(), N, P_t.stride(0), p.data_ptr() if want_proba else None,
                                                     None, lab.data_ptr() if want_labels else None))
        return p, lab

    def gnofix_device(self, X_t, B_t, max_it=50):
        """X_t (2n, C) int8 CUDA tensor re-phased IN PLACE, B_t (2n, W, A) float64 -> (labels i32 (2n, W), n_switches i32 (n,))"""
        import torch
        assert X_t.is_cuda and X_t.dtype == torch.int8 and X_t.stride(1) == 1
        assert B_t.is_cuda and B_t.dtype == torch.float64 and B_t.is_contiguous()
        self._bind_torch_stream()
        n_ind = X_t.shape[0] // 2
        Y = torch.empty((2 * n_ind, self.W), dtype=torch.int32, device=X_t.device)
        ns = torch.empty((n_ind,), dtype=torch.int32, device=X_t.device)
        self.ctx.check(self.lib.gnx_gnofix_dev(self.h, X_t.data_ptr(), X_t.stride(0), B_t.data_ptr(), n_ind, int(max_it),
                                               Y.data_ptr(), ns.data_ptr()))
        return Y, ns

    def gnofix_packed_device(self, P_t, B_t, max_it=50):
        """P_t (2n, ldp) uint8 CUDA tensor of 2-bit rows (gnx_pack_x layout) re-phased IN PLACE, B_t (2n, W, A) float64 ->
        (labels i32 (2n, W), n_switches i32 (n,))  (gnx_gnofix_packed_dev)"""
        import torch
        assert P_t.is_cuda and P_t.dtype == torch.uint8 and P_t.stride(1) == 1
        assert B_t.is_cuda and B_t.dtype == torch.float64 and B_t.is_contiguous()
        self._bind_torch_stream()
        n_ind = P_t.shape[0] // 2
        Y = torch.empty((2 * n_ind, self.W), dtype=torch.int32, device=P_t.device)
        ns = torch.empty((n_ind,), dtype=torch.int32, device=P_t.device)
        self.ctx.check(self.lib.gnx_gnofix_packed_dev(self.h, P_t.data_ptr(), P_t.stride(0), B_t.data_ptr(), n_ind, int(max_it),
                                                      Y.data_ptr(), ns.data_ptr()))
        return Y, ns

    def smooth_predict_device(self, B_t):
        import torch
        assert B_t.is_cuda and B_t.is_contiguous() and B_t.dtype in (torch.float32, torch.float64)
        self._bind_torch_stream()
        N = B_t.shape[0]
        p = torch.empty((N, self.W, self.A), dtype=torch.float32, device=B_t.device)
        lab = torch.empty((N, self.W), dtype=torch.int32, device=B_t.device)
        self.ctx.check(self.lib.gnx_smooth_predict_dev(self.h, B_t.data_ptr(), int(B_t.dtype == torch.float64), N,
                                                       p.data_ptr(), None, lab.data_ptr()))
        return p, lab

"""ctypes front-end of the CPU oracle (oracle/gnx_oracle.c) + the host-side restatements that
are control flow rather than arithmetic (CovSample, Gnofix loop).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  gnomix_amd/ must never import this module.

Reference rows restated: a1-a9.  Pinning status is in gnx_oracle.c's header.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgnx_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds).  Returns the .so path."""
    src = os.path.join(_HERE, "gnx_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libgnx_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.gnxo_pad_src.restype = C.c_int64
        _lib.gnxo_pad_src.argtypes = [C.c_int64] * 3
        _lib.gnxo_slide_src.restype = C.c_int64
        _lib.gnxo_slide_src.argtypes = [C.c_int64] * 3
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _chk(rc, what):
    if rc != 0:
        raise ValueError(f"oracle {what} failed with code {rc}")


# ------------------------------------------------------------------------------------------------
# a1-a3 logistic base
# ------------------------------------------------------------------------------------------------
def base_lr(X, M, ctx, coef, intercept):
    """X (N,C) int8; coef (W,A,ldc) f64 zero-padded per window; intercept (W,A) -> B (N,W,A) f64.
    Restates base.py:146-180 + models.py:12-21 (sklearn OvR logistic)."""
    X = np.ascontiguousarray(X, dtype=np.int8)
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    intercept = np.ascontiguousarray(intercept, dtype=np.float64)
    N, Cn = X.shape
    W, A, ldc = coef.shape
    assert W == Cn // M, (W, Cn, M)
    B = np.empty((N, W, A), dtype=np.float64)
    rc = lib().gnxo_base_lr(_p(X), C.c_int64(N), C.c_int64(Cn), C.c_int64(Cn), C.c_int64(M), C.c_int64(ctx),
                            C.c_int64(A), _p(coef), C.c_int64(ldc), _p(intercept), _p(B))
    _chk(rc, "base_lr")
    return B


def base_lr_windows(X, M, ctx, coef, intercept, w0, w1, B):
    """windows [w0, w1) of base_lr into a caller-owned B (N, W, A) float64: the unit of work bench.py's CPU baseline hands
    to a host thread (a window's weights stay in that core's cache)."""
    N, Cn = X.shape
    W, A, ldc = coef.shape
    assert X.dtype == np.int8 and X.flags.c_contiguous and B.shape == (N, W, A) and B.dtype == np.float64 and B.flags.c_contiguous
    rc = lib().gnxo_base_lr_range(_p(X), C.c_int64(N), C.c_int64(Cn), C.c_int64(Cn), C.c_int64(M), C.c_int64(ctx), C.c_int64(A),
                                  _p(coef), C.c_int64(ldc), _p(intercept), C.c_int64(w0), C.c_int64(w1), _p(B))
    _chk(rc, "base_lr_windows")
    return B


def base_lr_pad(X, ctx):
    """Base.pad (src/Base/base.py:41-44): ctx reflected SNPs on both sides"""
    X = np.asarray(X)
    return np.concatenate([X[:, :ctx][:, ::-1], X, X[:, X.shape[1] - ctx:][:, ::-1]], axis=1) if ctx else X


def base_lr_blas(X, M, ctx, coef, intercept, w0=0, w1=None, out=None, Xp=None):
    """The same step through the host's BLAS, window by window, the way the reference computes it: sklearn's
    LogisticRegression.predict_proba of an OvR model = expit(X_w.astype(float64) @ coef_.T + intercept_), normalised
    (src/Base/models.py:12-21 through src/Base/base.py:146-180; reflect padding of base.py:41-44).  Summation order is the
    BLAS library's, so values differ from base_lr in the last bits (tests allow 1e-12).  `Xp`: the reflect-padded matrix from an
    earlier call (base_lr_pad), so that window ranges running on several threads share one copy."""
    X = np.asarray(X)
    N, Cn = X.shape
    W, A, ldc = coef.shape
    rem = Cn - M * W
    M_ = M + 2 * ctx
    w1 = W if w1 is None else w1
    B = np.empty((N, W, A), np.float64) if out is None else out
    if Xp is None:
        Xp = base_lr_pad(X, ctx)
    for i in range(w0, w1):
        lo, n = (i * M, M_) if i < W - 1 else (Xp.shape[1] - (M_ + rem), M_ + rem)
        z = Xp[:, lo:lo + n].astype(np.float64) @ coef[i, :, :n].T + intercept[i]
        p = 1.0 / (1.0 + np.exp(-z))
        B[:, i, :] = p / p.sum(axis=1, keepdims=True)
    return B


# ------------------------------------------------------------------------------------------------
# a5 slide_window
# ------------------------------------------------------------------------------------------------
def slide_window(B, S):
    """(N,W,A) -> (N*W, S*A) float32   (Smooth/utils.py:4-29)"""
    B = np.ascontiguousarray(B)
    assert B.dtype in (np.float64, np.float32)
    N, W, A = B.shape
    out = np.empty((N * W, S * A), dtype=np.float32)
    rc = lib().gnxo_slide_window(_p(B), C.c_int(B.dtype == np.float64), C.c_int64(N), C.c_int64(W), C.c_int64(A),
                                 C.c_int64(S), _p(out))
    _chk(rc, "slide_window")
    return out


# ------------------------------------------------------------------------------------------------
# a6 xgboost-schema tree ensemble
# ------------------------------------------------------------------------------------------------
class _CTrees(C.Structure):
    _fields_ = [("n_trees", C.c_int32), ("n_class", C.c_int32), ("tree_off", C.c_void_p), ("left", C.c_void_p),
                ("right", C.c_void_p), ("feat", C.c_void_p), ("cond", C.c_void_p), ("default_left", C.c_void_p),
                ("tree_class", C.c_void_p), ("base_score", C.c_float)]


@dataclass
class Trees:
    """An xgboost-schema ensemble: per-tree node arrays concatenated, tree_off = node offsets.
    left/right are child indices WITHIN the tree (-1 at leaves); cond holds the split condition
    at internal nodes and the leaf value at leaves; tree_class = tree_info."""
    tree_off: np.ndarray
    left: np.ndarray
    right: np.ndarray
    feat: np.ndarray
    cond: np.ndarray
    tree_class: np.ndarray
    n_class: int
    base_score: float = 0.5
    default_left: np.ndarray | None = None

    def __post_init__(self):
        self.tree_off = np.ascontiguousarray(self.tree_off, dtype=np.int32)
        self.left = np.ascontiguousarray(self.left, dtype=np.int32)
        self.right = np.ascontiguousarray(self.right, dtype=np.int32)
        self.feat = np.ascontiguousarray(self.feat, dtype=np.int32)
        self.cond = np.ascontiguousarray(self.cond, dtype=np.float32)
        self.tree_class = np.ascontiguousarray(self.tree_class, dtype=np.int32)
        if self.default_left is not None:
            self.default_left = np.ascontiguousarray(self.default_left, dtype=np.uint8)

    @property
    def n_trees(self):
        return len(self.tree_off) - 1

    def _c(self):
        return _CTrees(self.n_trees, self.n_class, _p(self.tree_off).value, _p(self.left).value, _p(self.right).value,
                       _p(self.feat).value, _p(self.cond).value,
                       _p(self.default_left).value if self.default_left is not None else None,
                       _p(self.tree_class).value, self.base_score)


def xgb_predict_proba(trees: Trees, feats):
    """model.predict_proba on explicit features (R,F) -> (R,n_class) float32."""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    R, F = feats.shape
    out = np.empty((R, trees.n_class), dtype=np.float32)
    ct = trees._c()
    _chk(lib().gnxo_xgb_predict_proba(C.byref(ct), _p(feats), C.c_int64(R), C.c_int64(F), _p(out)), "xgb_predict_proba")
    return out


def smooth_xgb(trees: Trees, B, S):
    """Smoother.predict_proba for XGB_Smoother (smooth.py:40-56): -> proba (N,W,A) f32, labels (N,W) int64."""
    B = np.ascontiguousarray(B)
    assert B.dtype in (np.float64, np.float32)
    N, W, A = B.shape
    proba = np.empty((N, W, A), dtype=np.float32)
    labels = np.empty((N, W), dtype=np.int32)
    ct = trees._c()
    _chk(lib().gnxo_smooth_xgb(C.byref(ct), _p(B), C.c_int(B.dtype == np.float64), C.c_int64(N), C.c_int64(W),
                               C.c_int64(A), C.c_int64(S), _p(proba), _p(labels)), "smooth_xgb")
    return proba, labels.astype(np.int64)


def base_forest(trees: Trees, win_tree0, X, M, ctx, A, missing=2):
    """XGBBase.predict_proba (Base/models.py:24-35): per-window xgboost forests on the window's SNPs -> B (N,W,A) f32"""
    X = np.ascontiguousarray(X, dtype=np.int8)
    win_tree0 = np.ascontiguousarray(win_tree0, dtype=np.int32)
    N, Cn = X.shape
    W = Cn // M
    B = np.empty((N, W, A), dtype=np.float32)
    ct = trees._c()
    _chk(lib().gnxo_base_forest(C.byref(ct), _p(win_tree0), _p(X), C.c_int64(N), C.c_int64(Cn), C.c_int64(Cn), C.c_int64(M),
                                C.c_int64(ctx), C.c_int64(A), C.c_int(missing), _p(B)), "base_forest")
    return B


def base_rforest(rf, X, M, ctx, A):
    """RFBase.predict_proba (Base/models.py:54-66): rf = dict(win_tree0, tree_off, left, right, feat, thr, value) with
    sklearn's tree arrays concatenated over trees and windows, value = normalised leaf rows (n_nodes, A) -> B (N,W,A) f64"""
    X = np.ascontiguousarray(X, dtype=np.int8)
    N, Cn = X.shape
    W = Cn // M
    i32 = lambda k: np.ascontiguousarray(rf[k], dtype=np.int32)
    f64 = lambda k: np.ascontiguousarray(rf[k], dtype=np.float64)
    a = [i32("win_tree0"), i32("tree_off"), i32("left"), i32("right"), i32("feat"), f64("thr"), f64("value")]
    B = np.empty((N, W, A), dtype=np.float64)
    _chk(lib().gnxo_base_rforest(*[_p(x) for x in a], _p(X), C.c_int64(N), C.c_int64(Cn), C.c_int64(Cn), C.c_int64(M),
                                 C.c_int64(ctx), C.c_int64(A), _p(B)), "base_rforest")
    return B


def random_trees(n_rounds, n_class, n_feat, depth=4, seed=0, thr_lo=0.0, thr_hi=1.0, leaf_scale=0.3,
                 p_early_leaf=0.15):
    """Synthetic xgboost-schema ensemble (round-major, tree t has class t % n_class — the layout
    multi:softprob produces).  Some branches stop early (leaf at depth < max) as real boosters do."""
    rng = np.random.RandomState(seed)
    offs, L, R_, Fe, Cd, cls = [0], [], [], [], [], []
    for t in range(n_rounds * n_class):
        nodes = []  # (left,right,feat,cond)

        def grow(d):
            idx = len(nodes)
            nodes.append(None)
            if d == depth or (d > 0 and rng.rand() < p_early_leaf):
                nodes[idx] = (-1, -1, 0, np.float32(rng.randn() * leaf_scale))
            else:
                f = rng.randint(n_feat)
                thr = np.float32(rng.uniform(thr_lo, thr_hi))
                l = grow(d + 1)
                r = grow(d + 1)
                nodes[idx] = (l, r, f, thr)
            return idx

        grow(0)
        for (l, r, f, c) in nodes:
            L.append(l); R_.append(r); Fe.append(f); Cd.append(c)
        offs.append(len(L))
        cls.append(t % n_class)
    return Trees(np.array(offs), np.array(L), np.array(R_), np.array(Fe), np.array(Cd, dtype=np.float32),
                 np.array(cls), n_class)


# ------------------------------------------------------------------------------------------------
# a7 CRF marginals
# ------------------------------------------------------------------------------------------------
def smooth_crf(B, state, trans):
    """(N,W,A) f64 -> marginals (N,W,A) f64, labels (N,W).  state[a][y], trans[y'][y]."""
    B = np.ascontiguousarray(B, dtype=np.float64)
    state = np.ascontiguousarray(state, dtype=np.float64)
    trans = np.ascontiguousarray(trans, dtype=np.float64)
    N, W, A = B.shape
    proba = np.empty((N, W, A), dtype=np.float64)
    labels = np.empty((N, W), dtype=np.int32)
    _chk(lib().gnxo_smooth_crf(_p(B), C.c_int64(N), C.c_int64(W), C.c_int64(A), _p(state), _p(trans), _p(proba),
                               _p(labels)), "smooth_crf")
    return proba, labels.astype(np.int64)


# ------------------------------------------------------------------------------------------------
# a4 CovRSK + SVC probability
# ------------------------------------------------------------------------------------------------
def smooth_cnn(B, weight, bias):
    """CNN_Smoother.predict_proba / predict (Smooth/cnn.py): B (N,W,A) -> proba (N,W,A) f32, labels (N,W) int64"""
    B = np.ascontiguousarray(B, dtype=np.float64)
    wgt = np.ascontiguousarray(weight, dtype=np.float32)
    bs = np.ascontiguousarray(bias, dtype=np.float32)
    N, W, A = B.shape
    S = wgt.shape[2]
    proba = np.empty((N, W, A), dtype=np.float32)
    labels = np.empty((N, W), dtype=np.int64)
    _chk(lib().gnxo_smooth_cnn(_p(B), C.c_int64(N), C.c_int64(W), C.c_int64(A), C.c_int64(S), _p(wgt), _p(bs), _p(proba),
                               _p(labels)), "smooth_cnn")
    return proba, labels


def crf_objective(B, y, state, trans, c2=1.0):
    """CRFsuite's training objective for the reference's CRF smoother (src/Smooth/crf.py:9-15, 51-54: lbfgs, c1 = 0, c2 = 1,
    all possible state and transition features; CRFsuite crf1d_encode.c / train_lbfgs.c restated, log-space forward-backward):
        f = - sum_n log p(y_n | x_n) + c2 |w|^2,   score(y | x) = sum_t sum_a state[a][y_t] x[t][a] + sum_{t>=1} trans[y_{t-1}][y_t]
    -> f, df/dstate (A, A) [attribute][label], df/dtrans (A, A) [from][to]; float64.  The model is the one smooth_crf evaluates."""
    from scipy.special import logsumexp
    X = np.asarray(B, dtype=np.float64)
    y = np.asarray(y).astype(np.int64)
    state = np.asarray(state, dtype=np.float64)
    trans = np.asarray(trans, dtype=np.float64)
    N, W, A = X.shape
    s = X @ state                                                     # (N, W, A) state scores
    la = np.empty_like(s)
    lb = np.zeros_like(s)
    la[:, 0] = s[:, 0]
    for t in range(1, W):
        la[:, t] = s[:, t] + logsumexp(la[:, t - 1][:, :, None] + trans[None], axis=1)
    for t in range(W - 2, -1, -1):
        lb[:, t] = logsumexp(trans[None] + (s[:, t + 1] + lb[:, t + 1])[:, None, :], axis=2)
    logz = logsumexp(la[:, W - 1], axis=1)
    score = np.take_along_axis(s, y[:, :, None], axis=2)[:, :, 0].sum(1) + trans[y[:, :-1], y[:, 1:]].sum(1)
    marg = np.exp(la + lb - logz[:, None, None])
    onehot = (np.arange(A)[None, None, :] == y[:, :, None]).astype(np.float64)
    g_state = np.einsum("nta,ntl->al", X, marg - onehot)
    g_trans = np.zeros((A, A))
    for t in range(1, W):
        g_trans += np.exp(la[:, t - 1][:, :, None] + trans[None] + (s[:, t] + lb[:, t])[:, None, :] - logz[:, None, None]).sum(0)
    np.subtract.at(g_trans, (y[:, :-1].ravel(), y[:, 1:].ravel()), 1.0)
    f = float(np.sum(logz - score) + c2 * (np.sum(state * state) + np.sum(trans * trans)))
    return f, g_state + 2.0 * c2 * state, g_trans + 2.0 * c2 * trans


def crf_fit(B, y, c2=1.0, gtol=1e-10, max_iterations=10000):
    """the minimiser of crf_objective from zeros (CRFsuite's starting point) by scipy's L-BFGS-B run to a tight gradient tolerance:
    an optimiser that shares nothing with the device's -> state, trans, f"""
    from scipy.optimize import minimize
    A = np.asarray(B).shape[2]

    def fg(w):
        f, gs, gt = crf_objective(B, y, w[:A * A].reshape(A, A), w[A * A:].reshape(A, A), c2)
        return f, np.concatenate([gs.ravel(), gt.ravel()])
    r = minimize(fg, np.zeros(2 * A * A), jac=True, method="L-BFGS-B", options=dict(maxiter=max_iterations, maxfun=10 * max_iterations, gtol=gtol, ftol=1e-16, maxcor=20))
    return r.x[:A * A].reshape(A, A), r.x[A * A:].reshape(A, A), float(r.fun)


def cnn_fit(B, y, weight, bias, epochs, batch=128, order=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, log_eps=1e-8):
    """CNN.fit (reference src/Smooth/cnn.py:104-118) restated in float32 numpy: Conv1d(A, A, S, zero padding (S-1)//2)
    (cnn.py:37-39; zero padding: see smooth_cnn / tests/golden/make_golden.py G11), loss = NLLLoss(log(softmax + 1e-8), y) with
    mean reduction (cnn.py:57-75), torch.optim.Adam's single-tensor step (exp_avg, exp_avg_sq, bias corrections in Python floats,
    denom = sqrt(v) / sqrt(bc2) + eps, p -= lr / bc1 * m / denom), mini-batches of `batch` rows in the order `order[ep]`
    (None: 0 .. N-1).  Pinned by G17 (the reference's own CNN.fit run under torch).  -> weight, bias, per-epoch mean batch loss"""
    f32 = np.float32
    Bt = np.ascontiguousarray(np.transpose(np.asarray(B), (0, 2, 1)), dtype=f32)           # as_torch_tensor: (N, A, W) float
    y = np.asarray(y)
    N, A, W = Bt.shape
    w = np.array(weight, dtype=f32)
    b = np.array(bias, dtype=f32)
    S = w.shape[2]
    pad = (S - 1) // 2
    Bp = np.zeros((N, A, W + 2 * pad), f32)
    Bp[:, :, pad:pad + W] = Bt
    mw, vw, mb, vb = np.zeros_like(w), np.zeros_like(w), np.zeros_like(b), np.zeros_like(b)
    b1, b2 = float(betas[0]), float(betas[1])
    t = 0
    losses = np.zeros(int(epochs), np.float64)
    for ep in range(int(epochs)):
        rows = np.arange(N) if order is None else np.asarray(order[ep])
        nbat = (N + batch - 1) // batch
        for k in range(nbat):
            idx = rows[k * batch:(k + 1) * batch]
            nb = len(idx)
            V = np.lib.stride_tricks.sliding_window_view(Bp[idx], S, axis=2)              # (nb, A_in, W, S)
            z = np.einsum("nawk,cak->ncw", V, w, optimize=True).astype(f32) + b[None, :, None]
            z = z - z.max(axis=1, keepdims=True)
            e = np.exp(z, dtype=f32)
            p = e / e.sum(axis=1, keepdims=True, dtype=f32)
            py = np.take_along_axis(p, y[idx][:, None, :].astype(np.int64), axis=1)[:, 0, :]   # (nb, W)
            losses[ep] += float(np.mean(-np.log(py + f32(log_eps), dtype=f32), dtype=np.float64)) / nbat
            k_ = (py / (py + f32(log_eps)) / f32(nb * W)).astype(f32)
            onehot = (np.arange(A)[None, :, None] == y[idx][:, None, :]).astype(f32)
            g = (k_[:, None, :] * (p - onehot)).astype(f32)                                  # dL/dlogit (nb, A_out, W)
            gw = np.einsum("ncw,nawk->cak", g, V, optimize=True).astype(f32)
            gb = g.sum(axis=(0, 2), dtype=f32)
            t += 1
            step = lr / (1.0 - b1 ** t)
            bc2s = np.sqrt(1.0 - b2 ** t)
            for prm, grd, m_, v_ in ((w, gw, mw, vw), (b, gb, mb, vb)):
                m_ *= f32(b1); m_ += grd * f32(1.0 - b1)
                v_ *= f32(b2); v_ += (grd * grd) * f32(1.0 - b2)
                denom = np.sqrt(v_) / f32(bc2s) + f32(eps)
                prm -= f32(step) * (m_ / denom)
    return w, b, losses


def cov_sample(M, alpha=0.6, beta=1.0, seed=37):
    """CovSample (string_kernel.py:80-89): legacy MT19937 stream, one draw per m in 2..M."""
    rs = np.random.RandomState(seed)  # == np.random.seed(seed); np.random.rand() draws
    Ms = [1]
    for m in range(2, M + 1):
        if (1 - (alpha ** (m - Ms[-1] + 1))) * (m ** (-beta)) >= rs.random_sample():
            Ms.append(m)
    return Ms


def ms_ohe(Ms, M):
    o = np.zeros(M + 2, dtype=np.uint8)
    o[np.asarray(Ms)] = 1
    return o


def covrsk(Xq, Xt, Ms=None):
    Xq = np.ascontiguousarray(Xq, dtype=np.int8)
    Xt = np.ascontiguousarray(Xt, dtype=np.int8)
    Nq, Mw = Xq.shape
    Nt = Xt.shape[0]
    if Ms is None:
        Ms = cov_sample(Mw)
    ohe = ms_ohe(Ms, Mw)
    K = np.empty((Nq, Nt), dtype=np.int64)
    _chk(lib().gnxo_covrsk(_p(Xq), C.c_int64(Nq), C.c_int64(Mw), _p(Xt), C.c_int64(Nt), C.c_int64(Mw), C.c_int64(Mw),
                           _p(ohe), _p(K)), "covrsk")
    return K


def string_kernel(Xq, Xt):
    Xq = np.ascontiguousarray(Xq, dtype=np.int8)
    Xt = np.ascontiguousarray(Xt, dtype=np.int8)
    Nq, Mw = Xq.shape
    Nt = Xt.shape[0]
    K = np.empty((Nq, Nt), dtype=np.int64)
    _chk(lib().gnxo_string_kernel(_p(Xq), C.c_int64(Nq), C.c_int64(Mw), _p(Xt), C.c_int64(Nt), C.c_int64(Mw),
                                  C.c_int64(Mw), _p(K)), "string_kernel")
    return K


def poly_kernel(Xq, Xt, run_value, p):
    """poly_kernel (string_kernel.py:40-61): run_value = np.arange(width+1) ** p -> K (Nq, Nt) int64"""
    Xq = np.ascontiguousarray(Xq, dtype=np.int8)
    Xt = np.ascontiguousarray(Xt, dtype=np.int8)
    rv = np.ascontiguousarray(run_value, dtype=np.float64)
    Nq, Mw = Xq.shape
    Nt = Xt.shape[0]
    assert len(rv) >= Mw + 1
    K = np.empty((Nq, Nt), dtype=np.int64)
    _chk(lib().gnxo_poly_kernel(_p(Xq), C.c_int64(Nq), C.c_int64(Mw), _p(Xt), C.c_int64(Nt), C.c_int64(Mw), C.c_int64(Mw),
                                _p(rv), C.c_double(p), _p(K)), "poly_kernel")
    return K


def svc_predict_proba(K, support, dual, intercept, probA, probB, n_support):
    K = np.ascontiguousarray(K, dtype=np.int64)
    support = np.ascontiguousarray(support, dtype=np.int32)
    dual = np.ascontiguousarray(dual, dtype=np.float64)
    intercept = np.ascontiguousarray(intercept, dtype=np.float64)
    probA = np.ascontiguousarray(probA, dtype=np.float64)
    probB = np.ascontiguousarray(probB, dtype=np.float64)
    n_support = np.ascontiguousarray(n_support, dtype=np.int32)
    Nq, Nt = K.shape
    k = len(n_support)
    out = np.empty((Nq, k), dtype=np.float64)
    _chk(lib().gnxo_svc_predict_proba(_p(K), C.c_int64(Nq), C.c_int64(Nt), C.c_int(k), C.c_int64(len(support)),
                                      _p(support), _p(dual), _p(intercept), _p(probA), _p(probB), _p(n_support),
                                      _p(out)), "svc_predict_proba")
    return out


def base_windows(X, M, ctx):
    """Yield (i, Xw) exactly as base.py:146-164 slices them (pad + window i; last window wider)."""
    X = np.asarray(X)
    N, Cn = X.shape
    W = Cn // M
    rem = Cn - M * W
    M_ = M + 2 * ctx
    idx = np.array([lib().gnxo_pad_src(p, Cn, ctx) for p in range(Cn + 2 * ctx)], dtype=np.int64)
    for i in range(W):
        ln = M_ + rem if i == W - 1 else M_
        yield i, X[:, idx[i * M:i * M + ln]]


def base_covrsk(X, M, ctx, windows):
    """CovRSKBase.predict_proba: windows = list of dicts with Xfit, Ms, support, dual, intercept,
    probA, probB, n_support per window -> B (N,W,A) f64 (models.py:195-215 + sklearn SVC)."""
    out = []
    for i, Xw in base_windows(X, M, ctx):
        w = windows[i]
        K = poly_kernel(Xw, w["Xfit"], w["run_value"], w["poly_p"]) if w.get("poly_p") else covrsk(Xw, w["Xfit"], w["Ms"])
        out.append(svc_predict_proba(K, w["support"], w["dual"], w["intercept"], w["probA"], w["probB"], w["n_support"]))
    return np.swapaxes(np.array(out), 0, 1)


# ------------------------------------------------------------------------------------------------
# a9 Gnofix control loop (gnofix.py:58-208 with its default arguments, phasing.py:182-198)
# ------------------------------------------------------------------------------------------------
def gnofix(M_hap, P_hap, B, S, predict_rows, predict_labels, max_it=50):
    """One individual.  M_hap/P_hap: (C,) SNP vectors; B: (2,W,A) base probabilities.
    predict_rows(rows (R,S*A)) -> (R,A) = smoother.model.predict_proba (gnofix.py:157);
    predict_labels(B (2,W,A)) -> (2,W) = smoother.predict (gnofix.py:80,190).
    Returns X_m, X_p, Y_m, Y_p, tracker(2,W), n_switches."""
    B = np.array(B, copy=True)
    _, W, A = B.shape
    window_size = len(M_hap) // W  # gnofix.py:74 (C//W, may differ from M)
    X_m = np.array(M_hap, dtype=int, copy=True)
    X_p = np.array(P_hap, dtype=int, copy=True)
    Y_m, Y_p = predict_labels(B).reshape(2, W)
    half = (S - 1) // 2
    c_lo, c_hi = half, W - 1 - half  # centers[0], centers[-1]  (gnofix.py:83)
    trk_m, trk_p = np.zeros(W, dtype=int), np.ones(W, dtype=int)
    seen = []
    n_switch = 0
    for _ in range(max_it):
        if any(np.array_equal(X_m, s) for s in seen):  # gnofix.py:108-113
            break
        seen.append(X_m)
        for w in range(1, W):
            if Y_m[w] != Y_m[w - 1] or Y_p[w] != Y_p[w - 1]:  # check(): "disc_smooth" (gnofix.py:32)
                center = min(max(w, c_lo), c_hi)  # gnofix.py:122-127
                lo, hi = center - half, center + half + 1  # scope (gnofix.py:130)
                m_o, p_o = B[0, lo:hi], B[1, lo:hi]
                m_s = np.concatenate([B[0, lo:w], B[1, w:hi]])  # single switch at w (gnofix.py:134,144-153)
                p_s = np.concatenate([B[1, lo:w], B[0, w:hi]])
                rows = np.stack([m_o, p_o, m_s, p_s]).reshape(4, -1)
                outs = np.asarray(predict_rows(rows)).reshape(2, 2, A)
                probs = outs.max(axis=2).max(axis=1)  # prob_comp="max" (gnofix.py:162-163)
                if probs[1] * 0.5 > probs[0] * 0.5:  # prior_switch_prob=0.5 (gnofix.py:171)
                    Bm = np.concatenate([B[0, :w], B[1, w:]])
                    Bp = np.concatenate([B[1, :w], B[0, w:]])
                    B = np.stack([Bm, Bp])
                    trk_m, trk_p = (np.concatenate([trk_m[:w], trk_p[w:]]), np.concatenate([trk_p[:w], trk_m[w:]]))
                    i = w * window_size  # correct_phase_error (phasing.py:188-198)
                    X_m, X_p = (np.concatenate([X_m[:i], X_p[i:]]), np.concatenate([X_p[:i], X_m[i:]]))
                    Y_m, Y_p = predict_labels(B).reshape(2, W)
                    n_switch += 1
    return X_m, X_p, Y_m, Y_p, np.stack([trk_m, trk_p]), n_switch


class OracleXGBSmoother:
    """Duck-typed stand-in for the reference's XGB_Smoother (smooth.py:40-65 + models.py:8-24):
    .S .W .A .gnofix .model.predict_proba / .predict_proba / .predict — lets the REFERENCE's own
    gnofix()/Gnomix.phase() run with the oracle tree walker plugged in (golden G4/G5)."""

    def __init__(self, trees: Trees, W, A, S):
        self.trees, self.W, self.A, self.S = trees, W, A, S
        self.gnofix = True
        self.calibrate = False
        self.model = self

    def _model_predict_proba(self, rows):
        return xgb_predict_proba(self.trees, np.asarray(rows, dtype=np.float32))

    def predict_proba(self, B):
        B = np.asarray(B)
        if B.ndim == 2:  # used as .model.predict_proba(rows)
            return self._model_predict_proba(B)
        return smooth_xgb(self.trees, B, self.S)[0]

    def predict(self, B):
        return np.argmax(self.predict_proba(B), axis=-1)


# ----------------------------------------------------------------------------------------------------------------------
# f4  Base.train for the logistic base (src/Base/base.py:104-127 -> sklearn LogisticRegression(penalty="l2", C=3.,
#     solver="liblinear", max_iter=1000).fit per window, src/Base/models.py:12-21).  liblinear's primal L2R_LR problem per
#     class (one-vs-rest; A == 2: one problem, positive class 1), bias = regularised extra feature of value 1:
#         f(w) = 1/2 w'w + C sum_i log(1 + exp(-y_i w'x_i))
#     f is strictly convex: the fit is its unique minimiser, which liblinear approximates to tol = 1e-4.  This restatement is
#     a dense exact Newton method in numpy (small windows only: it factors the (width+1)^2 Hessian), run to 1e-12.
# ----------------------------------------------------------------------------------------------------------------------
def lr_objective(w, Xb, ypm, C_reg=3.0):
    z = Xb @ w
    return 0.5 * float(w @ w) + C_reg * float(np.sum(np.logaddexp(0.0, -ypm * z)))


def _lr_fit_one(Xb, ypm, C_reg, tol=1e-12, max_it=100):
    n = Xb.shape[1]
    w = np.zeros(n)
    g0 = None
    for _ in range(max_it):
        t = ypm * (Xb @ w)
        sg = 1.0 / (1.0 + np.exp(-t))
        g = w + C_reg * (Xb.T @ ((sg - 1.0) * ypm))
        gn = float(np.linalg.norm(g))
        if g0 is None:
            g0 = gn
        if gn <= tol * g0:
            break
        D = C_reg * sg * (1.0 - sg)
        H = np.eye(n) + (Xb.T * D) @ Xb
        s = -np.linalg.solve(H, g)
        f0, a = lr_objective(w, Xb, ypm, C_reg), 1.0
        while lr_objective(w + a * s, Xb, ypm, C_reg) - f0 > 0.01 * a * float(g @ s) and a > 1e-12:
            a *= 0.5
        w = w + a * s
    return w


def train_lr(X, y, M, ctx, A, C_reg=3.0):
    """-> (coef (W, A, ldc), intercept (W, A)) in the layout of GnxModelData.lr_coef / lr_intercept (A == 2: rows -w, +w)"""
    X = np.ascontiguousarray(X, dtype=np.int8)
    N, Cn = X.shape
    W = Cn // M
    rem = Cn - M * W
    ldc = M + 2 * ctx + rem
    coef, icpt = np.zeros((W, A, ldc)), np.zeros((W, A))
    for i, Xw in base_windows(X, M, ctx):
        Xb = np.concatenate([Xw.astype(np.float64), np.ones((N, 1))], axis=1)
        for a in ([1] if A == 2 else range(A)):
            w = _lr_fit_one(Xb, np.where(y[:, i] == a, 1.0, -1.0), C_reg)
            if A == 2:
                coef[i, 0, :Xw.shape[1]], coef[i, 1, :Xw.shape[1]] = -w[:-1], w[:-1]
                icpt[i] = [-w[-1], w[-1]]
            else:
                coef[i, a, :Xw.shape[1]] = w[:-1]
                icpt[i, a] = w[-1]
    return coef, icpt


# ------------------------------------------------------------------------------------------------
# f4 training the tree smoother (histogram gradient boosting, fixed-point sums: see gnx_oracle.c)
# ------------------------------------------------------------------------------------------------
class _CGbtParams(C.Structure):
    _fields_ = [("n_rounds", C.c_int32), ("max_depth", C.c_int32), ("max_bin", C.c_int32), ("reserved", C.c_int32),
                ("eta", C.c_double), ("lam", C.c_double), ("gamma", C.c_double), ("min_child_weight", C.c_double),
                ("base_score", C.c_double)]


def train_gbt(B, y, S, n_rounds=100, max_depth=4, eta=0.1, lam=1.0, gamma=0.0, min_child_weight=1.0, max_bin=256,
              base_score=0.5, exact=False):
    """Smoother.train of XGB_Smoother (src/Smooth/smooth.py:28-38, src/Smooth/models.py:14-20) in the histogram form, or with
    exact=True as exact greedy split enumeration (every boundary between two distinct feature values of a node's rows).
    B (N, W, A) base probabilities, y (N, W) labels.  Returns (Trees, losses (n_rounds+1,))."""
    B = np.ascontiguousarray(B)
    is64 = B.dtype == np.float64
    if not is64:
        B = np.ascontiguousarray(B, dtype=np.float32)
    N, W, A = B.shape
    y = np.ascontiguousarray(y, dtype=np.int32).reshape(N, W)
    T = n_rounds * A
    tree_off = np.zeros(T + 1, np.int32); tree_class = np.zeros(T, np.int32)
    left = np.zeros(T * 63, np.int32); right = np.zeros(T * 63, np.int32); feat = np.zeros(T * 63, np.int32)
    cond = np.zeros(T * 63, np.float32); loss = np.zeros(n_rounds + 1, np.float64)
    P = _CGbtParams(n_rounds, max_depth, max_bin, 1 if exact else 0, eta, lam, gamma, min_child_weight, base_score)
    fn = lib().gnxo_train_gbt
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(_CGbtParams)] + [C.c_void_p] * 7
    nn = fn(_p(B), int(is64), _p(y), N, W, A, S, C.byref(P), _p(tree_off), _p(tree_class), _p(left), _p(right), _p(feat),
            _p(cond), _p(loss))
    if nn < 0:
        raise ValueError(f"oracle train_gbt failed with code {nn}")
    return Trees(tree_off, left[:nn].copy(), right[:nn].copy(), feat[:nn].copy(), cond[:nn].copy(), tree_class, A, base_score), loss

"""Training the tree smoother: Smoother.train of XGB_Smoother (src/Smooth/smooth.py:28-38,
src/Smooth/models.py:14-20).  xgboost is absent, so the checker is the CPU restatement of the same histogram boosting
(oracle/gnx_oracle.c: gnxo_train_gbt); every sum on both sides is fixed point, so the HIP trainer must return IDENTICAL
trees.  The algorithm itself is sanity-checked against scikit-learn's histogram boosting on the same rows."""
import numpy as np
import pytest


def _problem(N, W, A, seed, noise=0.5):
    """admixed haplotypes: piecewise-constant ancestry, noisy base probabilities, labels = the true ancestry"""
    from gnomix_amd import synth
    rng = np.random.RandomState(seed)
    B = synth.synthetic_phased_individuals((N + 1) // 2, W, A, seed=seed, phase_errors=0, noise=0.02)[:N]
    y = np.argmax(B, -1).astype(np.int32)
    y[0, :A] = np.arange(A)                                   # every population present (smooth.py:30)
    Bn = B + rng.normal(0, noise, B.shape)
    Bn = np.clip(Bn, 1e-4, None)
    Bn /= Bn.sum(-1, keepdims=True)
    return Bn, y


def test_oracle_gbt_learns_and_is_deterministic(oracle):
    B, y = _problem(40, 60, 3, seed=1)
    T1, loss1 = oracle.train_gbt(B, y, 11, n_rounds=15)
    T2, loss2 = oracle.train_gbt(B, y, 11, n_rounds=15)
    assert np.array_equal(T1.cond, T2.cond) and np.array_equal(T1.feat, T2.feat) and np.array_equal(loss1, loss2)
    assert abs(loss1[0] - np.log(3)) < 1e-12                  # uniform start: base_score for every class
    assert np.all(np.diff(loss1) < 0) and loss1[-1] < 0.4 * loss1[0]
    _, lab = oracle.smooth_xgb(T1, B, 11)
    assert (lab == y).mean() > (np.argmax(B, -1) == y).mean() + 0.02     # the smoother beats the raw arg-max
    assert T1.n_trees == 45 and np.array_equal(T1.tree_class, np.arange(45) % 3)
    # thresholds live on the 1/65536 grid and features inside the sliding window
    internal = T1.left >= 0
    assert np.all(T1.cond[internal] * 65536 == np.round(T1.cond[internal] * 65536))
    assert T1.feat.max() < 11 * 3


def test_oracle_gbt_tracks_sklearn_histogram_boosting(oracle):
    """same rows, depth and rounds: the log loss after training is in the same range as scikit-learn's
    HistGradientBoostingClassifier.  xgboost's softmax hessian is 2p(1-p), scikit-learn's p(1-p): a Newton leaf -G/(H+l) of
    ours is half of theirs, so their learning rate is halved for the comparison.  Not a parity statement."""
    ens = pytest.importorskip("sklearn.ensemble")
    B, y = _problem(60, 70, 3, seed=5)
    S = 11
    T, loss = oracle.train_gbt(B, y, S, n_rounds=20)
    rows = oracle.slide_window(B, S)
    clf = ens.HistGradientBoostingClassifier(max_iter=20, max_depth=4, learning_rate=0.05, l2_regularization=0.5, max_bins=255,
                                             early_stopping=False, min_samples_leaf=1, random_state=1)
    clf.fit(rows.reshape(-1, rows.shape[-1]), y.reshape(-1))
    p = clf.predict_proba(rows.reshape(-1, rows.shape[-1]))
    sk_loss = -np.mean(np.log(p[np.arange(y.size), y.reshape(-1)]))
    assert 0.6 * sk_loss < loss[-1] < 1.6 * sk_loss


def test_oracle_exact_greedy_root_split_vs_numpy_enumeration(oracle):
    """tree_method = exact: the first tree's root split against a numpy enumeration that shares nothing with the C code — every
    feature, every boundary between two distinct values, float64 sums of the first round's gradients (all classes at p = 1/A)"""
    B, y = _problem(14, 30, 3, seed=3)
    B = B.astype(np.float32)
    S, A, lam = 7, 3, 1.0
    T, loss = oracle.train_gbt(B, y, S, n_rounds=2, max_depth=3, exact=True)
    assert np.all(np.diff(loss) < 0)
    rows = oracle.slide_window(B, S).reshape(-1, S * A)                   # (N*W, S*A) float32, the matrix xgboost would see
    yy = y.reshape(-1)
    p0 = 1.0 / A                                                           # margins start equal
    g = p0 - (yy == 0)                                                     # tree 0 = class 0
    h = np.full(len(yy), 2 * p0 * (1 - p0))
    G, H = g.sum(), h.sum()
    best = (1e-6, -1, None)
    for f in range(S * A):
        o = np.argsort(rows[:, f], kind="stable")
        v, gl, hl = rows[o, f], np.cumsum(g[o]), np.cumsum(h[o])
        cut = np.nonzero(v[:-1] < v[1:])[0]
        for q in cut:
            if hl[q] < 1.0 or H - hl[q] < 1.0:
                continue
            gain = gl[q] ** 2 / (hl[q] + lam) + (G - gl[q]) ** 2 / (H - hl[q] + lam) - G * G / (H + lam)
            if gain > best[0] + 1e-9:                                      # first best wins (1e-9: float64 vs fixed-point sums)
                thr = np.float32((v[q] + v[q + 1]) * np.float32(0.5))
                best = (gain, f, thr if thr > v[q] else v[q + 1])
    assert T.left[0] >= 0 and T.feat[0] == best[1] and T.cond[0] == best[2], (T.feat[0], T.cond[0], best)
    # the split really separates the root's rows as the sums said: left rows are exactly those with value < threshold
    left_rows = rows[:, best[1]] < T.cond[0]
    assert 0 < left_rows.sum() < len(yy)
    # exact thresholds are midpoints of data values, not grid points; the histogram form's are grid points
    Th, _ = oracle.train_gbt(B, y, S, n_rounds=2, max_depth=3)
    internal = T.left >= 0
    assert not np.all(T.cond[internal] * 65536 == np.round(T.cond[internal] * 65536))
    assert np.all(Th.cond[Th.left >= 0] * 65536 == np.round(Th.cond[Th.left >= 0] * 65536))


def test_oracle_gbt_rejects_bad_geometry(oracle):
    B, y = _problem(4, 20, 3, seed=2)
    with pytest.raises(ValueError):
        oracle.train_gbt(B, y, 11)          # W < 2 S (Smooth/models.py:13)
    with pytest.raises(ValueError):
        oracle.train_gbt(B, y, 4)           # even S


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def ga():
    import gnomix_amd
    return gnomix_amd


CASES = [
    dict(N=24, W=50, A=3, S=11, kw=dict(n_rounds=8)),
    dict(N=30, W=64, A=7, S=15, kw=dict(n_rounds=5)),
    dict(N=16, W=40, A=2, S=5, kw=dict(n_rounds=12, max_depth=3)),
    dict(N=20, W=70, A=4, S=21, kw=dict(n_rounds=4, max_depth=5, max_bin=64)),
    dict(N=18, W=48, A=3, S=9, kw=dict(n_rounds=6, gamma=0.5, min_child_weight=3.0, reg_lambda=0.0, learning_rate=0.3)),
    dict(N=9, W=170, A=12, S=75, kw=dict(n_rounds=2)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "N%dW%dA%dS%d" % (c["N"], c["W"], c["A"], c["S"]))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gbt_trees_identical_to_oracle(ga, oracle, case, dtype):
    from gnomix_amd import train
    B, y = _problem(case["N"], case["W"], case["A"], seed=case["N"] + case["S"])
    B = B.astype(dtype)
    kw = dict(case["kw"])
    okw = dict(kw)
    if "reg_lambda" in okw: okw["lam"] = okw.pop("reg_lambda")
    if "learning_rate" in okw: okw["eta"] = okw.pop("learning_rate")
    T, loss_ref = oracle.train_gbt(B, y, case["S"], **okw)
    trees, loss = train.train_gbt_arrays(B, y, case["S"], **kw)
    assert np.array_equal(trees["tree_off"], T.tree_off)
    assert np.array_equal(trees["left"], T.left) and np.array_equal(trees["right"], T.right)
    assert np.array_equal(trees["feat"], T.feat)
    assert np.array_equal(trees["cond"].view(np.uint32), T.cond.view(np.uint32))          # thresholds AND leaf values, bit for bit
    assert np.array_equal(trees["tree_class"], T.tree_class)
    assert np.allclose(loss, loss_ref, rtol=0, atol=1e-6)


EXACT_CASES = [
    dict(N=14, W=40, A=3, S=11, kw=dict(n_rounds=6)),
    dict(N=10, W=64, A=7, S=15, kw=dict(n_rounds=3)),
    dict(N=12, W=30, A=2, S=5, kw=dict(n_rounds=8, max_depth=3)),
    dict(N=8, W=60, A=4, S=21, kw=dict(n_rounds=2, max_depth=5)),
    dict(N=12, W=36, A=3, S=9, kw=dict(n_rounds=5, gamma=0.5, min_child_weight=3.0, reg_lambda=0.0, learning_rate=0.3)),
    dict(N=3, W=170, A=12, S=75, kw=dict(n_rounds=1)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", EXACT_CASES, ids=lambda c: "N%dW%dA%dS%d" % (c["N"], c["W"], c["A"], c["S"]))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gbt_exact_greedy_trees_identical_to_oracle(ga, oracle, case, dtype):
    """tree_method="exact" (k_gbt_exact_scan: sorted strip positions per class column, masked wave scans per node) against the
    oracle's qsort-per-node-and-feature restatement: identical trees, thresholds and leaf values bit for bit.  Inputs with repeated
    values (rounded probabilities) exercise the distinct-value rule."""
    from gnomix_amd import train
    B, y = _problem(case["N"], case["W"], case["A"], seed=case["N"] + case["S"])
    if case["S"] == 9:
        B = np.round(B, 2) + 1e-3                                            # many ties
        B /= B.sum(-1, keepdims=True)
    B = B.astype(dtype)
    kw = dict(case["kw"])
    okw = dict(kw)
    if "reg_lambda" in okw: okw["lam"] = okw.pop("reg_lambda")
    if "learning_rate" in okw: okw["eta"] = okw.pop("learning_rate")
    T, loss_ref = oracle.train_gbt(B, y, case["S"], exact=True, **okw)
    trees, loss = train.train_gbt_arrays(B, y, case["S"], tree_method="exact", **kw)
    assert np.array_equal(trees["tree_off"], T.tree_off)
    assert np.array_equal(trees["left"], T.left) and np.array_equal(trees["right"], T.right)
    assert np.array_equal(trees["feat"], T.feat)
    assert np.array_equal(trees["cond"].view(np.uint32), T.cond.view(np.uint32))
    assert np.allclose(loss, loss_ref, rtol=0, atol=1e-6)
    # the trained ensemble is an ordinary smoother model: labels on the training rows through the HIP smoother
    d = ga.GnxModelData(C=case["W"] * 10 + 3, M=10, A=case["A"], S=case["S"], context=0, smooth_kind="xgb", **trees)
    _, lab = ga.DeviceModel(d).smooth_predict(B.astype(np.float32))
    _, l_ref = oracle.smooth_xgb(T, B.astype(np.float32), case["S"])
    assert np.array_equal(lab, l_ref)


@pytest.mark.gpu
def test_gbt_device_tensors_and_smoother_plugin(ga, oracle):
    """CUDA tensors in; HipSmoother.train swaps in a model that predicts exactly what the oracle predicts with the same trees"""
    import torch
    from gnomix_amd import synth, train
    N, W, A, S = 40, 90, 5, 21
    B, y = _problem(N, W, A, seed=9)
    trees_h, loss_h = train.train_gbt_arrays(B, y, S, n_rounds=6)
    trees_d, loss_d = train.train_gbt_arrays(torch.from_numpy(B).cuda(), torch.from_numpy(y).cuda(), S, n_rounds=6)
    for k in trees_h:
        assert np.array_equal(trees_h[k], trees_d[k]), k
    assert np.array_equal(loss_h, loss_d)
    d = synth.synthetic_model(C=W * 10 + 3, M=10, A=A, S=S, n_rounds=2, seed=3)
    sm = ga.HipSmoother(ga.DeviceModel(d))
    sm.train(B, y, n_rounds=6)
    T = oracle.Trees(trees_h["tree_off"], trees_h["left"], trees_h["right"], trees_h["feat"], trees_h["cond"], trees_h["tree_class"], A, 0.5)
    p_ref, l_ref = oracle.smooth_xgb(T, B, S)
    assert np.array_equal(sm.predict(B), l_ref)
    assert np.max(np.abs(sm.predict_proba(B) - p_ref)) <= 1e-5
    assert (l_ref == y).mean() > (np.argmax(B, -1) == y).mean()
    assert sm.gnofix and sm.model.predict_proba(np.asarray(oracle.slide_window(B[:1], S)).reshape(-1, S * A)).shape == (W, A)
    with pytest.raises(AssertionError, match="does not include all populations"):
        sm.train(B, np.zeros_like(y))


@pytest.mark.gpu
def test_gnomix_train_end_to_end(ga, oracle):
    """Gnomix.train (src/model.py:104-167): base on train1, smoother on the base's probabilities of train2, base again on
    everything — on the device, and the trained model labels held-out admixed haplotypes better than its base alone"""
    from gnomix_amd import synth
    A, M, W, S = 3, 40, 30, 7
    C = M * W + 13
    rng = np.random.RandomState(4)
    freq = np.clip(rng.uniform(0.2, 0.8, size=(1, C)) + rng.normal(0, 0.13, size=(A, C)), 0.02, 0.98)   # weakly differentiated populations

    def haplotypes(n, seed):
        r = np.random.RandomState(seed)
        y = np.zeros((n, W), np.int32)
        for i in range(n):
            cuts = np.sort(r.choice(np.arange(3, W - 3), size=2, replace=False))
            a = r.randint(A)
            for lo, hi in zip([0, *cuts], [*cuts, W]):
                y[i, lo:hi] = a
                a = (a + 1 + r.randint(A - 1)) % A
        ysnp = np.concatenate([np.repeat(y, M, axis=1), np.repeat(y[:, -1:], C - M * W, axis=1)], axis=1)
        X = (r.uniform(size=(n, C)) < freq[ysnp, np.arange(C)[None, :]]).astype(np.int8)
        return X, y

    t1, t2, v = haplotypes(120, 1), haplotypes(80, 2), haplotypes(40, 3)
    d = ga.GnxModelData(C=C, M=M, A=A, S=S, context=M // 2)
    d.base_kind, d.lr_coef, d.lr_intercept = "logistic", np.zeros((W, A, M + 2 * (M // 2) + C - M * W)), np.zeros((W, A))
    for k, val in synth.synthetic_trees(1, A, S * A, seed=1).items():
        setattr(d, k, val)
    g = ga.HipGnomix(d)
    g.train((t1, t2, v), n_rounds=25)
    assert set(g.accuracies) == {k + s for k in ('base_train_acc', 'smooth_train_acc', 'base_val_acc', 'smooth_val_acc') for s in ('', '_bal')}
    assert g.Confusion_Matrices['val'][0].shape == (A, A)
    # held-out haplotypes, scored BEFORE the base is refitted on everything (src/model.py:127-151): the smoother beats its base
    assert g.accuracies['smooth_val_acc'] > 80 and g.accuracies['smooth_val_acc'] > g.accuracies['base_val_acc'] + 3, g.accuracies
    assert g.predict(v[0]).shape == v[1].shape
    assert np.all(np.diff(g.smooth.train_loss) < 0)
    # a trained model survives save / load
    import tempfile, os
    with tempfile.TemporaryDirectory() as td:
        path = g.save(os.path.join(td, "trained.gnx"))
        g2 = ga.HipGnomix.load(path)
        assert np.array_equal(g2.predict(v[0][:10]), g.predict(v[0][:10]))
    # the calibrated variant (config `calibrate: True`): Gnomix.train also fits the isotonic maps on train1 (src/model.py:119-124)
    gc = ga.HipGnomix(d, calibrate=True)
    np.random.seed(5)
    gc.train((t1, t2, v), n_rounds=8, evaluate=False)
    assert gc.dev.data.calib_off is not None and gc.smooth.calibrator
    pc = gc.predict_proba(v[0][:6])
    assert np.allclose(pc.sum(-1), 1.0, atol=1e-6)
    # ... and the FINAL model (base refitted last, src/model.py:153-167) still calibrates: every re-binding of the device
    # model carries the switch over (it used to come back uncalibrated float32)
    assert gc.smooth.calibrate and gc.dev.calibrated and pc.dtype == np.float64
    gc.smooth.calibrate = False
    raw = gc.predict_proba(v[0][:6])
    gc.smooth.calibrate = True
    assert raw.dtype == np.float32 and not np.allclose(raw, pc, atol=1e-4)
    want = gc.dev.calibrate_rows(raw.reshape(-1, A)).reshape(raw.shape)
    assert np.allclose(pc, want, atol=1e-12) and np.array_equal(gc.predict_proba(v[0][:6]), pc)


@pytest.mark.gpu
def test_gbt_rejects_bad_arguments(ga):
    from gnomix_amd import train, _lib
    B, y = _problem(6, 40, 3, seed=2)
    with pytest.raises(_lib.GnxError, match="Smoother size to large"):
        train.train_gbt_arrays(B, y, 31)
    with pytest.raises(_lib.GnxError, match="label outside"):
        train.train_gbt_arrays(B, y + 5, 11)
    with pytest.raises(_lib.GnxError, match="max_depth"):
        train.train_gbt_arrays(B, y, 11, max_depth=9)
    with pytest.raises(ValueError):
        train.train_gbt_arrays(B, y[:, :-1], 11)


@pytest.mark.gpu
def test_gbt_full_size_properties(ga, oracle):
    """config-2 geometry (W = 370, A = 7, S = 75), 3 000 haplotypes = 1.1 M rows x 525 features: too large for the scalar oracle, so
    size-independent properties — the loss falls every round, two runs give identical trees (fixed-point sums: no dependence on
    scheduling), the first trees equal the oracle's on a subset small enough for it when trained on that subset, and the
    trained smoother labels the training haplotypes better than the raw arg-max."""
    from gnomix_amd import synth, train
    N, W, A, S = 3000, 370, 7, 75
    B, y = _problem(N, W, A, seed=77, noise=0.45)
    t1, l1 = train.train_gbt_arrays(B, y, S, n_rounds=10)
    t2, l2 = train.train_gbt_arrays(B, y, S, n_rounds=10)
    for k in t1:
        assert np.array_equal(t1[k], t2[k]), k
    assert np.array_equal(l1, l2) and np.all(np.diff(l1) < 0) and abs(l1[0] - np.log(A)) < 1e-6
    assert t1["feat"].max() < S * A and np.array_equal(t1["tree_class"], np.arange(10 * A) % A)
    # a subset the oracle can do: same trees
    Bs, ys = B[:24], y[:24].copy()
    ys[0, :A] = np.arange(A)
    T, _ = oracle.train_gbt(Bs, ys, S, n_rounds=2)
    ts, _ = train.train_gbt_arrays(Bs, ys, S, n_rounds=2)
    assert np.array_equal(ts["feat"], T.feat) and np.array_equal(ts["cond"].view(np.uint32), T.cond.view(np.uint32))
    d = synth.synthetic_model(C=W * 10 + 3, M=10, A=A, S=S, n_rounds=1, seed=3)
    for k, v in t1.items():
        setattr(d, k, v)
    lab = ga.DeviceModel(d).smooth_predict(B[:400], want_proba=False)[1]
    assert (lab == y[:400]).mean() > (np.argmax(B[:400], -1) == y[:400]).mean() + 0.05


@pytest.mark.gpu
def test_train_calibrator_end_to_end(ga, oracle):
    """Smoother.train_calibrator (smooth.py:81-92) on the device model: the maps fitted by gnx_fit_isotonic_f32 on the smoother's own
    probabilities equal scikit-learn's on the same rows, the re-loaded model applies them (k_calibrate == the reference's transform
    arithmetic, pinned by G8 elsewhere), rows still sum to one, and Gnomix.train(calibrate=True) runs the whole sequence."""
    iso = pytest.importorskip("sklearn.isotonic")
    from gnomix_amd import synth
    N, W, A, S = 120, 60, 4, 11
    B, y = _problem(N, W, A, seed=21)
    d = synth.synthetic_model(C=W * 10 + 3, M=10, A=A, S=S, n_rounds=2, seed=3)
    sm = ga.HipSmoother(ga.DeviceModel(d))
    sm.train(B, y, n_rounds=5)
    raw = sm.predict_proba(B)
    np.random.seed(11)
    idxs = np.random.choice(N, int(0.25 * N), replace=False)
    np.random.seed(11)
    sm.train_calibrator(B, y, frac=0.25)
    data = sm.dev.data
    assert data.calib_is_f32 and len(data.calib_off) == A + 1
    for i in range(A):
        m = iso.IsotonicRegression(out_of_bounds="clip").fit(raw[idxs].reshape(-1, A)[:, i], (y[idxs].reshape(-1) == i).astype(float))
        sl = slice(data.calib_off[i], data.calib_off[i + 1])
        assert np.array_equal(data.calib_x[sl], m.X_thresholds_.astype(np.float64)) and np.array_equal(data.calib_y[sl], m.y_thresholds_.astype(np.float64))
    assert np.array_equal(sm.predict_proba(B), raw)                      # calibrate flag still off
    sm.calibrate = True
    cal = sm.predict_proba(B)
    assert cal.shape == raw.shape and np.allclose(cal.sum(-1), 1.0, atol=1e-6) and not np.array_equal(cal, raw)
    ref = sm.dev.calibrate_rows(raw.reshape(-1, A)).reshape(raw.shape)
    assert np.array_equal(cal, ref)

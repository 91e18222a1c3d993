"""SURVEY.md §8 f4 — training the logistic base.  The reference's fit is sklearn/liblinear's approximation (tol 1e-4) of the
unique minimiser of a strictly convex objective, so parity is stated on that objective:

  CPU   the oracle's exact Newton restatement (oracle.train_lr) vs golden G16 = the reference's own
        LogisticRegressionBase.train: the oracle's objective is <= the reference's for every problem, and the two fits agree to
        the size of the reference's stopping error (coefficients, Base.predict_proba of held-out haplotypes, argmax labels);
  GPU   gnx_train_logistic vs the oracle to 1e-7 (both converge to the same point), vs G16 as above, through
        HipBase.train / HipGnomix.train_base end to end, and at chr22 size through the optimality conditions (gradient norm).
"""
import numpy as np
import pytest

from conftest import load_golden


def _problems(g, tag):
    C, M, A, ctx = (int(g[tag + "_" + k]) for k in ("C", "M", "A", "ctx"))
    return C, M, A, ctx, g[tag + "_Xt"], g[tag + "_yt"], g[tag + "_coef"], g[tag + "_intercept"]


def _objectives(O, X, y, M, ctx, A, coef, icpt):
    out = []
    for i, Xw in O.base_windows(X, M, ctx):
        Xb = np.concatenate([Xw.astype(np.float64), np.ones((len(Xw), 1))], axis=1)
        for a in ([1] if A == 2 else range(A)):
            w = np.concatenate([coef[i, a, :Xw.shape[1]], [icpt[i, a]]])
            out.append(O.lr_objective(w, Xb, np.where(y[:, i] == a, 1.0, -1.0)))
    return np.array(out)


def check_fit_against_reference(O, g, tag, coef, icpt):
    C, M, A, ctx, Xt, yt, rcoef, ricpt = _problems(g, tag)
    f_ours, f_ref = _objectives(O, Xt, yt, M, ctx, A, coef, icpt), _objectives(O, Xt, yt, M, ctx, A, rcoef, ricpt)
    assert np.all(f_ours <= f_ref * (1 + 1e-12))                       # at least as optimal as liblinear's answer, every problem
    assert np.max((f_ref - f_ours) / f_ref) < 1e-4                     # ... which is itself close to the optimum
    assert np.max(np.abs(coef - rcoef)) < 2e-2 and np.max(np.abs(icpt - ricpt)) < 2e-2
    B_ref = g[tag + "_B"]
    B = O.base_lr(g[tag + "_Xq"], M, ctx, coef, icpt)
    assert np.max(np.abs(B - B_ref)) < 5e-3
    assert np.mean(np.argmax(B, -1) == np.argmax(B_ref, -1)) > 0.995
    if A == 2:
        assert np.array_equal(coef[:, 0], -coef[:, 1]) and np.array_equal(icpt[:, 0], -icpt[:, 1])


@pytest.mark.parametrize("tag", ["m", "b"])
def test_oracle_fit_vs_reference_G16(oracle, tag):
    g = load_golden("G16_lr_train.npz")
    C, M, A, ctx, Xt, yt, _, _ = _problems(g, tag)
    coef, icpt = oracle.train_lr(Xt, yt, M, ctx, A)
    check_fit_against_reference(oracle, g, tag, coef, icpt)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["m", "b"])
def test_hip_fit_vs_oracle_and_reference_G16(oracle, tag):
    import gnomix_amd
    from gnomix_amd import train
    g = load_golden("G16_lr_train.npz")
    C, M, A, ctx, Xt, yt, _, _ = _problems(g, tag)
    coef, icpt, info = train.train_logistic_arrays(Xt, yt, M, ctx, A, tol=1e-10)
    assert info["n_problems"] == (C // M) * (1 if A == 2 else A) and info["worst_rel_gradient"] <= 1e-10
    ocoef, oicpt = oracle.train_lr(Xt, yt, M, ctx, A)
    assert np.max(np.abs(coef - ocoef)) < 1e-7 and np.max(np.abs(icpt - oicpt)) < 1e-7      # the same minimiser
    check_fit_against_reference(oracle, g, tag, coef, icpt)


@pytest.mark.gpu
def test_hip_base_train_end_to_end(oracle):
    """HipBase.train / HipGnomix.train_base: fit on the device, reload, predict — against the reference's Base.predict_proba
    of ITS fit (G16) and against the oracle's fit run through the oracle's predictor"""
    import gnomix_amd
    from gnomix_amd import synth
    g = load_golden("G16_lr_train.npz")
    C, M, A, ctx, Xt, yt, _, _ = _problems(g, "m")
    d = synth.synthetic_model(C=C, M=M, A=A, S=5, context=ctx, seed=1, smooth=None)
    d.lr_coef, d.lr_intercept = np.zeros_like(d.lr_coef), np.zeros_like(d.lr_intercept)
    hg = gnomix_amd.HipGnomix(d)
    hg.train_base(Xt, yt)
    assert hg.base.train_info["newton_iterations"] > 0
    B = hg.base.predict_proba(g["m_Xq"])
    assert np.max(np.abs(B - g["m_B"])) < 5e-3
    ocoef, oicpt = oracle.train_lr(Xt, yt, M, ctx, A)
    assert np.max(np.abs(B - oracle.base_lr(g["m_Xq"], M, ctx, ocoef, oicpt))) < 1e-6
    with pytest.raises(ValueError):
        gnomix_amd.train.train_logistic_arrays(Xt, yt[:, :-1], M, ctx, A)
    with pytest.raises(ValueError):
        gnomix_amd.train.train_logistic_arrays(Xt, yt + 5, M, ctx, A)


@pytest.mark.gpu
def test_hip_fit_edge_geometries(oracle):
    """no context / context wider than a window (reflect padding reaches inner windows) / A = 7 / missing code 2 as a value"""
    from gnomix_amd import train
    rng = np.random.RandomState(5)
    for (C, M, ctx, A, N) in ((437, 50, 0, 3, 90), (333, 40, 60, 4, 70), (1037, 100, 50, 7, 120)):
        W = C // M
        X = (rng.random_sample((N, C)) < rng.uniform(0.2, 0.8, size=C)).astype(np.int8)
        X[rng.random_sample(X.shape) < 0.03] = 2
        y = rng.randint(0, A, size=(N, W)).astype(np.int32)
        y[:A] = np.arange(A)[:, None]
        coef, icpt, info = train.train_logistic_arrays(X, y, M, ctx, A, tol=1e-10)
        ocoef, oicpt = oracle.train_lr(X, y, M, ctx, A)
        assert np.max(np.abs(coef - ocoef)) < 1e-7 and np.max(np.abs(icpt - oicpt)) < 1e-7, (C, M, ctx, A)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_hip_fit_chr22_size_optimality():
    """config-2 geometry (C = 370 500, W = 370, A = 7, 2 000 training haplotypes): the returned point satisfies the optimality
    condition of every one of the 2 590 problems (|grad| <= 1e-8 |grad(0)|), and refitting is deterministic to rounding"""
    import torch
    from gnomix_amd import synth, train
    C, M, A, ctx, N = 370_500, 1000, 7, 500, 2000
    X = synth.synthetic_X(N, C, seed=3, miss=0.01)
    rng = np.random.RandomState(1)
    y = np.repeat(rng.randint(0, A, size=(N, 1)), C // M, axis=1).astype(np.int32)
    flip = rng.random_sample(y.shape) < 0.1
    y[flip] = rng.randint(0, A, size=int(flip.sum()))
    coef, icpt, info = train.train_logistic_arrays(X, y, M, ctx, A, tol=1e-8)
    assert info["n_problems"] == 2590 and info["worst_rel_gradient"] <= 1e-8
    assert np.isfinite(coef).all() and np.isfinite(icpt).all()
    coef2, icpt2, _ = train.train_logistic_arrays(X, y, M, ctx, A, tol=1e-8)
    assert np.max(np.abs(coef - coef2)) < 1e-9

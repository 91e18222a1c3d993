"""SURVEY.md §8 f4 — training the CRF smoother (CRF.fit, reference src/Smooth/crf.py:51-58 -> CRFsuite lbfgs, c1 = 0, c2 = 1).

CRFsuite is a third-party fitter that is absent here (parity unpinned until tests/golden/make_golden.py G13 runs on a host that has
it: tests/test_pins_thirdparty.py then compares its fit with ours).  What IS checked:
  CPU   the oracle's objective (oracle.crf_objective): its gradient against central differences, and its node marginals against
        oracle.smooth_crf — the restated CRFsuite inference that the device's CRF kernels are already held to;
  GPU   gnx_train_crf: objective and gradient norm at a given point vs the oracle (1e-10), the fit vs the oracle's independent
        optimiser (scipy L-BFGS-B: 5e-6 on the weights, objectives equal to 1e-10, the oracle's gradient at the device's answer < 1e-6), float32 / float64 inputs, A > 8 (several
        pairs per lane), HipSmoother.train end to end, argument checks.
"""
import numpy as np
import pytest


def _tracts(rng, N, W, A, noise=0.6):
    y = np.empty((N, W), np.int32)
    for i in range(N):
        a = rng.randint(A)
        for w in range(W):
            if rng.rand() < 0.12:
                a = rng.randint(A)
            y[i, w] = a
    y[:A, :] = np.arange(A)[:, None]
    B = rng.dirichlet(np.ones(A) * noise, size=(N, W))
    B[np.arange(N)[:, None], np.arange(W)[None, :], y] += rng.random_sample((N, W))
    return B / B.sum(-1, keepdims=True), y


def test_oracle_crf_gradient_and_marginals(oracle):
    rng = np.random.RandomState(3)
    N, W, A = 6, 11, 4
    B, y = _tracts(rng, N, W, A)
    st, tr = rng.normal(0, 0.7, (A, A)), rng.normal(0, 0.7, (A, A))
    f, gs, gt = oracle.crf_objective(B, y, st, tr)
    h = 1e-6
    for (arr, grad) in ((st, gs), (tr, gt)):
        for idx in [(0, 0), (1, 3), (3, 2)]:
            a = arr.copy(); a[idx] += h
            b = arr.copy(); b[idx] -= h
            fa = oracle.crf_objective(B, y, a if arr is st else st, a if arr is tr else tr)[0]
            fb = oracle.crf_objective(B, y, b if arr is st else st, b if arr is tr else tr)[0]
            assert abs((fa - fb) / (2 * h) - grad[idx]) < 1e-6 * max(1.0, abs(grad[idx]))
    # d f / d state[a][l] without the regulariser = sum (marginal - onehot) x: the marginals are smooth_crf's
    p, _ = oracle.smooth_crf(B, st, tr)
    onehot = (np.arange(A)[None, None, :] == y[:, :, None]).astype(np.float64)
    assert np.allclose(np.einsum("nta,ntl->al", B, p - onehot) + 2.0 * st, gs, rtol=0, atol=1e-10)


def test_oracle_crf_fit_is_a_stationary_point(oracle):
    rng = np.random.RandomState(4)
    B, y = _tracts(rng, 30, 25, 3)
    st, tr, f = oracle.crf_fit(B, y)
    f2, gs, gt = oracle.crf_objective(B, y, st, tr)
    assert abs(f - f2) < 1e-9 and max(np.abs(gs).max(), np.abs(gt).max()) < 1e-6
    assert f < oracle.crf_objective(B, y, np.zeros((3, 3)), np.zeros((3, 3)))[0]


@pytest.mark.gpu
@pytest.mark.parametrize("N,W,A,dtype", [(40, 30, 4, np.float64), (33, 317, 7, np.float32), (21, 19, 12, np.float64), (9, 5, 2, np.float64)])
def test_hip_crf_objective_and_fit_vs_oracle(oracle, N, W, A, dtype):
    from gnomix_amd.train import train_crf_arrays
    rng = np.random.RandomState(N + A)
    B, y = _tracts(rng, N, W, A)
    B = B.astype(dtype)
    # the evaluation alone: zero iterations at a random point
    st0, tr0 = rng.normal(0, 0.5, (A, A)), rng.normal(0, 0.5, (A, A))
    s_, t_, info = train_crf_arrays(B, y, max_iterations=0, state0=st0, trans0=tr0)
    assert np.array_equal(s_, st0) and np.array_equal(t_, tr0) and info["evaluations"] == 1
    f, gs, gt = oracle.crf_objective(B, y, st0, tr0)
    assert abs(info["objective"] - f) <= 1e-10 * abs(f)
    gn = np.sqrt(np.sum(gs * gs) + np.sum(gt * gt))
    assert abs(info["grad_norm"] - gn) <= 1e-9 * gn
    # the fit, from zeros
    st, tr, info = train_crf_arrays(B, y)
    so, to, fo = oracle.crf_fit(B, y)
    assert info["converged"] and info["grad_norm"] < 1e-7 * max(1.0, np.sqrt(np.sum(st * st) + np.sum(tr * tr))) * 1.01
    assert abs(info["objective"] - fo) <= 1e-10 * abs(fo)
    assert np.max(np.abs(st - so)) < 5e-6 and np.max(np.abs(tr - to)) < 5e-6
    _, gs, gt = oracle.crf_objective(B, y, st, tr)                  # the oracle's gradient at the device's answer
    assert max(np.abs(gs).max(), np.abs(gt).max()) < 1e-6


@pytest.mark.gpu
def test_hip_smoother_train_crf_end_to_end(oracle):
    import gnomix_amd as ga
    rng = np.random.RandomState(11)
    N, W, A = 120, 60, 5
    B, y = _tracts(rng, N, W, A, noise=0.8)
    d = ga.GnxModelData(C=W * 10 + 3, M=10, A=A, S=5, context=0, smooth_kind="crf", crf_state=np.zeros((A, A)), crf_trans=np.zeros((A, A)))
    sm = ga.HipSmoother(ga.DeviceModel(d))
    acc_base = np.mean(np.argmax(B, -1) == y)
    sm.train(B, y)
    assert sm.train_info["converged"]
    acc = np.mean(sm.predict(B) == y)
    assert acc > acc_base + 0.03                                   # the chain smooths the base's per-window calls
    p_o, _ = oracle.smooth_crf(B, sm.dev.data.crf_state, sm.dev.data.crf_trans)
    assert np.abs(sm.predict_proba(B) - p_o).max() < 1e-9
    assert np.all(np.diag(sm.dev.data.crf_trans) > sm.dev.data.crf_trans.mean())   # staying in an ancestry is what it learned


@pytest.mark.gpu
def test_train_crf_rejects_bad_arguments():
    import ctypes as C
    import gnomix_amd as ga
    from gnomix_amd import _lib
    from gnomix_amd.train import train_crf_arrays
    B = np.full((4, 10, 3), 1 / 3)
    y = np.zeros((4, 10), np.int32)
    y[0, 0] = 3
    with pytest.raises(ga.GnxError, match="label outside"):
        train_crf_arrays(B, y)
    y[0, 0] = 0
    with pytest.raises(ga.GnxError, match="c2 > 0"):
        train_crf_arrays(B, y, c2=0.0)
    ctx = _lib.default_context(0)
    P = _lib.CrfParams(0.1, 1.0, 1e-8, 10, 10)
    st = np.zeros((3, 3)); tr = np.zeros((3, 3))
    rc = ctx.lib.gnx_train_crf(ctx.h, B.ctypes.data, 1, y.ctypes.data, 4, 10, 3, C.byref(P), st.ctypes.data, tr.ctypes.data, None)
    assert rc != 0 and b"c1" in ctx.lib.gnx_last_error(ctx.h)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["crf", "cnn"])
def test_gnomix_train_end_to_end_fast_and_large_modes(kind):
    """Gnomix.train (src/model.py:104-167) with the smoothers of the reference's "fast" (CRF) and "large" (CNN) modes: logistic base on
    train1, smoother on the base's probabilities of train2, base again on everything — all on the device; the smoother beats its
    base on held-out admixed haplotypes, and the trained model survives save / load"""
    import os
    import tempfile
    import gnomix_amd as ga
    from gnomix_amd.train import cnn_init
    A, M, W, S = 3, 40, 30, 7
    C = M * W + 13
    rng = np.random.RandomState(4)
    freq = np.clip(rng.uniform(0.2, 0.8, size=(1, C)) + rng.normal(0, 0.13, size=(A, C)), 0.02, 0.98)

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
    d = ga.GnxModelData(C=C, M=M, A=A, S=S, context=M // 2, smooth_kind=kind)
    d.base_kind, d.lr_coef, d.lr_intercept = "logistic", np.zeros((W, A, M + 2 * (M // 2) + C - M * W)), np.zeros((W, A))
    if kind == "crf":
        d.crf_state, d.crf_trans = np.zeros((A, A)), np.zeros((A, A))
        kw = {}
    else:
        d.cnn_weight, d.cnn_bias = cnn_init(A, S, seed=0)
        kw = dict(max_ep=150, seed=1)
    g = ga.HipGnomix(d)
    g.train((t1, t2, v), **kw)
    assert g.accuracies["smooth_val_acc"] > 80 and g.accuracies["smooth_val_acc"] > g.accuracies["base_val_acc"] + 3, g.accuracies
    assert g.predict(v[0]).shape == v[1].shape
    with tempfile.TemporaryDirectory() as td:
        g2 = ga.HipGnomix.load(g.save(os.path.join(td, "trained.gnx")))
        assert np.array_equal(g2.predict(v[0][:10]), g.predict(v[0][:10]))

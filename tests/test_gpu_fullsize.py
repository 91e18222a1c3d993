"""-m gpu: full-size passes of the round-2 paths (BASELINE geometries), checked through the oracle on a few rows and through
bit-equality between independent routes to the same answer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ga():
    import gnomix_amd
    gnomix_amd.load_library()
    return gnomix_amd


def test_host_pipelines_at_chr22_size(ga, oracle):
    """int8 host path (three-stream pipeline, pageable memory, ragged last batch), 2-bit packed path and the device-resident
    path give bit-identical outputs at config-2 geometry; three rows against the oracle"""
    import torch
    from gnomix_amd import synth, _lib
    d = synth.synthetic_model(seed=0, n_rounds=20, **synth.CHR22)
    dev = ga.DeviceModel(d, ctx=_lib.Context(0))
    T = oracle.Trees(d.tree_off, d.left, d.right, d.feat, d.cond, d.tree_class, d.A, d.base_score)
    for N in (4097, 2050):
        X = synth.synthetic_X(N, d.C, seed=N, miss=0.02)
        p0, l0 = dev.infer_device(torch.from_numpy(X).cuda())
        torch.cuda.synchronize()
        p0, l0 = p0.cpu().numpy(), l0.cpu().numpy()
        p1, l1 = dev.infer(X)
        p2, l2 = dev.infer_packed(dev.pack_x(X))
        assert np.array_equal(p0, p1) and np.array_equal(l0, l1) and np.array_equal(p0, p2) and np.array_equal(l0, l2), N
        idx = [0, N // 2, N - 1]
        pr, lr_ = oracle.smooth_xgb(T, oracle.base_lr(X[idx], d.M, d.context, d.lr_coef, d.lr_intercept), d.S)
        assert np.array_equal(l0[idx], lr_) and np.max(np.abs(p0[idx] - pr)) <= 1e-5


def test_forest_base_at_chr22_size_both_wave_group_layouts(ga, oracle, monkeypatch):
    """XGBBase shape (140 depth-4 trees per window, W = 370): one and two wave groups per LDS tile agree bit for bit, six rows
    (tile edges included) against the oracle"""
    from gnomix_amd import synth, _lib
    cfg = synth.CHR22
    df = synth.synthetic_forest_model(cfg["C"], cfg["M"], 7, n_rounds=20, depth=4, seed=1, S=75, smooth=None)
    X = synth.synthetic_X(3000, cfg["C"], seed=5, miss=0.03)
    Tf = oracle.Trees(df.fb_tree_off, df.fb_left, df.fb_right, df.fb_feat, df.fb_cond, df.fb_tree_class, df.A, df.fb_base_score,
                      default_left=df.fb_default_left)
    rows = [0, 1, 255, 256, 1500, 2999]
    ref = oracle.base_forest(Tf, df.fb_win_tree0, X[rows], cfg["M"], df.context, 7, missing=2)
    outs = []
    for h in ("0", "1"):
        monkeypatch.setenv("GNX_FOREST_H", h)
        b32, _ = ga.DeviceModel(df, ctx=_lib.Context(0)).base_predict(X, want_f32=True, want_f64=False)
        assert np.max(np.abs(b32[rows] - ref)) <= 2.4e-7, h
        outs.append(b32)
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("A", [7, 12, 20])
def test_crf_long_chains(ga, oracle, A):
    """W = 1431 (chr1): the one-lane-per-haplotype scan (A <= 8), the A-lane kernel with the psi pre-pass, and its > 16 label form"""
    from gnomix_amd import _lib
    rng = np.random.RandomState(A)
    W = 1431
    B = rng.dirichlet(np.ones(A) * 0.5, size=(130, W))
    st = rng.standard_normal((A, A)) * 2 + 4 * np.eye(A)
    tr = rng.standard_normal((A, A)) * 0.5 + 3 * np.eye(A)
    dm = ga.GnxModelData(C=W * 10 + 3, M=10, A=A, S=75, context=5, smooth_kind="crf", crf_state=st, crf_trans=tr)
    p, lab = ga.DeviceModel(dm, ctx=_lib.Context(0)).smooth_predict(B)
    sel = [0, 64, 129]
    pr, lr_ = oracle.smooth_crf(B[sel], st, tr)
    assert np.max(np.abs(p[sel] - pr)) <= 1e-10 and np.array_equal(lab[sel], lr_)
    assert np.allclose(p.sum(-1), 1.0, atol=1e-9)

"""-m gpu: seeded random geometries through the whole path (base -> smoother) against the oracle.  Every case draws the
chromosome length, window size, context ratio, number of ancestries, smoother width / kind, tree depth, missing rate and
haplotype count at random; sizes stay small enough for the scalar oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.RandomState(1000 + seed)
    M = int(rng.choice([16, 37, 50, 64, 100, 128, 250]))
    ratio = float(rng.choice([0.0, 0.25, 0.5, 0.5, 1.0, 1.3]))
    ctx = int(M * ratio)
    S = int(rng.choice([5, 11, 21, 31, 75]))
    W = int(rng.randint(2 * S, 2 * S + 120))
    rem = int(rng.randint(1, M))
    A = int(rng.choice([2, 3, 4, 5, 7, 8, 12, 16, 24]))
    N = int(rng.choice([1, 2, 7, 33, 64, 65, 130, 513, 700]))
    smooth = str(rng.choice(["xgb", "xgb", "crf", "cnn"]))
    depth = int(rng.choice([1, 2, 3, 4, 4, 5, 6]))
    rounds = int(rng.randint(1, 9))
    miss = float(rng.choice([0.0, 0.01, 0.1]))
    return dict(C=W * M + rem, M=M, ctx=ctx, S=S, A=A, N=N, smooth=smooth, depth=depth, rounds=rounds, miss=miss, seed=seed)


@pytest.mark.parametrize("seed", range(160))
def test_random_geometry_vs_oracle(oracle, seed):
    import gnomix_amd
    from gnomix_amd import synth
    gnomix_amd.load_library()
    c = _case(seed)
    R = (c["M"] + 2 * c["ctx"] + c["M"] - 1) // c["M"]
    if R * c["A"] > 64:
        pytest.skip("more than 64 class columns per SNP: rejected at model load (covered by test_base_rejects_bad_geometry)")
    d = synth.synthetic_model(C=c["C"], M=c["M"], A=c["A"], S=c["S"], context=c["ctx"], seed=c["seed"], smooth=c["smooth"],
                              n_rounds=c["rounds"], depth=c["depth"])
    X = synth.synthetic_X(c["N"], c["C"], seed=c["seed"] + 7, miss=c["miss"])
    dev = gnomix_amd.DeviceModel(d)
    proba, labels = dev.infer(X)
    b32, b64 = dev.base_predict(X, want_f32=True, want_f64=True)
    n_chk = min(c["N"], 6)                               # oracle rows: first few and (when there are more) the last ones
    rows = np.unique(np.concatenate([np.arange(n_chk), np.arange(max(0, c["N"] - 3), c["N"])]))
    Bo = oracle.base_lr(X[rows], c["M"], c["ctx"], d.lr_coef, d.lr_intercept)
    assert np.max(np.abs(b64[rows] - Bo)) < 1e-12, c
    if c["smooth"] == "xgb":
        T = oracle.Trees(d.tree_off, d.left, d.right, d.feat, d.cond, d.tree_class, d.A, d.base_score)
        po, lo = oracle.smooth_xgb(T, b32[rows], c["S"])     # from the device's own float32 B: isolates the smoother
    elif c["smooth"] == "cnn":
        po, lo = oracle.smooth_cnn(b64[rows], d.cnn_weight, d.cnn_bias)
    else:
        po, lo = oracle.smooth_crf(b64[rows], d.crf_state, d.crf_trans)
    assert np.max(np.abs(proba[rows] - po)) < 1e-5, c
    if c["smooth"] == "cnn":   # float32 sums in a different order: a label may flip only where the top two are within the tolerance
        srt = np.sort(po, -1)
        clear = srt[..., -1] - srt[..., -2] > 2e-5
        assert np.array_equal(labels[rows][clear], lo[clear]), c
    else:
        assert np.array_equal(labels[rows], lo), c
    dev.close()


@pytest.mark.parametrize("seed", range(60))
def test_random_geometry_tree_bases_vs_oracle(oracle, seed):
    """the same draw for the boosted-tree and random-forest bases (base probabilities only)"""
    import gnomix_amd
    from gnomix_amd import synth
    c = _case(500 + seed)
    rng = np.random.RandomState(seed)
    A = min(c["A"], 16)
    trees, depth = int(rng.randint(1, 8)), int(rng.randint(1, 7))
    X = synth.synthetic_X(c["N"], c["C"], seed=seed + 3, miss=max(c["miss"], 0.03))
    rows = np.unique(np.concatenate([np.arange(min(c["N"], 5)), np.arange(max(0, c["N"] - 2), c["N"])]))
    if seed % 2 == 0:
        d = synth.synthetic_forest_model(c["C"], c["M"], A, context=c["ctx"], n_rounds=trees, depth=depth, seed=seed, p_early_leaf=0.2)
        b32, b64 = gnomix_amd.DeviceModel(d).base_predict(X, want_f32=True, want_f64=True)
        T = oracle.Trees(d.fb_tree_off, d.fb_left, d.fb_right, d.fb_feat, d.fb_cond, d.fb_tree_class, d.A, d.fb_base_score,
                         default_left=d.fb_default_left)
        ref = oracle.base_forest(T, d.fb_win_tree0, X[rows], c["M"], c["ctx"], A, missing=2)
        assert np.max(np.abs(b32[rows] - ref)) <= 2.4e-7, c
        assert np.array_equal(b64, b32.astype(np.float64))
    else:
        d = synth.synthetic_rforest_model(c["C"], c["M"], A, context=c["ctx"], n_trees=trees, depth=depth, seed=seed, p_early_leaf=0.2)
        b32, b64 = gnomix_amd.DeviceModel(d).base_predict(X, want_f32=True, want_f64=True)
        rf = {k[3:]: getattr(d, k) for k in ("rf_win_tree0", "rf_tree_off", "rf_left", "rf_right", "rf_feat", "rf_thr", "rf_value")}
        ref = oracle.base_rforest(rf, X[rows], c["M"], c["ctx"], A)
        assert np.array_equal(b64[rows], ref), c
        assert np.array_equal(b32, b64.astype(np.float32))


@pytest.mark.parametrize("seed", range(24))
def test_random_gnofix_vs_oracle(oracle, seed):
    """random smoother geometry, random individuals (some with identical haplotypes), random iteration cap"""
    import gnomix_amd
    from gnomix_amd import synth
    rng = np.random.RandomState(7000 + seed)
    A = int(rng.choice([2, 3, 5, 7, 12]))
    S = int(rng.choice([5, 11, 31, 75]))
    W = int(rng.randint(2 * S, 2 * S + 90))
    M = int(rng.choice([3, 7, 16]))
    C = W * M + int(rng.randint(1, M))
    max_it = int(rng.choice([1, 3, 6, 50]))
    d = gnomix_amd.GnxModelData(C=C, M=M, A=A, S=S, context=0, smooth_kind="xgb")
    trained = seed % 3 == 0   # a smoother that behaves like a trained one (few switches) or a chaotic random one (many)
    trees = synth.synthetic_smoothing_trees(6, A, S, seed=seed, reach=min(8, S - 1 - (S + 1) // 2)) if trained else \
        synth.synthetic_trees(int(rng.randint(2, 7)), A, S * A, seed=seed, thr_lo=0.0, thr_hi=0.6, leaf_scale=1.0)
    for k, v in trees.items():
        setattr(d, k, v)
    dev = gnomix_amd.DeviceModel(d)
    T = oracle.Trees(d.tree_off, d.left, d.right, d.feat, d.cond, d.tree_class, d.A, d.base_score)
    n_ind = int(rng.randint(1, 6))
    X = rng.randint(0, 3, size=(2 * n_ind, C)).astype(np.int8)
    if n_ind > 1:
        X[2:4] = X[2:3]
    B = rng.dirichlet(np.ones(A) * 0.3, size=(2 * n_ind, W))
    Xo, Y, nsw = dev.gnofix(X, B, max_it=max_it)
    rows = lambda r: oracle.xgb_predict_proba(T, r)
    labs = lambda b: oracle.smooth_xgb(T, b, S)[1]
    for i in range(n_ind):
        Xm, Xp, Ym, Yp, _, ns = oracle.gnofix(X[2 * i], X[2 * i + 1], B[2 * i:2 * i + 2], S, rows, labs, max_it=max_it)
        assert np.array_equal(Xo[2 * i], Xm) and np.array_equal(Xo[2 * i + 1], Xp), (seed, i)
        assert np.array_equal(Y[2 * i], Ym) and np.array_equal(Y[2 * i + 1], Yp), (seed, i)
        assert int(nsw[i]) == ns, (seed, i)


@pytest.mark.parametrize("seed", range(48))
def test_random_crf_vs_oracle(oracle, seed):
    """random CRF-smoother geometries: label counts on both sides of the kernels' row widths (8 / 12 / 16, the 17+ kernel), chain lengths
    around the segment length (8) and its multiples, weight ranges that select every forward-scale interval (8, 4, 2, 1 windows),
    float32 and float64 base probabilities"""
    import gnomix_amd
    rng = np.random.RandomState(9000 + seed)
    A = int(rng.choice([2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 20]))
    W = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 31, 64, 65, 100, 257, 1000]))
    N = int(rng.choice([1, 3, 4, 5, 17, 64, 130]))
    scale = float(rng.choice([0.3, 1.0, 3.0, 9.0, 25.0]))
    state = rng.standard_normal((A, A)) * scale
    trans = rng.standard_normal((A, A)) * scale * float(rng.choice([0.2, 1.0]))
    B = rng.dirichlet(np.ones(A) * float(rng.choice([0.2, 1.0, 5.0])), size=(N, W))
    if rng.randint(2):
        B = B.astype(np.float32)
    d = gnomix_amd.GnxModelData(C=W * 10 + 3, M=10, A=A, S=75, context=5, smooth_kind="crf", crf_state=state, crf_trans=trans)
    dev = gnomix_amd.DeviceModel(d)
    p_ref, l_ref = oracle.smooth_crf(B.astype(np.float64), state, trans)
    p, lab = dev.smooth_predict(B)
    assert np.isfinite(p).all() and np.max(np.abs(p - p_ref)) < 1e-10, seed
    top = np.sort(p_ref, -1)
    clear = top[..., -1] - top[..., -2] > 1e-9
    assert np.array_equal(lab[clear], l_ref[clear]), seed


@pytest.mark.parametrize("A,S,W", [(7, 75, 310), (5, 31, 260), (12, 75, 230)])
def test_gnofix_where_switches_come_thick(oracle, A, S, W):
    """a chaotic smoother on unstructured haplotypes: a label change at nearly every window and dozens of accepted switches per sweep,
    i.e. the regime in which k_gnofix only marks the rows of a switch and brings them up to date when the scan reads them (short
    cleaning batches, full batches at the start of a sweep).  Labels, phased SNPs and switch counts must be the reference loop's."""
    import gnomix_amd
    from gnomix_amd import synth
    rng = np.random.RandomState(A * 1000 + W)
    M = 5
    C = W * M + 3
    d = gnomix_amd.GnxModelData(C=C, M=M, A=A, S=S, context=0, smooth_kind="xgb")
    for k, v in synth.synthetic_trees(4, A, S * A, seed=A + S, thr_lo=0.0, thr_hi=0.6, leaf_scale=1.0).items():
        setattr(d, k, v)
    dev = gnomix_amd.DeviceModel(d)
    T = oracle.Trees(d.tree_off, d.left, d.right, d.feat, d.cond, d.tree_class, d.A, d.base_score)
    n_ind = 3
    X = rng.randint(0, 2, size=(2 * n_ind, C)).astype(np.int8)
    B = rng.dirichlet(np.ones(A), size=(2 * n_ind, W))
    Xo, Y, nsw = dev.gnofix(X, B, max_it=8)
    rows = lambda r: oracle.xgb_predict_proba(T, r)
    labs = lambda b: oracle.smooth_xgb(T, b, S)[1]
    total = 0
    for i in range(n_ind):
        Xm, Xp, Ym, Yp, _, ns = oracle.gnofix(X[2 * i], X[2 * i + 1], B[2 * i:2 * i + 2], S, rows, labs, max_it=8)
        assert np.array_equal(Xo[2 * i], Xm) and np.array_equal(Xo[2 * i + 1], Xp), i
        assert np.array_equal(Y[2 * i], Ym) and np.array_equal(Y[2 * i + 1], Yp), i
        assert int(nsw[i]) == ns, i
        total += ns
    assert total >= 20 * n_ind, total   # (else the inputs no longer exercise the thick regime)


@pytest.mark.parametrize("seed", range(20))
def test_random_covrsk_vs_oracle(oracle, seed):
    """random window widths (canonical-length fast path and generic path), class counts and support-vector sets"""
    import gnomix_amd
    from gnomix_amd import synth
    rng = np.random.RandomState(9000 + seed)
    M = int(rng.choice([20, 33, 64, 100, 175, 260]))
    ctx = int(M * float(rng.choice([0.0, 0.5, 1.0])))
    A = int(rng.choice([2, 3, 4, 7]))
    W = int(rng.randint(3, 9))
    C = W * M + int(rng.randint(1, M))
    N = int(rng.choice([1, 5, 64, 70]))
    d = synth.synthetic_svc_model(C, M, A, context=ctx, n_fit_per_class=int(rng.randint(3, 12)), seed=seed)
    X = synth.synthetic_X(N, C, seed=seed, miss=0.03)
    for n in range(0, N, 2):   # related haplotypes: long match runs
        w = rng.randint(d.W)
        src = d.svc[w]["xfit"][rng.randint(d.svc[w]["xfit"].shape[0])]
        start = w * M - ctx
        seg = src if start >= 0 else src[-start:]
        lo = max(0, start)
        ln = min(len(seg), C - lo)
        X[n, lo:lo + ln] = seg[:ln]
    _, b64 = gnomix_amd.DeviceModel(d).base_predict(X)
    ow = [dict(Xfit=w["xfit"], Ms=list(w["ms"]), support=w["support"], dual=w["dual_coef"], intercept=w["intercept"],
               probA=w["prob_a"], probB=w["prob_b"], n_support=w["n_support"]) for w in d.svc]
    ref = oracle.base_covrsk(X, M, ctx, ow)
    assert np.max(np.abs(b64 - ref)) < 1e-12, seed
    assert np.array_equal(np.argmax(b64, -1), np.argmax(ref, -1))

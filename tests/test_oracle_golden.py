"""The oracle (oracle/) against the golden vectors produced by RUNNING THE REFERENCE
(tests/golden/make_golden.py).  CPU only.  These tests pin the oracle; the -m gpu tests then
compare the HIP path with the oracle."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, trees_from_npz


def test_G1_logistic_base(oracle):
    g = load_golden("G1_lr.npz")
    B = oracle.base_lr(g["X"], int(g["M"]), int(g["ctx"]), g["coef"], g["intercept"])
    assert B.shape == g["B"].shape and B.dtype == np.float64
    # sklearn uses BLAS (unknown summation order): agreement is at f64 round-off, far below 1e-5
    assert np.max(np.abs(B - g["B"])) < 1e-13
    # what the smoother consumes is float32(B) (Smooth/utils.py:20): count bit differences
    diff = np.count_nonzero(B.astype(np.float32) != g["B"].astype(np.float32))
    assert diff <= 1, diff
    assert np.array_equal(np.argmax(B, -1), np.argmax(g["B"], -1))
    assert (g["X"] == 2).any()  # the fixture exercises the missing code


def test_G15_binary_logistic_base(oracle):
    """A = 2: sklearn stores one row and returns [1 - expit(z), expit(z)]; the converter's (-coef, +coef) rows through the
    OvR form must reproduce the REFERENCE's Base.predict_proba (copying the row into both classes gives 0.5/0.5)"""
    from gnomix_amd.convert import lr_rows_from_sklearn
    g = load_golden("G15_lr_binary.npz")
    assert int(g["A"]) == 2
    W = int(g["C"]) // int(g["M"])
    for i in range(W):  # the stored two-row arrays ARE what the converter makes of the raw sklearn arrays
        c2, b2 = lr_rows_from_sklearn(g["raw_coef"][i], g["raw_intercept"][i], 2)
        assert np.array_equal(c2, g["coef"][i]) and np.array_equal(b2, g["intercept"][i])
    B = oracle.base_lr(g["X"], int(g["M"]), int(g["ctx"]), g["coef"], g["intercept"])
    assert np.max(np.abs(B - g["B"])) < 1e-13
    assert np.array_equal(np.argmax(B, -1), np.argmax(g["B"], -1))
    assert np.abs(g["B"][..., 0] - 0.5).max() > 0.4   # the fixture is far from the degenerate 0.5/0.5 answer
    with pytest.raises(ValueError):
        lr_rows_from_sklearn(np.zeros((2, 5)), np.zeros(2), 3)


def test_G1_rejects_C_multiple_of_M(oracle):
    X = np.zeros((2, 40), dtype=np.int8)
    with pytest.raises(ValueError):
        oracle.base_lr(X, 10, 5, np.zeros((4, 2, 30)), np.zeros((4, 2)))


def test_G2_covsample_anchors(oracle):
    g = load_golden("G2_covrsk.npz")
    for m in (8, 20, 349, 499, 2000, 2500):
        assert oracle.cov_sample(m) == list(g["Ms_%d" % m])
    assert oracle.cov_sample(349) == [1, 4, 8, 39, 42, 117]
    assert oracle.cov_sample(2000) == [1, 4, 8, 39, 42, 117, 376, 866]


def test_G2_kernel_known_answers(oracle):
    g = load_golden("G2_covrsk.npz")
    a, b = g["a"][None], g["b"][None]
    assert oracle.covrsk(a, b)[0, 0] == g["k_ab"][0, 0] == 9
    assert oracle.string_kernel(a, b)[0, 0] == g["k_ab_plain"][0, 0] == 16
    z = np.zeros((1, 8), dtype=np.int8)
    assert oracle.covrsk(z, z)[0, 0] == g["k_eq"][0, 0] == 14
    assert oracle.string_kernel(z, z)[0, 0] == g["k_eq_plain"][0, 0] == 36


def _g2_windows(g):
    W = int(g["C"]) // int(g["M"])
    ws = []
    for i in range(W):
        xf = g["w%d_Xfit" % i]
        ws.append(dict(Xfit=xf, Ms=None, support=g["w%d_support" % i], dual=g["w%d_dual" % i],
                       intercept=g["w%d_intercept" % i], probA=g["w%d_probA" % i], probB=g["w%d_probB" % i],
                       n_support=g["w%d_nsv" % i]))
    return ws


def test_G2_covrsk_base(oracle):
    g = load_golden("G2_covrsk.npz")
    M, ctx = int(g["M"]), int(g["ctx"])
    ws = _g2_windows(g)
    for w in ws:
        w["Ms"] = oracle.cov_sample(w["Xfit"].shape[1])
    wins = dict(oracle.base_windows(g["X"], M, ctx))
    assert np.array_equal(oracle.covrsk(wins[3], ws[3]["Xfit"], ws[3]["Ms"]), g["K_w3"])
    B = oracle.base_covrsk(g["X"], M, ctx, ws)
    assert B.shape == g["B"].shape
    assert np.max(np.abs(B - g["B"])) < 1e-12
    assert (g["X"] == 2).any()


def test_G3_slide_window(oracle):
    g = load_golden("G3_slide.npz")
    assert np.array_equal(oracle.slide_window(g["tiny_B"], 3), g["tiny_S3"])
    assert g["tiny_S3"].tolist() == [[1, 0, 0], [0, 0, 1], [0, 1, 2], [1, 2, 3], [2, 3, 4], [3, 4, 5]]
    for name in "abcd":
        B, S = g[name + "_B"], int(g[name + "_S"])
        o = oracle.slide_window(B, S)
        assert o.dtype == np.float32
        if name + "_sha" in g:
            assert hashlib.sha256(o.tobytes()).digest() == g[name + "_sha"].tobytes()
            assert np.array_equal(o[g[name + "_rows"]], g[name + "_out"])
        else:
            assert np.array_equal(o, g[name + "_out"])


def test_G4_smoother_glue(oracle):
    g = load_golden("G4_smooth.npz")
    T = trees_from_npz(oracle, g, "t_")
    proba, labels = oracle.smooth_xgb(T, g["B"], int(g["S"]))
    assert np.array_equal(proba, g["proba"])  # same walker on both sides: pins slide/cast/reshape/argmax
    assert np.array_equal(labels, g["labels"])
    assert np.allclose(proba.sum(-1), 1, atol=1e-6)


@pytest.mark.parametrize("name", ["none", "one", "two", "edges", "many", "rand"])
def test_G5_gnofix(oracle, name):
    g = load_golden("G5_gnofix.npz")
    W, A, S = int(g["W"]), int(g["A"]), int(g["S"])
    T = trees_from_npz(oracle, g, "r_" if name == "rand" else "t_")
    rows = lambda r: oracle.xgb_predict_proba(T, r)
    labs = lambda B: oracle.smooth_xgb(T, B, S)[1]
    Xm, Xp, Ym, Yp, trk, nsw = oracle.gnofix(g[name + "_Xm"], g[name + "_Xp"], g[name + "_B"], S, rows, labs,
                                             max_it=4 if name == "rand" else 50)
    assert np.array_equal(Xm, g[name + "_oXm"]) and np.array_equal(Xp, g[name + "_oXp"])
    assert np.array_equal(Ym, g[name + "_oYm"]) and np.array_equal(Yp, g[name + "_oYp"])
    assert np.array_equal(trk, g[name + "_trk"])
    assert nsw == int(g[name + "_nhist"]) - 2  # history = initial + one per accepted switch + final


def test_G5_phase_wrapper(oracle):
    g = load_golden("G5_gnofix.npz")
    W, A, S = int(g["W"]), int(g["A"]), int(g["S"])
    T = trees_from_npz(oracle, g, "t_")
    rows = lambda r: oracle.xgb_predict_proba(T, r)
    labs = lambda B: oracle.smooth_xgb(T, B, S)[1]
    X, B = g["phase_X"], g["phase_B"]
    for i in range(X.shape[0] // 2):
        Xm, Xp, Ym, Yp, _, _ = oracle.gnofix(X[2 * i], X[2 * i + 1], B[2 * i:2 * i + 2], S, rows, labs)
        assert np.array_equal(np.stack([Xm, Xp]), g["phase_oX"][2 * i:2 * i + 2])
        assert np.array_equal(np.stack([Ym, Yp]), g["phase_oY"][2 * i:2 * i + 2])


def test_forest_base_hand_computed(oracle):
    """XGBBase restatement (PARITY UNPINNED: xgboost absent) checked against values worked out by hand:
    C=5, M=2, ctx=1 -> W=2, padded X = [x0 | x0..x4 | x4]; window 0 = padded[0:4], window 1 = padded[2:7]."""
    O = oracle
    X = np.array([[1, 0, 2, 1, 0]], np.int8)            # padded: 1 1 0 2 1 0 0
    # window 0: one stump on SNP 2 (x=0): left 0.3 / right -0.1;  window 1: stump on SNP 1 (x=2, missing) default left
    tr = O.Trees(tree_off=[0, 3, 6], left=[1, -1, -1, 1, -1, -1], right=[2, -1, -1, 2, -1, -1], feat=[2, 0, 0, 1, 0, 0],
                 cond=[0.5, 0.3, -0.1, 0.5, 0.7, -0.4], tree_class=[0, 0], n_class=2, base_score=0.5,
                 default_left=[0, 0, 0, 1, 0, 0])
    B = O.base_forest(tr, [0, 1, 2], X, 2, 1, 2, missing=2)
    p0 = np.float32(1) / (np.float32(1) + np.exp(np.float32(-0.3)))
    p1 = np.float32(1) / (np.float32(1) + np.exp(np.float32(-0.7)))
    assert np.allclose(B[0, 0], [1 - p0, p0], atol=1e-7)
    assert np.allclose(B[0, 1], [1 - p1, p1], atol=1e-7)
    tr.default_left = np.array([0, 0, 0, 0, 0, 0], np.uint8)   # missing now goes right
    B = O.base_forest(tr, [0, 1, 2], X, 2, 1, 2, missing=2)
    p1 = np.float32(1) / (np.float32(1) + np.exp(np.float32(0.4)))
    assert np.allclose(B[0, 1], [1 - p1, p1], atol=1e-7)
    # A = 3: softmax over per-class sums, base_score cancels
    tr3 = O.Trees(tree_off=[0, 3, 4, 5, 8, 9, 10], left=[1, -1, -1, -1, -1, 1, -1, -1, -1, -1],
                  right=[2, -1, -1, -1, -1, 2, -1, -1, -1, -1], feat=[0, 0, 0, 0, 0, 4, 0, 0, 0, 0],
                  cond=[0.5, 0.2, 0.9, 0.1, -0.3, 1.5, 0.6, -0.6, 0.0, 0.25], tree_class=[0, 1, 2, 0, 1, 2], n_class=3)
    B = O.base_forest(tr3, [0, 3, 6], X, 2, 1, 3)
    m0 = np.array([0.9, 0.1, -0.3])       # window 0: SNP0 = 1 -> right leaf 0.9
    m1 = np.array([0.6, 0.0, 0.25])       # window 1: SNP4 of padded[2:7] = 0 < 1.5 -> left leaf 0.6
    for w, mm in enumerate((m0, m1)):
        e = np.exp(mm - mm.max())
        assert np.allclose(B[0, w], e / e.sum(), atol=1e-6)


def test_rforest_base_golden_G9(oracle):
    """RFBase restatement against the REFERENCE's own RFBase.predict_proba (sklearn forests trained by Base.train)"""
    g = load_golden("G9_rf.npz")
    rf = {k[3:]: g[k] for k in g.files if k.startswith("rf_")}
    B = oracle.base_rforest(rf, g["X"], int(g["M"]), int(g["ctx"]), int(g["A"]))
    assert B.shape == g["B"].shape
    assert np.array_equal(B, g["B"])           # bit-exact: same float64 adds in estimator order, same division
    assert np.allclose(B.sum(-1), 1.0, atol=1e-12)


def test_numpy_pairwise_sum_order(oracle):
    """np.sum of a float64 vector is a pairwise sum: the restatement must reproduce numpy bit for bit at every length"""
    import ctypes as C
    L = oracle.lib()
    L.gnxo_np_sum.restype = C.c_double
    rng = np.random.RandomState(1)
    for n in list(range(1, 300)) + [511, 512, 513, 1000, 1023, 1024, 1025, 2047, 2500, 4096, 5001]:
        a = (rng.random_sample(n) * rng.choice([1, 1e3, 1e-3])) ** 1.2
        assert L.gnxo_np_sum(a.ctypes.data_as(C.c_void_p), C.c_int64(n)) == np.sum(a), n


def _poly_windows(g):
    wins = []
    for i in range(int(g["n_win"])):
        pre = "svc%d_" % i
        w = {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}
        wins.append(w)
    return wins


def test_poly_string_kernel_golden_G10(oracle):
    """polynomial string kernel + SVC probabilities against the reference's PolynomialStringKernelBase output"""
    g = load_golden("G10_poly.npz")
    wins = _poly_windows(g)
    C_, M, ctx = int(g["C"]), int(g["M"]), int(g["ctx"])
    X = g["X"]
    Xp = np.concatenate([X[:, :ctx][:, ::-1], X, X[:, -ctx:][:, ::-1]], axis=1)
    K0 = oracle.poly_kernel(Xp[:, :M + 2 * ctx], wins[0]["xfit"], wins[0]["run_value"], float(wins[0]["poly_p"]))
    assert np.array_equal(K0, g["K0"])                    # the reference's own poly_kernel matrix, exactly
    ow = [dict(Xfit=w["xfit"], support=w["support"], dual=w["dual_coef"], intercept=w["intercept"], probA=w["prob_a"],
               probB=w["prob_b"], n_support=w["n_support"], run_value=w["run_value"], poly_p=float(w["poly_p"])) for w in wins]
    B = oracle.base_covrsk(X, M, ctx, ow)
    assert np.max(np.abs(B - g["B"])) < 1e-12
    assert np.array_equal(np.argmax(B, -1), np.argmax(g["B"], -1))


def test_cnn_smoother_golden_G11(oracle):
    """CNN smoother restatement against the reference's CNN.predict_proba / predict (torch conv1d + softmax)"""
    g = load_golden("G11_cnn.npz")
    proba, labels = oracle.smooth_cnn(g["B"], g["weight"], g["bias"])
    assert proba.dtype == np.float32 and proba.shape == g["proba"].shape
    assert np.max(np.abs(proba - g["proba"])) < 1e-5      # the backend's tap order is not defined: a few float32 ulps
    assert np.array_equal(labels, g["labels"])


# ---- CRF smoother (a7): an anchor that does not come from any forward-backward recursion ------------------------------------------
def _crf_brute_force(B, state, trans):
    """marginals and log Z of the linear chain score(y | x) = sum_t sum_a state[a][y_t] x[t][a] + sum_{t>=1} trans[y_{t-1}][y_t]
    (src/Smooth/crf.py:9-15: all possible states and transitions) by ENUMERATING all A^W labelings"""
    import itertools
    W, A = B.shape
    s = B @ state                                       # (W, A) state scores
    scores, labelings = [], []
    for y in itertools.product(range(A), repeat=W):
        sc = sum(s[t, y[t]] for t in range(W)) + sum(trans[y[t - 1], y[t]] for t in range(1, W))
        scores.append(sc)
        labelings.append(y)
    scores = np.array(scores)
    mx = scores.max()
    p = np.exp(scores - mx)
    logz = mx + np.log(p.sum())
    p /= p.sum()
    marg = np.zeros((W, A))
    for pr, y in zip(p, labelings):
        for t in range(W):
            marg[t, y[t]] += pr
    return marg, logz, labelings, scores


@pytest.mark.parametrize("A,W,seed", [(2, 1, 0), (2, 6, 1), (3, 2, 2), (3, 5, 3), (3, 6, 4)])
def test_crf_restatement_vs_brute_force_enumeration(oracle, A, W, seed):
    """smooth_crf's marginals / labels and crf_objective's value and gradient against the enumeration of all
    A^W labelings — nothing of the oracle's (or the kernels') scaled forward-backward is shared with this check"""
    rng = np.random.RandomState(seed)
    B = rng.dirichlet(np.ones(A), size=(3, W))
    state = rng.standard_normal((A, A)) * 1.5
    trans = rng.standard_normal((A, A)) * 1.5
    proba, labels = oracle.smooth_crf(B, state, trans)
    logz = []
    for n in range(B.shape[0]):
        marg, lz, _, _ = _crf_brute_force(B[n], state, trans)
        logz.append(lz)
        assert np.max(np.abs(proba[n] - marg)) <= 1e-13
        assert np.array_equal(labels[n], np.argmax(marg, axis=1)) or np.sort(marg, axis=1)[:, -1].min() - np.sort(marg, axis=1)[:, -2].max() < 1e-12
    # the training objective: f = -sum_n log p(y_n | x_n) + c2 |w|^2 by enumeration, gradient by central differences of THAT
    y = rng.randint(0, A, size=(B.shape[0], W))

    def f_enum(st, tr):
        f = 0.0
        for n in range(B.shape[0]):
            s = B[n] @ st
            sc = sum(s[t, y[n, t]] for t in range(W)) + sum(tr[y[n, t - 1], y[n, t]] for t in range(1, W))
            f -= sc - _crf_brute_force(B[n], st, tr)[1]
        return f + 1.0 * (np.sum(st * st) + np.sum(tr * tr))
    f, gs, gt = oracle.crf_objective(B, y, state, trans, c2=1.0)
    assert abs(f - f_enum(state, trans)) <= 1e-11 * max(1.0, abs(f))
    h = 1e-6
    for (i, j) in [(0, 0), (A - 1, 0), (0, A - 1)]:
        d = np.zeros((A, A)); d[i, j] = h
        assert abs((f_enum(state + d, trans) - f_enum(state - d, trans)) / (2 * h) - gs[i, j]) <= 1e-6
        assert abs((f_enum(state, trans + d) - f_enum(state, trans - d)) / (2 * h) - gt[i, j]) <= 1e-6

"""Third-party pins for SURVEY §8 rows a6 (xgboost smoother), a7 (CRFsuite smoother), XGBBase, LGBMBase and CBBase.

The arithmetic of those rows lives in xgboost==1.1.1 / sklearn-crfsuite==0.3.6 (reference requirements.txt:9,11),
which are absent from the build image, so the oracle's restatement of them is "parity unpinned" (oracle/gnx_oracle.c
header).  The fixtures these tests read are produced by ONE command on any host that has the packages:

    pip install xgboost==1.1.1 sklearn-crfsuite==0.3.6
    python tests/golden/make_golden.py G12 G13 G14          # writes tests/golden/G1{2,3,4}_*.npz
    pip install lightgbm catboost && python tests/golden/make_golden.py G18 G19      # LGBMBase / CBBase (src/Base/models.py:38-52, 68-81)
    python -m pytest tests/test_pins_thirdparty.py           # CPU: oracle + booster-bytes parser vs the real packages
    python -m pytest tests/test_pins_thirdparty.py -m gpu    # GPU box: the HIP kernels vs the same fixtures

Until then every fixture-reading test SKIPS (it does not pass).  The one test that always runs
(`test_generator_plumbing_with_standin_xgboost`) executes the G12 / G14 generators and this file's own checks against a
STAND-IN `xgboost` module assembled from the oracle's tree walker: it proves the generator and the test code execute end
to end — it pins NOTHING about xgboost and says so.
"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

HOW = "generate it on a host with the package: python tests/golden/make_golden.py G12 G13 G14 G18 G19 (see this file's docstring)"


def _load(name, directory=GOLDEN):
    path = os.path.join(directory, name)
    if not os.path.exists(path):
        pytest.skip("%s is not generated yet (third-party package absent from the build image): %s" % (name, HOW))
    return np.load(path, allow_pickle=False)


# ---------------------------------------------------------------------------------------------------------------------
# the checks proper: shared by the real fixtures and by the plumbing self-check
# ---------------------------------------------------------------------------------------------------------------------
def _trees_of(O, raw, dumps, n_class):
    """booster bytes -> oracle Trees; the JSON dump of the same booster must describe the same trees"""
    from gnomix_amd import convert, refpickle
    t = refpickle.parse_xgb_raw(bytes(np.asarray(raw, dtype=np.uint8).tobytes()))
    j = convert.trees_from_xgb_json([json.dumps(x) for x in json.loads(str(dumps))], max(int(t["n_class"]), 1), t["base_score"])
    for k in ("tree_off", "left", "right", "feat", "tree_class", "default_left"):
        assert np.array_equal(t[k], j[k]), "booster bytes and JSON dump disagree on " + k
    # the text dump prints float32 values with enough digits to round-trip
    assert np.array_equal(t["cond"], j["cond"])
    n_out = max(int(t["n_class"]), 1)
    assert n_out == (n_class if n_class > 2 or n_out > 1 else 1)
    return t, O.Trees(t["tree_off"], t["left"], t["right"], t["feat"], t["cond"], t["tree_class"], n_out, t["base_score"],
                      default_left=t["default_left"])


def check_xgb_smoother(O, g, hip=None):
    """G12: oracle (and HIP when given) vs XGBClassifier.predict_proba / Smoother.predict on slide_window rows"""
    A, S, W = int(g["A"]), int(g["S"]), int(g["W"])
    t, T = _trees_of(O, g["raw"], g["dumps"], A)
    assert T.n_class == A            # multi:softprob num_class=A also for A == 2 (src/Smooth/models.py:14-20)
    ref_p, ref_l = g["proba"], g["labels"]
    assert ref_p.dtype == np.float32
    p, l = O.smooth_xgb(T, g["B"], S)
    assert np.array_equal(l, ref_l)
    assert np.max(np.abs(p - ref_p)) <= 1e-6      # same float32 sums; only exp() rounding may differ
    if hip is not None:
        d = hip.GnxModelData(C=W * 10 + 3, M=10, A=A, S=S, smooth_kind="xgb", tree_off=t["tree_off"], left=t["left"],
                             right=t["right"], feat=t["feat"], cond=t["cond"], tree_class=t["tree_class"], base_score=t["base_score"])
        ph, lh = hip.DeviceModel(d).smooth_predict(np.asarray(g["B"]))
        assert np.array_equal(lh, ref_l)
        assert np.max(np.abs(ph - ref_p)) <= 1e-6


def check_crf_smoother(O, g, hip=None):
    """G13: oracle (and HIP) vs sklearn_crfsuite.CRF.predict_marginals"""
    ref = g["proba"]
    p = O.smooth_crf(g["B"], g["state"], g["trans"])
    p = p[0] if isinstance(p, tuple) else p
    assert np.max(np.abs(p - ref)) <= 1e-9
    assert np.array_equal(np.argmax(p, -1), np.argmax(ref, -1))
    if hip is not None:
        A, W = int(g["A"]), int(g["W"])
        d = hip.GnxModelData(C=W * 10 + 3, M=10, A=A, S=5, smooth_kind="crf", crf_state=g["state"], crf_trans=g["trans"])
        ph, lh = hip.DeviceModel(d).smooth_predict(np.asarray(g["B"]))
        assert np.max(np.abs(ph - ref)) <= 1e-9
        assert np.array_equal(lh, np.argmax(ref, -1))


def check_crf_trainer(O, g, hip=None):
    """G13's training set: CRFsuite's own fit (libLBFGS stopped at its epsilon / delta / max_iterations) against the oracle's
    objective and the minimiser the oracle (and the device) find: never worse than CRFsuite's, and as close as CRFsuite's
    stopping error allows"""
    if "Bt" not in g.files:
        pytest.skip("G13 was generated before the training set was recorded: regenerate it")
    Bt, yt = g["Bt"], g["yt"]
    f_ref = O.crf_objective(Bt, yt, g["state"], g["trans"])[0]
    st, tr, f = O.crf_fit(Bt, yt)
    assert f <= f_ref * (1 + 1e-12)
    assert (f_ref - f) / f_ref < 1e-3
    assert np.max(np.abs(st - g["state"])) < 5e-2 and np.max(np.abs(tr - g["trans"])) < 5e-2
    if hip is not None:
        from gnomix_amd.train import train_crf_arrays
        sd, td, info = train_crf_arrays(Bt, yt)
        assert info["converged"] and info["objective"] <= f_ref * (1 + 1e-12)
        assert np.max(np.abs(sd - st)) < 1e-5 and np.max(np.abs(td - tr)) < 1e-5


def check_xgb_base(O, g, hip=None):
    """G14: per-window XGBClassifier(missing=2) — multi:softprob (A = 3) and binary:logistic (A = 2)"""
    from gnomix_amd import convert, refpickle
    for tag in ("m", "b"):
        C, M, A, ctx = (int(g[tag + "_" + k]) for k in ("C", "M", "A", "ctx"))
        off = g[tag + "_raw_off"]
        parts = []
        for w in range(len(off) - 1):
            t, _ = _trees_of(O, g[tag + "_raw"][off[w]:off[w + 1]], g[tag + "_dumps"][w], A)
            parts.append(t)
        fb = convert.forest_from_parts(parts, missing=2)
        T = O.Trees(fb["fb_tree_off"], fb["fb_left"], fb["fb_right"], fb["fb_feat"], fb["fb_cond"], fb["fb_tree_class"],
                    A if A > 2 else 1, fb["fb_base_score"], default_left=fb["fb_default_left"])
        ref = g[tag + "_B"]
        B = O.base_forest(T, fb["fb_win_tree0"], g[tag + "_X"], M, ctx, A, missing=2)
        assert np.max(np.abs(B - ref)) <= 1e-6
        assert np.array_equal(np.argmax(B, -1), np.argmax(ref, -1))
        assert (g[tag + "_X"] == 2).any()          # the missing code takes the default direction somewhere
        if hip is not None:
            d = hip.GnxModelData(C=C, M=M, A=A, S=5, context=ctx, base_kind="forest", **fb)
            b32, _ = hip.DeviceModel(d).base_predict(g[tag + "_X"], want_f32=True, want_f64=False)
            assert np.max(np.abs(b32 - ref)) <= 1e-6
            assert np.array_equal(np.argmax(b32, -1), np.argmax(ref, -1))


def _check_forest_base(O, g, convert_fn, hip=None, tol=2e-6):
    """G18 / G19: per-window tree models of another library converted to the forest base's arrays -> oracle (and HIP) vs the library's
    own predict_proba.  The library sums in float64, the forest kernels in float32 (xgboost's arithmetic): 2e-6 on probabilities."""
    for tag in ("m", "b"):
        C, M, A, ctx = (int(g[tag + "_" + k]) for k in ("C", "M", "A", "ctx"))
        fb = convert_fn([str(x) for x in g[tag + "_models"]], A)
        T = O.Trees(fb["fb_tree_off"], fb["fb_left"], fb["fb_right"], fb["fb_feat"], fb["fb_cond"], fb["fb_tree_class"], max(A, 2),
                    default_left=fb["fb_default_left"])
        ref = g[tag + "_B"]
        X = g[tag + "_X"]
        B = O.base_forest(T, fb["fb_win_tree0"], X, M, ctx, A, missing=2)
        assert B.shape == ref.shape and np.max(np.abs(B - ref)) <= tol
        sure = np.abs(np.sort(ref, -1)[..., -1] - np.sort(ref, -1)[..., -2]) > 10 * tol      # (an exact tie has no defined arg-max)
        assert np.array_equal(np.argmax(B, -1)[sure], np.argmax(ref, -1)[sure])
        assert (X == 2).any()                      # the missing code is compared as the number 2 somewhere
        if hip is not None:
            d = hip.GnxModelData(C=C, M=M, A=A, S=5, context=ctx, base_kind="forest", **fb)
            b32, _ = hip.DeviceModel(d).base_predict(X, want_f32=True, want_f64=False)
            assert np.max(np.abs(b32 - ref)) <= tol
            assert np.array_equal(np.argmax(b32, -1)[sure], np.argmax(ref, -1)[sure])


def check_lgbm_base(O, g, hip=None):
    from gnomix_amd import convert
    _check_forest_base(O, g, lambda models, A: convert.forest_from_lgbm_text(models, A, missing=2), hip)


def check_catboost_base(O, g, hip=None):
    from gnomix_amd import convert
    _check_forest_base(O, g, lambda models, A: convert.forest_from_catboost_json(models, A, missing=2), hip)


# ---------------------------------------------------------------------------------------------------------------------
# the real pins (skip until generated)
# ---------------------------------------------------------------------------------------------------------------------
def test_pin_G18_lightgbm_base_vs_oracle(oracle):
    check_lgbm_base(oracle, _load("G18_lgbm_base.npz"))


def test_pin_G19_catboost_base_vs_oracle(oracle):
    check_catboost_base(oracle, _load("G19_catboost_base.npz"))


@pytest.mark.gpu
def test_pin_G18_lightgbm_base_vs_hip(oracle):
    import gnomix_amd
    check_lgbm_base(oracle, _load("G18_lgbm_base.npz"), hip=gnomix_amd)


@pytest.mark.gpu
def test_pin_G19_catboost_base_vs_hip(oracle):
    import gnomix_amd
    check_catboost_base(oracle, _load("G19_catboost_base.npz"), hip=gnomix_amd)


def test_pin_G12_xgboost_smoother_vs_oracle(oracle):
    check_xgb_smoother(oracle, _load("G12_xgb_smoother.npz"))


def test_pin_G13_crfsuite_smoother_vs_oracle(oracle):
    check_crf_smoother(oracle, _load("G13_crf_smoother.npz"))


def test_pin_G13_crfsuite_fit_vs_oracle(oracle):
    check_crf_trainer(oracle, _load("G13_crf_smoother.npz"))


def test_pin_G14_xgboost_base_vs_oracle(oracle):
    check_xgb_base(oracle, _load("G14_xgb_base.npz"))


@pytest.mark.gpu
def test_pin_G12_xgboost_smoother_vs_hip(oracle):
    import gnomix_amd
    check_xgb_smoother(oracle, _load("G12_xgb_smoother.npz"), hip=gnomix_amd)


@pytest.mark.gpu
def test_pin_G13_crfsuite_fit_vs_hip(oracle):
    import gnomix_amd
    check_crf_trainer(oracle, _load("G13_crf_smoother.npz"), hip=gnomix_amd)


@pytest.mark.gpu
def test_pin_G13_crfsuite_smoother_vs_hip(oracle):
    import gnomix_amd
    check_crf_smoother(oracle, _load("G13_crf_smoother.npz"), hip=gnomix_amd)


@pytest.mark.gpu
def test_pin_G14_xgboost_base_vs_hip(oracle):
    import gnomix_amd
    check_xgb_base(oracle, _load("G14_xgb_base.npz"), hip=gnomix_amd)


# ---------------------------------------------------------------------------------------------------------------------
# plumbing self-check: a STAND-IN xgboost (oracle trees + this repo's legacy-binary writer).  Pins nothing.
# ---------------------------------------------------------------------------------------------------------------------
_STANDIN = textwrap.dedent('''
    """STAND-IN for xgboost, assembled from the oracle — exists only to execute the G12/G14 generator code paths."""
    import json, sys
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
    from oracle import gnx_oracle as O
    from test_refpickle import _legacy_bytes
    __version__ = "0.0-standin"

    class Booster:
        def __init__(self, T, F, objective):
            self.T, self.F, self.objective = T, F, objective
        def save_raw(self, *a, **k):
            d = dict(tree_off=self.T.tree_off, left=self.T.left, right=self.T.right, feat=self.T.feat, cond=self.T.cond,
                     tree_class=self.T.tree_class)
            b = _legacy_bytes(d, self.F, default_left=self.T.default_left)
            if self.objective != "multi:softprob":   # patch num_class = 0 and the objective name for binary:logistic
                b = b.replace(b"multi:softprob", b"binary:logistic")
                import struct
                i = b.index(b"binf") + 4
                b = b[:i + 8] + struct.pack("<i", 0) + b[i + 12:]
                b = b.replace(struct.pack("<Q", 14) + b"binary:logistic", struct.pack("<Q", 15) + b"binary:logistic")
            return bytearray(b)
        def get_dump(self, dump_format="json"):
            out = []
            T = self.T
            for t in range(T.n_trees):
                o = T.tree_off[t]
                def node(k):
                    if T.left[o + k] == -1:
                        return {"nodeid": int(k), "leaf": float(T.cond[o + k])}
                    l, r = int(T.left[o + k]), int(T.right[o + k])
                    return {"nodeid": int(k), "depth": 0, "split": "f%%d" %% T.feat[o + k], "split_condition": float(T.cond[o + k]),
                            "yes": l, "no": r, "missing": l if T.default_left[o + k] else r, "children": [node(l), node(r)]}
                out.append(json.dumps(node(0)))
            return out

    class XGBClassifier:
        def __init__(self, **kw):
            self.kw = dict(kw)
        def set_params(self, **kw):
            self.kw.update(kw); return self
        def fit(self, X, y):
            A = int(np.max(y)) + 1
            self.n_class = A
            multi = self.kw.get("objective") == "multi:softprob" or A > 2
            T = O.random_trees(int(self.kw.get("n_estimators", 10)), A if multi else 1, X.shape[1], depth=int(self.kw.get("max_depth", 4)),
                               seed=int(self.kw.get("random_state", 0)), thr_lo=0.0, thr_hi=1.5 if X.dtype == np.int8 else 1.0)
            T.default_left = (np.random.RandomState(1).random_sample(len(T.left)) < 0.5).astype(np.uint8)
            T.default_left[T.left == -1] = 0
            self.T, self.multi, self.F = T, multi, X.shape[1]
            return self
        def get_booster(self):
            return Booster(self.T, self.F, "multi:softprob" if self.multi else "binary:logistic")
        def predict_proba(self, X):
            X = np.asarray(X)
            if X.dtype == np.int8:   # XGBBase: SNP windows with missing = 2 -> the oracle's forest walker on one window
                W1 = np.array([0, self.T.n_trees], np.int32)
                C = X.shape[1]
                return O.base_forest(self.T, W1, np.concatenate([X, X[:, :1]], axis=1), C, 0, self.n_class, missing=2)[:, 0, :]
            return O.xgb_predict_proba(self.T, X)
''')


def test_generator_plumbing_with_standin_xgboost(oracle, tmp_path):
    """NOT a pin.  Runs make_golden.py G12 / G14 in a fresh interpreter whose `xgboost` is the stand-in above (and without the
    reference checkout, so the generators take their constructor-arguments branch), then runs this file's checks on what they
    wrote: generator + parser + oracle comparison execute end to end and agree with each other."""
    pkg = tmp_path / "xgboost"
    pkg.mkdir()
    (pkg / "__init__.py").write_text(_STANDIN % {"root": ROOT, "tests": os.path.join(ROOT, "tests")})
    outdir = tmp_path / "out"
    outdir.mkdir()
    code = textwrap.dedent('''
        import sys, os
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        os.environ["GNOMIX_REFERENCE"] = "/nonexistent"
        import make_golden as mg
        assert not mg.import_reference()
        mg._stub_modules()
        assert mg.have_real("xgboost") and not mg.have_real("sklearn_crfsuite")
        assert mg.make_G12(os.path.join(%r, "G12_xgb_smoother.npz"))
        assert mg.make_G14(os.path.join(%r, "G14_xgb_base.npz"))
        assert mg.make_G13(os.path.join(%r, "G13_crf_smoother.npz")) is False
    ''') % (str(tmp_path), os.path.join(ROOT, "tests", "golden"), str(outdir), str(outdir), str(outdir))
    env = dict(os.environ)
    env["GNOMIX_REFERENCE"] = "/nonexistent"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g12 = _load("G12_xgb_smoother.npz", str(outdir))
    assert str(g12["xgboost_version"]) == "0.0-standin" and not bool(g12["via_reference"])
    check_xgb_smoother(oracle, g12)
    check_xgb_base(oracle, _load("G14_xgb_base.npz", str(outdir)))
    assert not os.path.exists(os.path.join(str(outdir), "G13_crf_smoother.npz"))


# ---------------------------------------------------------------------------------------------------------------------
# plumbing self-check for G18 / G19: STAND-IN lightgbm / catboost modules that write each library's documented model format around
# random trees and evaluate the library's documented prediction rule directly.  Pins nothing.
# ---------------------------------------------------------------------------------------------------------------------
_STANDIN_LGBM = textwrap.dedent('''
    """STAND-IN for lightgbm: random leaf-wise trees in Booster.model_to_string()'s text layout + LightGBM's own rule on them"""
    import sys
    import numpy as np
    sys.path.insert(0, %(tests)r)
    from test_host_cpu import _lgbm_model_string, _lgbm_predict
    __version__ = "0.0-standin"

    class _Booster:
        def __init__(self, s):
            self._s = s
        def model_to_string(self):
            return self._s

    class LGBMClassifier:
        def __init__(self, **kw):
            self.kw = dict(kw)
        def fit(self, X, y):
            self.A = int(np.max(y)) + 1
            rng = np.random.RandomState(int(self.kw.get("random_state", 0)) + X.shape[1])
            s, self.trees, self.per_iter = _lgbm_model_string(rng, X.shape[1], self.A, rounds=int(self.kw.get("n_estimators", 10)))
            self.booster_ = _Booster(s)
            return self
        def predict_proba(self, X):
            return np.stack([_lgbm_predict(self.trees, self.per_iter, self.A, x) for x in np.asarray(X)])
''')

_STANDIN_CB = textwrap.dedent('''
    """STAND-IN for catboost: random oblivious trees in save_model(format="json")'s layout + CatBoost's own rule on them"""
    import json
    import numpy as np
    __version__ = "0.0-standin"

    class CatBoostClassifier:
        def __init__(self, **kw):
            self.kw = dict(kw)
        def fit(self, X, y):
            self.A = int(np.max(y)) + 1
            self.dims = 1 if self.A == 2 else self.A
            rng = np.random.RandomState(7 + X.shape[1])
            trees = []
            for t in range(int(self.kw.get("n_estimators", 10))):
                d = int(rng.randint(0, int(self.kw.get("max_depth", 4)) + 1))
                splits = [{"float_feature_index": int(rng.randint(X.shape[1])), "border": float(rng.choice([0.5, 1.5, 0.25])),
                           "split_type": "FloatFeature", "split_index": i} for i in range(d)]
                trees.append({"splits": splits, "leaf_values": [float(np.float32(v)) for v in rng.randn(self.dims << d) * 0.4],
                              "leaf_weights": [1] * (1 << d)})
            self.model = {"oblivious_trees": trees, "scale_and_bias": [0.75, [float(b) for b in rng.randn(self.dims) * 0.2]],
                          "features_info": {"float_features": []}}
            return self
        def save_model(self, fn, format="cbm"):
            assert format == "json"
            json.dump(self.model, open(fn, "w"))
        def predict_proba(self, X):
            m, out = self.model, []
            for x in np.asarray(X):
                raw = np.array(m["scale_and_bias"][1], dtype=np.float64)
                for tr in m["oblivious_trees"]:
                    idx = sum((1 << i) for i, sp in enumerate(tr["splits"]) if float(x[sp["float_feature_index"]]) > sp["border"])
                    raw = raw + m["scale_and_bias"][0] * np.array(tr["leaf_values"][idx * self.dims:(idx + 1) * self.dims])
                if self.dims == 1:
                    p1 = 1.0 / (1.0 + np.exp(-raw[0]))
                    out.append([1 - p1, p1])
                else:
                    e = np.exp(raw - raw.max())
                    out.append(e / e.sum())
            return np.array(out)
''')


def test_generator_plumbing_with_standin_lightgbm_and_catboost(oracle, tmp_path):
    """NOT a pin.  Runs make_golden.py G18 / G19 in a fresh interpreter whose `lightgbm` / `catboost` are the stand-ins above (no
    reference checkout: the constructor-arguments branch), then this file's checks on what they wrote: generator, converters and the
    oracle comparison execute end to end and agree with each other."""
    for name, src in (("lightgbm", _STANDIN_LGBM % {"tests": os.path.join(ROOT, "tests")}), ("catboost", _STANDIN_CB)):
        pkg = tmp_path / name
        pkg.mkdir()
        (pkg / "__init__.py").write_text(src)
    outdir = tmp_path / "out"
    outdir.mkdir()
    code = textwrap.dedent('''
        import sys, os
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        os.environ["GNOMIX_REFERENCE"] = "/nonexistent"
        import make_golden as mg
        assert not mg.import_reference()
        assert mg.have_real("lightgbm") and mg.have_real("catboost")
        assert mg.make_G18(os.path.join(%r, "G18_lgbm_base.npz"))
        assert mg.make_G19(os.path.join(%r, "G19_catboost_base.npz"))
    ''') % (str(tmp_path), os.path.join(ROOT, "tests", "golden"), str(outdir), str(outdir))
    env = dict(os.environ)
    env["GNOMIX_REFERENCE"] = "/nonexistent"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    g18 = _load("G18_lgbm_base.npz", str(outdir))
    assert str(g18["lightgbm_version"]) == "0.0-standin" and not bool(g18["via_reference"])
    check_lgbm_base(oracle, g18)
    check_catboost_base(oracle, _load("G19_catboost_base.npz", str(outdir)))

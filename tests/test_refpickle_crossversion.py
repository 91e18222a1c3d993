"""A reference `model.pkl` pickled by an OLDER scikit-learn / numpy than the one reading it (the closable half of SURVEY 8(f).1:
the published bundles were written by scikit-learn 1.0.1 / numpy 1.20.3, requirements.txt:2,6; the test interpreter has 1.7.2 / 2.2).

tests/golden/make_refpickle_py39.py runs in the image's second environment (/opt/conda/bin/python3.9: scikit-learn 0.24.2,
numpy 1.26.4), imports the reference from /root/reference, trains small `src.model.Gnomix` objects with the reference's own
LogisticRegressionBase / CovRSKBase / RFBase, pickles them as Gnomix.save does and stores the reference's OWN predict_proba next to
them — all into pytest's tmp_path (nothing pickled is committed or travels).  Here, under Python 3.10:

    load_reference_pickle (use_sklearn = False AND True; cli.load_model's fallback order) -> from_reference_model -> oracle
    == the old environment's reference output (<= 1e-12, arg-max labels identical); under -m gpu the HIP path as well.

Skipped when the reference tree or the conda interpreter is absent (the GPU box has neither: the -m gpu leg documents the intent and
runs wherever both a GPU and the reference exist)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PY39 = os.environ.get("GNX_OLD_PYTHON", "/opt/conda/bin/python3.9")
pytestmark = pytest.mark.skipif(not (os.path.isdir(REF) and os.path.exists(PY39)), reason="needs /root/reference and the image's older Python environment")


@pytest.fixture(scope="module")
def old_pickles(tmp_path_factory):
    out = tmp_path_factory.mktemp("refpickle_py39")
    r = subprocess.run([PY39, os.path.join(ROOT, "tests", "golden", "make_refpickle_py39.py"), str(out)], capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "PYTHONHOME")})
    if r.returncode != 0:
        pytest.skip("the older environment could not run the reference: " + (r.stderr or r.stdout)[-400:])
    meta = np.load(os.path.join(out, "lr.npz"))
    import sklearn
    # the point of the test: a scikit-learn MAJOR version gap between writer and reader
    assert str(meta["sklearn_version"]).split(".")[0] != sklearn.__version__.split(".")[0], (str(meta["sklearn_version"]), sklearn.__version__)
    return str(out)


def _load_both_modes(path):
    """(mode, model or exception) for use_sklearn False and True — cli.load_model tries True first and falls back to False"""
    from gnomix_amd import refpickle
    out = []
    for use in (False, True):
        try:
            out.append((use, refpickle.load_reference_pickle(path, use_sklearn=use)))
        except Exception as e:  # noqa: BLE001 - an old estimator the new scikit-learn refuses to rebuild is a legitimate outcome of mode True
            out.append((use, e))
    return out


def _oracle_base(oracle, d, Xq):
    if d.base_kind == "logistic":
        return oracle.base_lr(Xq, d.M, d.context, d.lr_coef, d.lr_intercept)
    if d.base_kind == "covrsk":
        wins = [dict(Xfit=w["xfit"], Ms=list(w["ms"]), support=w["support"], dual=w["dual_coef"], intercept=w["intercept"],
                     probA=w["prob_a"], probB=w["prob_b"], n_support=w["n_support"]) for w in d.svc]
        return oracle.base_covrsk(Xq, d.M, d.context, wins)
    if d.base_kind == "rforest":
        rf = {k[3:]: getattr(d, k) for k in ("rf_win_tree0", "rf_tree_off", "rf_left", "rf_right", "rf_feat", "rf_thr", "rf_value")}
        return oracle.base_rforest(rf, Xq, d.M, d.context, d.A)
    raise AssertionError(d.base_kind)


@pytest.mark.parametrize("name,kind", [("lr", "logistic"), ("covrsk", "covrsk"), ("rf", "rforest")])
def test_pickle_from_older_sklearn_converts_and_reproduces_the_reference(old_pickles, oracle, name, kind):
    from gnomix_amd import convert, refpickle
    z = np.load(os.path.join(old_pickles, name + ".npz"))
    Xq, B_ref = z["Xq"], z["B_ref"]
    path = os.path.join(old_pickles, name + ".pkl")
    modes = _load_both_modes(path)
    assert not isinstance(modes[0][1], Exception), "attribute-bag mode (use_sklearn=False) must always read the pickle: %r" % (modes[0][1],)
    worked = 0
    for use, model in modes:
        if isinstance(model, Exception):
            continue            # mode True may refuse an estimator of another major version: the command line then falls back to mode False
        worked += 1
        # nothing of the reference's `src` package (nor of xgboost / crfsuite) was imported to read it
        assert type(model).__module__ == "src.model" and isinstance(model, refpickle.Stub)
        d = convert.from_reference_model(model)
        assert d.base_kind == kind and d.smooth_kind == "crf" and (d.C, d.M, d.A) == (int(z["C"]), int(z["M"]), int(z["A"]))
        assert np.array_equal(d.crf_state, z["crf_state"]) and np.array_equal(d.crf_trans, z["crf_trans"])
        B = _oracle_base(oracle, d, Xq)
        assert B.shape == B_ref.shape
        assert np.max(np.abs(B - B_ref)) <= 1e-12, (name, use, float(np.max(np.abs(B - B_ref))))
        assert np.array_equal(B.argmax(-1), B_ref.argmax(-1))
    assert worked >= 1


def test_command_line_loader_reads_the_old_pickle(old_pickles, monkeypatch):
    """gnomix_amd.cli.load_model's own order (rebuild scikit-learn objects, fall back to attribute bags) up to the point where a device
    is needed: the converted model of the old LogisticRegression pickle round-trips through .gnx"""
    from gnomix_amd import convert, refpickle
    from gnomix_amd.model import GnxModelData
    path = os.path.join(old_pickles, "lr.pkl")
    try:
        ref_model = refpickle.load_reference_pickle(path)
    except Exception:
        ref_model = refpickle.load_reference_pickle(path, use_sklearn=False)
    d = convert.from_reference_model(ref_model)
    out = os.path.join(old_pickles, "lr.gnx")
    d.save(out)
    d2 = GnxModelData.load(out)
    assert np.array_equal(d2.lr_coef, d.lr_coef) and d2.smooth_kind == "crf" and list(d2.population_order) == ["p0", "p1", "p2"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lr", "covrsk", "rf"])
def test_hip_path_on_the_old_pickle(old_pickles, name):
    import gnomix_amd
    from gnomix_amd import convert, refpickle
    z = np.load(os.path.join(old_pickles, name + ".npz"))
    d = convert.from_reference_model(refpickle.load_reference_pickle(os.path.join(old_pickles, name + ".pkl"), use_sklearn=False))
    dev = gnomix_amd.DeviceModel(d)
    B = dev.base_predict(z["Xq"])
    assert np.max(np.abs(B - z["B_ref"])) <= 1e-12 and np.array_equal(B.argmax(-1), z["B_ref"].argmax(-1))

"""CPU: gnomix_amd.convert against REAL reference objects (needs /root/reference and sklearn; skipped elsewhere).
A model trained by the reference's own LogisticRegressionBase / CovRSKBase is exported to flat arrays and run through
the oracle: the export must reproduce the reference's predict_proba."""
import os
import sys
import types

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as mg
    assert mg.import_reference()
    return mg


class CRF_Smoother:  # name-compatible stand-ins: from_reference_model dispatches on the class name like the pickle would
    pass


class _FakeCRF:
    pass


def _fake_crf_smoother(A, S, rng):
    sm = CRF_Smoother()
    sm.S = S
    sm.calibrator = None
    sm.model = types.SimpleNamespace(CRF=_FakeCRF())
    st = rng.standard_normal((A, A))
    tr = rng.standard_normal((A, A))
    sm.model.CRF.state_features_ = {(str(a), str(y)): st[a, y] for a in range(A) for y in range(A)}
    sm.model.CRF.transition_features_ = {(str(a), str(y)): tr[a, y] for a in range(A) for y in range(A)}
    return sm, st, tr


def test_export_logistic_model_reproduces_reference(ref, oracle):
    import warnings
    from src.Base.models import LogisticRegressionBase
    from gnomix_amd import convert, GnxModelData
    rng = np.random.RandomState(1)
    C, M, A = 1237, 50, 4
    W, ctx = C // M, 25
    Xt, yt = ref.synth_admixed(rng, 200, C, A, W, M)
    for w in range(W):
        yt[:A, w] = np.arange(A)
    base = LogisticRegressionBase(chm_len=C, window_size=M, num_ancestry=A, missing_encoding=2, context=ctx, n_jobs=1,
                                  seed=1, verbose=False)
    base.base_multithread = False
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        base.train(Xt, yt)
    Xq, _ = ref.synth_admixed(rng, 9, C, A, W, M, miss=0.05)
    B_ref = base.predict_proba(Xq)
    sm, st, tr = _fake_crf_smoother(A, 75, rng)
    model = types.SimpleNamespace(C=C, M=M, A=A, context=ctx, base=base, smooth=sm, snp_pos=np.arange(C), snp_ref=None,
                                  snp_alt=None, population_order=["a", "b", "c", "d"], gen_map_df={})
    d = convert.from_reference_model(model)
    assert isinstance(d, GnxModelData) and d.base_kind == "logistic" and d.smooth_kind == "crf"
    assert d.lr_coef.shape == (W, A, M + 2 * ctx + C - M * W)
    B = oracle.base_lr(Xq, d.M, d.context, d.lr_coef, d.lr_intercept)
    assert np.max(np.abs(B - B_ref)) < 1e-13
    assert np.array_equal(d.crf_state, st) and np.array_equal(d.crf_trans, tr)
    desc, keep = d.to_desc()   # the description the C ABI receives
    assert desc.lr_ldc == d.lr_coef.shape[2] and desc.C == C


def test_export_calibrator_and_string_kernel_lengths(ref):
    from sklearn.isotonic import IsotonicRegression
    from gnomix_amd import convert
    rng = np.random.RandomState(2)
    iso = [IsotonicRegression(out_of_bounds="clip").fit(rng.rand(200).astype(np.float32), rng.rand(200) < 0.5) for _ in range(3)]
    c = convert.calibrator_arrays(iso)
    assert c["calib_is_f32"] and c["calib_off"][0] == 0 and len(c["calib_x"]) == c["calib_off"][-1]
    assert list(convert.string_kernel_lengths(349, "CovRSK_DP_triangular_numbers")) == [1, 4, 8, 39, 42, 117]
    assert list(convert.string_kernel_lengths(5, "string_kernel_DP_triangular_numbers_multithread")) == [1, 2, 3, 4, 5]
    with pytest.raises(NotImplementedError):
        convert.string_kernel_lengths(5, "poly_kernel")


def test_restricted_unpickle_of_real_reference_objects(ref, oracle, tmp_path):
    """pickle REAL reference objects (src.Base.models.LogisticRegressionBase / RFBase trained by Base.train), read the
    pickle back with gnomix_amd.refpickle (which never imports `src`), convert, and reproduce the reference's output"""
    import pickle
    import warnings
    from src.Base.models import LogisticRegressionBase, RFBase
    from gnomix_amd import convert, refpickle
    rng = np.random.RandomState(2)
    C, M, A = 937, 50, 3
    W, ctx = C // M, 25
    Xt, yt = ref.synth_admixed(rng, 150, C, A, W, M)
    for w in range(W):
        yt[:A, w] = np.arange(A)
    Xq, _ = ref.synth_admixed(rng, 7, C, A, W, M, miss=0.05)
    for cls in (LogisticRegressionBase, RFBase):
        base = cls(chm_len=C, window_size=M, num_ancestry=A, missing_encoding=2, context=ctx, n_jobs=1, seed=1, verbose=False)
        base.base_multithread = False
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            base.train(Xt, yt)
        B_ref = base.predict_proba(Xq)
        sm, st, tr = _fake_crf_smoother(A, 75, rng)
        model = types.SimpleNamespace(C=C, M=M, A=A, context=ctx, base=base, smooth=sm, snp_pos=np.arange(C), snp_ref=None,
                                      snp_alt=None, population_order=["a", "b", "c"], gen_map_df=None)
        p = tmp_path / (cls.__name__ + ".pkl")
        with open(p, "wb") as f:
            pickle.dump({"base": base}, f)
        got = refpickle.load_reference_pickle(str(p))["base"]
        assert isinstance(got, refpickle.Stub) and type(got).__name__ == cls.__name__ and type(got).__module__ == "src.Base.models"
        model.base = got
        d = convert.from_reference_model(model)
        if cls is LogisticRegressionBase:
            B = oracle.base_lr(Xq, M, ctx, d.lr_coef, d.lr_intercept)
            assert np.max(np.abs(B - B_ref)) < 1e-13
        else:
            rf = {k[3:]: getattr(d, k) for k in ("rf_win_tree0", "rf_tree_off", "rf_left", "rf_right", "rf_feat", "rf_thr", "rf_value")}
            assert np.array_equal(oracle.base_rforest(rf, Xq, M, ctx, A), B_ref)

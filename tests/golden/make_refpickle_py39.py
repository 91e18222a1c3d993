#!/usr/bin/env python3
"""Pickle REAL reference models under an OLDER scikit-learn than the one that will read them.

  /opt/conda/bin/python3.9 tests/golden/make_refpickle_py39.py <outdir>

The published Gnomix bundles (reference gnomix.py:26-35, download_pretrained_models.sh:2-10) were pickled by scikit-learn 1.0.1 /
numpy 1.20.3 (requirements.txt:2,6); the image's test interpreter has scikit-learn 1.7.2 / numpy 2.2.  This script runs in the
image's SECOND environment (/opt/conda: Python 3.9, scikit-learn 0.24.2, numpy 1.26.4 — SURVEY.md 8c), imports the reference
read-only from /root/reference, trains three small `src.model.Gnomix` objects with the reference's own bases

    LogisticRegressionBase   CovRSKBase   RFBase          (src/Base/models.py:12-21, 195-215, 54-66)

and writes, per model, <outdir>/<name>.pkl (pickle.dump of the Gnomix object, exactly what Gnomix.save does: model.py:100-102)
and <outdir>/<name>.npz with the query haplotypes and the reference's OWN base.predict_proba(Xq) computed in that environment.
tests/test_refpickle_crossversion.py reads them under Python 3.10 with gnomix_amd.refpickle (both modes), converts, and compares.

The smoother is a CRF_Smoother carrying hand-set weights in the attributes gnomix_amd.convert reads (xgboost / sklearn_crfsuite
are absent from both environments: the booster bytes of a real bundle stay unpinned, VERDICT r5 "missing" 2).
NOTHING written here is committed or travels: the test writes into pytest's tmp_path.
"""
import os
import pickle
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("GNOMIX_REFERENCE", "/root/reference")


class _Placeholder:
    def __init__(self, *a, **k):
        pass


def _stub_third_party():
    """module-level (hence picklable by reference) placeholders for the third-party modules the reference imports at import time
    and that are absent here; a host that has the real ones keeps them"""
    import importlib
    for name in ["xgboost", "allel", "seaborn", "calibration", "sklearn_crfsuite"]:
        try:
            importlib.import_module(name)
            continue
        except Exception:
            pass
        m = types.ModuleType(name)
        sys.modules[name] = m
        if name == "xgboost":
            m.XGBClassifier = type("XGBClassifier", (_Placeholder,), {"__module__": "xgboost"})
        if name == "sklearn_crfsuite":
            m.CRF = type("CRF", (_Placeholder,), {"__module__": "sklearn_crfsuite"})


def synth_admixed(rng, n, C, A, W, M, miss=0.01, switch_p=0.02):
    freq = rng.uniform(0.05, 0.95, size=(A, C))
    y = np.empty((n, W), dtype=int)
    for i in range(n):
        a = rng.randint(A)
        for w in range(W):
            if rng.rand() < switch_p:
                a = rng.randint(A)
            y[i, w] = a
    ysnp = np.repeat(y, M, axis=1)
    ysnp = np.concatenate([ysnp, np.repeat(y[:, -1:], C - ysnp.shape[1], axis=1)], axis=1)
    p = freq[ysnp, np.arange(C)[None, :]]
    X = (rng.uniform(size=(n, C)) < p).astype(np.int8)
    X[rng.uniform(size=(n, C)) < miss] = 2
    return X, y


def main(outdir):
    if not os.path.isdir(REF):
        print("reference absent: nothing written")
        return 3
    _stub_third_party()
    sys.path.insert(0, REF)
    import sklearn
    from src.model import Gnomix
    from src.Base.models import LogisticRegressionBase, CovRSKBase, RFBase
    from src.Smooth.models import CRF_Smoother
    import sklearn_crfsuite
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.RandomState(39)
    A, M, S = 3, 40, 5
    C = 13 * M + 17
    W, ctx = C // M, M // 2
    Xt, yt = synth_admixed(rng, 150, C, A, W, M)
    for w in range(W):
        yt[:A, w] = np.arange(A)          # every window sees every class (no classes_ remap in the vectorized path, base.py:176)
    Xq, _ = synth_admixed(rng, 11, C, A, W, M, miss=0.05, switch_p=0.1)
    st = rng.standard_normal((A, A))
    tr = rng.standard_normal((A, A))
    for name, cls in (("lr", LogisticRegressionBase), ("covrsk", CovRSKBase), ("rf", RFBase)):
        g = Gnomix.__new__(Gnomix)       # (the constructor would build an XGB_Smoother around the absent xgboost)
        g.C, g.M, g.A, g.S, g.W, g.context = C, M, A, S, W, ctx
        g.snp_pos, g.snp_ref, g.snp_alt = np.arange(C) * 37 + 1000, np.array(["A"] * C), np.array(["C"] * C)
        g.population_order, g.calibrate, g.gen_map_df = ["p0", "p1", "p2"], False, {}
        g.path, g.n_jobs, g.seed, g.verbose, g.time, g.accuracies = None, 1, 1, False, {}, {}
        base = cls(chm_len=C, window_size=M, num_ancestry=A, missing_encoding=2, context=ctx, n_jobs=1, seed=1, verbose=False)
        base.base_multithread = False
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            base.train(Xt, yt)
            B_ref = base.predict_proba(Xq)
        g.base = base
        sm = CRF_Smoother.__new__(CRF_Smoother)
        sm.S, sm.W, sm.A, sm.gnofix, sm.calibrate, sm.calibrator = S, W, A, False, False, None
        import src.Smooth.crf as crfmod
        sm.model = crfmod.CRF.__new__(crfmod.CRF)
        sm.model.CRF = sklearn_crfsuite.CRF()
        sm.model.CRF.state_features_ = {(str(a), str(y)): float(st[a, y]) for a in range(A) for y in range(A)}
        sm.model.CRF.transition_features_ = {(str(a), str(y)): float(tr[a, y]) for a in range(A) for y in range(A)}
        g.smooth = sm
        with open(os.path.join(outdir, name + ".pkl"), "wb") as f:
            pickle.dump(g, f)
        np.savez(os.path.join(outdir, name + ".npz"), Xq=Xq, B_ref=np.asarray(B_ref, dtype=np.float64), crf_state=st, crf_trans=tr,
                 C=C, M=M, A=A, S=S, context=ctx, sklearn_version=np.array(sklearn.__version__), numpy_version=np.array(np.__version__),
                 python_version=np.array("%d.%d" % sys.version_info[:2]))
        print("%s: pickled under python %d.%d / scikit-learn %s / numpy %s, B_ref %s" %
              (name, sys.version_info[0], sys.version_info[1], sklearn.__version__, np.__version__, B_ref.shape))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "_refpickle_py39")))

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the survey
container (it is imported read-only from /root/reference; nothing of it is copied here).

The reference has no tests and no golden vectors of its own, so every pin is an
input/output pair produced by executing its own functions:

  G1_lr.npz        LogisticRegressionBase.train + Base.predict_proba   (base.py:146-180, models.py:12-21)
  G2_covrsk.npz    CovSample / CovRSK kernel / CovRSKBase.predict_proba (string_kernel.py:80-110, models.py:195-215)
  G3_slide.npz     slide_window                                         (Smooth/utils.py:4-29)
  G4_smooth.npz    XGB_Smoother.predict_proba/predict glue with the oracle tree walker plugged in
                   as `.model` (smooth.py:40-65, models.py:22-24) — pins slide/f32 cast/reshape/argmax,
                   NOT xgboost arithmetic (xgboost is absent: parity unpinned)
  G5_gnofix.npz    gnofix() and Gnomix.phase() control flow with the same plug (gnofix.py:58-208,
                   model.py:188-214, phasing.py:182-198)
  G6_writers/      get_meta_data / write_msp / write_fb text              (postprocess.py:25-126)
  G7_vcf.npz       vcf_to_npy on a synthetic allel-style dict            (utils.py:104-159)
  G8_calib_sk.npz  Calibrator.fit/transform (Smooth/Calibration.py:19-69); StringKernelBase train+predict (models.py:161-176)
  G9 / G10 / G11   RFBase, PolynomialStringKernelBase, CNN smoother (see the functions)
  G12 / G13 / G14  third-party pins: xgboost smoother, CRFsuite smoother, XGBBase — generated only on a host that has
                   xgboost / sklearn_crfsuite (absent here: they print "skipped"); tests/test_pins_thirdparty.py consumes them
  G18 / G19        third-party pins of the other tree bases: LGBMBase (lightgbm model strings), CBBase (catboost JSON exports)
                   (models.py:38-52, 68-81) — likewise generated only where the packages exist
  G15_lr_binary.npz  A = 2 logistic base (sklearn's one-row binary form)
  G16_lr_train.npz   LogisticRegressionBase.train (base.py:104-127): training data + the reference's fitted coefficients
  G17_cnn_train.npz  CNN.fit (Smooth/cnn.py:104-118): data, initial and trained Conv1d parameters, the DataLoader's row order

Third-party modules the reference imports at module import time but that are absent here
(xgboost, allel, seaborn, calibration, sklearn_crfsuite) are replaced by empty stubs; no code path
used below calls into them.  Skips cleanly when /root/reference is absent (e.g. on the GPU box).
"""
import os
import sys
import types
import hashlib

import numpy as np

REF = os.environ.get("GNOMIX_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _stub_modules():
    """empty stand-ins ONLY for third-party modules that really are absent (a host with xgboost / sklearn_crfsuite keeps
    the real ones: the G12-G14 generators below need them)"""
    import importlib
    for name in ["xgboost", "allel", "seaborn", "calibration", "sklearn_crfsuite"]:
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
            continue
        except Exception:
            pass
        m = types.ModuleType(name)
        m.__gnx_stub__ = True
        sys.modules[name] = m
        if name == "xgboost":
            class _XGBClassifier:  # constructor-compatible placeholder; never fitted or called
                def __init__(self, *a, **k):
                    self.kw = k
            m.XGBClassifier = _XGBClassifier
        if name == "sklearn_crfsuite":
            class _CRF:
                def __init__(self, *a, **k):
                    pass
            m.CRF = _CRF


def have_real(name):
    """True when the third-party module `name` imports for real (not one of the stubs above)"""
    import importlib
    try:
        m = importlib.import_module(name)
    except Exception:
        return False
    return not getattr(m, "__gnx_stub__", False)


def import_reference():
    if not os.path.isdir(REF):
        return False
    _stub_modules()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    return True


def synth_admixed(rng, n, C, A, W, M, miss=0.01, switch_p=0.02):
    """Toy phased haplotypes: per-ancestry allele frequencies, ancestry piecewise-constant over windows."""
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


def make_G1(out):
    from src.Base.models import LogisticRegressionBase
    rng = np.random.RandomState(94305)
    C, M, A = 4037, 100, 7
    W, ctx = C // M, 50
    Xt, yt = synth_admixed(rng, 420, C, A, W, M)
    # make sure every window sees every class (no classes_ remap in the vectorized path, base.py:176)
    for w in range(W):
        for a in range(A):
            yt[a, w] = a
    base = LogisticRegressionBase(chm_len=C, window_size=M, num_ancestry=A, missing_encoding=2, context=ctx,
                                  n_jobs=1, seed=94305, verbose=False)
    base.base_multithread = False  # serial dispatch: same arithmetic, no spawn pool (base.py:170-174)
    base.train(Xt, yt)
    Xq, _ = synth_admixed(rng, 24, C, A, W, M, miss=0.03, switch_p=0.1)
    B = base.predict_proba(Xq)
    M_ = M + 2 * ctx
    rem = C - M * W
    ldc = M_ + rem
    coef = np.zeros((W, A, ldc))
    icpt = np.zeros((W, A))
    for i, m in enumerate(base.models):
        assert list(m.classes_) == list(range(A))
        coef[i, :, :m.coef_.shape[1]] = m.coef_
        icpt[i] = m.intercept_
    np.savez_compressed(out, C=C, M=M, A=A, ctx=ctx, X=Xq, coef=coef, intercept=icpt, B=B)
    print("G1", B.shape, B.dtype)


def make_G2(out):
    import numpy
    from src.Base import string_kernel as sk
    anchors = {}
    for m in (8, 20, 349, 499, 2000, 2500):
        anchors[str(m)] = np.array(sk.CovSample(m, 0.6, 1.0, 37))
    a = np.array([0, 1, 1, 0, 2, 2, 1, 0, 0, 1], dtype=np.int8)
    b = np.array([0, 1, 0, 0, 2, 2, 1, 1, 0, 1], dtype=np.int8)
    k_ab = sk.CovRSK_DP_triangular_numbers(a[None], b[None])
    k_ab_plain = sk.string_kernel_DP_triangular_numbers(a[None], b[None])
    ones = np.zeros((1, 8), dtype=np.int8)
    k_eq = sk.CovRSK_DP_triangular_numbers(ones, ones)
    k_eq_plain = sk.string_kernel_DP_triangular_numbers(ones, ones)

    rng = np.random.RandomState(7)
    C, M, A = 537, 50, 3
    W, ctx = C // M, 25
    Xt, yt = synth_admixed(rng, 60, C, A, W, M, switch_p=0.0)
    for w in range(W):
        for c in range(A):
            yt[c * 2:(c * 2 + 2), w] = c
    real_ver = numpy.__version__
    numpy.__version__ = "1.26.4"  # models.py:200 parses the MINOR version ("2.2.6" -> 2 < 20)
    try:
        from src.Base.models import CovRSKBase
        base = CovRSKBase(chm_len=C, window_size=M, num_ancestry=A, missing_encoding=2, context=ctx, n_jobs=1,
                          seed=94305, verbose=False)
    finally:
        numpy.__version__ = real_ver
    base.base_multithread = False
    base.log_inference = False
    np.random.seed(11)
    base.train(Xt, yt)
    Xq, _ = synth_admixed(rng, 10, C, A, W, M, miss=0.05, switch_p=0.2)
    B = base.predict_proba(Xq)
    d = dict(C=C, M=M, A=A, ctx=ctx, X=Xq, B=B, k_ab=k_ab, k_ab_plain=k_ab_plain, k_eq=k_eq, k_eq_plain=k_eq_plain,
             a=a, b=b)
    for k, v in anchors.items():
        d["Ms_%s" % k] = v
    # a kernel matrix for window 3, computed by the reference kernel
    M_ = M + 2 * ctx
    Xp = base.pad(Xq)
    Xw3 = Xp[:, 3 * M:3 * M + M_]
    for i, m in enumerate(base.models):
        assert list(m.classes_) == list(range(A))
        xf = getattr(m, "_BaseLibSVM__Xfit")
        d["w%d_Xfit" % i] = np.asarray(xf, dtype=np.int8)
        d["w%d_support" % i] = m.support_.astype(np.int32)
        d["w%d_dual" % i] = m._dual_coef_
        d["w%d_intercept" % i] = m._intercept_
        d["w%d_probA" % i] = m._probA
        d["w%d_probB" % i] = m._probB
        d["w%d_nsv" % i] = m._n_support.astype(np.int32)
    d["K_w3"] = sk.CovRSK_DP_triangular_numbers(Xw3, d["w3_Xfit"])
    np.savez_compressed(out, **d)
    print("G2", B.shape, {k: list(v) for k, v in anchors.items() if int(k) < 600})


def make_G3(out):
    from src.Smooth.utils import slide_window
    rng = np.random.RandomState(3)
    d = {}
    B0 = np.arange(6, dtype=float).reshape(1, 6, 1)
    d["tiny_B"] = B0
    d["tiny_S3"] = slide_window(B0, 3)[0]
    for name, (N, W, A, S) in {"a": (2, 9, 3, 5), "b": (3, 14, 2, 7), "c": (1, 80, 7, 75), "d": (2, 151, 4, 75)}.items():
        B = rng.uniform(size=(N, W, A))
        o = slide_window(B, S)[0]
        d[name + "_B"] = B
        d[name + "_S"] = S
        if o.size < 20000:
            d[name + "_out"] = o
        else:  # big: keep a digest + strided sample of rows
            d[name + "_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(o).tobytes()).digest(), dtype=np.uint8)
            d[name + "_rows"] = np.arange(0, o.shape[0], 7)
            d[name + "_out"] = o[::7]
    np.savez_compressed(out, **d)
    print("G3 ok")


def handmade_smoothing_trees(A, S, reach=6):
    """A tiny 'sensible' smoother in xgboost schema: class-c trees vote +/- by thresholding the base
    probability of class c at windows near the centre of the sliding window (centre index = pad-1)."""
    sys.path.insert(0, ROOT)
    from oracle import gnx_oracle as O
    pad = (S + 1) // 2
    centre = pad - 1  # slide_window is centred on w-1+...: feature s=pad-1 is window w-1
    offs, L, R, F, Cd, cls = [0], [], [], [], [], []
    for k in range(-reach, reach + 1):
        for c in range(A):
            s = centre + 1 + k  # window w+k
            f = s * A + c
            wgt = 0.6 / (1 + abs(k))
            # depth-2 tree: root on class-c prob at w+k; right child refines
            nodes = [(1, 2, f, 0.5), (-1, -1, 0, -wgt), (3, 4, f, 0.8), (-1, -1, 0, wgt), (-1, -1, 0, 1.5 * wgt)]
            for (l, r, ff, cc) in nodes:
                L.append(l); R.append(r); F.append(ff); Cd.append(cc)
            offs.append(len(L))
            cls.append(c)
    return O.Trees(np.array(offs), np.array(L), np.array(R), np.array(F), np.array(Cd, dtype=np.float32),
                   np.array(cls), A)


def trees_to_npz(prefix, T):
    return {prefix + "tree_off": T.tree_off, prefix + "left": T.left, prefix + "right": T.right,
            prefix + "feat": T.feat, prefix + "cond": T.cond, prefix + "tree_class": T.tree_class,
            prefix + "n_class": T.n_class, prefix + "base_score": T.base_score}


def make_G4(out):
    sys.path.insert(0, ROOT)
    from oracle import gnx_oracle as O
    from src.Smooth.models import XGB_Smoother
    rng = np.random.RandomState(4)
    N, W, A, S = 5, 163, 7, 75
    B = rng.dirichlet(np.ones(A) * 0.3, size=(N, W))
    T = O.random_trees(6, A, S * A, depth=4, seed=5)
    sm = XGB_Smoother(n_windows=W, num_ancestry=A, smooth_window_size=S, n_jobs=1, calibrate=False, mode_filter=0,
                      seed=1, verbose=False)
    sm.model = O.OracleXGBSmoother(T, W, A, S)  # the oracle tree walker stands in for XGBClassifier
    proba = sm.predict_proba(B)
    labels = sm.predict(B)
    d = dict(B=B, S=S, proba=proba, labels=labels)
    d.update(trees_to_npz("t_", T))
    np.savez_compressed(out, **d)
    print("G4", proba.shape, proba.dtype, labels.dtype)


def make_G5(out):
    sys.path.insert(0, ROOT)
    from oracle import gnx_oracle as O
    from src.Gnofix.gnofix import gnofix
    from src.model import Gnomix
    rng = np.random.RandomState(5)
    W, A, S, Mw = 160, 4, 75, 6
    Cn = W * Mw + 3
    T = handmade_smoothing_trees(A, S)
    sm = O.OracleXGBSmoother(T, W, A, S)
    d = dict(W=W, A=A, S=S, C=Cn)
    d.update(trees_to_npz("t_", T))

    def individual(switch_points, seg_m, seg_p, noise=0.15):
        """true ancestry per haplotype from segment lists, then scramble phase at switch_points"""
        def expand(seg):
            y = np.empty(W, dtype=int)
            for (a, b, c) in seg:
                y[a:b] = c
            return y
        ym, yp = expand(seg_m), expand(seg_p)
        Bm = np.full((W, A), noise / (A - 1)); Bm[np.arange(W), ym] = 1 - noise
        Bp = np.full((W, A), noise / (A - 1)); Bp[np.arange(W), yp] = 1 - noise
        Bm = Bm * rng.uniform(0.8, 1.2, size=Bm.shape); Bm /= Bm.sum(1, keepdims=True)
        Bp = Bp * rng.uniform(0.8, 1.2, size=Bp.shape); Bp /= Bp.sum(1, keepdims=True)
        Xm = rng.randint(0, 2, size=Cn); Xp = rng.randint(0, 2, size=Cn)
        for s in switch_points:
            Bm, Bp = np.concatenate([Bm[:s], Bp[s:]]), np.concatenate([Bp[:s], Bm[s:]])
            i = s * (Cn // W)
            Xm, Xp = np.concatenate([Xm[:i], Xp[i:]]), np.concatenate([Xp[:i], Xm[i:]])
        return Xm, Xp, np.stack([Bm, Bp])

    cases = {
        "none": individual([], [(0, W, 0)], [(0, W, 1)]),
        "one": individual([70], [(0, W, 0)], [(0, W, 1)]),
        "two": individual([50, 110], [(0, W, 2)], [(0, 90, 1), (90, W, 3)]),
        "edges": individual([5, 152], [(0, W, 0)], [(0, W, 3)]),
        "many": individual([20, 40, 41, 77, 120, 121, 122], [(0, 80, 0), (80, W, 2)], [(0, W, 1)]),
    }
    for name, (Xm, Xp, Bi) in cases.items():
        r = gnofix(Xm, Xp, B=Bi, smoother=sm)
        X_m, X_p, Y_m, Y_p, history, tracker = r
        d[name + "_Xm"], d[name + "_Xp"], d[name + "_B"] = Xm, Xp, Bi
        d[name + "_oXm"], d[name + "_oXp"], d[name + "_oYm"], d[name + "_oYp"] = X_m, X_p, Y_m, Y_p
        d[name + "_trk"] = np.array(tracker)
        d[name + "_nhist"] = history.shape[-1] if history.ndim == 3 else 1
        print("G5", name, "history", d[name + "_nhist"])
    # chaotic smoother (random trees): exercises many accepted/rejected switches and the max_it stop
    Tr = O.random_trees(3, A, S * A, depth=4, seed=9, leaf_scale=1.0)
    smr = O.OracleXGBSmoother(Tr, W, A, S)
    Xm, Xp, Bi = individual([30, 90], [(0, W, 0)], [(0, W, 1)], noise=0.5)
    r = gnofix(Xm, Xp, B=Bi, smoother=smr, max_it=4)
    d.update(trees_to_npz("r_", Tr))
    d["rand_Xm"], d["rand_Xp"], d["rand_B"] = Xm, Xp, Bi
    d["rand_oXm"], d["rand_oXp"], d["rand_oYm"], d["rand_oYp"] = r[0], r[1], r[2], r[3]
    d["rand_trk"] = np.array(r[5]); d["rand_nhist"] = r[4].shape[-1]
    print("G5 rand history", d["rand_nhist"])
    # Gnomix.phase wrapper (model.py:188-214) over 3 individuals
    g = Gnomix.__new__(Gnomix)
    g.smooth, g.W, g.A, g.base = sm, W, A, None
    Xs, Bs = [], []
    for name in ("one", "two", "none"):
        Xm, Xp, Bi = cases[name]
        Xs += [Xm, Xp]; Bs += [Bi[0], Bi[1]]
    Xs, Bs = np.array(Xs), np.array(Bs)
    Xph, Yph = g.phase(Xs, B=Bs)
    d["phase_X"], d["phase_B"], d["phase_oX"], d["phase_oY"] = Xs, Bs, Xph, Yph
    np.savez_compressed(out, **d)


def make_G6(outdir):
    import pandas as pd
    from src.postprocess import get_meta_data, write_msp, write_fb
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.RandomState(6)
    W, M, A, n_ind = 7, 10, 3, 2
    Cn = W * M + 4
    model_pos = np.sort(rng.choice(np.arange(10000, 900000), size=Cn, replace=False))
    query_pos = np.sort(rng.choice(model_pos, size=Cn - 9, replace=False))
    gen_map_df = pd.DataFrame({"chm": ["22"] * 5, "pos": [5000, 200000, 400000, 700000, 1000000],
                               "pos_cm": [0.0, 0.31, 0.7345678, 1.2, 2.05]})
    meta = get_meta_data("22", model_pos, query_pos, W, M, gen_map_df)
    proba = rng.dirichlet(np.ones(A), size=(2 * n_ind, W)).astype(np.float32)
    labels = np.argmax(proba, axis=-1)
    pops = ["AFR", "EUR", "EAS"]
    samples = np.array(["HG001", "NA002"])
    write_msp(os.path.join(outdir, "ref"), meta, labels, pops, samples)
    write_fb(os.path.join(outdir, "ref"), meta, proba, pops, samples)
    np.savez_compressed(os.path.join(outdir, "inputs.npz"), model_pos=model_pos, query_pos=query_pos,
                        gm_pos=gen_map_df.pos.values, gm_cm=gen_map_df.pos_cm.values, proba=proba, labels=labels,
                        W=W, M=M, A=A, pops=np.array(pops), samples=samples)
    print("G6 ok")


def make_G7(out):
    """vcf_to_npy (src/utils.py:104-159) on a synthetic scikit-allel-style dict: SNP intersection with the model,
    REF-mismatch flip, missing calls and multi-allelic codes -> 2"""
    from src.utils import vcf_to_npy
    rng = np.random.RandomState(7)
    n_var, n_ind, Cm = 60, 5, 50
    all_pos = np.sort(rng.choice(np.arange(1000, 9000), size=80, replace=False))
    model_pos = np.sort(rng.choice(all_pos, size=Cm, replace=False))
    vcf_pos = np.sort(rng.choice(all_pos, size=n_var, replace=False))
    bases = np.array(list("ACGT"))
    model_ref = bases[rng.randint(4, size=Cm)]
    vcf_ref = bases[rng.randint(4, size=n_var)]
    common, mi, vi = np.intersect1d(model_pos, vcf_pos, return_indices=True)
    agree = rng.rand(len(common)) < 0.7
    vcf_ref[vi[agree]] = model_ref[mi[agree]]
    gt = rng.randint(0, 2, size=(n_var, n_ind, 2)).astype(np.int8)
    gt[rng.rand(*gt.shape) < 0.05] = -1
    gt[rng.rand(*gt.shape) < 0.03] = 2
    vd = {"calldata/GT": gt, "variants/POS": vcf_pos, "variants/REF": vcf_ref}
    X, vcf_idx, fmt_idx = vcf_to_npy(vd, model_pos, model_ref, return_idx=True, verbose=False)
    X2 = vcf_to_npy({"calldata/GT": gt.copy(), "variants/POS": vcf_pos, "variants/REF": vcf_ref}, verbose=False)
    np.savez_compressed(out, gt=gt, vcf_pos=vcf_pos, vcf_ref=vcf_ref.astype("U1"), model_pos=model_pos,
                        model_ref=model_ref.astype("U1"), X=X, vcf_idx=vcf_idx, fmt_idx=fmt_idx, X_nofmt=X2)
    print("G7", X.shape, X.dtype)


def make_G8(out):
    """Calibrator.fit / transform (Smooth/Calibration.py:19-69: per-class sklearn IsotonicRegression + renormalise) and the
    plain triangular string kernel base (StringKernelBase, Base/models.py:161-176)"""
    from src.Smooth.Calibration import Calibrator
    rng = np.random.RandomState(8)
    A = 5
    proba_fit = rng.dirichlet(np.ones(A) * 0.5, size=3000).astype(np.float32)
    y = np.array([rng.choice(A, p=p / p.sum()) for p in proba_fit.astype(np.float64) ** 0.7])
    cal = Calibrator(A)
    cal.fit(proba_fit, y)
    P = rng.dirichlet(np.ones(A) * 0.4, size=(6, 40)).astype(np.float32)
    P[0, 0] = [1, 0, 0, 0, 0]
    out_p = cal.transform(P)
    d = dict(A=A, P=P, out=out_p)
    for i, m in enumerate(cal.models):
        d["x%d" % i] = np.asarray(m.X_thresholds_, dtype=np.float64)
        d["y%d" % i] = np.asarray(m.y_thresholds_, dtype=np.float64)
    # StringKernelBase on a tiny problem
    import numpy
    real_ver = numpy.__version__
    numpy.__version__ = "1.26.4"
    try:
        from src.Base.models import StringKernelBase
        C, M, A2 = 237, 30, 3
        W, ctx = C // M, 15
        base = StringKernelBase(chm_len=C, window_size=M, num_ancestry=A2, missing_encoding=2, context=ctx, n_jobs=1,
                                seed=94305, verbose=False)
    finally:
        numpy.__version__ = real_ver
    base.log_inference = False
    Xt, yt = synth_admixed(rng, 36, C, A2, W, M, switch_p=0.0)
    for w in range(W):
        for c in range(A2):
            yt[c * 2:(c * 2 + 2), w] = c
    np.random.seed(3)
    base.train(Xt, yt)
    Xq, _ = synth_admixed(rng, 7, C, A2, W, M, miss=0.05, switch_p=0.2)
    Bq = base.predict_proba(Xq)
    d.update(sk_C=C, sk_M=M, sk_A=A2, sk_ctx=ctx, sk_X=Xq, sk_B=Bq)
    for i, m in enumerate(base.models):
        d["sk%d_Xfit" % i] = np.asarray(getattr(m, "_BaseLibSVM__Xfit"), dtype=np.int8)
        d["sk%d_support" % i] = m.support_.astype(np.int32)
        d["sk%d_dual" % i] = m._dual_coef_
        d["sk%d_intercept" % i] = m._intercept_
        d["sk%d_probA" % i] = m._probA
        d["sk%d_probB" % i] = m._probB
        d["sk%d_nsv" % i] = m._n_support.astype(np.int32)
    np.savez_compressed(out, **d)
    print("G8", out_p.shape, out_p.dtype, Bq.shape)


def make_G9(out):
    """RFBase (src/Base/models.py:54-66) trained and run by the reference's own Base machinery: scikit-learn is
    installed, so the tree-ensemble base family has a real pin.  base_multithread is switched off exactly as in G1
    (serial dispatch, same arithmetic); the forests themselves run single-threaded (models.py:62-63)."""
    from src.Base.models import RFBase
    sys.path.insert(0, ROOT)
    from gnomix_amd.convert import rforest_from_sklearn
    rng = np.random.RandomState(94309)
    C, M, A = 2537, 100, 5
    W, ctx = C // M, 50
    Xt, yt = synth_admixed(rng, 300, C, A, W, M)
    for w in range(W):
        for a in range(A):
            yt[a, w] = a
    base = RFBase(chm_len=C, window_size=M, num_ancestry=A, missing_encoding=2, context=ctx, n_jobs=1, seed=94309, verbose=False)
    base.base_multithread = False
    np.random.seed(94309)  # RandomForestClassifier(random_state=None) draws from numpy's global state
    base.train(Xt, yt)
    Xq, _ = synth_admixed(rng, 40, C, A, W, M, miss=0.05, switch_p=0.1)
    B = base.predict_proba(Xq)
    rf = rforest_from_sklearn(base.models, A)
    np.savez_compressed(out, C=C, M=M, A=A, ctx=ctx, X=Xq, B=B, **rf)
    print("G9", B.shape, B.dtype, "trees", len(rf["rf_tree_off"]) - 1, "nodes", len(rf["rf_left"]))


def make_G10(out):
    """PolynomialStringKernelBase (src/Base/models.py:178-193): SVC(kernel=poly_kernel, probability=True) per window, trained
    and run by the reference's own Base machinery; also the raw kernel matrix of one window (string_kernel.py:40-61)."""
    from src.Base.models import PolynomialStringKernelBase
    from src.Base.string_kernel import poly_kernel
    sys.path.insert(0, ROOT)
    from gnomix_amd.convert import svc_window_from_sklearn
    rng = np.random.RandomState(94310)
    C, M, A = 457, 50, 3
    W, ctx = C // M, 25
    Xt, yt = synth_admixed(rng, 45, C, A, W, M)
    for w in range(W):
        for a in range(A):
            yt[a, w] = a
    import numpy
    real_ver = numpy.__version__
    numpy.__version__ = "1.26.4"  # models.py:183 parses the MINOR version ("2.2.6" -> 2 < 20)
    try:
        base = PolynomialStringKernelBase(chm_len=C, window_size=M, num_ancestry=A, missing_encoding=2, context=ctx, n_jobs=1,
                                          seed=94310, verbose=False)
    finally:
        numpy.__version__ = real_ver
    base.base_multithread = False
    base.log_inference = False
    np.random.seed(94310)   # libsvm's probability cross-validation shuffles with rand(): sklearn seeds it from numpy
    base.train(Xt, yt)
    Xq, _ = synth_admixed(rng, 14, C, A, W, M, miss=0.05, switch_p=0.1)
    Xq[0] = Xt[3]; Xq[1, :200] = Xt[5, :200]     # long matching runs
    B = base.predict_proba(Xq)
    M_ = M + 2 * ctx
    wins = [svc_window_from_sklearn(m, M_ + (C - M * W if i == W - 1 else 0), "poly_kernel") for i, m in enumerate(base.models)]
    flat = {}
    for i, w in enumerate(wins):
        for k, v in w.items():
            flat["svc%d_%s" % (i, k)] = np.asarray(v)
    Xp = np.concatenate([Xq[:, :ctx][:, ::-1], Xq, Xq[:, -ctx:][:, ::-1]], axis=1)
    K0 = poly_kernel(Xp[:, :M_], wins[0]["xfit"], p=1.2)
    np.savez_compressed(out, C=C, M=M, A=A, ctx=ctx, X=Xq, B=B, K0=K0, n_win=W, **flat)
    print("G10", B.shape, "K0", K0.shape, K0.max())


def make_G11(out):
    """CNN_Smoother ("large" mode, src/Smooth/models.py:35-42, cnn.py).  The reference builds nn.Conv1d with
    padding_mode="reflection", which torch <= 1.4 silently treats as zero padding and torch >= 1.5 rejects; the layer is
    constructed here the only way it ever ran — zero padding — by translating that one string while the reference's own
    CNN class is instantiated; everything else (as_torch_tensor, forward_tensors, Softmax, swapaxes) is the reference's."""
    import torch
    from torch import nn
    real_conv = nn.Conv1d

    class Conv1dOldTorch(real_conv):
        def __init__(self, *a, padding_mode="zeros", **k):
            super().__init__(*a, padding_mode="zeros" if padding_mode == "reflection" else padding_mode, **k)

    from src.Smooth import cnn as ref_cnn
    nn.Conv1d = Conv1dOldTorch
    try:
        torch.manual_seed(94311)
        A, S, W, N = 5, 21, 90, 9
        model = ref_cnn.CNN(num_classes=A, num_features=S)
    finally:
        nn.Conv1d = real_conv
    model.eval()
    rng = np.random.RandomState(94311)
    B = rng.dirichlet(np.ones(A) * 0.4, size=(N, W))
    with torch.no_grad():
        proba = model.predict_proba(B)
        labels = model.predict(B)
    wgt = model.smoothNet[0].weight.detach().numpy().copy()
    bias = model.smoothNet[0].bias.detach().numpy().copy()
    np.savez_compressed(out, A=A, S=S, W=W, B=B, weight=wgt, bias=bias, proba=proba, labels=labels)
    print("G11", proba.shape, proba.dtype, labels.shape)


# ----------------------------------------------------------------------------------------------------------------------
# Third-party pins.  xgboost (requirements.txt:11 pins 1.1.1) and sklearn-crfsuite
# (requirements.txt:9 pins 0.3.6) are absent from the build image, so rows a6 / a7 / XGBBase of SURVEY §8 are "parity
# unpinned".  The three generators below turn the first host that has them into a one-command pin:
#
#     pip install xgboost==1.1.1 sklearn-crfsuite==0.3.6        (any host; the reference checkout is optional)
#     python tests/golden/make_golden.py G12 G13 G14 && python -m pytest tests/test_pins_thirdparty.py
#
# They skip cleanly (print + return) when the package is absent.  With /root/reference present the reference's OWN classes
# are trained and run (XGB_Smoother, CRF, XGBBase); without it the same third-party estimator is constructed with the
# constructor arguments the reference uses (cited) and the window slicing / slide_window come from the oracle, which
# G1 / G3 pin bit-for-bit against the reference.  `via_reference` in the fixture says which.
# ----------------------------------------------------------------------------------------------------------------------
def _booster_artifacts(booster):
    """(raw model bytes as uint8, JSON tree dumps joined as one JSON list string)"""
    import json as _json
    try:
        raw = booster.save_raw()          # xgboost 1.1.1 ... 1.5: legacy binary; >= 2.0: UBJSON by default
    except TypeError:
        raw = booster.save_raw("deprecated")
    raw = bytes(raw)
    if raw[:1] not in (b"{", b"b", b"C") and hasattr(booster, "save_raw"):
        try:
            raw = bytes(booster.save_raw(raw_format="json"))   # newer xgboost: ask for the JSON schema parse_xgb_raw reads
        except TypeError:
            pass
    dumps = booster.get_dump(dump_format="json")
    return np.frombuffer(raw, dtype=np.uint8).copy(), np.array(_json.dumps([_json.loads(x) for x in dumps]))


def make_G12(out):
    """XGB smoother: XGBClassifier(multi:softprob) on slide_window rows -> booster bytes + JSON dump +
    predict_proba / predict exactly as Smoother.predict_proba / predict return them (smooth.py:40-65)."""
    if not have_real("xgboost"):
        print("G12 skipped: xgboost is not installed (pip install xgboost==1.1.1)")
        return False
    import xgboost
    sys.path.insert(0, ROOT)
    from oracle import gnx_oracle as O
    rng = np.random.RandomState(94312)
    N, W, A, S = 60, 48, 4, 11
    via_ref = os.path.isdir(REF)
    y = np.empty((N, W), dtype=int)
    for i in range(N):
        a = rng.randint(A)
        for w in range(W):
            if rng.rand() < 0.08:
                a = rng.randint(A)
            y[i, w] = a
    y[:A, :] = np.arange(A)[:, None]
    B = np.full((N, W, A), 0.25 / (A - 1)); B[np.arange(N)[:, None], np.arange(W)[None, :], y] = 0.75
    B = B * rng.uniform(0.5, 1.5, size=B.shape); B /= B.sum(-1, keepdims=True)
    Bq = rng.dirichlet(np.ones(A) * 0.5, size=(9, W))
    Bq[:4] = B[:4]
    if via_ref:
        from src.Smooth.models import XGB_Smoother
        sm = XGB_Smoother(n_windows=W, num_ancestry=A, smooth_window_size=S, n_jobs=1, calibrate=False, mode_filter=0,
                          seed=94312, verbose=False)
        sm.model.set_params(n_estimators=12)    # 12 rounds instead of 100: same arithmetic, small fixture
        sm.train(B, y)
        proba, labels, model = sm.predict_proba(Bq), sm.predict(Bq), sm.model
    else:  # constructor arguments of src/Smooth/models.py:14-20
        model = xgboost.XGBClassifier(n_estimators=12, max_depth=4, learning_rate=0.1, reg_lambda=1, reg_alpha=0, nthread=1,
                                      random_state=94312, num_class=A, objective="multi:softprob", eval_metric="mlogloss")
        model.fit(O.slide_window(B, S), y.reshape(-1))
        proba = model.predict_proba(O.slide_window(Bq, S)).reshape(-1, W, A)
        labels = np.argmax(proba, axis=-1)
    raw, dumps = _booster_artifacts(model.get_booster())
    np.savez_compressed(out, A=A, S=S, W=W, B=Bq, proba=np.asarray(proba), labels=np.asarray(labels), raw=raw, dumps=dumps,
                        xgboost_version=np.array(xgboost.__version__), via_reference=via_ref)
    print("G12 xgboost", xgboost.__version__, "proba", np.asarray(proba).dtype, np.asarray(proba).shape, "raw bytes", raw.size)
    return True


def make_G13(out):
    """CRF smoother: sklearn_crfsuite.CRF(all_possible_transitions / states) -> state_features_,
    transition_features_ and predict_marginals as CRF.predict_proba returns them (crf.py:9-15, 62-67)."""
    if not have_real("sklearn_crfsuite"):
        print("G13 skipped: sklearn_crfsuite is not installed (pip install sklearn-crfsuite==0.3.6)")
        return False
    import sklearn_crfsuite
    rng = np.random.RandomState(94313)
    N, W, A = 40, 30, 4
    y = np.empty((N, W), dtype=int)
    for i in range(N):
        a = rng.randint(A)
        for w in range(W):
            if rng.rand() < 0.1:
                a = rng.randint(A)
            y[i, w] = a
    y[:A, :] = np.arange(A)[:, None]
    B = np.full((N, W, A), 0.3 / (A - 1)); B[np.arange(N)[:, None], np.arange(W)[None, :], y] = 0.7
    B = B * rng.uniform(0.5, 1.5, size=B.shape); B /= B.sum(-1, keepdims=True)
    Bq = rng.dirichlet(np.ones(A) * 0.6, size=(7, W))
    via_ref = os.path.isdir(REF)
    if via_ref:
        from src.Smooth.crf import CRF
        m = CRF(max_it=200)
        m.fit(B, y)
        proba, crf = m.predict_proba(Bq), m.CRF
    else:  # constructor arguments of src/Smooth/crf.py:9-15, feature dicts of crf.py:17-34
        crf = sklearn_crfsuite.CRF(algorithm="lbfgs", max_iterations=200, all_possible_transitions=True, all_possible_states=True)
        feats = lambda Z: [[{str(a): Z[i, b, a] for a in range(A)} for b in range(W)] for i in range(len(Z))]
        crf.fit(feats(B), [[str(v) for v in row] for row in y])
        mg = crf.predict_marginals(feats(Bq))
        proba = np.array([[[mg[i][b][str(a)] for a in range(A)] for b in range(W)] for i in range(len(Bq))])
    state = np.zeros((A, A)); trans = np.zeros((A, A))
    for (attr, label), w in crf.state_features_.items():
        state[int(attr), int(label)] = w
    for (y0, y1), w in crf.transition_features_.items():
        trans[int(y0), int(y1)] = w
    # Bt / yt = the training set: the fit itself is the pin of the trainer (gnx_train_crf, tests/test_pins_thirdparty.py)
    np.savez_compressed(out, A=A, W=W, B=Bq, proba=np.asarray(proba, dtype=np.float64), state=state, trans=trans, Bt=B, yt=y.astype(np.int32),
                        via_reference=via_ref, version=np.array(getattr(sklearn_crfsuite, "__version__", "?")))
    print("G13 crf marginals", np.asarray(proba).shape)
    return True


def make_G14(out):
    """XGBBase (SURVEY §8 a4'' / f3): one XGBClassifier(n_estimators=20, max_depth=4, missing=2) per window on the
    window's SNPs (Base/models.py:24-35) -> per-window booster bytes + Base.predict_proba; A = 3 (multi:softprob) and
    A = 2 (binary:logistic, one tree per round)."""
    if not have_real("xgboost"):
        print("G14 skipped: xgboost is not installed (pip install xgboost==1.1.1)")
        return False
    import xgboost
    sys.path.insert(0, ROOT)
    from oracle import gnx_oracle as O
    via_ref = os.path.isdir(REF)
    d = dict(xgboost_version=np.array(xgboost.__version__), via_reference=via_ref)
    for tag, A in (("m", 3), ("b", 2)):
        rng = np.random.RandomState(94314 + A)
        C, M = 1237, 100
        W, ctx = C // M, 50
        Xt, yt = synth_admixed(rng, 150, C, A, W, M, miss=0.04)
        for w in range(W):
            for a in range(A):
                yt[a, w] = a
        Xq, _ = synth_admixed(rng, 21, C, A, W, M, miss=0.06, switch_p=0.1)
        if via_ref:
            from src.Base.models import XGBBase
            base = XGBBase(chm_len=C, window_size=M, num_ancestry=A, missing_encoding=2, context=ctx, n_jobs=1, seed=94314,
                           verbose=False)
            base.base_multithread = False
            base.train(Xt, yt)
            Bq, models = base.predict_proba(Xq), base.models
        else:  # constructor arguments of src/Base/models.py:31-34; window slicing from the oracle (pinned by G1)
            models, cols = [], []
            wins_t, wins_q = dict(O.base_windows(Xt, M, ctx)), dict(O.base_windows(Xq, M, ctx))
            for w in range(W):
                m = xgboost.XGBClassifier(n_estimators=20, max_depth=4, learning_rate=0.1, reg_lambda=1, reg_alpha=0, missing=2,
                                          random_state=94314)
                m.fit(wins_t[w], yt[:, w])
                models.append(m)
                cols.append(m.predict_proba(wins_q[w]))
            Bq = np.stack(cols, axis=1)
        raws, offs, dumps = [], [0], []
        for m in models:
            raw, dj = _booster_artifacts(m.get_booster())
            raws.append(raw); offs.append(offs[-1] + raw.size); dumps.append(str(dj))
        d.update({tag + "_C": C, tag + "_M": M, tag + "_A": A, tag + "_ctx": ctx, tag + "_X": Xq, tag + "_B": np.asarray(Bq),
                  tag + "_raw": np.concatenate(raws), tag + "_raw_off": np.array(offs, np.int64), tag + "_dumps": np.array(dumps)})
        print("G14", tag, "A", A, "B", np.asarray(Bq).shape, np.asarray(Bq).dtype)
    np.savez_compressed(out, **d)
    return True


def _tree_base_data(A, seed):
    """training / query haplotypes of the per-window tree bases' pins (G14, G18, G19): every window sees every class"""
    rng = np.random.RandomState(seed)
    C, M = 1237, 100
    W, ctx = C // M, 50
    Xt, yt = synth_admixed(rng, 150, C, A, W, M, miss=0.04)
    for w in range(W):
        for a in range(A):
            yt[a, w] = a
    Xq, _ = synth_admixed(rng, 21, C, A, W, M, miss=0.06, switch_p=0.1)
    return C, M, W, ctx, Xt, yt, Xq


def _fit_windows(make_model, base_cls_name, C, M, A, ctx, Xt, yt, Xq, seed):
    """the reference's own Base subclass when the checkout is present (src/Base/base.py:104-180), else one model per window with
    the constructor arguments of src/Base/models.py on the oracle's window slices (pinned by G1) -> (B (N, W, A), fitted models)"""
    sys.path.insert(0, ROOT)
    from oracle import gnx_oracle as O
    if os.path.isdir(REF):
        import src.Base.models as RM
        base = getattr(RM, base_cls_name)(chm_len=C, window_size=M, num_ancestry=A, missing_encoding=2, context=ctx, n_jobs=1,
                                          seed=seed, verbose=False)
        base.base_multithread = False
        base.train(Xt, yt)
        return np.asarray(base.predict_proba(Xq)), base.models
    models, cols = [], []
    wins_t, wins_q = dict(O.base_windows(Xt, M, ctx)), dict(O.base_windows(Xq, M, ctx))
    for w in range(C // M):
        m = make_model()
        m.fit(wins_t[w], yt[:, w])
        models.append(m)
        cols.append(np.asarray(m.predict_proba(wins_q[w])))
    return np.stack(cols, axis=1), models


def make_G18(out):
    """LGBMBase (src/Base/models.py:38-52): LGBMClassifier(n_estimators=20, max_depth=4, learning_rate=0.1, reg_lambda=1, reg_alpha=0)
    per window on the window's int8 SNPs (the missing code 2 is an ordinary number to LightGBM: "use np.nan for missing encoding" is
    only a comment there) -> per-window model strings (Booster.model_to_string()) + Base.predict_proba; A = 3 (multiclass) and A = 2
    (binary: one tree per round, sigmoid)."""
    if not have_real("lightgbm"):
        print("G18 skipped: lightgbm is not installed (pip install lightgbm)")
        return False
    import lightgbm
    d = dict(lightgbm_version=np.array(lightgbm.__version__), via_reference=os.path.isdir(REF))
    for tag, A in (("m", 3), ("b", 2)):
        C, M, W, ctx, Xt, yt, Xq = _tree_base_data(A, 94318 + A)
        Bq, models = _fit_windows(lambda: lightgbm.LGBMClassifier(n_estimators=20, max_depth=4, learning_rate=0.1, reg_lambda=1, reg_alpha=0,
                                                                  n_jobs=1, random_state=94318), "LGBMBase", C, M, A, ctx, Xt, yt, Xq, 94318)
        strs = [m.booster_.model_to_string() for m in models]
        d.update({tag + "_C": C, tag + "_M": M, tag + "_A": A, tag + "_ctx": ctx, tag + "_X": Xq, tag + "_B": Bq, tag + "_models": np.array(strs)})
        print("G18", tag, "A", A, "B", Bq.shape, Bq.dtype)
    np.savez_compressed(out, **d)
    return True


def make_G19(out):
    """CBBase (src/Base/models.py:68-81): catboost.CatBoostClassifier(n_estimators=20, max_depth=4, reg_lambda=1) per window ->
    per-window JSON exports (save_model(format="json"): oblivious trees, scale and bias) + Base.predict_proba; A = 3 (MultiClass)
    and A = 2 (Logloss)."""
    if not have_real("catboost"):
        print("G19 skipped: catboost is not installed (pip install catboost)")
        return False
    import json
    import tempfile
    import catboost
    d = dict(catboost_version=np.array(catboost.__version__), via_reference=os.path.isdir(REF))
    for tag, A in (("m", 3), ("b", 2)):
        C, M, W, ctx, Xt, yt, Xq = _tree_base_data(A, 94319 + A)
        Bq, models = _fit_windows(lambda: catboost.CatBoostClassifier(n_estimators=20, max_depth=4, reg_lambda=1, thread_count=1, verbose=0),
                                  "CBBase", C, M, A, ctx, Xt, yt, Xq, 94319)
        exports = []
        with tempfile.TemporaryDirectory() as td:
            for w, m in enumerate(models):
                fn = os.path.join(td, "w%d.json" % w)
                m.save_model(fn, format="json")
                exports.append(json.dumps(json.load(open(fn))))
        d.update({tag + "_C": C, tag + "_M": M, tag + "_A": A, tag + "_ctx": ctx, tag + "_X": Xq, tag + "_B": Bq, tag + "_models": np.array(exports)})
        print("G19", tag, "A", A, "B", Bq.shape, Bq.dtype)
    np.savez_compressed(out, **d)
    return True


def make_G15(out):
    """A = 2 logistic base: sklearn keeps ONE coefficient row for a binary LogisticRegression(liblinear) and
    `_predict_proba_lr` returns [1 - expit(z), expit(z)] (no OvR normalisation).  Pins
    gnomix_amd.convert.lr_rows_from_sklearn (the (-coef, +coef) two-row form) against the reference's own
    Base.predict_proba; the raw sklearn arrays are stored next to the converted ones."""
    from src.Base.models import LogisticRegressionBase
    sys.path.insert(0, ROOT)
    from gnomix_amd.convert import lr_rows_from_sklearn
    rng = np.random.RandomState(94315)
    C, M, A = 2237, 100, 2
    W, ctx = C // M, 50
    Xt, yt = synth_admixed(rng, 200, C, A, W, M)
    for w in range(W):
        for a in range(A):
            yt[a, w] = a
    base = LogisticRegressionBase(chm_len=C, window_size=M, num_ancestry=A, missing_encoding=2, context=ctx,
                                  n_jobs=1, seed=94315, verbose=False)
    base.base_multithread = False
    base.train(Xt, yt)
    Xq, _ = synth_admixed(rng, 30, C, A, W, M, miss=0.03, switch_p=0.1)
    B = base.predict_proba(Xq)
    ldc = M + 2 * ctx + (C - M * W)
    coef = np.zeros((W, A, ldc)); icpt = np.zeros((W, A))
    raw_coef = np.zeros((W, 1, ldc)); raw_icpt = np.zeros((W, 1))
    for i, m in enumerate(base.models):
        assert list(m.classes_) == [0, 1] and m.coef_.shape[0] == 1
        c2, b2 = lr_rows_from_sklearn(m.coef_, m.intercept_, A)
        coef[i, :, :c2.shape[1]] = c2
        icpt[i] = b2
        raw_coef[i, :, :m.coef_.shape[1]] = m.coef_
        raw_icpt[i] = m.intercept_
    np.savez_compressed(out, C=C, M=M, A=A, ctx=ctx, X=Xq, coef=coef, intercept=icpt, raw_coef=raw_coef,
                        raw_intercept=raw_icpt, B=B)
    print("G15", B.shape, "min/max P0", B[..., 0].min(), B[..., 0].max())


def make_G16(out):
    """Base.train of the logistic base: the reference's own LogisticRegressionBase.train (sklearn
    LogisticRegression(penalty="l2", C=3., solver="liblinear", max_iter=1000) per window, base.py:104-127) on a small
    problem, A = 3 (one-vs-rest) and A = 2 (sklearn's single-row binary form).  Stored: training data, the fitted
    coefficients in the (W, A, ldc) layout, and Base.predict_proba of held-out haplotypes."""
    from src.Base.models import LogisticRegressionBase
    sys.path.insert(0, ROOT)
    from gnomix_amd.convert import lr_rows_from_sklearn
    d = {}
    for tag, A in (("m", 3), ("b", 2)):
        rng = np.random.RandomState(94316 + A)
        C, M = 1237, 100
        W, ctx = C // M, 50
        Xt, yt = synth_admixed(rng, 160, C, A, W, M, miss=0.02)
        for w in range(W):
            for a in range(A):
                yt[a, w] = a
        base = LogisticRegressionBase(chm_len=C, window_size=M, num_ancestry=A, missing_encoding=2, context=ctx, n_jobs=1,
                                      seed=94316, verbose=False)
        base.base_multithread = False
        base.train(Xt, yt)
        Xq, _ = synth_admixed(rng, 20, C, A, W, M, miss=0.03, switch_p=0.1)
        B = base.predict_proba(Xq)
        ldc = M + 2 * ctx + (C - M * W)
        coef = np.zeros((W, A, ldc)); icpt = np.zeros((W, A))
        for i, m in enumerate(base.models):
            assert list(m.classes_) == list(range(A))
            c2, b2 = lr_rows_from_sklearn(m.coef_, m.intercept_, A)
            coef[i, :, :c2.shape[1]] = c2
            icpt[i] = b2
        d.update({tag + "_C": C, tag + "_M": M, tag + "_A": A, tag + "_ctx": ctx, tag + "_Xt": Xt, tag + "_yt": yt.astype(np.int32),
                  tag + "_coef": coef, tag + "_intercept": icpt, tag + "_Xq": Xq, tag + "_B": B})
        print("G16", tag, "A", A, "coef", coef.shape, "|coef|max", np.abs(coef).max())
    np.savez_compressed(out, **d)


def main():
    which = sys.argv[1:] or ["G1", "G2", "G3", "G4", "G5", "G6", "G7", "G8", "G9", "G10", "G11", "G12", "G13", "G14", "G15", "G16", "G17", "G18", "G19"]
    have_ref = import_reference()
    if not have_ref:
        _stub_modules()
        print("reference not found at", REF, "- only the third-party pins (G12-G14, G18, G19) can be generated")
        which = [w for w in which if w in ("G12", "G13", "G14", "G18", "G19")]
    if "G1" in which: make_G1(os.path.join(HERE, "G1_lr.npz"))
    if "G2" in which: make_G2(os.path.join(HERE, "G2_covrsk.npz"))
    if "G3" in which: make_G3(os.path.join(HERE, "G3_slide.npz"))
    if "G4" in which: make_G4(os.path.join(HERE, "G4_smooth.npz"))
    if "G5" in which: make_G5(os.path.join(HERE, "G5_gnofix.npz"))
    if "G6" in which: make_G6(os.path.join(HERE, "G6_writers"))
    if "G7" in which: make_G7(os.path.join(HERE, "G7_vcf.npz"))
    if "G8" in which: make_G8(os.path.join(HERE, "G8_calib_sk.npz"))
    if "G9" in which: make_G9(os.path.join(HERE, "G9_rf.npz"))
    if "G10" in which: make_G10(os.path.join(HERE, "G10_poly.npz"))
    if "G11" in which: make_G11(os.path.join(HERE, "G11_cnn.npz"))
    if "G12" in which: make_G12(os.path.join(HERE, "G12_xgb_smoother.npz"))
    if "G13" in which: make_G13(os.path.join(HERE, "G13_crf_smoother.npz"))
    if "G14" in which: make_G14(os.path.join(HERE, "G14_xgb_base.npz"))
    if "G15" in which: make_G15(os.path.join(HERE, "G15_lr_binary.npz"))
    if "G16" in which: make_G16(os.path.join(HERE, "G16_lr_train.npz"))
    if "G17" in which: make_G17(os.path.join(HERE, "G17_cnn_train.npz"))
    if "G18" in which: make_G18(os.path.join(HERE, "G18_lgbm_base.npz"))
    if "G19" in which: make_G19(os.path.join(HERE, "G19_catboost_base.npz"))
    return 0


def make_G17(out):
    """CNN.fit (src/Smooth/cnn.py:104-118) as Smoother.train runs it for CNN_Smoother: the reference's own class, optimiser and
    data generator under torch; the layer is constructed with zero padding as in G11.  LAIDataset.__getitem__ is wrapped only to
    RECORD the order in which the shuffling DataLoader visits the rows (300 rows = batches of 128, 128, 44 per epoch)."""
    import torch
    from torch import nn
    real_conv = nn.Conv1d

    class Conv1dOldTorch(real_conv):
        def __init__(self, *a, padding_mode="zeros", **k):
            super().__init__(*a, padding_mode="zeros" if padding_mode == "reflection" else padding_mode, **k)

    from src.Smooth import cnn as ref_cnn
    torch.manual_seed(170017)
    torch.set_num_threads(1)
    A, S, W, N, EP = 4, 9, 50, 300, 12
    nn.Conv1d = Conv1dOldTorch
    try:
        model = ref_cnn.CNN(num_classes=A, num_features=S)
    finally:
        nn.Conv1d = real_conv
    w0 = model.smoothNet[0].weight.detach().numpy().copy()
    b0 = model.smoothNet[0].bias.detach().numpy().copy()
    rng = np.random.RandomState(170017)
    # tract-structured labels and base probabilities that lean towards them: something a smoother can learn
    y = np.zeros((N, W), np.int64)
    for n in range(N):
        pos = 0
        while pos < W:
            ln = rng.randint(4, 20)
            y[n, pos:pos + ln] = rng.randint(A)
            pos += ln
    B = rng.dirichlet(np.ones(A) * 0.6, size=(N, W))
    B[np.arange(N)[:, None], np.arange(W)[None, :], y] += 0.8 * rng.random_sample((N, W))
    B = (B / B.sum(-1, keepdims=True)).astype(np.float32)
    visited = []
    real_get = ref_cnn.LAIDataset.__getitem__

    def logging_get(self, index):
        visited.append(int(index))
        return real_get(self, index)

    ref_cnn.LAIDataset.__getitem__ = logging_get
    try:
        model.fit(B, y, max_ep=EP)
    finally:
        ref_cnn.LAIDataset.__getitem__ = real_get
    order = np.asarray(visited, np.int64).reshape(EP, N)
    assert all(sorted(r) == list(range(N)) for r in order.tolist())
    w1 = model.smoothNet[0].weight.detach().numpy().copy()
    b1 = model.smoothNet[0].bias.detach().numpy().copy()
    model.eval()
    with torch.no_grad():
        proba = model.predict_proba(B[:16])
    np.savez_compressed(out, A=A, S=S, W=W, epochs=EP, B=B, y=y.astype(np.int32), order=order, w0=w0, b0=b0, w1=w1, b1=b1, proba16=proba)
    print("G17", "moved", float(np.abs(w1 - w0).max()), "train acc", float((np.argmax(model.predict_proba(B), -1) == y).mean()))


if __name__ == "__main__":
    sys.exit(main())

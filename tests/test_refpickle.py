"""CPU: reading a reference model.pkl without the reference's packages (gnomix_amd.refpickle) — restricted unpickling and the
xgboost booster-bytes decoder.  xgboost is absent: the byte layouts written here follow the same published 1.1.1 layout the
decoder documents (PARITY UNPINNED for that format); what these tests pin is the decoder's bookkeeping (offsets, wrappers,
default_left bit, tree_info) and the end-to-end conversion of a pickled model whose classes cannot be imported."""
import io
import json
import pickle
import struct
import sys
import types

import numpy as np
import pytest

from gnomix_amd import convert, refpickle


def _legacy_bytes(T, num_feature, magic=True, wrapper=True, default_left=None):
    """xgboost-schema arrays -> legacy binary model bytes (layout in gnomix_amd/refpickle.py's docstring)"""
    nt = len(T["tree_off"]) - 1
    n_class = int(T["tree_class"].max()) + 1 if nt else 1
    out = io.BytesIO()
    if magic:
        out.write(b"binf")
    out.write(struct.pack("<fIiiiII", 0.5, num_feature, n_class, 1, 0, 1, 1) + b"\0" * (27 * 4))
    for s in (b"multi:softprob", b"gbtree"):
        out.write(struct.pack("<Q", len(s)) + s)
    out.write(struct.pack("<iiiiqii", nt, 1, num_feature, 0, 0, n_class, 0) + b"\0" * (32 * 4))
    for t in range(nt):
        o, e = T["tree_off"][t], T["tree_off"][t + 1]
        nn = e - o
        out.write(struct.pack("<6i", 1, nn, 0, 4, num_feature, 0) + b"\0" * (31 * 4))
        parent = np.full(nn, -1, np.int32)
        for k in range(nn):
            if T["left"][o + k] != -1:
                parent[T["left"][o + k]] = k
                parent[T["right"][o + k]] = k
        for k in range(nn):
            leaf = T["left"][o + k] == -1
            sindex = 0 if leaf else int(T["feat"][o + k]) | ((int(default_left[o + k]) if default_left is not None else 0) << 31)
            out.write(struct.pack("<iiiIf", int(parent[k]), int(T["left"][o + k]), int(T["right"][o + k]) if not leaf else 0,
                                  sindex, float(T["cond"][o + k])))
        out.write(b"\0" * (16 * nn))
    out.write(np.asarray(T["tree_class"], "<i4").tobytes())
    out.write(struct.pack("<Q", 1) + struct.pack("<Q", 9) + b"objective" + struct.pack("<Q", 2) + b"{}")  # extra attributes
    model = out.getvalue()
    if not wrapper:
        return model
    return b"CONFIG-offset:" + struct.pack("<q", len(model)) + model + b'{"learner": {"generic_param": {}}}'


def _random_T(oracle, rounds=5, A=3, F=40, seed=1):
    t = oracle.random_trees(rounds, A, F, depth=4, seed=seed)
    return dict(tree_off=t.tree_off, left=t.left, right=t.right, feat=t.feat, cond=t.cond, tree_class=t.tree_class)


@pytest.mark.parametrize("magic,wrapper", [(True, True), (True, False), (False, False), (False, True)])
def test_parse_legacy_binary_roundtrip(oracle, magic, wrapper):
    T = _random_T(oracle)
    dl = (np.random.RandomState(0).random_sample(len(T["left"])) < 0.5).astype(np.uint8)
    got = refpickle.parse_xgb_raw(_legacy_bytes(T, 40, magic, wrapper, dl))
    internal = T["left"] != -1
    for k in ("tree_off", "left", "tree_class"):
        assert np.array_equal(got[k], T[k]), k
    assert np.array_equal(got["right"][internal], T["right"][internal]) and np.all(got["right"][~internal] == -1)
    assert np.array_equal(got["feat"][internal], T["feat"][internal])
    assert np.array_equal(got["cond"], T["cond"])
    assert np.array_equal(got["default_left"][internal], dl[internal]) and not got["default_left"][~internal].any()
    assert got["base_score"] == 0.5 and got["n_class"] == 3 and got["objective"] == "multi:softprob" and got["num_feature"] == 40


def test_parse_json_model(oracle):
    T = _random_T(oracle, rounds=2, A=2, F=9, seed=4)
    trees = []
    for t in range(len(T["tree_off"]) - 1):
        o, e = T["tree_off"][t], T["tree_off"][t + 1]
        trees.append({"left_children": T["left"][o:e].tolist(), "right_children": T["right"][o:e].tolist(),
                      "split_indices": T["feat"][o:e].tolist(), "split_conditions": [float(x) for x in T["cond"][o:e]],
                      "default_left": [1] * (e - o)})
    doc = {"learner": {"learner_model_param": {"base_score": "5E-1", "num_class": "2", "num_feature": "9"},
                       "objective": {"name": "multi:softprob"},
                       "gradient_booster": {"name": "gbtree", "model": {"trees": trees, "tree_info": T["tree_class"].tolist()}}}}
    got = refpickle.parse_xgb_raw(json.dumps(doc).encode())
    assert np.array_equal(got["left"], T["left"]) and np.array_equal(got["cond"], T["cond"]) and got["n_class"] == 2
    assert np.array_equal(got["tree_class"], T["tree_class"])


def test_parser_rejects_garbage():
    with pytest.raises(ValueError):
        refpickle.parse_xgb_raw(b"binf" + b"\x00" * 50)
    with pytest.raises(ValueError):
        refpickle.parse_xgb_raw(b"binf" + b"\xff" * 400)


def test_restricted_unpickler_refuses_code():
    evil = pickle.dumps(eval, protocol=2)
    with pytest.raises(pickle.UnpicklingError):
        refpickle.load_reference_pickle(io.BytesIO(evil))
    import os
    class E:
        def __reduce__(self):
            return (os.system, ("true",))
    with pytest.raises(pickle.UnpicklingError):
        refpickle.load_reference_pickle(io.BytesIO(pickle.dumps(E())))
    # protocol >= 4 walks dotted attribute paths below an allowed root: refused, as is any torch / sklearn
    # global that is not one of the classes the converter reads
    def stack_global(module, name, arg):
        return (b"\x80\x04" + b"\x8c" + bytes([len(module)]) + module.encode() + b"\x8c" + bytes([len(name)]) + name.encode() +
                b"\x93" + b"\x8c" + bytes([len(arg)]) + arg.encode() + b"\x85R.")
    for module, name in [("torch.serialization", "os.path.basename"), ("sklearn.base", "platform.os.path.basename"),
                         ("numpy", "testing._private.utils.os.path.basename"), ("torch.serialization", "load"),
                         ("torch", "load"), ("torch.hub", "load")]:
        with pytest.raises(pickle.UnpicklingError):
            refpickle.load_reference_pickle(io.BytesIO(stack_global(module, name, "/tmp/x")))
    # a sklearn global outside the estimator list is never imported: it becomes a Stub (constructed, not executed)
    got = refpickle.load_reference_pickle(io.BytesIO(stack_global("sklearn.utils", "check_array", "/tmp/x")))
    assert isinstance(got, refpickle.Stub)


def _fake_reference_modules():
    """stand-ins with the reference's module / class NAMES, only to produce a pickle that refers to them"""
    mods = {}
    def mk(modname, *classes):
        m = types.ModuleType(modname)
        for c in classes:
            cls = type(c, (), {"__module__": modname})
            setattr(m, c, cls)
        mods[modname] = m
        return m
    mk("src"); mk("src.Base"); mk("src.Smooth"); mk("xgboost")
    mk("src.model", "Gnomix"); mk("src.Base.models", "LogisticRegressionBase"); mk("src.Smooth.models", "XGB_Smoother")
    mk("xgboost.sklearn", "XGBClassifier"); mk("xgboost.core", "Booster")
    return mods


def test_model_pkl_without_reference_packages(oracle, tmp_path):
    """a pickled Gnomix-shaped object graph (real sklearn LogisticRegression per window, an XGBClassifier whose Booster holds
    raw booster bytes) -> restricted load (no `src`, no `xgboost` importable) -> GnxModelData identical to the source arrays"""
    from sklearn.linear_model import LogisticRegression
    rng = np.random.RandomState(0)
    C, M, A, S, ctx = 457, 50, 3, 5, 25
    W, rem, M_ = C // M, C % M, M + 2 * ctx
    mods = _fake_reference_modules()
    sys.modules.update(mods)
    try:
        lrs = []
        for i in range(W):
            width = M_ + (rem if i == W - 1 else 0)
            Xw = (rng.random_sample((60, width)) < 0.4).astype(np.int8)
            yw = np.arange(60) % A
            lrs.append(LogisticRegression(penalty="l2", C=3.0, solver="liblinear", max_iter=50).fit(Xw, yw))
        base = mods["src.Base.models"].LogisticRegressionBase()
        base.models, base.C, base.M, base.W, base.A, base.context, base.missing_encoding = lrs, C, M, W, A, ctx, 2
        T = _random_T(oracle, rounds=4, A=A, F=S * A, seed=7)
        booster = mods["xgboost.core"].Booster()
        booster.handle, booster.feature_names = bytearray(_legacy_bytes(T, S * A)), None
        clf = mods["xgboost.sklearn"].XGBClassifier()
        clf._Booster, clf.n_classes_, clf.classes_ = booster, A, np.arange(A)
        sm = mods["src.Smooth.models"].XGB_Smoother()
        sm.model, sm.S, sm.W, sm.A, sm.calibrator = clf, S, W, A, None
        g = mods["src.model"].Gnomix()
        g.C, g.M, g.A, g.W, g.context = C, M, A, W, ctx
        g.base, g.smooth = base, sm
        g.snp_pos, g.snp_ref, g.snp_alt = np.arange(C) * 10, np.array(["A"] * C), np.array(["G"] * C)
        g.population_order, g.gen_map_df = np.array(["P0", "P1", "P2"]), None
        path = tmp_path / "model.pkl"
        with open(path, "wb") as f:
            pickle.dump(g, f)
    finally:
        for k in mods:
            sys.modules.pop(k, None)
    assert "src" not in sys.modules and "xgboost" not in sys.modules
    obj = refpickle.load_reference_pickle(str(path))
    assert isinstance(obj, refpickle.Stub) and type(obj).__name__ == "Gnomix" and "xgboost" not in sys.modules
    for use_sklearn in (True, False):
        d = convert.from_reference_model(refpickle.load_reference_pickle(str(path), use_sklearn=use_sklearn))
        assert (d.C, d.M, d.A, d.S, d.context, d.base_kind, d.smooth_kind) == (C, M, A, S, ctx, "logistic", "xgb")
        for i, m in enumerate(lrs):
            assert np.array_equal(d.lr_coef[i, :, :m.coef_.shape[1]], m.coef_) and np.array_equal(d.lr_intercept[i], m.intercept_)
        for k in ("tree_off", "left", "feat", "cond", "tree_class"):
            got, want = getattr(d, k), T[k]
            if k == "feat":
                got, want = got[T["left"] != -1], want[T["left"] != -1]
            assert np.array_equal(got, want), k
        assert d.population_order == ["P0", "P1", "P2"] and len(d.snp_pos) == C


def test_cnn_smoother_pickle_roundtrip(tmp_path):
    """"large" mode: the pickled smoother holds a torch nn.Sequential(nn.Conv1d) — rebuilt by torch, converted to cnn_* arrays"""
    torch = pytest.importorskip("torch")
    from sklearn.linear_model import LogisticRegression
    rng = np.random.RandomState(1)
    C, M, A, S, ctx = 457, 50, 3, 5, 25
    W, rem, M_ = C // M, C % M, M + 2 * ctx
    mods = _fake_reference_modules()
    cnn_mod = types.ModuleType("src.Smooth.cnn")
    cnn_mod.CNN = type("CNN", (), {"__module__": "src.Smooth.cnn"})
    mods["src.Smooth.cnn"] = cnn_mod
    mods["src.Smooth.models"].CNN_Smoother = type("CNN_Smoother", (), {"__module__": "src.Smooth.models"})
    sys.modules.update(mods)
    try:
        lrs = []
        for i in range(W):
            width = M_ + (rem if i == W - 1 else 0)
            lrs.append(LogisticRegression(solver="liblinear").fit((rng.random_sample((40, width)) < 0.4).astype(np.int8), np.arange(40) % A))
        base = mods["src.Base.models"].LogisticRegressionBase()
        base.models = lrs
        net = cnn_mod.CNN()
        conv = torch.nn.Conv1d(A, A, S, padding=(S - 1) // 2)
        conv.padding_mode = "reflection"   # what the reference's constructor asks for (and old torch stored)
        net.smoothNet = torch.nn.Sequential(conv)
        sm = mods["src.Smooth.models"].CNN_Smoother()
        sm.model, sm.S, sm.calibrator = net, S, None
        g = mods["src.model"].Gnomix()
        g.C, g.M, g.A, g.W, g.context, g.base, g.smooth = C, M, A, W, ctx, base, sm
        g.snp_pos = g.snp_ref = g.snp_alt = None
        g.population_order, g.gen_map_df = None, None
        path = tmp_path / "large.pkl"
        with open(path, "wb") as f:
            pickle.dump(g, f)
    finally:
        for k in mods:
            sys.modules.pop(k, None)
    d = convert.from_reference_model(refpickle.load_reference_pickle(str(path)))
    assert d.smooth_kind == "cnn" and d.cnn_weight.shape == (A, A, S) and d.cnn_weight.dtype == np.float32
    assert np.array_equal(d.cnn_weight, conv.weight.detach().numpy()) and np.array_equal(d.cnn_bias, conv.bias.detach().numpy())
    desc, keep = d.to_desc()
    assert desc.cnn_weight and desc.cnn_bias


def test_random_forest_from_stubbed_pickle(oracle, tmp_path):
    """use_sklearn=False: RandomForestClassifier / DecisionTreeClassifier / Tree become attribute bags (Tree state = `nodes`
    structured array + `values`); the converter reads those and reproduces predict_proba through the oracle"""
    from sklearn.ensemble import RandomForestClassifier
    rng = np.random.RandomState(5)
    A, width, n = 3, 40, 80
    X = (rng.random_sample((n, width)) < 0.4).astype(np.int8)
    y = np.arange(n) % A
    rf = RandomForestClassifier(n_estimators=5, max_depth=3, n_jobs=1, random_state=0).fit(X, y)
    p = tmp_path / "rf.pkl"
    with open(p, "wb") as f:
        pickle.dump([rf], f)
    stub = refpickle.load_reference_pickle(str(p), use_sklearn=False)[0]
    assert isinstance(stub, refpickle.Stub) and type(stub).__name__ == "RandomForestClassifier"
    arrs = convert.rforest_from_sklearn([stub], A)
    real = convert.rforest_from_sklearn([rf], A)
    for k in real:
        assert np.array_equal(arrs[k], real[k]), k

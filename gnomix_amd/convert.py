"""Converters: reference objects -> GnxModelData (the export a maintainer runs once per model.pkl).

`from_reference_model` needs the packages the pickle itself needs (sklearn, and xgboost or
sklearn_crfsuite for the smoother): they are only touched through public fitted attributes.
`trees_from_xgb_json` parses xgboost's documented JSON dump (Booster.get_dump(dump_format="json")),
so it is testable without xgboost installed."""
from __future__ import annotations

import json

import numpy as np

from .model import GnxModelData


def cov_sample(width, alpha=0.6, beta=1.0, seed=37):
    """Substring lengths of the covering random string kernel for a window of `width` SNPs
    (reference src/Base/string_kernel.py:80-89: legacy MT19937 stream seeded per call, one uniform draw
    per m = 2..width).  The sequence is prefix-stable in `width`."""
    rs = np.random.RandomState(seed)
    ms = [1]
    for m in range(2, int(width) + 1):
        if (1 - (alpha ** (m - ms[-1] + 1))) * (m ** (-beta)) >= rs.random_sample():
            ms.append(m)
    return np.asarray(ms, dtype=np.int32)


def string_kernel_lengths(width, kernel_name="CovRSK"):
    """Substring lengths the window kernel counts: the covering sample for CovRSKBase, every length 1..width for the
    plain triangular-number kernel of StringKernelBase (string_kernel.py:5-24: K = sum over runs L(L+1)/2)."""
    if "CovRSK" in kernel_name:
        return cov_sample(width)
    if "string_kernel" in kernel_name:
        return np.arange(1, int(width) + 1, dtype=np.int32)
    raise NotImplementedError(f"string kernel {kernel_name}")


def svc_window_from_sklearn(svc, width, kernel_name="CovRSK"):
    """One fitted sklearn.svm.SVC(kernel=callable, probability=True) -> the dict GnxModelData.svc holds."""
    xfit = getattr(svc, "_BaseLibSVM__Xfit")
    d = dict(xfit=np.ascontiguousarray(xfit, dtype=np.int8), support=svc.support_.astype(np.int32),
             dual_coef=np.ascontiguousarray(svc._dual_coef_, dtype=np.float64),
             intercept=np.ascontiguousarray(svc._intercept_, dtype=np.float64),
             prob_a=np.ascontiguousarray(svc._probA, dtype=np.float64),
             prob_b=np.ascontiguousarray(svc._probB, dtype=np.float64),
             n_support=svc._n_support.astype(np.int32))
    if "poly" in kernel_name:   # PolynomialStringKernelBase (models.py:178-193): poly_kernel(X, Y, p=1.2)
        d.update(poly_run_values(width))
    else:
        d["ms"] = string_kernel_lengths(width, kernel_name)
    return d


def poly_run_values(width, p=1.2):
    """what a run of L equal SNPs contributes to the polynomial string kernel (string_kernel.py:52: contigs ** p), computed
    by numpy exactly as the reference computes it (int64 array ** float)"""
    return dict(poly_p=float(p), run_value=np.arange(int(width) + 1, dtype=np.int64) ** p)


def trees_from_xgb_json(dumps, n_class, base_score=0.5):
    """xgboost JSON tree dumps (list of str, one per tree, model order) -> xgboost-schema arrays.
    JSON nodes: {"nodeid", "split": "f12", "split_condition", "yes", "no", "missing", "children"} or {"nodeid","leaf"};
    `yes` is the child taken when feature < split_condition."""
    off, L, R, F, Cd, Dl, cls = [0], [], [], [], [], [], []
    for t, s in enumerate(dumps):
        root = json.loads(s) if isinstance(s, str) else s
        nodes = {}

        def visit(nd):
            nodes[nd["nodeid"]] = nd
            for ch in nd.get("children", []):
                visit(ch)

        visit(root)
        ids = sorted(nodes)
        remap = {nid: k for k, nid in enumerate(ids)}
        if remap[root["nodeid"]] != 0:
            raise ValueError("tree root must have the smallest node id")
        for nid in ids:
            nd = nodes[nid]
            if "leaf" in nd:
                L.append(-1); R.append(-1); F.append(0); Cd.append(np.float32(nd["leaf"])); Dl.append(0)
            else:
                f = nd["split"]
                f = int(f[1:]) if isinstance(f, str) and f.startswith("f") else int(f)
                L.append(remap[nd["yes"]]); R.append(remap[nd["no"]]); F.append(f)
                Cd.append(np.float32(nd["split_condition"]))
                Dl.append(int(nd.get("missing", nd["yes"]) == nd["yes"]))
        off.append(len(L))
        cls.append(t % n_class)
    return dict(tree_off=np.array(off, np.int32), left=np.array(L, np.int32), right=np.array(R, np.int32),
                feat=np.array(F, np.int32), cond=np.array(Cd, np.float32), tree_class=np.array(cls, np.int32),
                base_score=float(base_score), default_left=np.array(Dl, np.uint8))


def xgb_trees_of(xgb_obj, n_class):
    """xgboost-schema arrays of a fitted XGBClassifier.  Always through the booster's own serialised model
    (gnomix_amd.refpickle.parse_xgb_raw): the raw bytes a stubbed pickle carries, or `Booster.save_raw()` of a real
    object — so trees-per-round, tree_info (the class of every tree) and base_score come from the booster itself and are
    never guessed from A (an A == 2 smoother trained as multi:softprob num_class=2, src/Smooth/models.py:14-20, has two
    trees per round; XGBBase with A == 2 is binary:logistic with one)."""
    from .refpickle import booster_bytes, parse_xgb_raw
    raw = booster_bytes(xgb_obj)
    if raw is None:
        booster = xgb_obj.get_booster()
        try:
            raw = booster.save_raw(raw_format="json")   # xgboost >= 1.6
        except TypeError:
            raw = booster.save_raw()                     # xgboost 1.1.1: legacy binary
    t = parse_xgb_raw(bytes(raw))
    n_out = max(int(t["n_class"]), 1)
    if n_out > 1 and n_out != n_class:
        raise ValueError(f"xgboost model: num_class={n_out} but the Gnomix model has A={n_class}")
    if n_out == 1 and n_class != 2:
        raise ValueError("xgboost model: single-output booster for a model with more than 2 ancestries")
    if len(t["tree_class"]) != len(t["tree_off"]) - 1 or (len(t["tree_class"]) and t["tree_class"].max() >= n_out):
        raise ValueError("xgboost model: tree_info does not match the number of classes")
    return {k: t[k] for k in ("tree_off", "left", "right", "feat", "cond", "tree_class", "base_score", "default_left")}


def forest_from_xgb_json(window_dumps, n_class, base_score=0.5, missing=2):
    """Per-window xgboost JSON dumps (XGBBase: one XGBClassifier per window, src/Base/models.py:24-35) -> the fb_*
    arrays of GnxModelData.  multi:softprob lays trees out round-major (tree t -> class t % A); binary:logistic
    (A == 2) has one tree per round."""
    per_round = 1 if n_class == 2 else n_class
    parts = [trees_from_xgb_json(dumps, per_round, base_score) for dumps in window_dumps]
    return forest_from_parts(parts, missing=missing)


def forest_from_parts(parts, missing=2):
    """per-window xgboost-schema dicts (trees_from_xgb_json / refpickle.parse_xgb_raw) -> the fb_* arrays"""
    base_score = parts[0]["base_score"] if parts else 0.5
    wt0 = np.concatenate([[0], np.cumsum([len(p["tree_off"]) - 1 for p in parts])]).astype(np.int32)
    node0 = np.concatenate([[0], np.cumsum([p["tree_off"][-1] for p in parts])])
    return dict(fb_win_tree0=wt0,
                fb_tree_off=np.concatenate([p["tree_off"][:-1] + node0[i] for i, p in enumerate(parts)]
                                           + [[node0[-1]]]).astype(np.int32),
                fb_left=np.concatenate([p["left"] for p in parts]), fb_right=np.concatenate([p["right"] for p in parts]),
                fb_feat=np.concatenate([p["feat"] for p in parts]), fb_cond=np.concatenate([p["cond"] for p in parts]),
                fb_default_left=np.concatenate([p["default_left"] for p in parts]),
                fb_tree_class=np.concatenate([p["tree_class"] for p in parts]),
                fb_base_score=float(base_score), fb_missing=int(missing))


def _int_threshold_as_less_than(thr):
    """SNPs only take the values 0..3, the forest loader asks `x < cond` (xgboost): `x <= thr` of LightGBM / CatBoost is
    `x < floor(thr) + 1` on integers, and floor(thr) + 1 is exact in float32 wherever it matters (clamped to [-1, 5])"""
    return np.float32(min(max(np.floor(float(thr)) + 1.0, -1.0), 5.0))


def _fold_two_outputs(d):
    """A two-output softmax model (LightGBM multiclass with num_class=2, CatBoost MultiClass on two labels) for a two-ancestry
    Gnomix model: the forest kernels treat A == 2 as ONE margin through a sigmoid (xgboost's binary:logistic, k_base_forest.hip),
    and softmax([m0, m1])[1] = sigmoid(m1 - m0) — so class 0's trees enter with their leaves negated and every tree feeds the one
    margin."""
    leaf = d["left"] < 0
    cls_of_node = np.repeat(d["tree_class"], np.diff(d["tree_off"]))
    cond = d["cond"].copy()
    cond[leaf & (cls_of_node == 0)] *= np.float32(-1.0)
    d["cond"] = cond
    d["tree_class"] = np.zeros_like(d["tree_class"])
    return d


def trees_from_lgbm_text(model_str, n_class, missing=2):
    """A LightGBM model string (Booster.model_to_string(), the `handle` a pickled LGBMClassifier's booster carries) ->
    xgboost-schema arrays for the forest base (LGBMBase, src/Base/models.py:38-52: 20 rounds, max_depth 4).

    LightGBM's text format, per tree: `num_leaves`, and for the num_leaves - 1 internal nodes `split_feature`, `threshold`,
    `decision_type`, `left_child`, `right_child` (a child >= 0 is an internal node, a child < 0 is leaf ~child), then `leaf_value`
    per leaf; a numerical node sends `x <= threshold` LEFT.  decision_type: bit 0 categorical, bit 1 default-left, bits 2-3 the
    missing type (0 none, 1 zero, 2 NaN).  The reference feeds int8 SNPs with the missing code 2 as an ordinary number (its own
    comment: "use np.nan for missing encoding" — it does not), so value 2 simply compares: default direction = (2 <= threshold).
    Multiclass: tree t adds to class t % num_tree_per_iteration, probabilities = softmax of the sums (the constant xgboost's
    base_score adds to every class cancels); binary: ONE tree per round, P(class 1) = sigmoid(sigmoid_param * sum).
    The first round's leaves already contain the initial score (boost_from_average)."""
    head, *blocks = model_str.split("\nTree=")
    hp = dict(l.split("=", 1) for l in head.splitlines() if "=" in l)
    per_iter = int(hp.get("num_tree_per_iteration", 1))
    n_out = int(hp.get("num_class", 1))
    if n_out > 1 and n_out != n_class:
        raise ValueError(f"LightGBM model: num_class={n_out} but the Gnomix model has A={n_class}")
    if n_out == 1 and n_class != 2:
        raise ValueError("LightGBM model: single-output booster for a model with more than 2 ancestries")
    if "average_output" in head.split():
        raise NotImplementedError("LightGBM model: average_output (boosting=rf) averages its trees instead of summing them")
    obj = hp.get("objective", "")
    if n_out > 1 and obj.split()[:1] == ["multiclassova"]:
        raise NotImplementedError("LightGBM model: multiclassova (per-class sigmoids, not a softmax)")
    scale = 1.0
    if n_out == 1:
        for tok in obj.split():
            if tok.startswith("sigmoid:"):
                scale = float(tok.split(":", 1)[1])
    off, L, R, F, Cd, Dl, cls = [0], [], [], [], [], [], []
    for t, blk in enumerate(blocks):
        body = blk.split("end of trees")[0]
        kv = dict(l.split("=", 1) for l in body.splitlines() if "=" in l)
        nl = int(kv["num_leaves"])
        if int(kv.get("num_cat", 0)) != 0:
            raise NotImplementedError("LightGBM model: categorical splits (SNPs are numerical features)")
        if int(kv.get("is_linear", 0)) != 0:
            raise NotImplementedError("LightGBM model: linear trees")
        lv = [float(x) for x in kv["leaf_value"].split()]
        if nl == 1:
            L.append(-1); R.append(-1); F.append(0); Cd.append(np.float32(lv[0] * scale)); Dl.append(0)
        else:
            sf = [int(x) for x in kv["split_feature"].split()]
            th = [float(x) for x in kv["threshold"].split()]
            dt = [int(x) for x in kv["decision_type"].split()]
            lc = [int(x) for x in kv["left_child"].split()]
            rc = [int(x) for x in kv["right_child"].split()]
            if not (len(sf) == len(th) == len(dt) == len(lc) == len(rc) == nl - 1 and len(lv) == nl):
                raise ValueError(f"LightGBM model: tree {t} has inconsistent array lengths")
            # node ids of this tree: internal node i -> i, leaf j -> (nl - 1) + j; the root is internal node 0
            nid = lambda c: c if c >= 0 else (nl - 1) + (~c)
            for i in range(nl - 1):
                if dt[i] & 1:
                    raise NotImplementedError("LightGBM model: categorical split")
                if (dt[i] >> 2) & 3 == 1:
                    raise NotImplementedError("LightGBM model: zero_as_missing splits (value 0 would take the default direction)")
                L.append(nid(lc[i])); R.append(nid(rc[i])); F.append(sf[i]); Cd.append(_int_threshold_as_less_than(th[i]))
                Dl.append(int(float(missing) <= th[i]))
            for j in range(nl):
                L.append(-1); R.append(-1); F.append(0); Cd.append(np.float32(lv[j] * scale)); Dl.append(0)
        off.append(len(L))
        cls.append(t % per_iter if n_out > 1 else 0)
    d = dict(tree_off=np.array(off, np.int32), left=np.array(L, np.int32), right=np.array(R, np.int32),
             feat=np.array(F, np.int32), cond=np.array(Cd, np.float32), tree_class=np.array(cls, np.int32),
             base_score=0.5, default_left=np.array(Dl, np.uint8))
    return _fold_two_outputs(d) if n_out == 2 else d


def forest_from_lgbm_text(window_model_strs, n_class, missing=2):
    """Per-window LightGBM model strings (LGBMBase: one LGBMClassifier per window) -> the fb_* arrays of GnxModelData"""
    return forest_from_parts([trees_from_lgbm_text(s, n_class, missing) for s in window_model_strs], missing=missing)


def lgbm_text_of(lgbm_obj):
    """the model string of a fitted LGBMClassifier: from the live booster, or from the attribute bag a stubbed pickle carries
    (lightgbm.Booster.__getstate__ replaces the native handle by model_to_string())"""
    b = getattr(lgbm_obj, "_Booster", None) or getattr(lgbm_obj, "booster_", None)
    if b is None:
        raise ValueError("LGBMClassifier without a fitted booster")
    if hasattr(b, "model_to_string"):
        return b.model_to_string()
    h = getattr(b, "handle", None) or getattr(b, "_handle", None)
    if isinstance(h, bytes):
        h = h.decode()
    if not isinstance(h, str):
        raise ValueError("pickled LightGBM booster without its model string")
    return h


def trees_from_catboost_json(model, n_class, missing=2):
    """A CatBoost model exported with save_model(..., format="json") (dict or str) -> xgboost-schema arrays for the forest base
    (CBBase, src/Base/models.py:68-81: 20 oblivious trees of depth 4).

    An oblivious tree is a list of `splits` (float_feature_index, border) — the SAME split for every node of a level — and
    2^depth leaves; the leaf index has bit i set iff x[feature_i] > border_i.  MultiClass: leaf_values holds n_class numbers per
    leaf (leaf-major), raw_c = scale * sum over trees + bias_c, probabilities = softmax; Logloss (two classes): one number per leaf,
    P(class 1) = sigmoid(raw).  Every oblivious tree becomes one complete binary tree per class (split depth-1 at the root: the
    index's most significant bit first), `x > border` = right; the bias goes into the class's first tree."""
    m = json.loads(model) if isinstance(model, str) else model
    ot = m["oblivious_trees"]
    sb = m.get("scale_and_bias", [1.0, [0.0]])
    scale = float(sb[0])
    bias = sb[1] if isinstance(sb[1], (list, tuple)) else [sb[1]]
    dims = 1
    if ot:
        dims = max(1, len(ot[0]["leaf_values"]) >> len(ot[0].get("splits", [])))
    if dims > 1 and dims != n_class:
        raise ValueError(f"CatBoost model: {dims} output dimensions but the Gnomix model has A={n_class}")
    if dims == 1 and n_class != 2:
        raise ValueError("CatBoost model: single-output model for a model with more than 2 ancestries")
    bias = [float(b) for b in bias] + [0.0] * max(0, dims - len(bias))
    off, L, R, F, Cd, Dl, cls = [0], [], [], [], [], [], []
    for t, tree in enumerate(ot):
        splits = tree.get("splits", [])
        d = len(splits)
        for sp in splits:
            if sp.get("split_type", "FloatFeature") != "FloatFeature":
                raise NotImplementedError("CatBoost model: only float-feature splits (SNPs are numerical features)")
        lv = tree["leaf_values"]
        if len(lv) != (dims << d):
            raise ValueError(f"CatBoost model: tree {t} has {len(lv)} leaf values for depth {d}")
        for c in range(dims):
            base = len(L)
            n_int = (1 << d) - 1
            # heap order inside the tree: internal node h (0-based) at level l tests split d-1-l; children 2h+1, 2h+2
            for h in range(n_int):
                lvl = (h + 1).bit_length() - 1
                sp = splits[d - 1 - lvl]
                L.append(2 * h + 1); R.append(2 * h + 2); F.append(int(sp["float_feature_index"]))
                Cd.append(_int_threshold_as_less_than(sp["border"]))
                Dl.append(int(float(missing) <= float(sp["border"])))
            for leaf in range(1 << d):   # heap leaf order = index with the root's bit as the most significant one
                v = scale * float(lv[leaf * dims + c]) + (bias[c] if t == 0 else 0.0)
                L.append(-1); R.append(-1); F.append(0); Cd.append(np.float32(v)); Dl.append(0)
            assert len(L) - base == n_int + (1 << d)
            off.append(len(L))
            cls.append(c)
    d = dict(tree_off=np.array(off, np.int32), left=np.array(L, np.int32), right=np.array(R, np.int32),
             feat=np.array(F, np.int32), cond=np.array(Cd, np.float32), tree_class=np.array(cls, np.int32),
             base_score=0.5, default_left=np.array(Dl, np.uint8))
    return _fold_two_outputs(d) if dims == 2 else d


def forest_from_catboost_json(window_models, n_class, missing=2):
    """Per-window CatBoost JSON exports (CBBase: one CatBoostClassifier per window) -> the fb_* arrays of GnxModelData"""
    return forest_from_parts([trees_from_catboost_json(m, n_class, missing) for m in window_models], missing=missing)


class _TreeView:
    """sklearn.tree._tree.Tree, real or as the attribute bag a stubbed pickle carries (Tree.__getstate__: `nodes` is a
    structured array with left_child / right_child / feature / threshold, `values` the (n_nodes, n_outputs, n_classes) array)"""

    def __init__(self, t):
        if hasattr(t, "children_left"):
            self.children_left, self.children_right = np.asarray(t.children_left), np.asarray(t.children_right)
            self.feature, self.threshold, self.value = np.asarray(t.feature), np.asarray(t.threshold), np.asarray(t.value)
        else:
            nodes = t.nodes
            self.children_left, self.children_right = np.asarray(nodes["left_child"]), np.asarray(nodes["right_child"])
            self.feature, self.threshold, self.value = np.asarray(nodes["feature"]), np.asarray(nodes["threshold"]), np.asarray(t.values)
        self.node_count = len(self.children_left)


def rforest_from_sklearn(models, n_class):
    """Per-window fitted sklearn RandomForestClassifier (RFBase, src/Base/models.py:54-66) -> the rf_* arrays.
    rf_value[node] is what DecisionTreeClassifier.predict_proba returns for a sample that ends in `node`: tree_.value's
    row, divided by its sum on the scikit-learn versions whose predict_proba still normalises (counts in tree_.value)."""
    import inspect
    from sklearn.tree import DecisionTreeClassifier
    normalise = "normalizer" in inspect.getsource(DecisionTreeClassifier.predict_proba)
    # (a stubbed pickle of an older scikit-learn carries class COUNTS in tree_.value: rows that do not sum to 1 are
    # normalised below exactly as that version's predict_proba did)
    wt0, off, L, R, F, T, V = [0], [0], [], [], [], [], []
    for i, m in enumerate(models):
        if list(m.classes_) != list(range(n_class)):
            raise ValueError(f"window {i}: classes_ != 0..A-1 (the vectorized reference path has no remap, base.py:176)")
        for e in m.estimators_:
            t = _TreeView(e.tree_)
            proba = np.array(t.value[:, 0, :n_class], dtype=np.float64)
            if normalise or not np.allclose(proba.sum(axis=1), 1.0, atol=1e-9):
                normalizer = proba.sum(axis=1)[:, np.newaxis]
                normalizer[normalizer == 0.0] = 1.0
                proba /= normalizer
            L.append(np.asarray(t.children_left, np.int32)); R.append(np.asarray(t.children_right, np.int32))
            F.append(np.where(t.children_left == -1, 0, t.feature).astype(np.int32))
            T.append(np.asarray(t.threshold, np.float64)); V.append(proba)
            off.append(off[-1] + t.node_count)
        wt0.append(len(off) - 1)
    return dict(rf_win_tree0=np.array(wt0, np.int32), rf_tree_off=np.array(off, np.int32), rf_left=np.concatenate(L),
                rf_right=np.concatenate(R), rf_feat=np.concatenate(F), rf_thr=np.concatenate(T), rf_value=np.concatenate(V))


def calibrator_arrays(iso_models):
    """fitted sklearn IsotonicRegression per class (Calibration.py:55) -> calib_off / calib_x / calib_y"""
    off, xs, ys = [0], [], []
    for m in iso_models:
        xs.append(np.asarray(m.X_thresholds_, dtype=np.float64))
        ys.append(np.asarray(m.y_thresholds_, dtype=np.float64))
        off.append(off[-1] + len(xs[-1]))
    return dict(calib_off=np.array(off, np.int32), calib_x=np.concatenate(xs), calib_y=np.concatenate(ys),
                calib_is_f32=bool(np.asarray(iso_models[0].X_thresholds_).dtype == np.float32))


def lr_rows_from_sklearn(coef_, intercept_, n_class):
    """coef_ / intercept_ of one fitted LogisticRegression(solver="liblinear") -> the (A, width) / (A,) rows of the
    one-vs-rest form the kernel evaluates, P_a = expit(z_a) / sum_b expit(z_b).
    A >= 3: sklearn's `_predict_proba_lr` is exactly that form, the arrays pass through.
    A == 2: sklearn keeps ONE row (coef_ (1, n), intercept_ (1,)) and returns [1 - expit(z), expit(z)]; since
    1 - expit(z) = expit(-z) and expit(-z) + expit(z) = 1 the same probabilities are the OvR form of the two rows
    (-coef_, +coef_) / (-b, +b) — copying the single row into both classes would give 0.5 / 0.5 everywhere."""
    coef = np.asarray(coef_, dtype=np.float64)
    icpt = np.asarray(intercept_, dtype=np.float64).reshape(-1)
    if coef.ndim != 2:
        raise ValueError("coef_ must be 2-D")
    if n_class == 2 and coef.shape[0] == 1:
        return np.concatenate([-coef, coef], axis=0), np.array([-icpt[0], icpt[0]])
    if coef.shape[0] != n_class or icpt.shape[0] != n_class:
        raise ValueError(f"coef_ has {coef.shape[0]} rows for {n_class} classes")
    return coef, icpt


def from_reference_model(model) -> GnxModelData:
    """An unpickled reference `src.model.Gnomix` -> GnxModelData."""
    C, M, A = int(model.C), int(model.M), int(model.A)
    d = GnxModelData(C=C, M=M, A=A, S=int(model.smooth.S), context=int(model.context))
    W = C // M
    models = model.base.models
    first = type(models[0]).__name__
    if first == "LogisticRegression":
        ldc = d.M_ + d.rem
        d.base_kind = "logistic"
        d.lr_coef = np.zeros((W, A, ldc))
        d.lr_intercept = np.zeros((W, A))
        for i, m in enumerate(models):
            if list(m.classes_) != list(range(A)):
                raise ValueError(f"window {i}: classes_ != 0..A-1 (the vectorized reference path has no remap, base.py:176)")
            coef, icpt = lr_rows_from_sklearn(m.coef_, m.intercept_, A)
            d.lr_coef[i, :, :coef.shape[1]] = coef
            d.lr_intercept[i] = icpt
    elif first == "SVC":
        d.base_kind = "covrsk"
        kname = getattr(getattr(model.base, "kernel", None), "__name__", "CovRSK")
        d.svc = [svc_window_from_sklearn(m, d.window_width(i), kname) for i, m in enumerate(models)]
    elif first == "XGBClassifier":  # XGBBase (src/Base/models.py:24-35)
        d.base_kind = "forest"
        parts = [xgb_trees_of(m, A) for m in models]
        for k, v in forest_from_parts(parts, missing=int(getattr(model.base, "missing_encoding", 2))).items():
            setattr(d, k, v)
    elif first == "LGBMClassifier":  # LGBMBase (src/Base/models.py:38-52): the boosters' own model strings, no lightgbm needed
        d.base_kind = "forest"
        for k, v in forest_from_lgbm_text([lgbm_text_of(m) for m in models], A,
                                          missing=int(getattr(model.base, "missing_encoding", 2))).items():
            setattr(d, k, v)
    elif first == "CatBoostClassifier":  # CBBase (src/Base/models.py:68-81): through CatBoost's JSON export (needs catboost:
        import json as _json              # its pickle carries the binary model blob)
        import os as _os
        import tempfile
        d.base_kind = "forest"
        exports = []
        for m in models:
            if not hasattr(m, "save_model"):
                raise NotImplementedError("CBBase from a stubbed pickle: export every window with save_model(format='json') and use "
                                          "gnomix_amd.convert.forest_from_catboost_json")
            with tempfile.TemporaryDirectory() as td:
                fn = _os.path.join(td, "m.json")
                m.save_model(fn, format="json")
                exports.append(_json.load(open(fn)))
        for k, v in forest_from_catboost_json(exports, A, missing=int(getattr(model.base, "missing_encoding", 2))).items():
            setattr(d, k, v)
    elif first == "RandomForestClassifier":  # RFBase (src/Base/models.py:54-66)
        d.base_kind = "rforest"
        for k, v in rforest_from_sklearn(models, A).items():
            setattr(d, k, v)
    else:
        raise NotImplementedError(f"base model {first}")
    sm = type(model.smooth).__name__
    if sm == "XGB_Smoother":
        t = xgb_trees_of(model.smooth.model, A)
        if A == 2 and len(t["tree_class"]) and t["tree_class"].max() == 0:
            raise NotImplementedError("binary:logistic smoother (A == 2 trained by xgboost's sklearn wrapper)")
        d.smooth_kind = "xgb"
        for k, v in t.items():
            if k != "default_left":  # smoother features are probabilities, never missing
                setattr(d, k, v)
    elif sm == "CRF_Smoother":
        crf = model.smooth.model.CRF
        d.smooth_kind = "crf"
        d.crf_state = np.zeros((A, A))
        d.crf_trans = np.zeros((A, A))
        for (attr, label), w in crf.state_features_.items():
            d.crf_state[int(attr), int(label)] = w
        for (y0, y1), w in crf.transition_features_.items():
            d.crf_trans[int(y0), int(y1)] = w
    elif sm == "CNN_Smoother":  # mode "large" (src/model.py:65-67): src.Smooth.cnn.CNN holds nn.Sequential(nn.Conv1d(A, A, S))
        conv = model.smooth.model.smoothNet[0]
        d.smooth_kind = "cnn"
        d.cnn_weight = np.ascontiguousarray(conv.weight.detach().cpu().numpy(), dtype=np.float32)
        d.cnn_bias = np.ascontiguousarray(conv.bias.detach().cpu().numpy(), dtype=np.float32)
        if d.cnn_weight.shape != (A, A, d.S):
            raise ValueError(f"CNN smoother: Conv1d weight {d.cnn_weight.shape} != (A, A, S) = ({A}, {A}, {d.S})")
    else:
        raise NotImplementedError(f"smoother {sm}")
    cal = getattr(model.smooth, "calibrator", None)
    if cal is not None and all(mm is not None for mm in cal.models):
        for k, v in calibrator_arrays(cal.models).items():
            setattr(d, k, v)
    d.snp_pos, d.snp_ref, d.snp_alt = model.snp_pos, model.snp_ref, model.snp_alt
    d.population_order = list(model.population_order) if model.population_order is not None else None
    gm = getattr(model, "gen_map_df", None)
    if gm is not None and len(gm):
        d.gen_map_pos, d.gen_map_cm = np.asarray(gm["pos"]), np.asarray(gm["pos_cm"])
    return d

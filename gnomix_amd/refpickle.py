"""Reading a reference `model.pkl` WITHOUT the packages it was pickled with.

A pickled `src.model.Gnomix` (reference gnomix.py:26-35) refers to classes of the reference's `src` package, of
scikit-learn, of xgboost and (CRF models) of sklearn_crfsuite.  `load_reference_pickle` unpickles it with a
restricted `Unpickler`:

 * numpy arrays / dtypes / scalars and a short list of builtins are rebuilt for real;
 * every class or function of `src.*`, `xgboost.*`, `sklearn*`, `sklearn_crfsuite.*`, `pycrfsuite.*`, `scipy.*`
   becomes a `Stub` subclass carrying the pickled attributes (`stub.__dict__`), with the original `__name__` /
   `__module__`, so `gnomix_amd.convert.from_reference_model` duck-types it exactly like the real object
   (`type(m).__name__ == "LogisticRegression"`, `m.coef_`, ...);  scikit-learn classes are rebuilt for real when
   `use_sklearn=True` and scikit-learn imports;
 * torch objects (the CNN smoother of the "large" mode pickles an `nn.Sequential(nn.Conv1d)`) are rebuilt by torch
   through an explicit (module, name) allow-list (`_TORCH_OK`); the tensor bytes inside go through
   `torch.load(..., weights_only=True)`, never through a nested unrestricted pickle;
 * scikit-learn classes are rebuilt for real only when they are one of the estimator classes the converter reads
   (`_SKLEARN_OK`), everything else under `sklearn.*` becomes a `Stub`;
 * dotted names (`module="torch.serialization", name="os.system"`: protocol >= 4 walks attribute paths) are refused
   outright, whatever the module;
 * anything else raises `pickle.UnpicklingError` (a pickle is code: nothing outside the lists above is ever
   imported or called).

`parse_xgb_raw` decodes the booster bytes a pickled `xgboost.Booster` holds in `state["handle"]`
(xgboost/core.py `Booster.__getstate__` -> `XGBoosterSerializeToBuffer`).  xgboost is a third-party dependency that
is absent from /root/reference and from this image (requirements.txt pins 1.1.1): the layout below is RESTATED from
xgboost 1.1.1's published sources (src/learner.cc `LearnerIO::Save/SaveModel`, src/gbm/gbtree_model.{h,cc},
include/xgboost/tree_model.h) and could not be checked against a real booster here — PARITY UNPINNED:

    ["CONFIG-offset:" int64 json_offset]                       serialisation wrapper (1.0 - 1.1), then the model:
    ["binf"]                                                    optional magic
    LearnerModelParamLegacy  136 B: f32 base_score, u32 num_feature, i32 num_class, i32 contain_extra_attrs,
                                    i32 contain_eval_metrics, u32 major, u32 minor, i32 reserved[27]
    u64 len + bytes   objective name;   u64 len + bytes   booster name ("gbtree")
    GBTreeModelParam         160 B: i32 num_trees, i32 x3 (deprecated / pad), i64 deprecated, i32 deprecated,
                                    i32 size_leaf_vector, i32 reserved[32]
    per tree: TreeParam      148 B: i32 deprecated_num_roots, i32 num_nodes, i32 num_deleted, i32 deprecated_max_depth,
                                    i32 num_feature, i32 size_leaf_vector, i32 reserved[31]
              num_nodes x Node        20 B: i32 parent, i32 cleft, i32 cright, u32 sindex (bit 31 = default_left),
                                            f32 leaf_value | split_cond
              num_nodes x NodeStat    16 B: f32 loss_chg, f32 sum_hess, f32 base_weight, i32 leaf_child_cnt
    num_trees x i32 tree_info (output group of every tree)
    ... (attributes, metrics, JSON config: ignored)

A model saved in xgboost's JSON schema (first byte `{`; doc/model.schema) is decoded too.
"""
from __future__ import annotations

import gzip
import io
import json
import pickle
import struct

import numpy as np

_SAFE_BUILTINS = {"set", "frozenset", "list", "dict", "tuple", "bytearray", "bytes", "complex", "slice", "range", "object",
                  "str", "int", "float", "bool"}
_STUB_ROOTS = ("src", "xgboost", "sklearn", "sklearn_crfsuite", "pycrfsuite", "scipy", "joblib", "lightgbm", "catboost")
_NUMPY_OK = {"_reconstruct", "ndarray", "dtype", "scalar", "_frombuffer", "frombuffer", "float64", "float32", "int64",
             "int32", "int8", "uint8", "bool_", "int16", "uint16", "uint32", "uint64", "str_", "bytes_", "object_",
             "RandomState", "__RandomState_ctor", "__randomstate_ctor", "__generator_ctor", "__bit_generator_ctor",
             "MT19937", "Generator", "_pickle"}


class Stub:
    """A pickled object of a class we refuse to import: its attributes, nothing else."""

    def __init__(self, *args, **kwargs):
        if args:
            self._ctor_args = args
        if kwargs:
            self._ctor_kwargs = kwargs

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[0], (dict, type(None))):
            self.__dict__.update(state[0] or {})
            self.__dict__.update(state[1] or {})
        else:
            self._state = state

    def __call__(self, *a, **k):
        raise TypeError(f"{type(self).__module__}.{type(self).__name__} is a placeholder for a pickled object and cannot be called")

    def __repr__(self):
        return f"<stub {type(self).__module__}.{type(self).__name__} {sorted(self.__dict__)[:8]}>"


_stub_cache: dict = {}


def _stub_class(module, name):
    key = (module, name)
    if key not in _stub_cache:
        _stub_cache[key] = type(name, (Stub,), {"__module__": module})
    return _stub_cache[key]


# torch globals a pickled nn.Sequential(nn.Conv1d) refers to (torch 1.4 ... 2.x), nothing else of torch is reachable
_TORCH_OK = {("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
             ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"),
             ("torch.nn.parameter", "Parameter"), ("torch.nn.modules.container", "Sequential"),
             ("torch.nn.modules.conv", "Conv1d"), ("torch.nn.modules.activation", "Softmax"),
             ("torch.nn.modules.activation", "ReLU"), ("torch", "FloatStorage"), ("torch", "DoubleStorage"),
             ("torch", "LongStorage"), ("torch", "Size"), ("torch", "device"), ("torch", "float32"), ("torch", "float64")}
# scikit-learn estimator classes the converter reads fitted attributes of (module path differs between 0.2x and 1.x,
# so the class NAME is matched inside the sklearn package); any other sklearn global becomes a Stub
_SKLEARN_OK = {"LogisticRegression", "SVC", "RandomForestClassifier", "DecisionTreeClassifier", "Tree",
               "IsotonicRegression", "LabelEncoder"}


def _torch_storage_from_bytes(b):
    """stand-in for torch.storage._load_from_bytes (which is torch.load(weights_only=False), i.e. a nested
    UNRESTRICTED pickle of attacker-controlled bytes): the same bytes through torch's own restricted loader"""
    import torch
    return torch.load(io.BytesIO(b), weights_only=True)


class RefUnpickler(pickle.Unpickler):
    def __init__(self, f, use_sklearn=True):
        super().__init__(f)
        self.use_sklearn = use_sklearn

    def find_class(self, module, name):
        if "." in name or not name or not module:
            # protocol >= 4 resolves dotted names attribute by attribute ("os.path.basename" below an allowed module)
            raise pickle.UnpicklingError(f"dotted global {module}.{name} is refused")
        root = module.split(".")[0]
        if root == "numpy":
            if name in _NUMPY_OK:
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"numpy global {module}.{name} is not on the allow-list")
        if module in ("builtins", "__builtin__"):
            if name in _SAFE_BUILTINS:
                return super().find_class("builtins", name)
            raise pickle.UnpicklingError(f"builtin {name} is not on the allow-list")
        if module == "collections" and name in ("OrderedDict", "defaultdict", "deque"):
            return super().find_class(module, name)
        if module == "copyreg" and name == "_reconstructor":
            return super().find_class(module, name)
        if root == "torch":  # the CNN smoother pickles a real torch module: rebuilt by torch itself (tensors, Parameters)
            if (module, name) == ("torch.storage", "_load_from_bytes"):
                return _torch_storage_from_bytes
            if (module, name) not in _TORCH_OK:
                raise pickle.UnpicklingError(f"torch global {module}.{name} is not on the allow-list")
            try:
                return super().find_class(module, name)
            except Exception as e:
                raise pickle.UnpicklingError(f"the pickle holds torch objects and torch is not importable: {e}")
        if root == "sklearn" and self.use_sklearn and name in _SKLEARN_OK:
            try:
                return super().find_class(module, name)
            except Exception:
                return _stub_class(module, name)
        if root in _STUB_ROOTS:
            return _stub_class(module, name)
        raise pickle.UnpicklingError(f"global {module}.{name} is not on the allow-list")


def load_reference_pickle(path_or_file, use_sklearn=True):
    """-> the object graph of a reference model.pkl / model.pkl.gz with `Stub`s for foreign classes"""
    if hasattr(path_or_file, "read"):
        return RefUnpickler(path_or_file, use_sklearn).load()
    opener = gzip.open if str(path_or_file).endswith(".gz") else open
    with opener(path_or_file, "rb") as f:
        return RefUnpickler(f, use_sklearn).load()


# ------------------------------------------------------------------------------------------------------------------
# xgboost booster bytes
# ------------------------------------------------------------------------------------------------------------------
_SER_HEADER = b"CONFIG-offset:"


def _trees_dict(off, L, R, F, Cd, Dl, tree_info, base_score, n_class):
    return dict(tree_off=np.array(off, np.int32), left=np.array(L, np.int32), right=np.array(R, np.int32),
                feat=np.array(F, np.int32), cond=np.array(Cd, np.float32), default_left=np.array(Dl, np.uint8),
                tree_class=np.array(tree_info, np.int32), base_score=float(base_score), n_class=int(n_class))


def _parse_legacy_binary(buf):
    pos = 0
    if buf[:4] == b"binf":
        pos = 4
    if len(buf) < pos + 136:
        raise ValueError("xgboost model: truncated learner parameters")
    base_score, num_feature, num_class, extra_attrs, eval_metrics, major, minor = struct.unpack_from("<fIiiiII", buf, pos)
    pos += 136

    def read_str():
        nonlocal pos
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if n > 4096 or pos + n > len(buf):
            raise ValueError("xgboost model: implausible string length (not the legacy binary layout?)")
        s = bytes(buf[pos:pos + n]).decode("utf-8", "replace")
        pos += n
        return s

    objective = read_str()
    booster = read_str()
    if not booster.startswith("gbtree") and booster != "dart":
        raise ValueError(f"xgboost model: booster {booster!r} is not a tree booster")
    (num_trees,) = struct.unpack_from("<i", buf, pos)
    pos += 160
    if num_trees < 0 or num_trees > 10_000_000:
        raise ValueError("xgboost model: implausible tree count")
    off, L, R, F, Cd, Dl = [0], [], [], [], [], []
    node_dt = np.dtype([("parent", "<i4"), ("cleft", "<i4"), ("cright", "<i4"), ("sindex", "<u4"), ("value", "<f4")])
    for _ in range(num_trees):
        _, num_nodes, num_deleted, _, _, size_leaf_vector = struct.unpack_from("<6i", buf, pos)
        pos += 148
        if num_nodes <= 0 or pos + num_nodes * 36 > len(buf):
            raise ValueError("xgboost model: truncated tree")
        if size_leaf_vector not in (0, 1):
            raise ValueError("xgboost model: vector leaves are not supported")
        nodes = np.frombuffer(buf, dtype=node_dt, count=num_nodes, offset=pos)
        pos += num_nodes * 20 + num_nodes * 16
        leaf = nodes["cleft"] == -1
        L.append(np.where(leaf, -1, nodes["cleft"]).astype(np.int32))
        R.append(np.where(leaf, -1, nodes["cright"]).astype(np.int32))
        F.append(np.where(leaf, 0, nodes["sindex"] & 0x7FFFFFFF).astype(np.int32))
        Cd.append(nodes["value"].astype(np.float32))
        Dl.append(np.where(leaf, 0, nodes["sindex"] >> 31).astype(np.uint8))
        off.append(off[-1] + num_nodes)
    tree_info = np.frombuffer(buf, dtype="<i4", count=num_trees, offset=pos) if num_trees else np.zeros(0, np.int32)
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    d = _trees_dict(off, cat(L, np.int32), cat(R, np.int32), cat(F, np.int32), cat(Cd, np.float32), cat(Dl, np.uint8),
                    tree_info, base_score, max(num_class, 1))
    d["objective"] = objective
    d["num_feature"] = int(num_feature)
    return d


def _parse_json_model(doc):
    learner = doc["learner"] if "learner" in doc else doc["Model"]["learner"]
    lp = learner["learner_model_param"]
    model = learner["gradient_booster"]["model"]
    off, L, R, F, Cd, Dl = [0], [], [], [], [], []
    for t in model["trees"]:
        left = np.asarray(t["left_children"], np.int32)
        leaf = left == -1
        L.append(left)
        R.append(np.asarray(t["right_children"], np.int32))
        F.append(np.where(leaf, 0, np.asarray(t["split_indices"], np.int64)).astype(np.int32))
        Cd.append(np.asarray(t["split_conditions"], np.float32))  # the leaf value at leaves
        Dl.append(np.where(leaf, 0, np.asarray(t["default_left"], np.int64)).astype(np.uint8))
        off.append(off[-1] + len(left))
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    d = _trees_dict(off, cat(L, np.int32), cat(R, np.int32), cat(F, np.int32), cat(Cd, np.float32), cat(Dl, np.uint8),
                    model.get("tree_info", []), float(lp.get("base_score", 0.5)), max(int(lp.get("num_class", 0)), 1))
    d["objective"] = learner.get("objective", {}).get("name", "")
    d["num_feature"] = int(lp.get("num_feature", 0))
    return d


def parse_xgb_raw(raw):
    """Booster bytes (bytes / bytearray / memoryview) -> xgboost-schema arrays: tree_off, left, right, feat, cond,
    default_left, tree_class (= tree_info), base_score, n_class, objective, num_feature."""
    buf = bytes(raw)
    if buf[:len(_SER_HEADER)] == _SER_HEADER:  # serialisation wrapper: model bytes, then the JSON configuration
        (json_offset,) = struct.unpack_from("<q", buf, len(_SER_HEADER))
        start = len(_SER_HEADER) + 8
        buf = buf[start:start + json_offset] if 0 < json_offset <= len(buf) - start else buf[start:]
    head = buf.lstrip()[:1]
    if head == b"{":
        return _parse_json_model(json.loads(buf.decode("utf-8")))
    return _parse_legacy_binary(buf)


def booster_bytes(xgb_obj):
    """The raw model held by a (stubbed) XGBClassifier / Booster, or None"""
    b = getattr(xgb_obj, "_Booster", xgb_obj)
    h = getattr(b, "handle", None)
    if isinstance(h, (bytes, bytearray, memoryview)):
        return bytes(h)
    return None

"""CPU-side tests (-m "not gpu"): the C-ABI library loads and exports every symbol the header
declares, the host mirror validates its inputs, the .gnx container round-trips, and the product
package never reaches into oracle/."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from gnomix_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gnomix_hip.h")).read()
    declared = set(re.findall(r"\b(gnx_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"gnx_ctx", "gnx_model"}
    assert declared, "no prototypes parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    assert lib.gnx_abi_version() == _lib.GNX_ABI_VERSION
    assert isinstance(lib, ctypes.CDLL)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import gnomix_amd
    with pytest.raises(gnomix_amd.GnxError):
        gnomix_amd.Context(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gnomix_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "gnx_oracle" not in src and "libgnx_oracle" not in src, f


def test_gnx_roundtrip(tmp_path):
    from gnomix_amd import synth, GnxModelData
    d = synth.synthetic_model(C=1237, M=50, A=5, S=11, n_rounds=3, seed=1)
    d.snp_pos = np.arange(1237) * 100
    p = tmp_path / "m.gnx"
    d.save(str(p))
    e = GnxModelData.load(str(p))
    assert (e.C, e.M, e.A, e.S, e.context, e.base_kind, e.smooth_kind) == (1237, 50, 5, 11, 25, "logistic", "xgb")
    assert np.array_equal(e.lr_coef, d.lr_coef) and np.array_equal(e.cond, d.cond)
    assert e.population_order == d.population_order and np.array_equal(e.snp_pos, d.snp_pos)
    assert e.W == 24 and e.rem == 37 and e.window_width(23) == 100 + 37


def test_desc_validation():
    from gnomix_amd import synth
    d = synth.synthetic_model(C=1237, M=50, A=5, S=11, n_rounds=2)
    desc, keep = d.to_desc()
    assert desc.C == 1237 and desc.lr_ldc == 50 + 50 + 37 and desc.n_trees == 10
    d.lr_coef = d.lr_coef[:-1]
    with pytest.raises(ValueError):
        d.to_desc()


def test_synthetic_trees_match_oracle_schema(oracle):
    from gnomix_amd import synth
    t = synth.synthetic_trees(4, 3, 33, seed=1)
    T = oracle.Trees(t["tree_off"], t["left"], t["right"], t["feat"], t["cond"], t["tree_class"], 3)
    f = np.random.RandomState(0).uniform(size=(20, 33)).astype(np.float32)
    p = oracle.xgb_predict_proba(T, f)
    assert p.shape == (20, 3) and np.allclose(p.sum(1), 1, atol=1e-6)

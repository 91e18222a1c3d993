"""CPU-side tests (-m "not gpu"): the C-ABI library loads and exports every symbol the header
declares, the host mirror validates its inputs, the .gnx container round-trips, and the product
package never reaches into oracle/."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from gnomix_amd import _lib
    hdr = "".join(open(os.path.join(ROOT, "include", f)).read() for f in sorted(os.listdir(os.path.join(ROOT, "include")))
                  if f.endswith(".h"))
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)   # prototypes only: comments mention entry points in prose
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


def test_cov_sample_matches_oracle_and_anchors(oracle):
    from gnomix_amd import convert
    for m in (8, 20, 349, 499, 2000, 2500):
        assert list(convert.cov_sample(m)) == oracle.cov_sample(m)
    assert list(convert.cov_sample(349)) == [1, 4, 8, 39, 42, 117]


def test_trees_from_xgb_json_roundtrip(oracle):
    """hand-written JSON in xgboost's dump format -> arrays -> same predictions as the arrays' own walk"""
    import json
    from gnomix_amd import convert
    t0 = {"nodeid": 0, "depth": 0, "split": "f3", "split_condition": 0.25, "yes": 1, "no": 2, "missing": 1, "children": [
        {"nodeid": 1, "leaf": -0.1},
        {"nodeid": 2, "depth": 1, "split": "f0", "split_condition": 0.5, "yes": 3, "no": 4, "missing": 3, "children": [
            {"nodeid": 3, "leaf": 0.2}, {"nodeid": 4, "leaf": 0.3}]}]}
    t1 = {"nodeid": 0, "leaf": 0.05}
    t = convert.trees_from_xgb_json([json.dumps(t0), json.dumps(t1)], n_class=2)
    assert t["left"].tolist() == [1, -1, 3, -1, -1, -1] and t["feat"].tolist()[:3] == [3, 0, 0]
    T = oracle.Trees(t["tree_off"], t["left"], t["right"], t["feat"], t["cond"], t["tree_class"], 2)
    f = np.array([[0.6, 0, 0, 0.3], [0.1, 0, 0, 0.9], [0.9, 0, 0, 0.1]], dtype=np.float32)
    m = np.log(oracle.xgb_predict_proba(T, f))
    want = np.array([0.3, 0.2, -0.1]) - 0.05  # margin difference class0 - class1
    assert np.allclose(m[:, 0] - m[:, 1], want, atol=1e-6)


def test_writers_match_reference_bytes(tmp_path):
    """G6: .msp / .fb text produced by the reference's own writers on the same inputs"""
    from gnomix_amd import postprocess as pp
    g = np.load(os.path.join(ROOT, "tests", "golden", "G6_writers", "inputs.npz"), allow_pickle=False)
    meta = pp.get_meta_data("22", g["model_pos"], g["query_pos"], int(g["W"]), int(g["M"]), g["gm_pos"], g["gm_cm"])
    out = str(tmp_path / "ours")
    pp.write_msp(out, meta, g["labels"], list(g["pops"]), list(g["samples"]))
    pp.write_fb(out, meta, g["proba"], list(g["pops"]), list(g["samples"]))
    ref = os.path.join(ROOT, "tests", "golden", "G6_writers", "ref")
    assert open(out + ".msp").read() == open(ref + ".msp").read()
    assert open(out + ".fb").read() == open(ref + ".fb").read()
    samples, snp = pp.msp_to_lai(out + ".msp", g["query_pos"], str(tmp_path / "o.lai"))
    assert snp.shape == (len(g["query_pos"]), 4) and samples[0] == "HG001.0"
    pp.msp_to_bed(out + ".msp", str(tmp_path / "bed"), pop_order=list(g["pops"]))
    assert os.path.exists(str(tmp_path / "bed" / "HG001_0.bed"))


def test_vcf_to_npy_matches_reference():
    """G7: the reference's own vcf_to_npy output on the same scikit-allel-style dict"""
    from gnomix_amd import vcfio
    g = np.load(os.path.join(ROOT, "tests", "golden", "G7_vcf.npz"), allow_pickle=False)
    vd = {"calldata/GT": g["gt"].copy(), "variants/POS": g["vcf_pos"], "variants/REF": g["vcf_ref"]}
    X, vi, fi = vcfio.vcf_to_npy(vd, g["model_pos"], g["model_ref"], return_idx=True, verbose=False)
    assert X.dtype == np.int8 and np.array_equal(X, g["X"])
    assert np.array_equal(vi, g["vcf_idx"]) and np.array_equal(fi, g["fmt_idx"])
    assert (X == 2).any() and set(np.unique(X)) <= {0, 1, 2}
    X2 = vcfio.vcf_to_npy({"calldata/GT": g["gt"].copy(), "variants/POS": g["vcf_pos"], "variants/REF": g["vcf_ref"]}, verbose=False)
    assert np.array_equal(X2, g["X_nofmt"])


def test_vcf_text_roundtrip(tmp_path):
    from gnomix_amd import vcfio
    p = tmp_path / "q.vcf"
    lines = ["##fileformat=VCFv4.2", "##contig=<ID=22>", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2",
             "22\t100\trs1\tA\tG\t.\tPASS\t.\tGT\t0|1\t1|1", "22\t150\trs2\tC\tT,G\t50\tPASS\t.\tGT:DP\t.|0:3\t2|0:4",
             "21\t10\trs0\tA\tC\t.\tPASS\t.\tGT\t0|0\t0|1"]
    p.write_text("\n".join(lines) + "\n")
    d = vcfio.read_vcf(str(p), chm="22")
    assert d["calldata/GT"].shape == (2, 2, 2) and list(d["samples"]) == ["S1", "S2"]
    assert d["calldata/GT"][1, 0, 0] == -1 and d["calldata/GT"][1, 1, 0] == 2 and list(d["variants/POS"]) == [100, 150]
    X = vcfio.vcf_to_npy(d, verbose=False)
    assert X.tolist() == [[0, 2], [1, 0], [1, 2], [1, 0]]
    assert vcfio.read_headers(str(p)).count("##") == 2
    out = vcfio.npy_to_vcf(d, X, str(tmp_path / "phased"), headers=vcfio.read_headers(str(p)))
    d2 = vcfio.read_vcf(out, chm="22")
    assert np.array_equal(vcfio.vcf_to_npy(d2, verbose=False), X)
    assert vcfio.read_vcf(str(p), chm="7")["calldata/GT"].shape[0] == 3  # unknown region -> whole file (utils.py:72-78)


def test_cli_usage_and_training_mode_messages(capsys):
    from gnomix_amd import cli
    assert cli.main(["gnomix.py"]) == 0
    assert "Usage when using a pre-trained model" in capsys.readouterr().out
    assert cli.main(["gnomix.py", "a", "b"]) == 0
    assert "Incorrect number of arguments" in capsys.readouterr().out
    assert cli.main(["gnomix.py"] + ["x"] * 7) == 2


def test_forest_model_gnx_roundtrip_and_desc(tmp_path):
    """forest-base arrays survive the .gnx container and fill the gnx_model_desc fields the header declares"""
    from gnomix_amd import synth, GnxModelData, _lib
    d = synth.synthetic_forest_model(1237, 100, 3, n_rounds=2, depth=3, seed=4)
    p = tmp_path / "f.gnx"
    d.save(p)
    e = GnxModelData.load(p)
    assert e.base_kind == "forest" and e.fb_missing == 2 and e.fb_base_score == 0.5
    for k in ("fb_win_tree0", "fb_tree_off", "fb_left", "fb_right", "fb_feat", "fb_cond", "fb_default_left", "fb_tree_class"):
        assert np.array_equal(getattr(d, k), getattr(e, k)), k
    desc, keep = e.to_desc()
    assert desc.base_kind == _lib.BASE_FOREST and desc.fb_n_trees == len(d.fb_tree_off) - 1
    assert desc.fb_win_tree0 and desc.fb_default_left and desc.fb_tree_class


def test_forest_from_xgb_json(oracle):
    """per-window JSON dumps -> fb_* arrays; the dump's "missing" child becomes the default direction"""
    import json
    from gnomix_amd import convert
    w0 = [{"nodeid": 0, "split": "f1", "split_condition": 0.5, "yes": 1, "no": 2, "missing": 2, "children": [
        {"nodeid": 1, "leaf": 0.4}, {"nodeid": 2, "leaf": -0.2}]}]
    w1 = [{"nodeid": 0, "split": "f3", "split_condition": 1.5, "yes": 1, "no": 2, "missing": 1, "children": [
        {"nodeid": 1, "leaf": 0.1}, {"nodeid": 2, "leaf": 0.6}]}, {"nodeid": 0, "leaf": -0.05}]
    f = convert.forest_from_xgb_json([[json.dumps(t) for t in w0], [json.dumps(t) for t in w1]], n_class=2)
    assert f["fb_win_tree0"].tolist() == [0, 1, 3] and f["fb_tree_off"].tolist() == [0, 3, 6, 7]
    assert f["fb_default_left"].tolist() == [0, 0, 0, 1, 0, 0, 0]
    T = oracle.Trees(f["fb_tree_off"], f["fb_left"], f["fb_right"], f["fb_feat"], f["fb_cond"], f["fb_tree_class"], 2,
                     default_left=f["fb_default_left"])
    X = np.array([[0, 2, 1, 2, 0]], np.int8)            # C=5, M=2, ctx=1: padded 0 0 2 1 2 0 0
    B = oracle.base_forest(T, f["fb_win_tree0"], X, 2, 1, 2)
    # window 0 = padded[0:4] = 0 0 2 1: SNP1 = 0 < 0.5 -> 0.4;  window 1 = padded[2:7] = 2 1 2 0 0: SNP3 = 0 < 1.5 -> 0.1 - 0.05
    assert np.allclose(B[0, :, 1], 1 / (1 + np.exp(-np.array([0.4, 0.05]))), atol=1e-6)
    X[0, 0] = 2                                          # padded 2 2 2 1 ...: window 0 SNP1 missing -> "no" child (-0.2)
    B = oracle.base_forest(T, f["fb_win_tree0"], X, 2, 1, 2)
    assert np.allclose(B[0, 0, 1], 1 / (1 + np.exp(0.2)), atol=1e-6)


def test_rforest_converter_against_sklearn(oracle):
    """rforest_from_sklearn + the oracle's RF walk == RandomForestClassifier.predict_proba on every window (bit-exact)"""
    from sklearn.ensemble import RandomForestClassifier
    from gnomix_amd import convert
    rng = np.random.RandomState(3)
    C, M, ctx, A, N = 457, 50, 25, 4, 60
    W, rem, M_ = C // M, C % M, M + 2 * ctx
    X = (rng.random_sample((200, C)) < 0.4).astype(np.int8)
    X[rng.random_sample(X.shape) < 0.05] = 2
    y = rng.randint(0, A, size=(200, W))
    y[:A] = np.arange(A)[:, None]
    Xp = np.concatenate([X[:, :ctx][:, ::-1], X, X[:, -ctx:][:, ::-1]], axis=1)   # Base.pad (base.py:41-44)
    models, ref = [], []
    for i in range(W):
        sl = slice(i * M, i * M + M_) if i < W - 1 else slice(Xp.shape[1] - (M_ + rem), Xp.shape[1])
        m = RandomForestClassifier(n_estimators=7, max_depth=4, n_jobs=1, random_state=i).fit(Xp[:, sl], y[:, i])
        models.append(m)
        ref.append(m.predict_proba(Xp[:N, sl]))
    rf = convert.rforest_from_sklearn(models, A)
    B = oracle.base_rforest({k[3:]: v for k, v in rf.items()}, X[:N], M, ctx, A)
    assert np.array_equal(B, np.swapaxes(np.array(ref), 0, 1))


def test_graft_entry_build_passes():
    """the driver's "does it build" hook: compiles (no-op when up to date), loads the library, checks the ABI version"""
    import __graft_entry__ as g
    g.build()


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """compile a C probe against include/gnomix_hip.h and compare sizeof / offsetof of every struct field with the
    ctypes mirrors in gnomix_amd/_lib.py (a drifting field would corrupt every model load silently)"""
    import ctypes as C
    import subprocess
    from gnomix_amd import _lib
    structs = {"gnx_model_desc": _lib.ModelDesc, "gnx_svc_window": _lib.SvcWindow, "gnx_model_info": _lib.ModelInfo,
               "gnx_vcf_info": _lib.VcfInfo, "gnx_train_info": _lib.TrainInfo, "gnx_cnn_params": _lib.CnnParams,
               "gnx_crf_params": _lib.CrfParams, "gnx_crf_info": _lib.CrfInfo}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "gnomix_hip.h"', '#include "gnomix_io.h"', 'int main(void) {']
    for cname, ct in structs.items():
        src.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            src.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src += ['  return 0;', '}']
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, ct in structs.items():
        assert int(out[cname]) == C.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"


def test_bench_refuses_gpu_count_it_cannot_have():
    """`python bench.py --gpus N` spawns its own N ranks or refuses: it must never print a line for a smaller world
    (--gpus 8 silently ran 1 GPU).  No GPU here -> any N >= 2 is refused before anything is launched."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2 and "refusing to run" in r.stderr
    assert not any(line.startswith("{") for line in r.stdout.splitlines())
    env = dict(os.environ, RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


def test_pack_x_roundtrip_and_validation():
    """gnx_pack_x is a host-only utility (no context, no GPU): 4 SNPs per byte, SNP j in bits 2*(j%4).. of byte j/4"""
    import gnomix_amd
    lib = gnomix_amd.load_library()
    rng = np.random.RandomState(0)
    for Cn in (1, 3, 4, 5, 16, 17, 63, 64, 1001, 37_053):
        N = 37
        X = rng.randint(0, 3, size=(N, Cn)).astype(np.int8)
        ldp = int(lib.gnx_packed_row_bytes(Cn))
        assert ldp % 4 == 0 and ldp >= (Cn + 3) // 4
        P = np.full((N, ldp), 255, np.uint8)
        for threads in (1, 3):
            assert lib.gnx_pack_x(X.ctypes.data, N, Cn, Cn, P.ctypes.data, ldp, threads) == 0
            U = np.zeros((N, ldp * 4), np.int8)
            for k in range(4):
                U[:, k::4] = (P >> (2 * k)) & 3
            assert np.array_equal(U[:, :Cn], X) and not U[:, Cn:].any()
    Xs = np.zeros((4, 40), np.int8)                      # strided input rows (ldx > C)
    Xs[:, :33] = rng.randint(0, 3, size=(4, 33))
    P = np.zeros((4, 12), np.uint8)
    assert lib.gnx_pack_x(Xs.ctypes.data, 4, 40, 33, P.ctypes.data, 12, 1) == 0
    assert ((P[:, 8] >> 0) & 3).tolist() == Xs[:, 32].tolist()
    X[5, 7] = 4                                           # not representable in 2 bits
    assert lib.gnx_pack_x(X.ctypes.data, N, Cn, Cn, np.zeros((N, ldp), np.uint8).ctypes.data, ldp, 2) == gnomix_amd._lib.GNX_EINVAL
    X[5, 7] = -1
    assert lib.gnx_pack_x(X.ctypes.data, N, Cn, Cn, np.zeros((N, ldp), np.uint8).ctypes.data, ldp, 2) == gnomix_amd._lib.GNX_EINVAL
    assert lib.gnx_pack_x(X.ctypes.data, N, Cn, Cn, P.ctypes.data, 3, 1) == gnomix_amd._lib.GNX_EINVAL   # ldp too small


def test_metrics_equal_sklearn():
    """the scores and the confusion matrix of the reference's evaluate() / conf_matrix() (sklearn.metrics) in numpy"""
    skm = pytest.importorskip("sklearn.metrics")
    import warnings
    from gnomix_amd.metrics import accuracy_pair, confusion
    rng = np.random.RandomState(0)
    for _ in range(10):
        y = rng.randint(0, 5, 800)
        yp = np.where(rng.rand(800) < 0.7, y, rng.randint(0, 6, 800))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = (round(skm.accuracy_score(y, yp) * 100, 2), round(skm.balanced_accuracy_score(y, yp) * 100, 2))
            cm = skm.confusion_matrix(y, yp)
        assert accuracy_pair(y.reshape(20, 40), yp.reshape(20, 40)) == ref
        got, labels = confusion(y, yp)
        assert np.array_equal(got, cm) and labels == sorted(set(y) | set(yp))


def test_isotonic_fit_equals_the_references_calibrator():
    """gnx_fit_isotonic_f32 (host arithmetic in the library) vs golden G8: the thresholds the REFERENCE's Calibrator.fit produced
    on the same (regenerated) inputs, bit for bit; plus random problems against scikit-learn when it is installed"""
    from conftest import load_golden
    from gnomix_amd import calibrate
    g = load_golden("G8_calib_sk.npz")
    rng = np.random.RandomState(8)                       # make_golden.py: make_G8 draws these first
    A = int(g["A"])
    proba_fit = rng.dirichlet(np.ones(A) * 0.5, size=3000).astype(np.float32)
    y = np.array([rng.choice(A, p=p / p.sum()) for p in proba_fit.astype(np.float64) ** 0.7])
    d = calibrate.fit_calibrator(proba_fit, y, A)
    assert d["calib_is_f32"] is True
    for i in range(A):
        sl = slice(d["calib_off"][i], d["calib_off"][i + 1])
        assert np.array_equal(d["calib_x"][sl], g["x%d" % i]) and np.array_equal(d["calib_y"][sl], g["y%d" % i]), i
    with pytest.raises(ValueError):
        calibrate.fit_calibrator(proba_fit, np.zeros_like(y), A)
    iso = pytest.importorskip("sklearn.isotonic")
    for seed in range(12):
        r = np.random.RandomState(seed)
        n = int(r.randint(1, 400))
        x = r.beta(0.3, 0.3, size=n).astype(np.float32)
        if seed % 3 == 0:
            x = np.round(x, 2)                             # heavy ties
        if seed % 4 == 0:
            x[: n // 2] += np.float32(3e-7) * r.randint(0, 3, size=n // 2)   # values closer than float32's resolution
        t = (r.rand(n) < x).astype(np.float32)
        m = iso.IsotonicRegression(out_of_bounds="clip").fit(x, t)
        xt, yt = calibrate.fit_isotonic(x, t)
        assert np.array_equal(xt, m.X_thresholds_) and np.array_equal(yt, m.y_thresholds_), seed


def test_no_default_dispatch_kernel_carries_scratch():
    """scripts/scratch_audit.py --check: the code objects' metadata of libgnomix_hip.so (no GPU needed) must show
    .private_segment_fixed_size == 0 for every kernel a default dispatch reaches — a kernel with scratch is a correctness-only
    kernel (the forest bases, k_crf_psi, k_calibrate and the generic CovRSK instances spilled)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "scratch_audit.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 of those on a default dispatch" in r.stdout


# ---------------------------------------------------------------- LGBMBase / CBBase: third-party tree formats -> forest base ----
def _lgbm_model_string(rng, n_feat, n_class, rounds, max_leaves=16, softmax2=False):
    """a LightGBM model string with random leaf-wise trees (the text layout of Booster.model_to_string()), and the same trees as
    Python structures for the direct evaluator below (softmax2: two classes trained with objective=multiclass, num_class=2)"""
    binary = n_class == 2 and not softmax2
    per_iter = 1 if binary else n_class
    head = ["tree", "version=v3", "num_class=%d" % (1 if binary else n_class), "num_tree_per_iteration=%d" % per_iter,
            "label_index=0", "max_feature_idx=%d" % (n_feat - 1),
            "objective=" + ("binary sigmoid:1" if binary else "multiclass num_class:%d" % n_class),
            "feature_names=" + " ".join("Column_%d" % i for i in range(n_feat)), "feature_infos=" + " ".join("[0:2]" for _ in range(n_feat)), ""]
    out, trees = ["\n".join(head)], []
    for t in range(rounds * per_iter):
        nl = int(rng.randint(1, max_leaves + 1))
        # grow leaf-wise: split a random leaf until nl leaves; LightGBM numbers internal nodes in creation order and keeps the
        # split leaf's index for the left child, the new leaf index for the right one
        depth_of = {0: 0}
        parent_slot = {}                       # leaf -> (internal node, "l" / "r")
        sf, th, dt, lc, rc = [], [], [], [], []
        for i in range(nl - 1):
            cand = [l for l, d in depth_of.items() if d < 4]
            if not cand:
                nl = i + 1
                break
            leaf = cand[rng.randint(len(cand))]
            new_leaf = len(depth_of)
            sf.append(int(rng.randint(n_feat))); th.append(float(rng.choice([0.5, 1.5, 1.0000000180025095e-35, 2.5, 0.9999])))
            dt.append(int(rng.choice([2, 0, 8, 10])))          # default-left / none, missing type none / NaN
            lc.append(~leaf); rc.append(~new_leaf)
            if leaf in parent_slot:
                p, side = parent_slot[leaf]
                (lc if side == "l" else rc)[p] = i
            parent_slot[leaf] = (i, "l"); parent_slot[new_leaf] = (i, "r")
            depth_of[new_leaf] = depth_of[leaf] = depth_of[leaf] + 1
        lv = [float(np.float32(rng.randn() * 0.3)) for _ in range(nl)]
        blk = ["Tree=%d" % t, "num_leaves=%d" % nl, "num_cat=0"]
        if nl > 1:
            blk += ["split_feature=" + " ".join(map(str, sf)), "split_gain=" + " ".join("1" for _ in sf),
                    "threshold=" + " ".join(repr(x) for x in th), "decision_type=" + " ".join(map(str, dt)),
                    "left_child=" + " ".join(map(str, lc)), "right_child=" + " ".join(map(str, rc))]
        blk += ["leaf_value=" + " ".join(repr(x) for x in lv), "is_linear=0", "shrinkage=0.1", "", ""]
        out.append("\n".join(blk))
        trees.append((sf, th, lc, rc, lv))
    out.append("end of trees\n\nfeature_importances:\n\nparameters:\n[boosting: gbdt]\nend of parameters\n\npandas_categorical:null\n")
    return "\n".join(out), trees, per_iter


def _lgbm_predict(trees, per_iter, n_class, x):
    """LightGBM's own rule on the structures above: x <= threshold -> left child; a negative child c is leaf ~c"""
    raw = np.zeros(per_iter)
    for t, (sf, th, lc, rc, lv) in enumerate(trees):
        if not sf:
            raw[t % per_iter] += lv[0]
            continue
        node = 0
        while node >= 0:
            node = lc[node] if float(x[sf[node]]) <= th[node] else rc[node]
        raw[t % per_iter] += lv[~node]
    if n_class == 2 and per_iter == 1:
        p1 = 1.0 / (1.0 + np.exp(-raw[0]))
        return np.array([1 - p1, p1])
    e = np.exp(raw - raw.max())
    return e / e.sum()


@pytest.mark.parametrize("A", [2, 3, 7, -2])
def test_forest_from_lgbm_text(oracle, A):
    """LGBMBase (src/Base/models.py:38-52): LightGBM model strings -> fb_* arrays; the oracle's forest on them == LightGBM's own
    prediction rule evaluated directly on the strings' trees (x <= threshold left, leaf ~child, class = tree % num_tree_per_iteration).
    Unpinned to lightgbm itself (absent from this image): the format is restated from its documented text layout."""
    from gnomix_amd import convert
    softmax2 = A < 0          # -2: a two-class model trained as multiclass (two trees per round, softmax) — folded into one margin
    A = abs(A)
    rng = np.random.RandomState(100 + A + 50 * softmax2)
    M, ctx, W = 6, 2, 4
    C = M * W + 1
    strs, structs = [], []
    for w in range(W):
        width = M + 2 * ctx + (1 if w == W - 1 else 0)
        s, trees, per_iter = _lgbm_model_string(rng, width, A, rounds=5, softmax2=softmax2)
        strs.append(s); structs.append((trees, per_iter))
    f = convert.forest_from_lgbm_text(strs, A)
    T = oracle.Trees(f["fb_tree_off"], f["fb_left"], f["fb_right"], f["fb_feat"], f["fb_cond"], f["fb_tree_class"], max(A, 2),
                     default_left=f["fb_default_left"])
    X = rng.randint(0, 3, size=(40, C)).astype(np.int8)
    B = oracle.base_forest(T, f["fb_win_tree0"], X, M, ctx, A, missing=2)
    Xp = np.concatenate([X[:, :ctx][:, ::-1], X, X[:, -ctx:][:, ::-1]], axis=1)     # Base.pad (base.py:41-44)
    for w in range(W):
        width = M + 2 * ctx + (1 if w == W - 1 else 0)
        for n in range(len(X)):
            ref = _lgbm_predict(structs[w][0], structs[w][1], A, Xp[n, w * M: w * M + width])
            assert np.allclose(B[n, w], ref, atol=2e-6), (w, n)
    with pytest.raises(ValueError, match="num_class|single-output"):
        convert.trees_from_lgbm_text(strs[0], A + 1 if A > 2 else 3)
    with pytest.raises(NotImplementedError, match="average_output"):
        convert.trees_from_lgbm_text(strs[0].replace("label_index=0", "average_output\nlabel_index=0", 1), A)


@pytest.mark.parametrize("A", [2, 5, -2])
def test_forest_from_catboost_json(oracle, A):
    """CBBase (src/Base/models.py:68-81): CatBoost JSON exports (oblivious trees) -> fb_* arrays; the oracle's forest on them ==
    CatBoost's own rule evaluated directly (leaf index bit i = x[feature_i] > border_i, leaf-major class values, scale and bias).
    Unpinned to catboost itself (absent from this image)."""
    from gnomix_amd import convert
    softmax2 = A < 0          # -2: MultiClass on two labels (two values per leaf, softmax) — folded into one margin
    A = abs(A)
    rng = np.random.RandomState(7 + A + 50 * softmax2)
    M, ctx, W = 5, 1, 3
    C = M * W + 2
    dims = 1 if (A == 2 and not softmax2) else A
    models = []
    for w in range(W):
        width = M + 2 * ctx + (2 if w == W - 1 else 0)
        trees = []
        for t in range(6):
            d = int(rng.randint(0, 5))
            splits = [{"float_feature_index": int(rng.randint(width)), "border": float(rng.choice([0.5, 1.5, 0.25])),
                       "split_type": "FloatFeature", "split_index": i} for i in range(d)]
            trees.append({"splits": splits, "leaf_values": [float(np.float32(v)) for v in rng.randn(dims << d) * 0.4],
                          "leaf_weights": [1] * (1 << d)})
        models.append({"oblivious_trees": trees, "scale_and_bias": [0.75, [float(b) for b in rng.randn(dims) * 0.2]],
                       "features_info": {"float_features": []}})
    f = convert.forest_from_catboost_json([models[0], __import__("json").dumps(models[1]), models[2]], A)
    T = oracle.Trees(f["fb_tree_off"], f["fb_left"], f["fb_right"], f["fb_feat"], f["fb_cond"], f["fb_tree_class"], max(A, 2),
                     default_left=f["fb_default_left"])
    X = rng.randint(0, 3, size=(30, C)).astype(np.int8)
    B = oracle.base_forest(T, f["fb_win_tree0"], X, M, ctx, A, missing=2)
    Xp = np.concatenate([X[:, :ctx][:, ::-1], X, X[:, -ctx:][:, ::-1]], axis=1)
    for w in range(W):
        width = M + 2 * ctx + (2 if w == W - 1 else 0)
        m = models[w]
        for n in range(len(X)):
            x = Xp[n, w * M: w * M + width]
            raw = np.array(m["scale_and_bias"][1], dtype=np.float64)
            for tr in m["oblivious_trees"]:
                idx = sum((1 << i) for i, sp in enumerate(tr["splits"]) if float(x[sp["float_feature_index"]]) > sp["border"])
                raw = raw + m["scale_and_bias"][0] * np.array(tr["leaf_values"][idx * dims:(idx + 1) * dims])
            if dims == 1:
                p1 = 1.0 / (1.0 + np.exp(-raw[0]))
                ref = np.array([1 - p1, p1])
            else:
                e = np.exp(raw - raw.max())
                ref = e / e.sum()
            assert np.allclose(B[n, w], ref, atol=2e-6), (w, n)


# ---------------------------------------------------------------- profile summaries -> bench.py's counter fields ----
def test_traffic_keys_are_exact_kernel_names_and_bench_refuses_impossible_counters():
    """scripts/summarize_prof.py filed the parked k_smooth_xgb_bs (197 MB per launch) under the rank walk's key and the
    driver-run line printed it as the dominant kernel's traffic (truth: 591 MB).  Keys are exact kernel names now; bench.py refuses a
    counter below 0.95 x the kernel's algorithmic bytes."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import importlib
    sp = importlib.import_module("summarize_prof")
    import bench

    def k(b, n=3):
        return {"FETCH_SIZE": {"avg_per_launch": 1.0, "launches": n}, "derived": {"hbm_bytes_per_launch": b}}
    pmc = {"k_smooth_xgb_rk<3, 8, 4, true, true>": k(590.75e6), "k_smooth_xgb_bs<512>": k(197.4e6), "k_bs_ranks": k(156e6),
           "k_base_logistic_i8<2, 1, 8, 2>": k(4.62e9), "k_base_logistic_p2<2, 8, 4, 3, 4, 4>": k(1.39e9),
           "k_smooth_xgb_rk<1, 8, 4, true, true>": k(1e6, n=1), "k_pack2": {"derived": {}}}
    t = sp.traffic_table(pmc)
    assert t["k_smooth_xgb"] == 590.75e6 and t["k_smooth_xgb_bs"] == 197.4e6 and t["k_base_logistic"] == 4.62e9
    assert t["k_base_logistic_p2"] == 1.39e9 and t["kernel_of_key"]["k_smooth_xgb"].startswith("k_smooth_xgb_rk<3")
    assert t["also_seen"] == ["k_smooth_xgb_rk<1, 8, 4, true, true>"]
    alg = 21090 * 10000
    assert bench.checked_traffic(t, "k_smooth_xgb", alg) == (590.75e6, None)
    val, note = bench.checked_traffic({"k_smooth_xgb": 197.4e6}, "k_smooth_xgb", alg)
    assert val is None and "below its algorithmic" in note
    assert bench.checked_traffic({}, "k_smooth_xgb", alg) == (None, None)
    # the printed line drops explanatory notes (the driver's record keeps ~8 KB) but keeps the roofline's
    c = bench._compact({"a": {"note": "x" * 500, "v": 1.23456789}, "roofline": {"note": "kept", "frac": 0.3333333}})
    assert c == {"a": {"v": 1.2346}, "roofline": {"note": "kept", "frac": 0.33333}}


def test_committed_counter_files_are_stamped_with_the_kernel_sources():
    """profiles/traffic_latest.json and trained_inputs_latest.json feed `roofline.traffic` / `roofline.trained_inputs` of the bench line
    and carry the hash of gnomix_amd/csrc at the time they were collected; bench.py prints null (and `counters.stale`) when the
    sources have moved on.  Whoever edits a kernel source re-collects them (scripts/collect_profiles.sh bench +
    scripts/dev/trained_inputs_counters.py, see profiles/README.md) — this test is the reminder."""
    import json
    import bench
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = bench.kernel_src_sha16()
    for name in ("traffic_latest.json", "trained_inputs_latest.json"):
        with open(os.path.join(here, "profiles", name)) as f:
            d = json.load(f)
        assert d.get("kernel_src_sha16") == sha, f"profiles/{name} was collected at other kernel sources ({d.get('kernel_src_sha16')} != {sha}): re-collect it"

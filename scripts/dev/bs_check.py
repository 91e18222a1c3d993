# k_smooth_xgb_bs (bit-sliced, no walks) against the rank kernel: bit-identity on small geometries with special values, then timing at config 2
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth, _lib

def model(W, A, S, rounds, depth, seed):
    d = gnomix_amd.GnxModelData(C=W * 10 + 3, M=10, A=A, S=S, context=5, smooth_kind="xgb")
    for k, v in synth.synthetic_trees(rounds, A, S * A, depth=depth, seed=seed, thr_lo=0.0, thr_hi=1.0, p_early_leaf=0.15).items():
        setattr(d, k, v)
    return d

def run(d, B, impl, **env):
    os.environ["GNX_SMOOTH_IMPL"] = impl
    for k, v in env.items(): os.environ[k] = str(v)
    m = gnomix_amd.DeviceModel(d)
    out = m.smooth_predict(B)
    for k in env: os.environ.pop(k)
    return out

bad = 0
if "check" in sys.argv or len(sys.argv) == 1:
    for (W, A, S, rounds, depth) in [(370, 7, 75, 12, 4), (131, 3, 31, 6, 4), (160, 12, 75, 4, 2), (500, 5, 75, 8, 4), (200, 4, 31, 7, 3), (700, 7, 75, 23, 4), (64, 2, 5, 3, 1), (1431, 12, 75, 5, 4)]:
        rng = np.random.RandomState(W + A)
        d = model(W, A, S, rounds, depth, W)
        N = 24
        B = rng.dirichlet(np.ones(A) * 0.4, size=(N, W)).astype(np.float32)
        thr = d.cond[d.left != -1]
        pick = rng.choice(thr, size=B.shape); m = rng.random_sample(B.shape)
        B = np.where(m < 0.25, pick, B)
        B = np.where((m >= 0.25) & (m < 0.35), np.nextafter(pick, np.float32(-1)), B)
        B = np.where((m >= 0.35) & (m < 0.45), np.nextafter(pick, np.float32(2)), B)
        special = np.array([0.0, 1.0, -0.25, 1.75, np.inf, -np.inf, np.nan, 1e-30, -0.0], np.float32)
        B = np.where(m > 0.97, rng.choice(special, size=B.shape), B).astype(np.float32)
        pr, lr = run(d, B, "rk")
        for wc in (128,):
            pb, lb = run(d, B, "bs")
            same = np.array_equal(pr, pb, equal_nan=True) and np.array_equal(lr, lb)
            nd = int((lr != lb).sum())
            print("W=%d A=%d S=%d rounds=%d depth=%d wc=%d identical=%s label diffs=%d maxdiff=%g" % (W, A, S, rounds, depth, wc, same, nd, float(np.nanmax(np.abs(pr - pb)))), flush=True)
            bad += not same
if "bench" in sys.argv or len(sys.argv) == 1:
    W, A, S, N = 370, 7, 75, int(os.environ.get("N", 10000))
    d = synth.synthetic_model(seed=0, n_rounds=100, **synth.CHR22)
    if os.environ.get("GEOM") == "chr1a12":   # config 5's smoother: chr1, 12 ancestries
        W, A, N = 1431, 12, int(os.environ.get("N", 4096))
        d = gnomix_amd.GnxModelData(C=W * 10 + 3, M=10, A=A, S=S, context=5, smooth_kind="xgb")
        for k, v in synth.synthetic_trees(100, A, S * A, depth=4, seed=0).items():
            setattr(d, k, v)
    rng = np.random.RandomState(1)
    B = rng.dirichlet(np.ones(A) * 0.5, size=(N, W)).astype(np.float32)
    Bd = torch.from_numpy(B).cuda()
    ref = None
    res, ident = {}, True
    for impl, env in [("rp", {}), ("bs", {})]:
        os.environ["GNX_SMOOTH_IMPL"] = impl
        os.environ.update(env)
        m = gnomix_amd.DeviceModel(d)
        p, l = m.smooth_predict_device(Bd); torch.cuda.synchronize()
        m.ctx.profile_reset(); m.ctx.profile_enable(True)
        for _ in range(10):
            m.smooth_predict_device(Bd)
        torch.cuda.synchronize(); m.ctx.profile_enable(False)
        ms, n = m.ctx.profile_get(_lib.K_SMOOTH_XGB)
        same = "" if ref is None else " identical to rp: %s" % (bool(torch.equal(p, ref[0]) and torch.equal(l, ref[1])))
        if ref is None: ref = (p, l)
        print("%-4s %s %.3f ms%s" % (impl, env, ms / n, same), flush=True)
        res[impl] = ms / n
        ident = ident and (same == "" or same.endswith("True"))
    import json
    print(json.dumps({"config": "tree smoother alone at config 2 (10 000 haplotypes x 370 windows, A = 7, 700 random depth-4 trees, B ~ Dirichlet(0.5)): "
                      "k_smooth_xgb_rk (pointer nodes) against k_smooth_xgb_bs + k_bs_ranks", "rp_ms": res.get("rp"), "bs_ms": res.get("bs"),
                      "outputs_identical": ident}))
sys.exit(1 if bad else 0)

# run the GPU fuzz tests (tests/test_gpu_fuzz.py) over seeds beyond the ones the suite pins:  python scripts/dev/extended_fuzz.py 1000 1040
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
from oracle import gnx_oracle as O
O.build()
import test_gpu_fuzz as F
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for name in ("test_random_geometry_vs_oracle", "test_random_geometry_tree_bases_vs_oracle", "test_random_gnofix_vs_oracle", "test_random_covrsk_vs_oracle"):
    fn = getattr(F, name)
    fn = getattr(fn, "__wrapped__", fn)
    ok = skipped = 0
    for seed in range(lo, hi):
        try:
            fn(O, seed); ok += 1
        except pytest.skip.Exception:
            skipped += 1
        except Exception:
            bad += 1
            print("FAIL", name, seed); traceback.print_exc(limit=3)
    print(name, "ok", ok, "skipped", skipped)

# the tree-smoother trainer against the oracle's on random small problems (trees must be identical, bit for bit)
import numpy as np
from gnomix_amd import train
import test_train_gbt as TG
ok = 0
for seed in range(lo, hi):
    rng = np.random.RandomState(seed)
    A = int(rng.choice([2, 3, 4, 5, 7, 9, 12])); S = int(rng.choice([3, 5, 9, 15, 31])); W = int(rng.randint(max(2 * S, A), max(2 * S, A) + 40)); N = int(rng.randint(4, 40))
    kw = dict(n_rounds=int(rng.randint(1, 6)), max_depth=int(rng.randint(1, 6)), max_bin=int(rng.choice([2, 7, 32, 256])),
              gamma=float(rng.choice([0.0, 0.3])), min_child_weight=float(rng.choice([0.0, 1.0, 4.0])), reg_lambda=float(rng.choice([0.0, 1.0, 10.0])),
              learning_rate=float(rng.choice([0.1, 0.5, 1.0])), base_score=float(rng.choice([0.5, 0.0])))
    B, y = TG._problem(N, W, A, seed=seed, noise=float(rng.choice([0.1, 0.5, 1.0])))
    if rng.rand() < 0.3:
        B = np.round(B, 2); B /= B.sum(-1, keepdims=True)      # heavy ties: few distinct values per class column
    if rng.rand() < 0.5:
        B = B.astype(np.float32)
    okw = dict(kw); okw["lam"] = okw.pop("reg_lambda"); okw["eta"] = okw.pop("learning_rate")
    try:
        for exact in (False, True):   # histogram form, and exact greedy (round 4)
            if exact and N * W > 900:
                continue              # (the oracle's exact form sorts per node and feature: keep it to small problems)
            T, lref = O.train_gbt(B, y, S, exact=exact, **okw)
            t, l = train.train_gbt_arrays(B, y, S, tree_method="exact" if exact else "hist", **kw)
            assert np.array_equal(t["tree_off"], T.tree_off) and np.array_equal(t["feat"], T.feat) and np.array_equal(t["left"], T.left), exact
            assert np.array_equal(t["cond"].view(np.uint32), T.cond.view(np.uint32)) and np.allclose(l, lref, atol=1e-6, rtol=0), exact
        ok += 1
    except Exception:
        bad += 1
        print("FAIL gbt", seed, dict(N=N, W=W, A=A, S=S), kw); traceback.print_exc(limit=2)
print("gbt trainer vs oracle ok", ok)

# the CNN and CRF smoother trainers against the oracle's restatements on random small problems
ok = 0
for seed in range(lo, min(hi, lo + 120)):
    r = np.random.RandomState(seed)
    A, S, W, N = int(r.randint(2, 14)), int(2 * r.randint(0, 12) + 1), int(r.randint(1, 80)), int(r.randint(1, 90))
    batch, ep = int(r.randint(1, N + 5)), int(r.randint(1, 5))
    y = r.randint(A, size=(N, W)).astype(np.int32)
    B = r.dirichlet(np.ones(A) * 0.5, size=(N, W)).astype(np.float32)
    w0, b0 = train.cnn_init(A, S, seed=seed)
    order = np.stack([r.permutation(N) for _ in range(ep)])
    try:
        w, b, loss = train.train_cnn_arrays(B, y, S, weight=w0, bias=b0, max_ep=ep, batch_size=batch, order=order)
        wo, bo, lo_ = O.cnn_fit(B, y, w0, b0, ep, batch=batch, order=order)
        assert np.abs(w - wo).max() < 3e-5 and np.abs(b - bo).max() < 3e-5 and np.allclose(loss, lo_, rtol=0, atol=1e-5)   # (float32 sums in another order, four epochs of Adam: seed 4084 reaches 1.05e-5)
        ok += 1
    except Exception:
        bad += 1
        print("FAIL cnn trainer", seed, (A, S, W, N, batch, ep)); traceback.print_exc(limit=3)
print("cnn trainer vs oracle ok", ok)
ok = 0
for seed in range(lo, min(hi, lo + 60)):
    r = np.random.RandomState(seed)
    A, W, N = int(r.randint(2, 13)), int(r.randint(1, 60)), int(r.randint(1, 40))
    y = r.randint(A, size=(N, W)).astype(np.int32)
    B = r.dirichlet(np.ones(A) * 0.5, size=(N, W))
    st0, tr0 = r.normal(0, 0.5, (A, A)), r.normal(0, 0.5, (A, A))
    try:
        _, _, info = train.train_crf_arrays(B, y, max_iterations=0, state0=st0, trans0=tr0)
        f, gs, gt = O.crf_objective(B, y, st0, tr0)
        gn = np.sqrt(np.sum(gs * gs) + np.sum(gt * gt))
        assert abs(info["objective"] - f) <= 1e-10 * max(1.0, abs(f)) and abs(info["grad_norm"] - gn) <= 1e-9 * max(1.0, gn)
        st, tr, info = train.train_crf_arrays(B, y)
        _, gs, gt = O.crf_objective(B, y, st, tr)
        assert info["converged"] and max(np.abs(gs).max(), np.abs(gt).max()) < 1e-6
        ok += 1
    except Exception:
        bad += 1
        print("FAIL crf trainer", seed, (A, W, N)); traceback.print_exc(limit=3)
print("crf trainer vs oracle ok", ok)
print("failures (all families):", bad)

sys.exit(1 if bad else 0)

# random geometries through k_smooth_xgb_bs against the rank walk, bit for bit:  python scripts/dev/bs_fuzz.py 0 300
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = ran = fell_back = 0
for seed in range(lo, hi):
    rng = np.random.RandomState(seed)
    A = int(rng.choice([2, 3, 5, 7, 8, 9, 12, 16]))
    S = int(rng.choice([1, 3, 5, 31, 75, 99, 127, 129]))
    W = int(rng.randint(max(2 * S, 2), max(2 * S, 2) + rng.choice([1, 40, 300, 1200])))
    rounds = int(rng.randint(1, 40))
    depth = int(rng.randint(1, 6))
    N = int(rng.choice([1, 2, 7, 33]))
    d = gnomix_amd.GnxModelData(C=W * 10 + 3, M=10, A=A, S=S, context=5, smooth_kind="xgb")
    T = synth.synthetic_trees(rounds, A, S * A, depth=depth, seed=seed, thr_lo=-0.1, thr_hi=1.1, p_early_leaf=float(rng.choice([0.0, 0.15, 0.6])))
    drop = int(rng.randint(0, A)) if rounds > 1 else 0
    if drop:
        n = len(T["tree_class"]) - drop
        nn = int(T["tree_off"][n])
        T = dict(tree_off=T["tree_off"][:n + 1], tree_class=T["tree_class"][:n], **{k: T[k][:nn] for k in ("left", "right", "feat", "cond")})
    for k, v in T.items():
        setattr(d, k, v)
    B = rng.dirichlet(np.ones(A) * rng.choice([0.05, 0.4, 3.0]), size=(N, W)).astype(np.float32)
    thr = d.cond[d.left != -1]
    if len(thr):
        m = rng.random_sample(B.shape)
        B = np.where(m < 0.2, rng.choice(thr, size=B.shape), B)
        B = np.where(m > 0.98, rng.choice(np.array([np.nan, np.inf, -np.inf, 0.0, 1.0, -1.0], np.float32), size=B.shape), B).astype(np.float32)
    if rng.rand() < 0.3:
        B[:, : W // 2] = B[:, :1]
    os.environ["GNX_SMOOTH_IMPL"] = "rk"
    pr, lr = gnomix_amd.DeviceModel(d).smooth_predict(B)
    os.environ["GNX_SMOOTH_IMPL"] = "bs"
    pb, lb = gnomix_amd.DeviceModel(d).smooth_predict(B)
    ran += 1
    fell_back += int(depth > 4 or S > 128)
    if not (np.array_equal(pr, pb, equal_nan=True) and np.array_equal(lr, lb)):
        bad += 1
        print("MISMATCH seed", seed, dict(A=A, S=S, W=W, rounds=rounds, depth=depth, N=N, drop=drop), int((lr != lb).sum()), flush=True)
print("bs fuzz: %d geometries, %d of them outside the kernel's range (rank walk on both sides), %d mismatches" % (ran, fell_back, bad))
sys.exit(1 if bad else 0)

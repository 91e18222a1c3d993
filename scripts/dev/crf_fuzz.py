# random CRF-smoother geometries against the oracle (marginals 1e-10, labels where the top two marginals are apart):
#   python scripts/dev/crf_fuzz.py 0 200
import os, sys, numpy as np
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
from oracle import gnx_oracle as O
O.build()
import gnomix_amd
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    rng = np.random.RandomState(9000 + seed)
    A = int(rng.choice([2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 20]))
    W = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 31, 64, 65, 100, 257, 1000]))
    N = int(rng.choice([1, 3, 4, 5, 17, 64, 130]))
    scale = float(rng.choice([0.3, 1.0, 3.0, 9.0, 25.0]))
    state = rng.standard_normal((A, A)) * scale
    trans = rng.standard_normal((A, A)) * scale * float(rng.choice([0.2, 1.0]))
    B = rng.dirichlet(np.ones(A) * float(rng.choice([0.2, 1.0, 5.0])), size=(N, W))
    f32 = bool(rng.randint(2))
    if f32: B = B.astype(np.float32)
    d = gnomix_amd.GnxModelData(C=W * 10 + 3, M=10, A=A, S=75, context=5, smooth_kind="crf", crf_state=state, crf_trans=trans)
    dev = gnomix_amd.DeviceModel(d)
    p_ref, l_ref = O.smooth_crf(B.astype(np.float64), state, trans)
    p, lab = dev.smooth_predict(B)
    top = np.sort(p_ref, -1)
    clear = (top[..., -1] - top[..., -2] > 1e-9) if A > 1 else np.ones_like(l_ref, bool)
    ok = np.isfinite(p).all() and np.max(np.abs(p - p_ref)) < 1e-10 and np.array_equal(lab[clear], l_ref[clear])
    if not ok:
        bad += 1; print("FAIL seed", seed, dict(A=A, W=W, N=N, scale=scale, f32=f32), float(np.max(np.abs(p - p_ref))), flush=True)
    dev.close()
print("crf fuzz:", hi - lo - bad, "ok,", bad, "bad")

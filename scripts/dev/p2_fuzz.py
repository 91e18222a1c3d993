"""random geometries through the 2-bit logistic kernels against the int8 kernels (B must be bit-identical):  python scripts/dev/p2_fuzz.py 0 200
Half of the draws have R * A == 24 class columns (the flat-tile kernel k_base_logistic_p2f: A in {2, 3, 4, 6, 8, 12} with the context that
makes R = 24 / A windows overlap), the rest any A <= 16 and any context (slot tiles / one tile).  GPU box only."""
import os
import sys

os.environ.setdefault("GNX_LR_P2", "2")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gnomix_amd
from gnomix_amd import synth

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = flat = 0
for seed in range(lo, hi):
    rng = np.random.RandomState(seed)
    M = int(rng.choice([40, 64, 100, 128, 175, 256, 300, 513, 700]))
    if seed % 2 == 0:
        A = int(rng.choice([2, 3, 4, 6, 8, 12]))
        R = 24 // A
        # R = ceil((M + 2 ctx) / M)  <=>  (R - 1) M < M + 2 ctx <= R M
        c_lo, c_hi = ((R - 2) * M) // 2 + 1, ((R - 1) * M) // 2
        ctx = int(rng.randint(max(c_lo, 0), c_hi + 1))
        flat += 1
    else:
        A = int(rng.randint(2, 17))
        ctx = int(rng.randint(0, 2 * M))
    W = int(rng.randint(3, 30))
    C = W * M + int(rng.randint(1, M))          # C % M != 0 (the reference rejects exact multiples)
    N = int(rng.choice([1, 2, 31, 33, 64, 200, 257, 600]))
    try:
        d = synth.synthetic_model(C=C, M=M, A=A, S=5, context=ctx, seed=seed, smooth=None)
        X = synth.synthetic_X(N, C, seed=seed + 1, miss=float(rng.choice([0.0, 0.03, 0.3])))
        dev = gnomix_amd.DeviceModel(d)
    except Exception as e:  # geometries the model loader rejects (e.g. a context wider than the chromosome)
        print("skip", seed, (C, M, A, ctx, N), str(e)[:80])
        continue
    Xt = torch.from_numpy(X).cuda()
    Pt = dev.pack_device(Xt)
    for f64 in (True, False):
        a = dev.base_predict_device(Xt, f64=f64)
        b = dev.base_predict_packed_device(Pt, f64=f64)
        torch.cuda.synchronize()
        if not torch.equal(a, b):
            bad += 1
            print("MISMATCH", seed, (C, M, A, ctx, N), "f64" if f64 else "f32", float((a.double() - b.double()).abs().max()), flush=True)
    dev.close()
print("p2 fuzz", lo, hi, "flat-eligible draws", flat, "FAILURES" if bad else "no failures", bad)

#!/usr/bin/env python3
"""Training the logistic base at config-2 geometry on ONE GPU next to sklearn/liblinear — the solver the
reference's Base.train calls per window (src/Base/models.py:17-21) — timed on a few windows of the same data on one host core.

  python scripts/bench_train.py [N_train]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from gnomix_amd import synth, train

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
C, M, A, ctx = 370_500, 1000, 7, 500
W = C // M
rng = np.random.RandomState(1)
# admixed-like training set: per-ancestry allele frequencies, piecewise-constant ancestry along the chromosome
freq = rng.uniform(0.05, 0.95, size=(A, C)).astype(np.float32)
y = np.empty((N, W), np.int32)
for i in range(N):
    a = rng.randint(A)
    for w in range(W):
        if rng.rand() < 0.02:
            a = rng.randint(A)
        y[i, w] = a
ysnp = np.repeat(y, M, axis=1)
ysnp = np.concatenate([ysnp, np.repeat(y[:, -1:], C - ysnp.shape[1], axis=1)], axis=1)
X = np.empty((N, C), np.int8)
for n0 in range(0, N, 64):
    sl = slice(n0, min(N, n0 + 64))
    X[sl] = rng.random_sample((sl.stop - sl.start, C)).astype(np.float32) < freq[ysnp[sl], np.arange(C)[None, :]]
X[rng.random_sample(X.shape) < 0.01] = 2
del ysnp
res = {"config": "train logistic base, chr22 geometry C=370500 M=1000 ctx=500 A=7", "N_train": N, "problems": W * A}
train.train_logistic_arrays(X[:64], y[:64], M, ctx, A, tol=1e-3, max_iter=2)   # warm-up (context, first allocations)
for tol in (1e-4, 1e-9):
    t0 = time.perf_counter()
    coef, icpt, info = train.train_logistic_arrays(X, y, M, ctx, A, tol=tol)
    dt = time.perf_counter() - t0
    res["gpu_tol_%g" % tol] = dict(seconds=dt, **info)
try:
    from sklearn.linear_model import LogisticRegression
    import warnings
    wins = [3, 120, 250]
    t0 = time.perf_counter()
    worst = 0.0
    for w in wins:
        Xw = X[:, w * M - ctx:w * M + M + ctx]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = LogisticRegression(penalty="l2", C=3., solver="liblinear", max_iter=1000).fit(Xw, y[:, w])
        worst = max(worst, float(np.max(np.abs(m.coef_ - coef[w, :, :2 * ctx + M]))))
    dt = (time.perf_counter() - t0) / len(wins)
    res["cpu_liblinear"] = {"seconds_per_window_one_core": dt, "seconds_all_windows_one_core": dt * W, "windows_timed": wins,
                            "max_abs_coef_diff_vs_gpu": worst}
except ImportError:
    pass
print(json.dumps(res))

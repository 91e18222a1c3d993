#!/usr/bin/env python3
"""Training the tree smoother at config-2 geometry (chr22: W = 370, A = 7, S = 75; the reference's XGBClassifier arguments:
100 rounds, depth 4) on ONE GPU, next to scikit-learn's histogram boosting on the same slide_window rows on
the host's cores — xgboost itself, what the reference's Smoother.train calls (src/Smooth/models.py:14-20), is not installed.

  python scripts/bench_train_gbt.py [N_train] [cpu_rounds]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gnomix_amd import synth, train

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
cpu_rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
W, A, S = 370, 7, 75
rng = np.random.RandomState(3)
B = synth.synthetic_phased_individuals((N + 1) // 2, W, A, seed=5, phase_errors=0, noise=0.02)[:N]
y = np.argmax(B, -1).astype(np.int32)
B = np.clip(B + rng.normal(0, 0.45, B.shape), 1e-4, None)
B /= B.sum(-1, keepdims=True)
res = {"config": "train tree smoother, chr22 geometry W=370 A=7 S=75, 100 rounds depth 4", "N_train": N, "rows": N * W, "features": S * A}
train.train_gbt_arrays(B[:16], y[:16], S, n_rounds=2)   # warm-up
t0 = time.perf_counter()
trees, loss = train.train_gbt_arrays(B, y, S)
res["gpu_host_arrays"] = {"seconds": time.perf_counter() - t0, "loss_first_last": [float(loss[0]), float(loss[-1])], "nodes": int(len(trees["left"]))}
Bd, yd = torch.from_numpy(B).cuda(), torch.from_numpy(y).cuda()
torch.cuda.synchronize()
t0 = time.perf_counter()
trees_d, loss_d = train.train_gbt_arrays(Bd, yd, S)
res["gpu_device_tensors"] = {"seconds": time.perf_counter() - t0, "identical_to_host_run": bool(all(np.array_equal(trees[k], trees_d[k]) for k in trees))}
# exact greedy (tree_method="exact"): a few rounds timed, scaled to 100 (a round costs the same whatever its number)
ex_rounds = int(os.environ.get("EXACT_ROUNDS", "5"))
if ex_rounds > 0:
    t0 = time.perf_counter()
    trees_e, loss_e = train.train_gbt_arrays(B, y, S, n_rounds=ex_rounds, tree_method="exact")
    dt = time.perf_counter() - t0
    _, loss_h = train.train_gbt_arrays(B, y, S, n_rounds=ex_rounds)
    res["gpu_exact_greedy"] = {"rounds_timed": ex_rounds, "seconds": dt, "seconds_scaled_to_100_rounds": dt * 100 / ex_rounds,
                               "loss_after": float(loss_e[-1]), "histogram_loss_after_same_rounds": float(loss_h[-1])}
try:
    if cpu_rounds <= 0:
        raise ImportError
    from sklearn.ensemble import HistGradientBoostingClassifier
    sys.path.insert(0, ROOT)
    pad = (S + 1) // 2
    Bp = np.concatenate([B[:, :pad][:, ::-1], B, B[:, -pad:][:, ::-1]], axis=1).astype(np.float32)
    rows = np.lib.stride_tricks.sliding_window_view(Bp, (S, A), axis=(1, 2))[:, :W, 0].reshape(N * W, S * A)
    clf = HistGradientBoostingClassifier(max_iter=cpu_rounds, max_depth=4, learning_rate=0.05, l2_regularization=0.5, max_bins=255,
                                         early_stopping=False, min_samples_leaf=1)
    t0 = time.perf_counter()
    clf.fit(rows, y.reshape(-1))
    dt = time.perf_counter() - t0
    res["cpu_sklearn_hist"] = {"rounds_timed": cpu_rounds, "seconds": dt, "seconds_scaled_to_100_rounds": dt * 100 / cpu_rounds,
                               "cores": os.cpu_count()}
except ImportError:
    pass
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "bench_train_gbt.json"), "w") as f:
    json.dump(res, f, indent=1)

# CRF.fit at the "fast" mode's shape on the device: N training rows x W = 317 windows, A = 7
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from gnomix_amd.train import train_crf_arrays
N, W, A = int(os.environ.get("N", 2000)), 317, 7
rng = np.random.RandomState(0)
y = np.repeat(rng.randint(A, size=(N, (W + 9) // 10)), 10, axis=1)[:, :W].astype(np.int32)
B = rng.dirichlet(np.ones(A) * 0.6, size=(N, W)); B[np.arange(N)[:, None], np.arange(W)[None, :], y] += 0.8 * rng.random_sample((N, W))
B = B / B.sum(-1, keepdims=True)
train_crf_arrays(B[:64], y[:64], max_iterations=2)
for eps in (1e-5, 1e-8):
    t0 = time.perf_counter(); st, tr, info = train_crf_arrays(B, y, epsilon=eps); dt = time.perf_counter() - t0
    print("N %d epsilon %g: %.3f s, %d iterations, %d evaluations (%.0f us each), objective %.6f, |g| %.2e" % (N, eps, dt, info["iterations"], info["evaluations"], dt / info["evaluations"] * 1e6, info["objective"], info["grad_norm"]), flush=True)

# CNN.fit at the "large" mode's shape on the device: N training rows x W windows, A = 7, S = 75, 250 epochs, batches of 128
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from gnomix_amd.train import train_cnn_arrays
N, W, A, S = int(os.environ.get("N", 2000)), 317, 7, 75
rng = np.random.RandomState(0)
y = np.repeat(rng.randint(A, size=(N, (W + 9) // 10)), 10, axis=1)[:, :W].astype(np.int32)
B = rng.dirichlet(np.ones(A) * 0.6, size=(N, W)); B[np.arange(N)[:, None], np.arange(W)[None, :], y] += 0.8 * rng.random_sample((N, W))
B = (B / B.sum(-1, keepdims=True)).astype(np.float32)
train_cnn_arrays(B[:256], y[:256], S, max_ep=2, seed=0)
for ep in (25, 250):
    t0 = time.perf_counter(); w, b, loss = train_cnn_arrays(B, y, S, max_ep=ep, seed=0); dt = time.perf_counter() - t0
    steps = ep * ((N + 127) // 128)
    print("N %d, %d epochs = %d Adam steps: %.2f s (%.0f us per step), loss %.4f -> %.4f" % (N, ep, steps, dt, dt / steps * 1e6, loss[0], loss[-1]), flush=True)

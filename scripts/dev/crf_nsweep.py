import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth
W, A = 1431, 12
d = synth.synthetic_model(C=W * 10 + 5, M=10, A=A, S=75, seed=4, smooth="crf")
m = gnomix_amd.DeviceModel(d)
rng = np.random.RandomState(0)
for N in [4096, 8192, 12288, 16384, 20480, 24576, 25000, 28672, 32768]:
    B = torch.rand((N, W, A), dtype=torch.float64, device="cuda"); B /= B.sum(-1, keepdim=True)
    for _ in range(2): m.smooth_predict_device(B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m.smooth_predict_device(B)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("N=%d waves=%d  %.3f ms  %.1f ns/hap" % (N, N // 4, dt * 1e3, dt * 1e9 / N), flush=True)
    del B

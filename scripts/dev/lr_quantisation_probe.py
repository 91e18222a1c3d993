# Is the logistic pass losing time to block-count quantisation?  One 8-wave block per CU (184 VGPRs), grid = haplotype tiles (512 rows)
# x window ranges (a multiple of 8): 10 000 haplotypes -> 20 x 32 = 640 blocks on 256 CUs = 2.5 "rounds".
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import gnomix_amd
from gnomix_amd import synth
from bench_configs import prof
C, M, A, S = 370500, 1000, 7, 75
d = synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=2, seed=1)
model = gnomix_amd.DeviceModel(d)
Xall = torch.from_numpy(synth.synthetic_X(14336, C, seed=2, miss=0.01)).cuda()
for N in (6144, 8192, 9216, 10000, 10240, 11264, 12288, 13312, 14336):
    X = Xall[:N]
    model.base_predict_device(X); torch.cuda.synchronize()
    model.ctx.profile_reset(); model.ctx.profile_enable(True)
    for _ in range(10):
        model.base_predict_device(X)
    torch.cuda.synchronize(); model.ctx.profile_enable(False)
    ms = prof(model.ctx)["k_base_logistic"]
    print("N %6d  tiles %3d  %.3f ms  %.1f M hap/s  %.2f TB/s of X" % (N, (N + 511) // 512, ms, N / ms / 1e3, N * C / ms / 1e9))

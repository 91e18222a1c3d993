"""forest base: does running two decoupled 128-haplotype blocks per CU beat one 256-haplotype block? (needs trees small
enough for 2 x (64 KB tile + trees) to fit the LDS: 10 rounds here)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gnomix_amd
from gnomix_amd import synth, _lib
C, M, A, N = 370_500, 1000, 7, 10000
rounds = int(os.environ.get("ROUNDS", "10"))
data = synth.synthetic_forest_model(C, M, A, n_rounds=rounds, depth=4, seed=0, S=75, smooth=None)
model = gnomix_amd.DeviceModel(data)
X = synth.synthetic_X_device(N, C, "cuda:0", seed=1)
for _ in range(2): model.base_predict_device(X)
torch.cuda.synchronize()
model.ctx.profile_reset(); model.ctx.profile_enable(True)
for _ in range(5): model.base_predict_device(X)
torch.cuda.synchronize()
ms, n = model.ctx.profile_get(_lib.K_BASE_FOREST)
print("T=%s rounds=%d: k_base_forest %.3f ms" % (os.environ.get("GNX_FOREST_T", "auto"), rounds, ms / n))

"""Does the logistic pass's HBM rate depend on the row stride / alignment (TLB, DRAM page and channel mapping)?
Same model, same haplotypes, X stored with different leading dimensions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gnomix_amd
from gnomix_amd import synth, _lib
C, M, N = 370_500, 1000, 10_000
data = synth.synthetic_model(C=C, M=M, A=7, S=5, seed=0, smooth=None)
model = gnomix_amd.DeviceModel(data)
X0 = synth.synthetic_X_device(N, C, "cuda:0", seed=1)
for ldx in (C, 370_560, 372_736, 393_216, 524_288, 2 * C, 370_500 + 64):
    buf = torch.zeros((N, ldx), dtype=torch.int8, device="cuda:0")
    buf[:, :C] = X0
    X = buf[:, :C]
    for _ in range(2): model.base_predict_device(X)
    torch.cuda.synchronize()
    model.ctx.profile_reset(); model.ctx.profile_enable(True)
    for _ in range(5): model.base_predict_device(X)
    torch.cuda.synchronize()
    model.ctx.profile_enable(False)
    ms, n = model.ctx.profile_get(_lib.K_BASE_LOGISTIC)
    print("ldx=%d (%%128=%d, %%4096=%d): %.3f ms  X %.2f TB/s" % (ldx, ldx % 128, ldx % 4096, ms / n, C * N / (ms / n * 1e-3) / 1e12), flush=True)
    del X, buf
    torch.cuda.empty_cache()

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gnomix_amd
from gnomix_amd import synth, _lib
A = int(os.environ.get("A", "24")); N = int(os.environ.get("NH", "8192"))
data = synth.synthetic_model(C=370_500, M=1000, A=A, S=75, seed=1, smooth=None)
model = gnomix_amd.DeviceModel(data)
X = synth.synthetic_X_device(N, data.C, "cuda:0", seed=1)
for _ in range(2): model.base_predict_device(X)
torch.cuda.synchronize()
model.ctx.profile_reset(); model.ctx.profile_enable(True)
for _ in range(4): model.base_predict_device(X)
torch.cuda.synchronize()
ms, n = model.ctx.profile_get(_lib.K_BASE_LOGISTIC)
print(os.environ.get("TAG", ""), "A=%d base avg_ms %.3f X GB/s %.0f" % (A, ms / n, 370500 * N / (ms / n * 1e-3) / 1e9))

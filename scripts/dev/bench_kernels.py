import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gnomix_amd
from gnomix_amd import synth, _lib
N = int(os.environ.get("NH", "10000"))
data = synth.synthetic_model(seed=0, n_rounds=100, **synth.CHR22)
model = gnomix_amd.DeviceModel(data)
X = synth.synthetic_X_device(N, data.C, "cuda:0", seed=1)
which = os.environ.get("WHICH", "base")
def run():
    if which == "base": return model.base_predict_device(X)
    return model.infer_device(X)
for _ in range(2): run()
torch.cuda.synchronize()
model.ctx.profile_reset(); model.ctx.profile_enable(True)
for _ in range(5): run()
torch.cuda.synchronize()
model.ctx.profile_enable(False)
for k in (_lib.K_BASE_LOGISTIC, _lib.K_SMOOTH_XGB):
    ms, n = model.ctx.profile_get(k)
    if n: print(os.environ.get("TAG",""), _lib.KERNEL_NAMES[k], "avg_ms %.4f" % (ms/n), "GB/s %.0f" % ((380860*N if k==0 else 21090*N)/(ms/n*1e-3)/1e9))

# The tree smoother on a TRAINED ensemble (splits concentrated on the central windows, leaves that matter) against the bench's
# random trees (uniform random features / thresholds: the worst case for divergence and LDS bank conflicts).  Same input B.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth, train
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
from bench_configs import prof

W, A, S, N = 370, 7, 75, 10000
rng = np.random.RandomState(3)
def noisy(Bc, sd):
    B = np.clip(Bc + rng.normal(0, sd, Bc.shape), 1e-4, None)
    return B / B.sum(-1, keepdims=True)
Bt = synth.synthetic_phased_individuals(500, W, A, seed=5, phase_errors=0, noise=0.02)
yt = np.argmax(Bt, -1).astype(np.int32)
Bt = noisy(Bt, 0.45)
trees, loss = train.train_gbt_arrays(Bt, yt, S)
Bq = noisy(synth.synthetic_phased_individuals(N // 2, W, A, seed=9, phase_errors=0, noise=0.02), 0.45).astype(np.float32)
Bd = torch.from_numpy(Bq).cuda()
d_rand = synth.synthetic_model(C=W * 1000 + 500, M=1000, A=A, S=S, n_rounds=100, seed=1)
d_tr = synth.synthetic_model(C=W * 1000 + 500, M=1000, A=A, S=S, n_rounds=1, seed=1)
for k, v in trees.items():
    setattr(d_tr, k, v)
for name, d in (("random trees (bench)", d_rand), ("trained ensemble", d_tr)):
    m = gnomix_amd.DeviceModel(d)
    m.smooth_predict_device(Bd); torch.cuda.synchronize()
    m.ctx.profile_reset(); m.ctx.profile_enable(True)
    for _ in range(10):
        m.smooth_predict_device(Bd)
    torch.cuda.synchronize(); m.ctx.profile_enable(False)
    print(name, "nodes", len(d.left), prof(m.ctx))

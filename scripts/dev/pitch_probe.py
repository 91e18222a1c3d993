# Does the row pitch of X matter to the logistic pass?  (C = 370500 is not a multiple of 64: most 128-byte row runs straddle
# three cache lines.)  Same data, pitch C vs pitch rounded up to 128 / 256 bytes.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth

C, M, A, S, N = 370500, 1000, 7, 75, 10000
d = synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=100, seed=1)
model = gnomix_amd.DeviceModel(d)
Xh = torch.from_numpy(synth.synthetic_X(N, C, seed=2, miss=0.01))
ref = None
for align in (1, 64, 128, 256, 4096):
    Cp = (C + align - 1) // align * align
    Xp = torch.zeros((N, Cp), dtype=torch.int8, device="cuda")
    Xp[:, :C] = Xh.cuda()
    X = Xp[:, :C]
    p, lab = model.infer_device(X)
    torch.cuda.synchronize()
    if ref is None:
        ref = lab.clone()
    assert torch.equal(ref, lab)
    model.ctx.profile_reset(); model.ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(10):
        model.infer_device(X)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    model.ctx.profile_enable(False)
    from scripts.bench_configs import prof
    print("pitch aligned to %5d: %.3f ms/step  %s" % (align, dt * 1e3, prof(model.ctx)))

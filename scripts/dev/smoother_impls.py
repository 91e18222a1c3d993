# the tree smoother of config 2 through its three kernels (GNX_SMOOTH_IMPL at model load): float, ranks (lane = window), h64 (lane = haplotype)
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth, _lib
W, A, S, N = 370, 7, 75, int(os.environ.get("N", 10000))
d = synth.synthetic_model(seed=0, n_rounds=100, **synth.CHR22)
rng = np.random.RandomState(1)
B = rng.dirichlet(np.ones(A) * 0.5, size=(N, W)).astype(np.float32)
Bd = torch.from_numpy(B).cuda()
ref = None
for impl in os.environ.get("IMPLS", "rk,rp,h64,f32").split(","):
    os.environ["GNX_SMOOTH_IMPL"] = impl
    m = gnomix_amd.DeviceModel(d)
    p, l = m.smooth_predict_device(Bd); torch.cuda.synchronize()
    m.ctx.profile_reset(); m.ctx.profile_enable(True)
    for _ in range(10):
        m.smooth_predict_device(Bd)
    torch.cuda.synchronize(); m.ctx.profile_enable(False)
    ms, n = m.ctx.profile_get(_lib.K_SMOOTH_XGB)
    same = "" if ref is None else " identical to rk: %s" % (bool(torch.equal(p, ref[0]) and torch.equal(l, ref[1])))
    if ref is None: ref = (p, l)
    print("%-4s %.3f ms%s" % (impl, ms / n, same), flush=True)

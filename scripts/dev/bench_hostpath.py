import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, gnomix_amd
from gnomix_amd import synth
N = int(os.environ.get("NH", "8192"))
data = synth.synthetic_model(seed=0, n_rounds=100, **synth.CHR22)
g = gnomix_amd.HipGnomix(data)
X = synth.synthetic_X_device(N, data.C, "cuda:0", seed=1).cpu().numpy()
for rep in range(3):
    t0 = time.perf_counter()
    p, lab = g.dev.infer(X)
    dt = time.perf_counter() - t0
    print("host-pointer gnx_infer: N=%d %.3f s  %.0f haplotypes/s  (%.1f GB/s of X over PCIe)" % (N, dt, N / dt, N * data.C / dt / 1e9), flush=True)

# the same through page-locked arrays (Context.pinned_empty): input and outputs
Xp = g.dev.ctx.pinned_empty(X.shape, np.int8)
Xp[...] = X
for rep in range(3):
    t0 = time.perf_counter()
    p, lab = g.dev.infer(Xp)
    dt = time.perf_counter() - t0
    print("pinned X        gnx_infer: N=%d %.3f s  %.0f haplotypes/s  (%.1f GB/s of X over PCIe)" % (N, dt, N / dt, N * data.C / dt / 1e9), flush=True)

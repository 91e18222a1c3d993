# phase clocks of k_gnofix: build csrc with `make CXXFLAGS+=-DGNX_GNOFIX_CLOCKS` first (see k_gnofix.hip); prints mean clocks per phase.
# argv[1] = c5b (default: chr1, A = 12, individuals with two switch errors) | worst (chr22, A = 7, random trees on unstructured haplotypes)
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth
which = (sys.argv[1:] or ["c5b"])[0]
S = 75
if which == "worst":
    W, A = 370, 7
    trees = synth.synthetic_trees(100, A, S * A, depth=4, seed=1)
else:
    W, A = 1431, 12
    trees = synth.synthetic_smoothing_trees(100, A, S, seed=6)
C = 1000 * W + 500
data = gnomix_amd.GnxModelData(C=C, M=1000, A=A, S=S, context=500, smooth_kind="xgb")
for k, v in trees.items():
    setattr(data, k, v)
model = gnomix_amd.DeviceModel(data)
n = int(os.environ.get("N_IND", "512"))
B = synth.synthetic_phased_individuals(n, W, A, seed=3) if which != "worst" else np.random.RandomState(5).dirichlet(np.ones(A), size=(2 * n, W))
Xd = torch.randint(0, 2, (2 * n, C), dtype=torch.int8, device="cuda"); Bd = torch.from_numpy(B).cuda()
Yd, ns = model.gnofix_device(Xd.clone(), Bd)
v = ns.cpu().numpy().astype(float).reshape(-1, 8) * 64
print(which, "mean clk per phase [load, converge, search, cand tile, cand walks, cand decide, accept, re-evaluate]:", v.mean(0).round(0), "sum", v.mean(0).sum().round(0))

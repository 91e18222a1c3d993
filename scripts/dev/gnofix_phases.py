# phase clocks of k_gnofix: build csrc with `make CXXFLAGS+=-DGNX_GNOFIX_CLOCKS` first (see k_gnofix.hip); prints mean clocks per phase
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth
W, A, S = 1431, 12, 75
C = 1000 * W + 500
data = gnomix_amd.GnxModelData(C=C, M=1000, A=A, S=S, context=500, smooth_kind="xgb")
for k, v in synth.synthetic_smoothing_trees(100, A, S, seed=6).items():
    setattr(data, k, v)
model = gnomix_amd.DeviceModel(data)
n = int(os.environ.get("N_IND", "512"))
B = synth.synthetic_phased_individuals(n, W, A, seed=3)
X = np.random.RandomState(1).randint(0, 2, size=(2 * n, C)).astype(np.int8)
Xd = torch.from_numpy(X).cuda(); Bd = torch.from_numpy(B).cuda()
Yd, ns = model.gnofix_device(Xd.clone(), Bd)
v = ns.cpu().numpy().astype(float).reshape(-1, 8) * 64
print("mean clk per phase [load, converge, search, cand rows, cand walks, cand decide, accept, re-evaluate]:", v.mean(0).round(0), "sum", v.mean(0).sum().round(0))

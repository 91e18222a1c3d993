import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnomix_amd as ga
from gnomix_amd import synth
d = synth.synthetic_model(seed=0, **synth.CHR22)
dev = ga.DeviceModel(d)
N = 512
Xd = synth.synthetic_X_device(N, d.C, "cuda:0", seed=94305)
def run(X):
    p, lab = dev.infer_device(X); torch.cuda.synchronize(); return p.cpu().numpy(), lab.cpu().numpy()
p1, _ = run(Xd); p1b, _ = run(Xd)
print("repeat identical:", np.array_equal(p1, p1b))
B1 = dev.base_predict_device(Xd); torch.cuda.synchronize(); B1 = B1.cpu().numpy()
B1b = dev.base_predict_device(Xd); torch.cuda.synchronize(); B1b = B1b.cpu().numpy()
print("base repeat identical:", np.array_equal(B1, B1b))
g = torch.Generator(device="cuda:0").manual_seed(1)
perm = torch.randperm(N, device="cuda:0", generator=g)
Xp = Xd[perm].contiguous(); torch.cuda.synchronize()
pn = perm.cpu().numpy()
print("X perm ok:", np.array_equal(Xp.cpu().numpy(), Xd.cpu().numpy()[pn]))
B2 = dev.base_predict_device(Xp); torch.cuda.synchronize(); B2 = B2.cpu().numpy()
bad = np.where((B2 != B1[pn]).reshape(N, -1).any(1))[0]
print("base perm mismatching rows:", len(bad), bad[:20], "-> orig idx", pn[bad[:20]])
if len(bad):
    r = bad[0]; w = np.where((B2[r] != B1[pn[r]]).any(1))[0]; print("windows", w[:20], B2[r, w[0]], B1[pn[r], w[0]])
Bt = torch.from_numpy(B1).cuda()
s1, _ = dev.smooth_predict_device(Bt); torch.cuda.synchronize(); s1 = s1.cpu().numpy()
s2, _ = dev.smooth_predict_device(Bt[perm].contiguous()); torch.cuda.synchronize(); s2 = s2.cpu().numpy()
bad = np.where((s2 != s1[pn]).reshape(N, -1).any(1))[0]
print("smooth perm mismatching rows:", len(bad), bad[:20])

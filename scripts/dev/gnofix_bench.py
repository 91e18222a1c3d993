# device-resident Gnofix timings: config 5b (chr1 WGS, A = 12, individuals with 2 switch errors) and the worst case
# (chr22, A = 7, unstructured haplotypes + random trees: a label change at almost every window, 50 sweeps)
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth

def run(name, W, A, n, structured, reps=3):
    S = 75
    C = 1000 * W + 500
    data = gnomix_amd.GnxModelData(C=C, M=1000, A=A, S=S, context=500, smooth_kind="xgb")
    if structured:
        for k, v in synth.synthetic_smoothing_trees(100, A, S, seed=6).items():
            setattr(data, k, v)
        B = synth.synthetic_phased_individuals(n, W, A, seed=3)
    else:
        for k, v in synth.synthetic_trees(100, A, S * A, depth=4, seed=1).items():
            setattr(data, k, v)
        B = np.random.RandomState(5).dirichlet(np.ones(A), size=(2 * n, W))
    model = gnomix_amd.DeviceModel(data)
    Xd = torch.randint(0, 2, (2 * n, C), dtype=torch.int8, device="cuda")
    Bd = torch.from_numpy(B).cuda()
    model.gnofix_device(Xd.clone(), Bd)
    best = 1e9
    for _ in range(reps):
        Xw = Xd.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Yd, ns = model.gnofix_device(Xw, Bd)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%s: %d individuals in %.2f ms = %.0f individuals/s; mean switches %.1f" % (name, n, best * 1e3, n / best, float(ns.float().mean())), flush=True)
    model.close()

which = sys.argv[1:] or ["c5b", "worst"]
if "c5b" in which: run("c5b chr1 A=12", 1431, 12, int(os.environ.get("N_IND", "2048")), True)
if "worst" in which: run("worst chr22 A=7", 370, 7, int(os.environ.get("N_WORST", "512")), False, reps=1)

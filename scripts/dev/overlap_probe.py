# Do the logistic pass (L1-miss-queue bound) and the tree smoother (LDS/VALU bound) overlap when issued on two streams?
# Two contexts = two streams; each runs the fused pipeline on half of the haplotypes, offset by one kernel.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth, _lib

C, M, A, S, N = 317500, 1000, 7, 75, 10000
d = synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=100, seed=1)
X = torch.from_numpy(synth.synthetic_X(N, C, seed=2, miss=0.01)).cuda()
ctxs = [_lib.Context(0), _lib.Context(0)]
models = [gnomix_amd.DeviceModel(d, ctx=c) for c in ctxs]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def run_single(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        models[0].infer_device(X)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K

def run_split(K, parts):
    n = N // parts
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        for p in range(parts):
            s = p % 2
            with torch.cuda.stream(streams[s]):
                models[s].infer_device(X[p * n:(p + 1) * n])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K

run_single(2); run_split(2, 2)
print("single stream, one batch     ms/step", round(run_single(10) * 1e3, 3))
for parts in (2, 4, 8):
    print("two streams, %d parts         ms/step" % parts, round(run_split(10, parts) * 1e3, 3))

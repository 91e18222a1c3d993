# Do the logistic pass (L1-miss-queue bound) and the tree smoother (LDS/VALU bound) overlap when issued on two streams?
# Two contexts = two streams; each runs the fused pipeline on half of the haplotypes.  GNX_LDS_PAD="lr,sm" (read at gnx_init)
# pads each kernel's dynamic LDS so that only ONE of its blocks fits a CU and the other kernel's block can sit beside it.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth, _lib

C, M, A, S, N = 370500, 1000, 7, 75, 10000
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
print("GNX_LDS_PAD", os.environ.get("GNX_LDS_PAD"), " single stream ms/step", round(run_single(10) * 1e3, 3),
      " two streams, 2 / 4 parts", round(run_split(10, 2) * 1e3, 3), round(run_split(10, 4) * 1e3, 3))

# anti-phase: the logistic pass of part k+1 may start when the logistic pass of part k is done, i.e. beside the smoother of part k
def run_antiphase(K, parts):
    n = N // parts
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prev = None
    for _ in range(K):
        for p in range(parts):
            s = p % 2
            with torch.cuda.stream(streams[s]):
                if prev is not None:
                    streams[s].wait_event(prev)
                B = models[s].base_predict_device(X[p * n:(p + 1) * n])
                prev = torch.cuda.Event(); prev.record(streams[s])
                models[s].smooth_predict_device(B)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K

def run_serial_two_calls(K):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        B = models[0].base_predict_device(X)
        models[0].smooth_predict_device(B)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K

run_antiphase(2, 4)
print("   base + smoother as two calls, one stream", round(run_serial_two_calls(10) * 1e3, 3),
      " anti-phase on two streams, 2 / 4 / 8 parts", round(run_antiphase(10, 2) * 1e3, 3), round(run_antiphase(10, 4) * 1e3, 3), round(run_antiphase(10, 8) * 1e3, 3))

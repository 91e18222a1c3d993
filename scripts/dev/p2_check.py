"""development check of k_base_logistic_p2: bit-equality with the int8 kernels on the parity suite's geometries, then timing at
config 2 (chr22, 10 k haplotypes) and config 5 geometry.  GPU box only:  python scripts/dev/p2_check.py [quick]"""
import os
import sys
import time

os.environ.setdefault("GNX_LR_P2", "2")  # build the 2-bit planes whatever the padding costs (small test geometries)
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gnomix_amd
from gnomix_amd import synth

GEOMS = [(20500, 1000, 7, 500, 2100), (30500, 1000, 12, 500, 1100), (4037, 100, 7, 50, 24), (4037, 100, 7, 0, 5), (2531, 100, 3, 30, 70), (1999, 64, 2, 32, 130), (3001, 100, 12, 50, 33),
         (2201, 100, 9, 50, 600), (1801, 60, 7, 45, 50), (1503, 100, 16, 50, 9), (2777, 100, 5, 120, 40), (1237, 50, 7, 25, 600),
         (1237, 50, 7, 25, 1), (937, 300, 4, 150, 66), (20500, 1000, 7, 500, 700), (20500, 1000, 12, 500, 300), (2401, 100, 8, 100, 70)]


def check():
    bad = 0
    for (C, M, A, ctx, N) in GEOMS:
        d = synth.synthetic_model(C=C, M=M, A=A, S=5, context=ctx, seed=C + A, smooth=None)
        X = synth.synthetic_X(N, C, seed=N, miss=0.03)
        dev = gnomix_amd.DeviceModel(d)
        Xt = torch.from_numpy(X).cuda()
        Pt = dev.pack_device(Xt)
        # cross-check the torch packer against the host packer
        Ph = dev.pack_x(X)
        assert np.array_equal(Pt.cpu().numpy(), np.asarray(Ph)), "pack_device != gnx_pack_x"
        for f64 in (True, False):
            b_ref = dev.base_predict_device(Xt, f64=f64)
            b_p2 = dev.base_predict_packed_device(Pt, f64=f64)
            torch.cuda.synchronize()
            same = torch.equal(b_ref, b_p2)
            if not same:
                bad += 1
                diff = (b_ref.double() - b_p2.double()).abs()
                print("MISMATCH", (C, M, A, ctx, N), "f64" if f64 else "f32", "max", float(diff.max()), "n", int((diff > 0).sum()), "of", diff.numel(),
                      "first", torch.nonzero(diff > 0)[:3].tolist())
        print("geom", (C, M, A, ctx, N), "ok" if not bad else "BAD(cumulative %d)" % bad, flush=True)
    return bad


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def bench(C, M, A, N, label):
    d = synth.synthetic_model(C=C, M=M, A=A, S=75, context=M // 2, seed=0, smooth=None)
    t0 = time.time()
    dev = gnomix_amd.DeviceModel(d)
    t_load = time.time() - t0
    g = torch.Generator(device="cuda").manual_seed(1)
    Xt = (torch.rand((N, C), device="cuda", generator=g) < 0.4).to(torch.int8)
    Pt = dev.pack_device(Xt)
    t_i8 = timeit(lambda: dev.base_predict_device(Xt))
    t_p2 = timeit(lambda: dev.base_predict_packed_device(Pt))
    same = torch.equal(dev.base_predict_device(Xt), dev.base_predict_packed_device(Pt))
    print(f"{label}: load {t_load:.2f} s  int8 {t_i8:.3f} ms  p2 {t_p2:.3f} ms  ({N * C / 4 / t_p2 / 1e6:.1f} GB/s of 2-bit X, {N * C / t_i8 / 1e6:.1f} GB/s of int8 X)  identical={same}  "
          f"tune={os.environ.get('GNX_P2_TUNE', '-')} bpc={os.environ.get('GNX_LR_BPC', '-')}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode in ("all", "check"):
        bad = check()
        print("CHECK", "FAILED" if bad else "PASSED")
    if mode in ("all", "bench"):
        bench(370500, 1000, 7, 10000, "config2 chr22 A=7")
    if mode in ("aligned",):
        bench(379392, 1024, 7, 10000, "aligned M=1024 C=379392")
    if mode in ("bench12",):
        bench(370500, 1000, 12, 16384, "chr22 A=12")
    if mode in ("c5",):
        bench(1431500, 1000, 12, 25000, "config5a chr1 A=12")

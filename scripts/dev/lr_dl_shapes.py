# the LDS-direct logistic kernel of config 2 (A = 7) in its block shapes: waves x rows per wave, ring depth, window ranges per CU
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth, _lib
N = int(os.environ.get("N", 10000))
d = synth.synthetic_model(seed=0, n_rounds=2, **synth.CHR22)
X = synth.synthetic_X_device(N, d.C, torch.device("cuda", 0), seed=94305)
ref = None
KEYS = ("GNX_LR_DL", "GNX_LR_TUNE", "GNX_LR_NBUF", "GNX_LR_BPC")
for name, env in (("i8 default", {}), ("dl default", {"GNX_LR_DL": "1"}), ("dl 1x16 nbuf3", {"GNX_LR_DL": "1", "GNX_LR_TUNE": "1,16"}),
                  ("dl 1x16 nbuf2", {"GNX_LR_DL": "1", "GNX_LR_TUNE": "1,16", "GNX_LR_NBUF": "2"}),
                  ("dl 1x16 bpc2", {"GNX_LR_DL": "1", "GNX_LR_TUNE": "1,16", "GNX_LR_BPC": "2"}),
                  ("dl 1x16 bpc8", {"GNX_LR_DL": "1", "GNX_LR_TUNE": "1,16", "GNX_LR_BPC": "8"}),
                  ("dl 2x12", {"GNX_LR_DL": "1", "GNX_LR_TUNE": "2,12"})):
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update(env)
    ctx = _lib.Context(0)
    m = gnomix_amd.DeviceModel(d, ctx=ctx)
    B = m.base_predict_device(X); torch.cuda.synchronize()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(20): m.base_predict_device(X)
    torch.cuda.synchronize(); ctx.profile_enable(False)
    ms, n = ctx.profile_get(_lib.K_BASE_LOGISTIC)
    same = "" if ref is None else " identical: %s" % bool(torch.equal(B, ref))
    if ref is None: ref = B.clone()
    print("%-14s %.3f ms  %.2f TB/s of X%s" % (name, ms / n, N * d.C / (ms / n * 1e-3) / 1e12, same), flush=True)
    m.close(); ctx.close()

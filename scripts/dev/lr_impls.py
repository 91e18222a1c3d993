# the logistic pass of config 2 through its int8 kernels (launch knobs are read at gnx_init: one context each)
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth, _lib
N = int(os.environ.get("N", 10000))
d = synth.synthetic_model(seed=0, n_rounds=2, **synth.CHR22)
X = synth.synthetic_X_device(N, d.C, torch.device("cuda", 0), seed=94305)
ref = None
for name, env in (("i8", {}), ("i8_dl", {"GNX_LR_DL": "1"}), ("i8_ws pw2", {"GNX_LR_WS": "1"}), ("i8_ws pw4", {"GNX_LR_WS": "1", "GNX_LR_WS_PW": "4"}),
                  ("i8_w512", {"GNX_LR_W512": "1"}), ("i8_w512 bpc3", {"GNX_LR_W512": "1", "GNX_LR_BPC": "3"}), ("i8_w512 bpc4", {"GNX_LR_W512": "1", "GNX_LR_BPC": "4"}),
                  ("i8_w512 nbuf2", {"GNX_LR_W512": "1", "GNX_LR_NBUF": "2"})):
    for k in ("GNX_LR_DL", "GNX_LR_WS", "GNX_LR_WS_PW", "GNX_LR_W512", "GNX_LR_BPC", "GNX_LR_NBUF"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = _lib.Context(0)
    m = gnomix_amd.DeviceModel(d, ctx=ctx)
    B = m.base_predict_device(X); torch.cuda.synchronize()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(20):
        m.base_predict_device(X)
    torch.cuda.synchronize(); ctx.profile_enable(False)
    ms, n = ctx.profile_get(_lib.K_BASE_LOGISTIC)
    same = "" if ref is None else " identical to i8: %s" % bool(torch.equal(B, ref))
    if ref is None: ref = B.clone()
    print("%-10s %.3f ms  %.2f TB/s of X%s" % (name, ms / n, N * d.C / (ms / n * 1e-3) / 1e12, same), flush=True)
    m.close(); ctx.close()

# the logistic pass on the X the FILE path builds (gnx_infer_gt2: row pitch round_up(C, 64), rows 256-byte aligned)
# next to the bench's resident batch (pitch C = 370 500: most 128-byte row runs straddle cache lines)
import os, sys, torch
sys.path.insert(0, os.getcwd())
import gnomix_amd
from gnomix_amd import synth, _lib
d = synth.synthetic_model(seed=0, n_rounds=100, **synth.CHR22)
m = gnomix_amd.DeviceModel(d, ctx=_lib.Context(0))
N, C = 10000, d.C
X = synth.synthetic_X_device(N, C, "cuda:0", seed=94305)
ldx = (C + 63) // 64 * 64
Xp = torch.zeros((N, ldx), dtype=torch.int8, device="cuda:0")
Xp[:, :C] = X
Xv = Xp[:, :C]
assert Xv.stride(0) == ldx and Xv.data_ptr() % 256 == 0
for name, x in (("pitch C = %d (bench)" % C, X), ("pitch %d (file path)" % ldx, Xv)):
    for _ in range(3): m.infer_device(x)
    torch.cuda.synchronize(); m.ctx.profile_reset(); m.ctx.profile_enable(True)
    for _ in range(40): out = m.infer_device(x)
    torch.cuda.synchronize(); m.ctx.profile_enable(False)
    ms, n = m.ctx.profile_get(_lib.K_BASE_LOGISTIC)
    ms /= n
    print("%-28s k_base_logistic %.4f ms  %.3f TB/s algorithmic = %.3f of 8 TB/s" % (name, ms, (C + d.W * d.A * 4) * N / ms / 1e9, (C + d.W * d.A * 4) * N / ms / 1e9 / 8))

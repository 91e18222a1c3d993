import time, os, sys
t0=time.perf_counter()
os.environ.setdefault("GNX_NO_TORCH","1")
sys.path.insert(0, os.getcwd())
import numpy as np
t1=time.perf_counter()
import gnomix_amd
from gnomix_amd import _lib, synth
t2=time.perf_counter()
ctx=_lib.Context(0)
t3=time.perf_counter()
from gnomix_amd.model import GnxModelData
p="/dev/shm/m.gnx"
if not os.path.exists(p):
    d = synth.synthetic_model(seed=0, n_rounds=100, **synth.CHR22); d.save(p); print("saved"); 
t4=time.perf_counter()
d=GnxModelData.load(p)
t5=time.perf_counter()
m=gnomix_amd.DeviceModel(d, ctx=ctx)
t6=time.perf_counter()
a=ctx.pinned_empty((230*1024*1024,), np.uint8)
t7=time.perf_counter()
print("numpy %.3f  gnomix_amd import %.3f  Context %.3f  model file load %.3f  DeviceModel %.3f  pinned 230MB %.3f"%(t1-t0,t2-t1,t3-t2,t5-t4,t6-t5,t7-t6))

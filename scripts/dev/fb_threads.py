# write_fb on chr22 x 10 000 haplotypes of random probabilities: seconds by formatter thread count (tmpfs), several repetitions
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from gnomix_amd import postprocess as pp, _lib
N, W, A = 10000, 370, 7
rng = np.random.default_rng(0)
proba = rng.dirichlet(np.ones(A) * 0.3, size=(N, W)).astype(np.float32)
meta = {"chm": ["22"] * W, "spos": np.arange(W) * 1000, "epos": np.arange(W) * 1000 + 999, "sgpos": np.arange(W) * 0.2, "egpos": np.arange(W) * 0.2 + 0.2, "wind_index": np.arange(W), "n_snps": [1000] * W}
samples, pops = ["I%d" % i for i in range(N // 2)], ["P%d" % a for a in range(A)]
out = "/dev/shm/fbt"
pp.write_fb(out, meta, proba, pops, samples)
for rep in range(3):
    row = []
    for nt in (0, 8, 12, 14, 15, 16, 20, 24, 32):
        t0 = time.perf_counter(); pp.write_fb(out, meta, proba, pops, samples, n_threads=nt); row.append("%d:%.3f" % (nt, time.perf_counter() - t0))
    print(" ".join(row), flush=True)
print(os.path.getsize(out + ".fb") / 1e6, "MB")
os.remove(out + ".fb")

"""read_vcf of a chr22 x 5 000-sample BGZF query under window sizes / thread counts (GPU box; development aid)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from gnomix_amd import synth, vcfio, _lib
C, N = 370500, 10000
rng = np.random.default_rng(0)
pos = 1000 + 37 * np.arange(C)
work = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
path = os.path.join(work, "probe.vcf")
X = (rng.random((N, 4096)) < 0.4).astype(np.int8)
X = np.tile(X, (1, C // 4096 + 1))[:, :C]
synth.write_vcf_gt2(path, vcfio.pack_gt2(X), N // 2, pos, np.array(["A"] * C), ["C"] * C)
gz = synth.bgzf_compress_file(path, path + ".gz", n_threads=16)
print("text %.2f GB, bgzf %.2f GB" % (os.path.getsize(path) / 1e9, os.path.getsize(gz) / 1e9), flush=True)
ctx = _lib.default_context(0)
for env in ({}, {"GNX_IO_CHUNK": str(4 << 20)}, {"GNX_IO_THREADS": "48"}, {"GNX_IO_THREADS": "64"}, {"GNX_IO_THREADS": "24"}, {"GNX_VCF_ZLIB": "1"}):
    for k in ("GNX_IO_CHUNK", "GNX_IO_THREADS", "GNX_VCF_ZLIB"):
        os.environ.pop(k, None)
    os.environ.update(env)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        v = vcfio.read_vcf(gz, chm="22", ctx=ctx)
        best = min(best, time.perf_counter() - t0)
        del v
    t0 = time.perf_counter(); v = vcfio.read_vcf(path, chm="22", ctx=ctx); tp = time.perf_counter() - t0; del v
    print(env, "bgzf %.3f s   plain %.3f s" % (best, tp), flush=True)

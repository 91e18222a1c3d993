"""Stage timings of the native VCF reader / writers at config-2 size on this box's host cores (no GPU needed unless --pinned).
   python scripts/dev/vcf_io_probe.py [--samples 5000 --variants 370500 --dir /dev/shm --pinned]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnomix_amd import vcfio, synth, postprocess as pp, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=5000)
ap.add_argument("--variants", type=int, default=370500)
ap.add_argument("--dir", default="/dev/shm")
ap.add_argument("--pinned", action="store_true")
ap.add_argument("--threads", default="16,32,64,128,256")
a = ap.parse_args()
ns, V = a.samples, a.variants
rng = np.random.default_rng(0)
ldg = (2 * ns + 15) // 16 * 4
G = rng.integers(0, 256, (V, ldg), dtype=np.uint8) & 0x55
G[rng.random((V, ldg)) < 0.02] = 2
pos = 16_000_000 + np.cumsum(rng.integers(1, 180, V))
path = os.path.join(a.dir, "gnx_probe.vcf")
for nt in (1, 8, 0):
    t = time.time(); synth.write_vcf_gt2(path, G, ns, pos, ["A"] * V, ["C"] * V, n_threads=nt); dt = time.time() - t
    sz = os.path.getsize(path)
    print("write_vcf threads=%d: %.2f s  %.2f GB/s (%.2f GB)" % (nt, dt, sz / dt / 1e9, sz / 1e9), flush=True)
ctx = _lib.Context(0) if a.pinned else None
for nt in [int(x) for x in a.threads.split(",")]:
    for rep in range(2):
        t = time.time(); d = vcfio.read_vcf(path, chm="22", n_threads=nt, ctx=ctx); dt = time.time() - t
        i = d.info
        print("read threads=%3d: %.3f s  %.2f GB/s | load %.3f parse %.3f alloc %.3f merge %.3f | fast %d pinned %d" %
              (nt, dt, sz / dt / 1e9, i.seconds_load, i.seconds_parse, i.seconds_alloc, i.seconds_merge, i.n_fast_lines, i.gt2_pinned), flush=True)
        t = time.time(); del d; print("   free %.3f s" % (time.time() - t))
# writers at chr22 size
N, W, A = 2 * ns, 370, 7
proba = rng.random((N, W, A), dtype=np.float32); proba /= proba.sum(-1, keepdims=True)
labels = rng.integers(0, A, (N, W)).astype(np.int32)
mpos = np.arange(W * 1000 + 500) * 50
meta = pp.get_meta_data("22", mpos, mpos[::3], W, 1000, np.array([0, 10 ** 9]), np.array([0.0, 70.0]))
out = os.path.join(a.dir, "gnx_probe_out")
for nt in (1, 16, 64, 0):
    t = time.time(); pp.write_msp(out, meta, labels, list("ABCDEFG"), ["S%d" % i for i in range(ns)], n_threads=nt); t1 = time.time()
    pp.write_fb(out, meta, proba, list("ABCDEFG"), ["S%d" % i for i in range(ns)], n_threads=nt); t2 = time.time()
    print("writers threads=%3d: msp %.3f s (%.1f MB)  fb %.3f s (%.1f MB, %.2f GB/s)" % (nt, t1 - t, os.path.getsize(out + ".msp") / 1e6, t2 - t1,
          os.path.getsize(out + ".fb") / 1e6, os.path.getsize(out + ".fb") / (t2 - t1) / 1e9), flush=True)
for f in (path, out + ".msp", out + ".fb"):
    os.remove(f)

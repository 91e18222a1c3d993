import sys, zlib, numpy as np, time
sys.path.insert(0, "/root/repo")
from gnomix_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(0)
# VCF-like text: fixed fields + genotypes of 5000 samples per line, allele freq varying per line
lines = []
for v in range(60):
    p = rng.random() * 0.9 + 0.05
    a = (rng.random(10000) < p).astype(np.uint8)
    g = "\t".join("%d|%d" % (a[2*i], a[2*i+1]) for i in range(5000))
    lines.append("22\t%d\trs%d\tA\tC\t.\tPASS\t.\tGT\t%s\n" % (1000 + 37 * v, v, g))
data = "".join(lines).encode()
for level in (1, 6):
    blocks = []
    for i in range(0, len(data), 65280):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        blocks.append((np.frombuffer(c.compress(data[i:i+65280]) + c.flush(), np.uint8).copy(), min(65280, len(data) - i)))
    out = np.zeros(65536, np.uint8)
    reps = 8
    zb = [bytes(z) for z, s in blocks]
    bg = bz = 1e9
    for _ in range(8):
        t0 = time.perf_counter()
        for _ in range(reps):
            for z, s in blocks:
                lib.gnx_io_inflate_raw(z.ctypes.data, len(z), out.ctypes.data, s)
        t1 = time.perf_counter()
        for _ in range(reps):
            for b in zb:
                zlib.decompress(b, -15)
        t2 = time.perf_counter()
        bg = min(bg, t1 - t0); bz = min(bz, t2 - t1)
    tot = reps * len(data) / 1e6
    print("level %d: gnx %.0f MB/s  zlib %.0f MB/s  ratio %.2f  compression %.1fx" % (level, tot / bg, tot / bz, bz / bg, len(data) / sum(len(z) for z, _ in blocks)))

import sys, zlib, ctypes as C, numpy as np, time
sys.path.insert(0, "/root/repo")
import gnomix_amd
lib = gnomix_amd.load_library()
rng = np.random.default_rng(0)
def raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
    return c.compress(data) + c.flush()
def check(data, **kw):
    z = raw(data, **kw)
    out = np.zeros(len(data) + 64, np.uint8); out[len(data):] = 0xAB
    zb = np.frombuffer(z + b"\0" * 0, np.uint8).copy()
    rc = lib.gnx_io_inflate_raw(zb.ctypes.data, len(z), out.ctypes.data, len(data))
    assert rc == 0, (rc, kw, len(data))
    assert bytes(out[:len(data)]) == data, kw
    assert (out[len(data):] == 0xAB).all(), "overrun"
    # wrong size / truncated input are rejected
    if len(data) > 10:
        assert lib.gnx_io_inflate_raw(zb.ctypes.data, len(z), out.ctypes.data, len(data) - 1) != 0
        assert lib.gnx_io_inflate_raw(zb.ctypes.data, len(z) // 2, out.ctypes.data, len(data)) != 0
n = 0
texts = []
gt = ("\t".join(rng.choice(["0|0", "0|1", "1|0", "1|1", ".|."], 4000, p=[.5, .2, .2, .09, .01])) + "\n").encode()
texts.append(gt * 12)
texts.append(bytes(rng.integers(0, 256, 70000, dtype=np.uint8)))          # incompressible
texts.append(bytes(rng.integers(0, 4, 65000, dtype=np.uint8)))            # small alphabet: long codes rare
texts.append(b"a" * 65536)                                                 # distance 1 runs
texts.append(b"abc" * 20000)
texts.append(b"")
texts.append(b"x")
texts.append(("".join(chr(32 + (i * 7919) % 90) for i in range(50000))).encode())
skew = rng.choice(256, 60000, p=(lambda p: p / p.sum())(1.0 / (1 + np.arange(256)) ** 2.5)).astype(np.uint8)
texts.append(bytes(skew))                                                  # skewed alphabet: 15-bit codes, subtables
for t in texts:
    for level in (0, 1, 4, 6, 9):
        for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            check(t, level=level, strategy=strat); n += 1
for i in range(300):
    L = int(rng.integers(1, 70000))
    kind = i % 3
    if kind == 0: d = bytes(rng.integers(0, int(rng.integers(2, 256)), L, dtype=np.uint8))
    elif kind == 1: d = (gt * 30)[int(rng.integers(0, 1000)):][:L]
    else: d = bytes(np.repeat(rng.integers(0, 256, L // 7 + 1, dtype=np.uint8), 7)[:L])
    check(d, level=int(rng.integers(1, 10))); n += 1
print("inflate ok:", n, "streams")
# throughput on genotype text, one thread
data = gt * 4
blocks = [raw(data[i:i + 65280]) for i in range(0, len(data), 65280)]
zs = [np.frombuffer(b, np.uint8).copy() for b in blocks]
out = np.zeros(65536, np.uint8)
sizes = [min(65280, len(data) - i) for i in range(0, len(data), 65280)]
t0 = time.perf_counter()
for _ in range(200):
    for z, s in zip(zs, sizes):
        lib.gnx_io_inflate_raw(z.ctypes.data, len(z), out.ctypes.data, s)
t1 = time.perf_counter()
for _ in range(200):
    for b in blocks:
        zlib.decompress(b, -15)
t2 = time.perf_counter()
tot = 200 * len(data) / 1e6
print("gnx %.0f MB/s   zlib %.0f MB/s  (ratio %.2f, compression %.1fx)" % (tot / (t1 - t0), tot / (t2 - t1), (t2 - t1) / (t1 - t0), len(data) / sum(map(len, blocks))))

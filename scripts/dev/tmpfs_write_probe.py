# how fast can ONE thread put 305 MB into a fresh file on tmpfs (the .fb writer's last step)? and into an existing one?
import os, time, numpy as np
buf = np.random.default_rng(0).integers(32, 127, 305_000_000, dtype=np.uint8).tobytes()
p = "/dev/shm/wprobe.bin"
for mode in ("fresh", "fresh", "rewrite", "rewrite"):
    flags = os.O_WRONLY | os.O_CREAT | (os.O_TRUNC if mode == "fresh" else 0)
    if mode == "fresh" and os.path.exists(p): os.remove(p)
    t0 = time.perf_counter(); fd = os.open(p, flags, 0o644)
    off = 0
    while off < len(buf): off += os.write(fd, memoryview(buf)[off:off + (1 << 30)])
    os.close(fd); dt = time.perf_counter() - t0
    print("%s: %.3f s, %.2f GB/s" % (mode, dt, len(buf) / dt / 1e9), flush=True)
os.remove(p)

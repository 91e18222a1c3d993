"""One process, several GPUs: the file path of SURVEY 8e / north_star ("haplotypes are sharded across the 8 GPUs of one node,
embarrassingly parallel, gather-only") without a collective.

The query is parsed ONCE into one page-locked variant-major 2-bit matrix (gnx_vcf_read); every device holds a replica of the
model in a context of its own and is driven by a host thread of its own (a gnx_ctx is not thread-safe, different contexts are
independent: include/gnomix_hip.h).  Individuals shard contiguously; a device uploads only ITS columns of the variant rows
(gnx_infer_gt2_range / gnx_phase_gt2_range: one strided copy), builds its X in its own HBM and writes its row block of the shared
page-locked outputs, which the writers then format once.  Nothing is exchanged between devices: the reference has no analogue
(gnomix.py:37-100 is one process on one matrix; src/model.py:205-210 loops over individuals), this is its outer loop cut by
individual.
"""
from __future__ import annotations

import os
import threading

import numpy as np

from . import _lib
from .model import DeviceModel


def visible_devices():
    """device ordinals to use: GNX_DEVICES="0,2,3" or every GPU the process sees.  Every ordinal is checked against
    gnx_device_count(): a wrong list fails here, with a message, instead of inside gnx_init on a worker thread."""
    n = max(int(_lib.load().gnx_device_count()), 0)
    env = os.environ.get("GNX_DEVICES", "").strip()
    if not env:
        return list(range(max(n, 1)))
    try:
        devs = [int(t) for t in env.split(",") if t.strip() != ""]
    except ValueError:
        raise _lib.GnxError(_lib.GNX_EINVAL, "GNX_DEVICES=%r: a comma-separated list of device ordinals is expected" % env) from None
    if not devs:
        raise _lib.GnxError(_lib.GNX_EINVAL, "GNX_DEVICES=%r names no device" % env)
    bad = [d for d in devs if d < 0 or d >= n]
    if bad:
        raise _lib.GnxError(_lib.GNX_EINVAL, "GNX_DEVICES=%r: ordinals %s are outside the %d visible device(s)" % (env, bad, n))
    return devs


def shard_individuals(n_ind, k, weights=None):
    """contiguous blocks of whole individuals, block starts EVEN (a block's first haplotype is a multiple of 4: whole bytes of
    the 2-bit rows).  -> [(first individual, count), ...] with no empty block (fewer blocks than k when n_ind is small)."""
    k = max(1, int(k))
    w = np.ones(k) if weights is None else np.asarray(weights, dtype=float)
    edges = np.round(np.cumsum(w) / w.sum() * n_ind).astype(int)
    edges = np.minimum((edges + 1) // 2 * 2, n_ind)   # even starts
    edges[-1] = n_ind
    out, lo = [], 0
    for hi in edges:
        hi = max(int(hi), lo)
        if hi > lo:
            out.append((lo, hi - lo))
        lo = hi
    return out


class DeviceGroupError(_lib.GnxError):
    """one or more devices of a DeviceGroup failed: .failures = [(position in the group, device ordinal, exception), ...]"""

    def __init__(self, failures):
        self.failures = failures
        code = next((e.code for _, _, e in failures if isinstance(e, _lib.GnxError)), _lib.GNX_EHIP)
        super().__init__(code, "; ".join("device %d (context %d of the group): %s" % (d, i, e) for i, d, e in failures))


class DeviceGroup:
    """replicas of one model, one per entry of `devices` (the same ordinal may appear several times: several contexts on one
    GPU — how the tests exercise the sharded path on a single-GPU box).

    `n_ind`: the number of individuals the group will be asked to shard, when known: no more replicas are made than there are
    shards (a one-sample query on an 8-GPU node makes none).  Replicas are loaded on one thread per device (a model upload is a
    host-side re-layout plus one copy: the devices do not wait for each other).  The group owns the contexts it made — close()
    (or a failed job) releases them — never the caller's `first`."""

    def __init__(self, data, devices=None, first=None, n_ind=None):
        devices = list(devices) if devices is not None else visible_devices()
        if not devices:
            raise _lib.GnxError(_lib.GNX_EINVAL, "DeviceGroup: no device")
        n_dev = max(int(_lib.load().gnx_device_count()), 0)
        bad = [d for d in devices if not isinstance(d, (int, np.integer)) or d < 0 or d >= max(n_dev, 1)]
        if bad:
            raise _lib.GnxError(_lib.GNX_EINVAL, "DeviceGroup: ordinals %s are outside the %d visible device(s)" % (bad, n_dev))
        if n_ind is not None:
            devices = devices[:max(1, len(shard_individuals(int(n_ind), len(devices))))]
        self.devices = devices
        self.data = data
        self._own = []
        models = [None] * len(devices)
        errs = []

        def load(i, d):
            try:
                models[i] = DeviceModel(data, ctx=_lib.Context(int(d)))
            except BaseException as e:   # noqa: BLE001 - reported below
                errs.append((i, int(d), e))
        ts = []
        for i, d in enumerate(devices):
            if i == 0 and first is not None:
                models[0] = first                       # a DeviceModel the caller already loaded (its context is kept)
            else:
                ts.append(threading.Thread(target=load, args=(i, d), daemon=True))
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        self.models = models
        self._own = [m for i, m in enumerate(models) if m is not None and not (i == 0 and first is not None)]
        if errs:
            self.close()
            raise DeviceGroupError(sorted(errs, key=lambda f: f[0]))
        m0 = self.models[0]
        self.W, self.A, self.C = m0.W, m0.A, m0.C
        self._sync_state()

    def _sync_state(self):
        """what gnomix.py pokes into the loaded model after the fact (gnomix.py:365-370: model.calibrate, model.smooth.calibrate)
        lives in the FIRST model: every replica follows it before each run, or the shards of one output would differ"""
        m0 = self.models[0]
        want = bool(getattr(m0, "calibrated", False))
        for m in self.models[1:]:
            if bool(getattr(m, "calibrated", False)) != want:
                m.set_calibrate(want)

    def close(self):
        """releases the replicas (and their contexts) this group made; the caller's `first` stays open"""
        for m in self._own:
            try:
                ctx = m.ctx
                m.close()
                ctx.close()
            except Exception:   # noqa: BLE001 - closing is best effort
                pass
        self._own = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _run(self, jobs):
        """jobs: [(model, callable)] -> each on its own thread (ctypes releases the GIL for the duration of a library call).  Every
        job runs to its end (a library call cannot be cancelled); if any failed the group closes its contexts and ONE error names
        every failed device."""
        errs = [None] * len(jobs)

        def work(i, fn):
            try:
                fn()
            except BaseException as e:   # noqa: BLE001 - re-raised on the caller's thread
                errs[i] = e
        ts = [threading.Thread(target=work, args=(i, fn), daemon=True) for i, (_, fn) in enumerate(jobs)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        failures = [(i, int(self.devices[self.models.index(m)]) if m in self.models else -1, e) for i, ((m, _), e) in enumerate(zip(jobs, errs)) if e is not None]
        if failures:
            self.close()
            for _, _, e in failures:
                if isinstance(e, (KeyboardInterrupt, SystemExit)):
                    raise e
            raise DeviceGroupError(failures)

    def _outputs(self, N, proba_dtype, out):
        m0 = self.models[0]
        pd = np.dtype(proba_dtype or m0.proba_dtype())
        if out is not None:
            p, lab = out
        else:
            p = m0.ctx.pinned_empty((N, self.W, self.A), pd)
            lab = m0.ctx.pinned_empty((N, self.W), np.int32)
        if p.shape != (N, self.W, self.A) or lab.shape != (N, self.W) or lab.dtype != np.int32 or not (p.flags.c_contiguous and lab.flags.c_contiguous):
            raise ValueError("out = (proba (N, W, A) float32/float64, labels (N, W) int32), C-contiguous")
        return p, lab

    def infer_gt2(self, G, N, src, proba_dtype=None, out=None):
        """DeviceModel.infer_gt2 with the individuals cut over the group's devices -> (proba, labels)"""
        m0 = self.models[0]
        self._sync_state()
        G, N, src = m0._gt2_args(G, N, src)
        if N % 2:
            raise ValueError("N = 2 * samples")
        p, lab = self._outputs(N, proba_dtype, out)
        WA, W = self.W * self.A, self.W
        jobs = []
        for m, (i0, n) in zip(self.models, shard_individuals(N // 2, len(self.models))):
            h0, nh = 2 * i0, 2 * n
            p32 = p.ctypes.data + h0 * WA * 4 if p.dtype == np.float32 else None
            p64 = p.ctypes.data + h0 * WA * 8 if p.dtype == np.float64 else None

            def fn(m=m, h0=h0, nh=nh, p32=p32, p64=p64):
                m.ctx.check(m.lib.gnx_infer_gt2_range(m.h, G.ctypes.data, G.shape[0], G.shape[1], h0, nh, src.ctypes.data, p32, p64,
                                                      lab.ctypes.data + h0 * W * 4))
            jobs.append((m, fn))
        self._run(jobs)
        return p, lab

    def phase_gt2(self, G, N, src, out_cols=None, max_it=50, proba_dtype=None, out=None):
        """DeviceModel.phase_gt2 cut the same way -> (G_phased or None, proba, labels, n_switches)"""
        m0 = self.models[0]
        self._sync_state()
        G, N, src = m0._gt2_args(G, N, src)
        if N % 2:
            raise ValueError("phase_gt2: N = 2 * individuals")
        p, lab = self._outputs(N, proba_dtype, out)
        nsw = m0.ctx.pinned_empty((N // 2,), np.int32)
        Go = cols = None
        if out_cols is not None:
            cols = np.ascontiguousarray(out_cols, dtype=np.int32)
            # page-locked (and portable: every device writes its byte columns of these rows with a strided D2H copy — pageable rows
            # would go through the runtime's staging buffer and serialise the devices)
            Go = m0.ctx.pinned_empty((len(cols), G.shape[1]), np.uint8)
            Go[...] = 0
        WA, W = self.W * self.A, self.W
        jobs = []
        for m, (i0, n) in zip(self.models, shard_individuals(N // 2, len(self.models))):
            h0, nh = 2 * i0, 2 * n
            p32 = p.ctypes.data + h0 * WA * 4 if p.dtype == np.float32 else None
            p64 = p.ctypes.data + h0 * WA * 8 if p.dtype == np.float64 else None

            def fn(m=m, h0=h0, nh=nh, p32=p32, p64=p64, i0=i0):
                m.ctx.check(m.lib.gnx_phase_gt2_range(m.h, G.ctypes.data, G.shape[0], G.shape[1], h0, nh, src.ctypes.data, int(max_it),
                                                      cols.ctypes.data if cols is not None else None, len(cols) if cols is not None else 0,
                                                      Go.ctypes.data if Go is not None else None, G.shape[1], p32, p64,
                                                      lab.ctypes.data + h0 * W * 4, nsw.ctypes.data + i0 * 4))
            jobs.append((m, fn))
        self._run(jobs)
        return Go, p, lab, nsw

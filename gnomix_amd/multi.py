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
    """device ordinals to use: GNX_DEVICES="0,2,3" or every GPU the process sees"""
    env = os.environ.get("GNX_DEVICES", "").strip()
    if env:
        return [int(t) for t in env.split(",") if t.strip() != ""]
    n = _lib.load().gnx_device_count()
    return list(range(max(n, 1)))


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


class DeviceGroup:
    """replicas of one model, one per entry of `devices` (the same ordinal may appear several times: several contexts on one
    GPU — how the tests exercise the sharded path on a single-GPU box)"""

    def __init__(self, data, devices=None, first=None):
        devices = list(devices) if devices is not None else visible_devices()
        self.devices = devices
        self.models = []
        for i, d in enumerate(devices):
            if i == 0 and first is not None:
                self.models.append(first)            # a DeviceModel the caller already loaded (its context is kept)
            else:
                self.models.append(DeviceModel(data, ctx=_lib.Context(d)))
        m0 = self.models[0]
        self.W, self.A, self.C = m0.W, m0.A, m0.C
        self.data = data

    def close(self):
        for m in self.models:
            m.close()

    def _run(self, jobs):
        """jobs: [(model, callable)] -> each on its own thread (ctypes releases the GIL for the duration of a library call)"""
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
        for e in errs:
            if e is not None:
                raise e

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
        G, N, src = m0._gt2_args(G, N, src)
        if N % 2:
            raise ValueError("phase_gt2: N = 2 * individuals")
        p, lab = self._outputs(N, proba_dtype, out)
        nsw = np.empty((N // 2,), np.int32)
        Go = cols = None
        if out_cols is not None:
            cols = np.ascontiguousarray(out_cols, dtype=np.int32)
            Go = np.zeros((len(cols), G.shape[1]), np.uint8)
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

"""-m gpu: every C-ABI entry binds its context's device ("one gnx_ctx per device ... different ctxs may run on different
threads"; 8e).  HIP's current device belongs to the calling THREAD and a new thread starts on device 0, while hipMalloc and kernel
launches go to the current device — so an entry point that does not bind would, on a multi-GPU node, put a context's workspaces on
GPU 0 and launch GPU-d kernels on them.  Each `_dev` entry (and the host-pointer / file forms) is driven here from a FRESH thread on
the LAST visible device, and every live workspace must report that device (gnx_debug_ws_devices).  On a one-GPU box the device
assertions are trivially true (the outputs are still compared); on the first 8-GPU node they are the tripwire."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ga():
    import gnomix_amd
    gnomix_amd.load_library()
    return gnomix_amd


def _in_fresh_thread(fn):
    box = {}

    def run():
        try:
            box["out"] = fn()
        except BaseException as e:  # noqa: BLE001 - re-raised in the caller's thread
            box["err"] = e

    t = threading.Thread(target=run)
    t.start()
    t.join()
    if "err" in box:
        raise box["err"]
    return box["out"]


def _last_device(ga):
    n = ga.load_library().gnx_device_count()
    assert n >= 1
    return n - 1


@pytest.mark.parametrize("smooth,A", [("xgb", 7), ("crf", 12), ("cnn", 3)])
def test_dev_entries_from_a_fresh_thread_stay_on_the_contexts_device(ga, smooth, A):
    import torch
    from gnomix_amd import synth, _lib
    dev_id = _last_device(ga)
    C, M, S, N = 6037, 100, 21, 66
    d = synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=5, seed=11, smooth=smooth)
    X = synth.synthetic_X(N, C, seed=5, miss=0.03)
    ctx = _lib.Context(dev_id)
    dm = ga.DeviceModel(d, ctx=ctx)
    p_ref, l_ref = dm.infer(X)                      # host-pointer form, main thread
    tdev = torch.device("cuda", dev_id)

    def job():
        # a fresh thread: torch / HIP current device is 0 here, whatever the context's device is
        before = torch.cuda.current_device()
        X_t = torch.from_numpy(X).to(tdev)
        out = {}
        with torch.cuda.device(0):                  # the CALLER's device stays 0 across every call
            p, lab = dm.infer_device(X_t)
            out["infer"] = (p.cpu().numpy(), lab.cpu().numpy())
            B32 = dm.base_predict_device(X_t)
            B64 = dm.base_predict_device(X_t, f64=True)
            out["smooth"] = tuple(t.cpu().numpy() for t in dm.smooth_predict_device(B64 if smooth == "crf" else B32))
            P_t = dm.pack_device(X_t)
            out["infer_packed"] = tuple(t.cpu().numpy() for t in dm.infer_packed_device(P_t))
            out["base_packed"] = dm.base_predict_packed_device(P_t, f64=True).cpu().numpy()
            out["b64"] = B64.cpu().numpy()
            if smooth == "xgb":
                Xg, Pg = X_t.clone(), P_t.clone()
                Y, ns = dm.gnofix_device(Xg, B64)
                Y2, ns2 = dm.gnofix_packed_device(Pg, B64)
                out["gnofix"] = (Y.cpu().numpy(), ns.cpu().numpy(), Xg.cpu().numpy(), Y2.cpu().numpy(), ns2.cpu().numpy(),
                                 dm.pack_device(Xg).cpu().numpy(), Pg.cpu().numpy())
            ctx.synchronize()
            assert torch.cuda.current_device() == 0
        assert torch.cuda.current_device() == before
        # host-pointer and file forms from this thread as well
        out["host"] = dm.infer(X)
        out["host_packed"] = dm.infer_packed(dm.pack_x(X))
        out["ws"] = ctx.workspace_devices()
        return out

    out = _in_fresh_thread(job)
    assert len(out["ws"]) >= 3 and all(v == dev_id for v in out["ws"]), out["ws"]
    for k in ("infer", "infer_packed", "host", "host_packed"):
        p, l = out[k]
        assert np.array_equal(l, l_ref), k
        assert np.array_equal(p.astype(p_ref.dtype), p_ref) or np.max(np.abs(p - p_ref)) <= 1e-6, k
    assert np.array_equal(out["smooth"][1], l_ref)
    assert np.array_equal(out["base_packed"], out["b64"])
    if smooth == "xgb":
        Y, ns, Xg, Y2, ns2, Pg_from_x, Pg = out["gnofix"]
        assert np.array_equal(Y, Y2) and np.array_equal(ns, ns2) and np.array_equal(Pg_from_x[:, :Pg.shape[1]], Pg[:, :Pg_from_x.shape[1]])
    dm.close()
    ctx.close()


def test_two_contexts_driven_from_one_thread(ga):
    """torch tensors on two devices, two contexts, ONE thread — calls interleaved, no hipSetDevice by the caller.
    With one visible GPU both contexts sit on device 0 (the interleaving is still exercised)."""
    import torch
    from gnomix_amd import synth, _lib
    d1 = _last_device(ga)
    C, M, A, S, N = 4037, 100, 5, 11, 40
    d = synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=4, seed=2)
    X = synth.synthetic_X(N, C, seed=9)
    ctxs = [_lib.Context(0), _lib.Context(d1)]
    dms = [ga.DeviceModel(d, ctx=c) for c in ctxs]
    p_ref, l_ref = dms[0].infer(X)
    Xs = [torch.from_numpy(X).to(torch.device("cuda", c.device)) for c in ctxs]
    outs = []
    for rep in range(2):
        for dm, X_t in zip(dms, Xs):
            outs.append((dm.ctx.device, dm.infer_device(X_t), dm.infer_packed_device(dm.pack_device(X_t))))
    for dev_id, (p, l), (p2, l2) in outs:
        assert np.array_equal(l.cpu().numpy(), l_ref) and np.array_equal(l2.cpu().numpy(), l_ref)
        assert np.array_equal(p.cpu().numpy(), p_ref.astype(np.float32)) and np.array_equal(p2.cpu().numpy(), p_ref.astype(np.float32))
        assert p.device.index == dev_id
    for c in ctxs:
        ws = c.workspace_devices()
        assert ws and all(v == c.device for v in ws), (c.device, ws)
    for dm in dms:
        dm.close()
    for c in ctxs:
        c.close()


def test_file_forms_from_a_fresh_thread(ga, tmp_path):
    """gnx_infer_gt2 / gnx_phase_gt2 (the one-process multi-GPU file path runs them on one thread per device, multi.py)"""
    from gnomix_amd import synth, vcfio, _lib
    dev_id = _last_device(ga)
    C, M, A, S, N = 3037, 100, 4, 11, 24
    d = synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=4, seed=4)
    X = synth.synthetic_X(N, C, seed=3)
    ctx = _lib.Context(dev_id)
    dm = ga.DeviceModel(d, ctx=ctx)
    p_ref, l_ref = dm.infer(X)
    G = vcfio.pack_gt2(X)
    src = np.arange(C, dtype=np.int32)

    def job():
        a = dm.infer_gt2(G, N, src)
        b = dm.phase_gt2(G, N, src, out_cols=np.arange(C, dtype=np.int32), max_it=5)
        return a, b, ctx.workspace_devices()

    (p, l), ph, ws = _in_fresh_thread(job)
    assert np.array_equal(l, l_ref) and np.array_equal(p, p_ref)
    assert ws and all(v == dev_id for v in ws), ws
    dm.close()
    ctx.close()


def test_packed_rows_in_the_int8_workspace_survive_the_widening_fallback(ga, monkeypatch):
    """The file routes build their 2-bit rows in the context's int8 workspace (ws_xu).  When the 2-bit launcher declines a model
    (no instantiation fits the LDS) the rows are widened to int8 — which must not happen in place (overlapping strides
    corrupted B silently).  GNX_P2_DECLINE=1 makes the launcher decline."""
    from gnomix_amd import synth, vcfio, _lib
    C, M, A, S, N = 6037, 100, 7, 21, 44
    d = synth.synthetic_model(C=C, M=M, A=A, S=S, n_rounds=5, seed=6)
    X = synth.synthetic_X(N, C, seed=8, miss=0.04)
    G = vcfio.pack_gt2(X)
    src = np.arange(C, dtype=np.int32)
    monkeypatch.setenv("GNX_LR_P2", "2")
    ref_ctx = _lib.Context(0)
    ref = ga.DeviceModel(d, ctx=ref_ctx)
    p_ref, l_ref = ref.infer_gt2(G, N, src)
    assert np.array_equal(l_ref, ref.infer(X)[1])
    ph_ref = ref.phase_gt2(G, N, src, out_cols=np.arange(C, dtype=np.int32), max_it=8)
    monkeypatch.setenv("GNX_P2_DECLINE", "1")
    monkeypatch.setenv("GNX_HOST_BATCH", "16")      # several batches: the workspace is reused
    ctx = _lib.Context(0)
    dm = ga.DeviceModel(d, ctx=ctx)
    p, l = dm.infer_gt2(G, N, src)
    assert np.array_equal(p, p_ref) and np.array_equal(l, l_ref)
    ph = dm.phase_gt2(G, N, src, out_cols=np.arange(C, dtype=np.int32), max_it=8)
    for a, b in zip(ph, ph_ref):
        if a is None or b is None:
            assert a is None and b is None
        else:
            assert np.array_equal(np.asarray(a), np.asarray(b))
    for m in (dm, ref):
        m.close()
    ctx.close()
    ref_ctx.close()

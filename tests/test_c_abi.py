"""The C ABI from plain C (examples/abi_smoke.c): compiles and links against include/gnomix_hip.h + libgnomix_hip.so with
gcc only; without a GPU it must fail loudly (exit 2, message on stderr); on a GPU its output must equal what the Python
layer computes for the identical model."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "gnomix_amd")
    subprocess.check_call(["gcc", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "abi_smoke.c"),
                           "-L", libdir, "-lgnomix_hip", "-Wl,-rpath," + libdir, "-lm", "-o", exe])
    return exe


def _lcg_stream():
    state = 12345
    while True:
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        yield (state >> 8) / 16777216.0


def test_c_program_builds_and_fails_loudly_without_gpu(tmp_path):
    import gnomix_amd
    gnomix_amd.load_library()   # the .so exists (built by __graft_entry__.build / make)
    exe = _build(tmp_path)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    r = subprocess.run([exe], capture_output=True, text=True)
    if has_gpu:
        assert r.returncode == 0 and r.stdout.startswith("labels ")
    else:
        assert r.returncode == 2 and "gnx_init failed" in r.stderr and r.stdout == ""


@pytest.mark.gpu
def test_c_program_matches_python_layer(tmp_path):
    import gnomix_amd
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    _, ls, _, ps = r.stdout.split()
    C, M, A, CTX, N = 1237, 50, 4, 25, 24
    W, rem = C // M, C % M
    ldc = M + 2 * CTX + rem
    g = _lcg_stream()
    take = lambda n: np.array([next(g) for _ in range(n)])
    coef = ((take(W * A * ldc) - 0.5) * 0.2).reshape(W, A, ldc)
    icpt = (take(W * A) - 0.5).reshape(W, A)
    eye = np.eye(A)
    state = (take(A * A) - 0.5).reshape(A, A) * 2.0 + 4.0 * eye
    trans = (take(A * A) - 0.5).reshape(A, A) + 3.0 * eye
    u = take(N * C)
    X = np.where(u < 0.02, 2, np.where(u < 0.45, 1, 0)).astype(np.int8).reshape(N, C)
    d = gnomix_amd.GnxModelData(C=C, M=M, A=A, S=75, context=CTX, base_kind="logistic", smooth_kind="crf", lr_coef=coef,
                                lr_intercept=icpt, crf_state=state, crf_trans=trans)
    proba, labels = gnomix_amd.DeviceModel(d).infer(X)
    want = int(np.sum((np.arange(labels.size, dtype=np.int64) + 1) * labels.reshape(-1).astype(np.int64)) & 0x7FFFFFFF)
    assert int(ls) == want
    assert abs(float(ps) - float(proba.sum())) < 1e-5

"""-m gpu: the RCCL (backend "nccl") flavour of the N>1 path, as far as ONE GPU can exercise it, and bench.py's
launch contract (`--gpus N` spawns its own ranks or refuses).  Every case runs in a fresh interpreter: a process
group / a HIP context per rank is the product's process model (one process per GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORLD1 = r'''
import os, sys
sys.path.insert(0, %(root)r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%(port)d")
import numpy as np, torch, torch.distributed as dist
import gnomix_amd
from gnomix_amd import synth, _lib
from gnomix_amd.dist import infer_sharded
from oracle import gnx_oracle as O
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
d = synth.synthetic_model(C=6037, M=100, A=7, S=21, n_rounds=12, seed=3)
dev = gnomix_amd.DeviceModel(d, ctx=_lib.Context(0))
X = synth.synthetic_X(70, d.C, seed=1)
Xd = torch.from_numpy(X).cuda()
p_all, l_all = infer_sharded(dev.infer_device, Xd, dst=None)     # all_gather over RCCL, CUDA tensors end to end
p_dst, l_dst = infer_sharded(dev.infer_device, Xd, dst=0)        # gather to rank 0
p_loc, l_loc, bounds = infer_sharded(dev.infer_device, Xd, gather=False)
torch.cuda.synchronize()
assert p_all.is_cuda and l_all.is_cuda and bounds == (0, 70)
O.build()
T = O.Trees(d.tree_off, d.left, d.right, d.feat, d.cond, d.tree_class, d.A, d.base_score)
p_ref, l_ref = O.smooth_xgb(T, O.base_lr(X, d.M, d.context, d.lr_coef, d.lr_intercept), d.S)
for p, l in ((p_all, l_all), (p_dst, l_dst), (p_loc, l_loc)):
    assert np.array_equal(l.cpu().numpy(), l_ref)
    assert np.max(np.abs(p.cpu().numpy() - p_ref)) <= 2.4e-7
t = torch.ones(4, device="cuda"); dist.all_reduce(t); assert float(t.sum()) == 4.0
print("NCCL_WORLD1_OK", dist.get_backend())
dist.destroy_process_group()
'''


def _env():
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    return e


def _port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_nccl_world1_infer_sharded_with_device_model():
    r = subprocess.run([sys.executable, "-c", _WORLD1 % {"root": ROOT, "port": _port()}], capture_output=True, text=True,
                       timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0 and "NCCL_WORLD1_OK nccl" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_force_dist_runs_the_rccl_path_on_one_gpu():
    e = _env()
    e["GNX_BENCH_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--haps", "512",
                        "--cpu-seconds", "0", "--e2e-steps", "0", "--configs", "0"], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    assert r.stdout.strip().splitlines()[-1].startswith("{"), r.stdout[-500:]   # the JSON line is the last line
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["dist_backend"] == "nccl" and line["value"] > 0
    assert isinstance(line["label_checksum"], int)


def test_bench_force_dist_with_the_file_leg_on_two_contexts():
    """first contact of the two multi-GPU paths on a 1-GPU box: the timed loop through torch.distributed / RCCL (world 1) AND the
    e2e_vcf leg through multi.DeviceGroup([0, 0]) (two contexts, two host threads, shared page-locked rows and outputs)"""
    e = _env()
    e["GNX_BENCH_FORCE_DIST"] = "1"
    e["GNX_BENCH_VCF_DEVICES"] = "0,0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--haps", "512",
                        "--cpu-seconds", "0", "--e2e-steps", "0", "--vcf-reps", "1", "--trained", "0", "--configs", "0"], capture_output=True, text=True,
                       timeout=900, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["config"]["dist_backend"] == "nccl" and line["value"] > 0
    v = line["e2e_vcf"]
    assert "error" not in v, v
    assert v["devices"] == [0, 0] and v["msp_labels_equal_device_path"] is True and v["haplotypes_per_s"] > 0


def test_bench_refuses_more_gpus_than_the_box_has():
    import torch
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--steps", "1", "--warmup", "0",
                        "--cpu-seconds", "0"], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode != 0
    assert "refusing to run" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())   # no JSON line for an experiment that did not run


def test_bench_refuses_world_size_mismatch():
    e = _env()
    e.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--cpu-seconds", "0"], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)

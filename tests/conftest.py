import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _library_is_built():
    """a fresh clone has no libgnomix_hip.so (build artefacts are git-ignored): build it once (hipcc cross-compiles
    gfx950 without a GPU; a no-op when it is up to date), exactly what __graft_entry__.build() does"""
    import subprocess
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "gnomix_amd", "csrc")])


@pytest.fixture(scope="session")
def oracle():
    from oracle import gnx_oracle as O
    O.build()
    return O


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def trees_from_npz(O, d, prefix):
    return O.Trees(d[prefix + "tree_off"], d[prefix + "left"], d[prefix + "right"], d[prefix + "feat"],
                   d[prefix + "cond"], d[prefix + "tree_class"], int(d[prefix + "n_class"]),
                   float(d[prefix + "base_score"]))

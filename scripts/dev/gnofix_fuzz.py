import os, sys, traceback
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
from oracle import gnx_oracle as O
O.build()
import test_gpu_fuzz as F
fn = getattr(F.test_random_gnofix_vs_oracle, "__wrapped__", F.test_random_gnofix_vs_oracle)
ok = bad = sk = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        fn(O, seed); ok += 1
    except pytest.skip.Exception:
        sk += 1
    except Exception:
        bad += 1; print("FAIL", seed); traceback.print_exc(limit=2)
print("gnofix fuzz ok", ok, "skipped", sk, "bad", bad)

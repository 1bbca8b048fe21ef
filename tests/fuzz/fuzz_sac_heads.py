import os, sys, random, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import emu_backend
emu_backend.install()
from conftest import Backend
import test_sac_heads as T

# the Gaussian policy head (forward, log-prob of a given action, backward) on random (batch, action_dim) — rows of one
# element, widths around the wavefront and beyond one workgroup — through the checks of tests/test_sac_heads.py
# (float64 autograd of the reference's formulas, actor.py:166-261)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
random.seed(seed)
bad = 0
for case in range(cases):
    B = random.choice([2, 3, 63, 64, 65, 257, 1000])  # (the test plants its clamp probes in rows 0 and 1)
    A = random.choice([2, 3, 16, 31, 32, 33, 64, 65, 128, 255, 256, 257, 300])
    try:
        T.test_gaussian_head_forward_backward(Backend("emu", "cpu"), B, A)
        print("OK ", B, A)
    except Exception:
        bad += 1
        print("BAD", B, A)
        traceback.print_exc(limit=2)
print("bad cases:", bad)
sys.exit(1 if bad else 0)

"""First device touch in a subprocess: tells a faulty NODE from a faulty product.

torch only (no reagent_amd kernel library is loaded by the child), so a failure here is the box's: the GPU tests and
`__graft_entry__.smoke()` run it first and print NODE_FAULT when it fails."""
import subprocess
import sys
import time

NODE_FAULT = "NODE FAULT: first device touch aborted before reagent_amd was loaded"

TOUCH = ("import torch; assert torch.cuda.is_available(), 'no GPU visible'; "
         "x = torch.ones(1 << 20, device='cuda').mul(2).sum(); torch.cuda.synchronize(); "
         "assert float(x) == float(2 << 20), float(x); "
         "y = torch.arange(1 << 16).to('cuda'); assert int(y.sum()) == (1 << 16) * ((1 << 16) - 1) // 2; print('touch ok')")


def device_preflight(code=TOUCH, tries=3, backoff=5.0, timeout=300):
    """(ok, log): run `code` in a fresh interpreter up to `tries` times"""
    log = []
    for k in range(tries):
        try:
            p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout)
            rc, tail = p.returncode, (p.stdout + p.stderr)[-600:]
        except subprocess.TimeoutExpired:
            rc, tail = -1, f"no answer within {timeout} s"
        log.append(f"attempt {k + 1}: rc {rc} {tail.strip()}")
        if rc == 0:
            return True, "\n".join(log)
        if k + 1 < tries:
            time.sleep(backoff)
    return False, "\n".join(log)

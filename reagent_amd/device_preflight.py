"""First device touch in a subprocess: tells a faulty NODE from a faulty product.

torch only (no reagent_amd kernel library is loaded by the child), so a failure here is the box's: the GPU tests and
`__graft_entry__.smoke()` run it first and print NODE_FAULT when it fails.  Seen on this pool about once in ten leases
(round 4's driver record, round 5 profiles/scripts run r05b): `Memory access fault by GPU node-2 ... on address (nil)` inside
torch's first kernel.  `python -m reagent_amd.device_preflight` is the same check for shell scripts (exit code 97 on a fault).

`settle()` goes one step further for the test session and smoke(): when the plain touch keeps faulting it tries the touch under
a few HSA runtime settings that route the first copies / fills differently (ALTERNATIVES); the first one that works is
exported into os.environ BEFORE this process initialises its own device, and named in the log — the run then continues on a
node that would otherwise have voided it.  On a healthy node none of this costs more than one 2-second child process."""
import os
import subprocess
import sys
import time

NODE_FAULT = "NODE FAULT: first device touch aborted before reagent_amd was loaded"

NO_GPU_RC = 3  # the child's answer when torch sees no device at all (a CPU box): not a fault
TOUCH = ("import sys, torch; torch.cuda.is_available() or sys.exit(3); "
         "x = torch.ones(1 << 20, device='cuda').mul(2).sum(); torch.cuda.synchronize(); "
         "assert float(x) == float(2 << 20), float(x); "
         "y = torch.arange(1 << 16).to('cuda'); assert int(y.sum()) == (1 << 16) * ((1 << 16) - 1) // 2; print('touch ok')")

# runtime settings tried, in order, when the plain touch faults (each is a documented ROCr / HIP switch)
ALTERNATIVES = (
    {"HSA_ENABLE_SDMA": "0"},                                  # copies through blit kernels instead of the SDMA engines
    {"HSA_ENABLE_SDMA": "0", "HIP_FORCE_DEV_KERNARG": "0"},    # + kernel arguments in host memory
    {"HSA_ENABLE_SDMA": "0", "GPU_MAX_HW_QUEUES": "1"},        # + a single hardware queue
)


def device_preflight(code=TOUCH, tries=3, backoff=5.0, timeout=300, env=None):
    """(ok, log): run `code` in a fresh interpreter up to `tries` times"""
    log = []
    child_env = dict(os.environ, **(env or {}))
    for k in range(tries):
        try:
            p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=child_env)
            rc, tail = p.returncode, (p.stdout + p.stderr)[-600:]
        except subprocess.TimeoutExpired:
            rc, tail = -1, f"no answer within {timeout} s"
        log.append(f"attempt {k + 1}{' under ' + str(env) if env else ''}: rc {rc} {tail.strip()}")
        if rc == 0:
            return True, "\n".join(log)
        if rc == NO_GPU_RC:
            return False, "\n".join(log + ["no GPU visible to torch"])
        if k + 1 < tries:
            time.sleep(backoff)
    return False, "\n".join(log)


def settle(code=TOUCH, tries=3, backoff=5.0, alternatives=ALTERNATIVES):
    """(ok, log, adopted): the plain touch, then the ALTERNATIVES; an alternative that works is exported into os.environ
    (call this BEFORE the first device call of the process — torch.cuda.is_available() included: the HSA runtime reads its
    settings when it starts).  ok is False with a log ending in "no GPU visible to torch" on a box without a device."""
    ok, log = device_preflight(code, tries=tries, backoff=backoff)
    if ok or log.endswith("no GPU visible to torch"):
        return ok, log, None
    for alt in alternatives:
        ok2, log2 = device_preflight(code, tries=1, env=alt)
        log += "\n" + log2
        if ok2:
            os.environ.update(alt)
            return True, log + f"\nNODE WORKAROUND: the plain first touch faulted {tries} times; continuing under {alt}", alt
    return False, log, None


if __name__ == "__main__":
    ok, log, adopted = settle()
    print(log)
    if adopted:
        print("export " + " ".join(f"{k}={v}" for k, v in adopted.items()))
    if not ok and not log.endswith("no GPU visible to torch"):
        print(NODE_FAULT)
    sys.exit(0 if ok else 97)

"""The MFMA kernels live at the edge of the register file (two waves per SIMD, 128 accumulator registers each): a
harmless-looking edit can tip the compiler into scratch spills that cost 20 % of a kernel without failing any
numerics test (it happened: a run-time `if (dz_dst)` in the backward epilogue, 245 spilled registers, 124 MB of
scratch traffic per launch).  This compiles the fused kernels for gfx950 with the resource remarks on and checks that
none of them spills."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src", ["mlp_fused.hip", "mlp_fused_x3.hip"])
def test_fused_kernels_do_not_spill(src, tmp_path):
    csrc = os.path.join(ROOT, "reagent_amd", "csrc")
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{csrc}", f"-I{ROOT}/include",
                          "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(csrc, src),
                          "-o", str(tmp_path / "o.o")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels = {}
    name = None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
        for key in ("VGPRs Spill", "ScratchSize [bytes/lane]", "VGPRs", "Occupancy [waves/SIMD]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and name:
                kernels[name].setdefault(key, int(m.group(1)))
    hot = {k: v for k, v in kernels.items() if re.search(r"mlp_(fwd|bwd)|wgrad_(group|frag|grouped)_kernel", k)}
    assert len(hot) >= 6, list(kernels)
    for k, v in hot.items():
        assert v.get("VGPRs Spill", 0) == 0 and v.get("ScratchSize [bytes/lane]", 0) == 0, (k, v)
        assert v.get("VGPRs", 0) <= 256, (k, v)

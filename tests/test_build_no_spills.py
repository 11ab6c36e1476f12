"""The persistent LDS-DMA kernels take work-queue tickets with an atomic issued from inline assembly
(sf_common.h: queue_take_async): its result register is valid only behind queue_ticket_wait, and
the compiler, which does not know that, must never copy it before -- a register spill would.  (A
build of one variant with 25 spilled registers faulted the GPU.)  This test compiles the three files
that use the tickets with the compiler's resource remarks on and asserts that no kernel of theirs
uses scratch memory or spills a vector register (other kernels of these files may)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tcn-transformer-gan-hybrid-model-for-predicting-streamflow-in-ungauged-basins_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src", ["sf_conv3x3.hip", "sf_conv_s2.hip", "sf_wgrad3x3.hip"])
def test_ticket_kernels_do_not_spill(src, tmp_path):
    r = subprocess.run([HIPCC, "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", "--cuda-device-only",
                        "-c", os.path.join(CSRC, src), "-o", str(tmp_path / "o.o"), "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    name, seen, bad = None, 0, []
    ticket = ("conv3_dma", "conv_s2_", "wgrad_wide", "wgrad3_dma")       # the kernels that draw tickets
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1) if any(t in m.group(1) for t in ticket) else None
            seen += name is not None
        if name is None:
            continue
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and int(m.group(1)) != 0:
            bad.append((name, "scratch", int(m.group(1))))
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and int(m.group(1)) != 0:
            bad.append((name, "vgpr spills", int(m.group(1))))
    assert seen >= 3 and not bad, bad

"""CPU suite: the pure arithmetic helpers the kernels are built from (nst_nt4 restatement, the x2 truncation of nw_alignment,
the four-characters-at-once read encoder, the RefSequence window fetches across both strand boundaries), compiled for the host with hipcc and run without a GPU (no HIP API call)."""
import os, shutil, subprocess
import pytest
import common


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_kernel_arithmetic_helpers_on_host(workdir):
    src = os.path.join(common.ROOT, "tests", "native", "host_checks.hip")
    exe = os.path.join(workdir, "host_checks")
    subprocess.run(["hipcc", "-O2", "--offload-arch=gfx950", "-std=c++17", "-w", "-o", exe, src], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("bad=0"), out

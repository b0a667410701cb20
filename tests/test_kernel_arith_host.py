"""CPU suite: the pure arithmetic helpers the kernels are built from (nst_nt4 restatement, the x2 truncation of nw_alignment,
the four-characters-at-once read encoder, the RefSequence window fetches across both strand boundaries, and the serial nw_alignment restatement d_nw against the oracle's on
20 000 random pairs), compiled for the host with hipcc and run without a GPU (no HIP API call)."""
import os, shutil, subprocess
import pytest
import common, oracle_py


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_kernel_arithmetic_helpers_on_host(workdir):
    src = os.path.join(common.ROOT, "tests", "native", "host_checks.hip")
    exe = os.path.join(workdir, "host_checks")
    oracle_py.build()                                   # the checker the harness links (orc_nw)
    odir = os.path.dirname(oracle_py.LIB)
    subprocess.run(["hipcc", "-O2", "--offload-arch=gfx950", "-std=c++17", "-w", "-o", exe, src, "-L" + odir, "-loracle", "-Wl,-rpath," + odir], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("bad=0") and "d_nw: 20000" in out, out

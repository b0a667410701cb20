"""CPU suite: the pieces of the host program's parallel FASTQ pipeline that need no GPU (dart_amd/csrc/host/fast_fastq.h): byte-string
kernels of the SAM formatter (AVX2 and scalar), the integer printer, and the parallel FASTQ record index against a sequential line
splitter on random, awkward text.  The pipeline as a whole runs on the GPU box (tests/test_gpu_cli.py, both host paths)."""
import os, subprocess
import common


def test_host_text_kernels_and_fastq_index(workdir):
    src = os.path.join(common.ROOT, "tests", "native", "host_text_checks.cpp")
    exe = os.path.join(workdir, "host_text_checks")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(common.ROOT, "include"), "-I", os.path.join(common.ROOT, "dart_amd", "csrc", "host"),
                    "-o", exe, src], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("bad=0"), out.stdout + out.stderr

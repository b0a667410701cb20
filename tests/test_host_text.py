"""CPU suite: the pieces of the host program's parallel FASTQ pipeline that need no GPU (dart_amd/csrc/host/fast_fastq.h): byte-string
kernels of the SAM formatter (AVX2 and scalar), the integer printer, and the parallel FASTQ record index against a sequential line
splitter on random, awkward text.  The pipeline as a whole runs on the GPU box (tests/test_gpu_cli.py, both host paths)."""
import gzip, os, re, subprocess
import common, bam_decode


def test_host_text_kernels_and_fastq_index(workdir):
    src = os.path.join(common.ROOT, "tests", "native", "host_text_checks.cpp")
    exe = os.path.join(workdir, "host_text_checks")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(common.ROOT, "include"), "-I", os.path.join(common.ROOT, "dart_amd", "csrc", "host"),
                    "-o", exe, src], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("bad=0"), out.stdout + out.stderr


def test_bam_writer_against_golden_sam(workdir):
    """`dart -bo` (dart_amd/csrc/host/bam_writer.h; the reference: sam_parse1 + sam_write1 of htslib on every SAM line, Mapping.cpp:655-662):
    the reference-generated golden SAM files written as BAM and read back by an independent decoder -- header, reference table, every
    field, bins, integer tag types, BGZF framing (CRC, ISIZE <= 0xff00, header in its own block, end-of-file block).  The strand tag the
    reference joins with a blank (" XS:A:+") is lost, as in the reference's BAM (strtol stops at the blank); a line whose quality and
    sequence differ in length is dropped, as sam_parse1 fails on it."""
    src = os.path.join(common.ROOT, "tests", "native", "bam_checks.cpp")
    exe = os.path.join(workdir, "bam_checks")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(common.ROOT, "dart_amd", "csrc", "host"), "-o", exe, src, "-lz"], check=True)
    bases = sorted(f[:-7] for f in os.listdir(common.GOLDEN) if f.endswith(".sam.gz"))
    assert len(bases) >= 5
    for k, base in enumerate(bases):
        text = common.golden_sam(base)
        if k == 0:            # awkward lines: quality shorter than the sequence (refused), a negative and a large tag value, no quality
            body = text.splitlines()
            first = next(i for i, l in enumerate(body) if not l.startswith("@"))
            f = body[first].split("\t")
            extra = ["\t".join(f[:10] + [f[10][:-1]] + f[11:]),
                     "\t".join(f[:10] + ["*", "NM:i:-3", "AS:i:-200", "XS:i:70000", "YS:i:-40000", "ZZ:Z:a b"]),
                     "\t".join(["r0", "4", "*", "0", "0", "*", "*", "0", "0", "*", "*"])]
            text = "\n".join(body[:first] + extra + body[first:]) + "\n"
        sam_path = os.path.join(workdir, "g.sam"); bam_path = os.path.join(workdir, "g.bam")
        open(sam_path, "w").write(text)
        out = subprocess.run([exe, sam_path, bam_path, str(1 + k % 4)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        hdr, refs, lines, bins = bam_decode.decode(open(bam_path, "rb").read())
        want_hdr = "".join(l + "\n" for l in text.splitlines() if l.startswith("@"))
        assert hdr == want_hdr
        assert ["@SQ\tSN:%s\tLN:%d" % r for r in refs] == [l for l in want_hdr.splitlines() if l.startswith("@SQ")]
        want = [re.sub(r" XS:A:[+-]$", "", l) for l in text.splitlines() if not l.startswith("@")]
        if k == 0:
            assert "refused=1" in out.stdout
            del want[0]
        else:
            assert "refused=0" in out.stdout
        assert len(lines) == len(want)
        for a, b in zip(lines, want):
            assert a == b, (base, a, b)
        assert any(b != 4680 for b in bins)

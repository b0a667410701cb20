"""CPU suite: the pieces of the host program's parallel FASTQ pipeline that need no GPU (dart_amd/csrc/host/fast_fastq.h): byte-string
kernels of the SAM formatter (AVX2 and scalar), the integer printer, and the parallel FASTQ record index against a sequential line
splitter on random, awkward text.  The pipeline as a whole runs on the GPU box (tests/test_gpu_cli.py, both host paths)."""
import gzip, os, re, subprocess
import common, bam_decode


def test_host_text_kernels_and_fastq_index(workdir):
    src = os.path.join(common.ROOT, "tests", "native", "host_text_checks.cpp")
    exe = os.path.join(workdir, "host_text_checks")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(common.ROOT, "include"), "-I", os.path.join(common.ROOT, "dart_amd", "csrc", "host"),
                    "-o", exe, src, "-ldl"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("bad=0"), out.stdout + out.stderr


def test_gz_library_inflated_whole_qualifies_only_when_both_readers_agree(workdir):
    """.gz FASTQ through the parallel pipeline (fast_fastq.h: MappedFile::open_gz with the system's libdeflate, FastqIndex::run(gz)): the inflated bytes are
    the file's (one member, several members), and a text the reference's gz reader (gzgets into 1024 bytes, GetData.cpp:181-210) would read differently
    from its plain reader -- a line of 1024 bytes, a header that is no '@' line, a record cut short, an entry without bases, a NUL -- does NOT qualify
    (it goes through the streaming reader, which restates gzgets)."""
    src = os.path.join(common.ROOT, "tests", "native", "host_text_checks.cpp")
    exe = os.path.join(workdir, "host_text_checks_gz")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(common.ROOT, "include"), "-I", os.path.join(common.ROOT, "dart_amd", "csrc", "host"),
                    "-o", exe, src, "-ldl"], check=True)
    rec = lambda i, seq, name=None: "@%s/1\n%s\n+\n%s\n" % (name or ("r%d" % i), seq, "I" * len(seq))
    good = "".join(rec(i, "ACGTN" * (5 + i % 17)) for i in range(5000))
    cases = {
        "good": (good, True),
        "crlf_and_long_but_legal": ("".join("@r%d some text\r\n%s\r\n+\r\n%s\r\n" % (i, "A" * 1000, "I" * 1000) for i in range(50)), True),
        "line_of_1024_bytes": (good + rec(1, "A" * 1023), False),
        "line_of_1023_bytes": (good + rec(1, "A" * 1022), True),
        "header_without_at": (good + "r7/1\nACGT\n+\nIIII\n", False),
        "empty_name": (good + "@/1\nACGT\n+\nIIII\n", False),
        "cut_short": (good + "@r9/1\nACGT\n+\n", False),
        "entry_without_bases": (good + "@r9/1\n\n+\n\n" + good, False),
        "nul_byte": (good + "@r9/1\nAC\0GT\n+\nIIIII\n", False),
        "no_final_newline": (good + "@r9/1\nACGT\n+\nIIII", True),
    }
    for name, (text, want) in cases.items():
        plain = os.path.join(workdir, name + ".fq"); gz = plain + ".gz"
        open(plain, "wb").write(text.encode("latin-1"))
        with gzip.open(gz, "wb", compresslevel=1 + len(name) % 9) as f:
            f.write(text.encode("latin-1"))
        if name == "good":            # several members, as `cat a.gz b.gz` makes them (gzread reads through all of them)
            with open(gz, "ab") as f:
                f.write(gzip.compress(good.encode()))
            open(plain, "ab").write(good.encode())
        out = subprocess.run([exe, "gz", gz, plain], capture_output=True, text=True)
        if "libdeflate missing" in out.stdout:
            import pytest
            pytest.skip("no libdeflate on this system: the streaming reader is the only one")
        assert out.returncode == 0, (name, out.stdout, out.stderr)
        assert ("qualifies=1 same_bytes=1" in out.stdout) == want, (name, out.stdout)
    # a file that is no clean gzip stream (bytes behind the member) is left to gzread
    bad = os.path.join(workdir, "trailing.fq.gz")
    open(bad, "wb").write(gzip.compress(good.encode()) + b"trailing bytes that are no gzip member")
    out = subprocess.run([exe, "gz", bad, os.path.join(workdir, "good.fq")], capture_output=True, text=True)
    assert "qualifies=0" in out.stdout, out.stdout


def test_bam_writer_against_golden_sam(workdir):
    """`dart -bo` (dart_amd/csrc/host/bam_writer.h; the reference: sam_parse1 + sam_write1 of htslib on every SAM line, Mapping.cpp:655-662):
    the reference-generated golden SAM files written as BAM and read back by an independent decoder -- header, reference table, every
    field, bins, integer tag types, BGZF framing (CRC, ISIZE <= 0xff00, header in its own block, end-of-file block).  The strand tag the
    reference joins with a blank (" XS:A:+") is lost, as in the reference's BAM (strtol stops at the blank); a line whose quality and
    sequence differ in length is dropped, as sam_parse1 fails on it."""
    src = os.path.join(common.ROOT, "tests", "native", "bam_checks.cpp")
    exe = os.path.join(workdir, "bam_checks")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(common.ROOT, "dart_amd", "csrc", "host"), "-o", exe, src, "-lz", "-ldl"], check=True)
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
        # BAM holds a base as one of sixteen codes (htslib's seq_nt16_table, as sam_parse1 applies it): case is lost, everything but "=ACMGRSVTWYHKDBN" -- a '-' too -- reads back as N
        def as_bam_stores_it(line):
            f = line.split("\t")
            if len(f) > 9 and f[9] != "*":
                f[9] = "".join(ch if ch in "=ACMGRSVTWYHKDBN" else "N" for ch in f[9].upper())
            return "\t".join(f)
        want = [as_bam_stores_it(l) for l in want]
        if k == 0:
            assert "refused=1" in out.stdout
            del want[0]
        else:
            assert "refused=0" in out.stdout
        assert len(lines) == len(want)
        for a, b in zip(lines, want):
            assert a == b, (base, a, b)
        assert any(b != 4680 for b in bins)

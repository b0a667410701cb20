"""CPU suite: libdartgpu.so loads and exports every symbol include/dartgpu.h declares; without a
device the product fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import pytest
import common
from dart_amd import host


def declared_symbols():
    text = open(os.path.join(common.ROOT, "include", "dartgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dg_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = C.CDLL(host.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), "libdartgpu.so does not export %s" % s


def test_index_library_exports_every_declared_symbol():
    """libdartindex.so (the offline indexer's device side) against include/dartindex.h"""
    import __graft_entry__ as ge
    from dart_amd import index_build
    ge.build()
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(common.ROOT, "include", "dartindex.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(di_[a-z_0-9]+)\s*\(", text)))
    assert len(syms) >= 9 and "di_build_files" in syms
    lib = index_build._index_lib()
    for s in syms:
        assert hasattr(lib, s), "libdartindex.so does not export %s" % s
    assert lib.di_text_words(64) == 4


def test_dart_index_host_half_and_loud_failure_without_a_device(workdir):
    """`dart index ref.fa prefix` (main.cpp:125-127): its host half -- FASTA records, holes, the lrand48 stream -- writes the reference indexer's
    .ann / .amb for the FASTA with ambiguous bases (tests/golden/index_holes.json); without a device the build then fails with a message and a
    non-zero exit, and writes no .bwt (no CPU fallback).  The five files on a GPU: tests/test_gpu_cli.py."""
    import json, subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    gold = json.load(open(os.path.join(common.GOLDEN, "index_holes.json")))
    fa = os.path.join(workdir, "holes_cli.fa")
    common.write_holes_fasta(fa)
    assert common.sha(fa) == gold["fasta_sha256"]
    prefix = os.path.join(workdir, "holes_cli")
    r = subprocess.run([os.path.join(common.ROOT, "dart_amd", "dart"), "index", fa, prefix], capture_output=True, text=True)
    assert r.returncode != 0 and "di_build_files failed" in r.stderr
    assert not os.path.exists(prefix + ".bwt")
    assert open(prefix + ".ann").read() == gold["ann"] and open(prefix + ".amb").read() == gold["amb"]
    r = subprocess.run([os.path.join(common.ROOT, "dart_amd", "dart"), "index", fa], capture_output=True, text=True)
    assert "usage:" in r.stderr and "index ref.fa prefix" in r.stderr


def test_index_builder_fasta_path_matches_reference_indexer(workdir):
    """dart_amd/index_build.py from a FASTA with ambiguous bases (CPU path): the five files are the reference bwt_index's bytes"""
    import json
    from dart_amd import index_build
    gold = json.load(open(os.path.join(common.GOLDEN, "index_holes.json")))
    fa = os.path.join(workdir, "holes_py.fa")
    common.write_holes_fasta(fa)
    prefix = os.path.join(workdir, "holes_py")
    index_build.build_index_from_fasta(fa, prefix, device="cpu")
    for ext, want in gold["index_sha256"].items():
        assert common.sha(prefix + "." + ext) == want, ext


def test_no_device_is_a_loud_failure(workdir):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    c = common.build_case("se100", workdir)
    with pytest.raises(RuntimeError) as e:
        host.DartGPU(host.Index(c["prefix"]))
    assert "no HIP device" in str(e.value) or "dg_init failed" in str(e.value)


def test_record_layouts_match_header():
    # sizes the C structs must have (include/dartgpu.h): 9 x i32, 6 x i32 + i64 + 2 x u32, 2 x i64 + 2 x i32
    assert host.READ_OUT.itemsize == 36 and host.REPORT_OUT.itemsize == 40 and host.SJ_OUT.itemsize == 24
    assert C.sizeof(host.Params) == 32


def test_compact_record_layout_round_trip(workdir):
    """include/dartgpu.h's compact layout (12 + 16 bytes, offsets as running sums, CIGAR ops in report order, none stored for a plain
    full-length match) against its reference expansion host.expand_compact: the oracle's records of a golden case are packed here
    by the header's rules -- independently of the kernels that do it on the GPU -- and must come back unchanged"""
    import numpy as np
    import oracle_py
    assert host.READ_C.itemsize == 12 and host.REPORT_C.itemsize == 16 and host.CIGAR_FULL_MATCH == 255
    c = common.build_case("pe101_spliced", workdir)
    orc = oracle_py.Oracle(c["prefix"])
    so, rl, flat = host.pack_reads(c["reads"])
    reads, rep, cig, sj = orc.map_batch(orc.params(paired=1, max_mismatch=5), so, rl, flat)
    orc.close()
    rc = np.zeros(len(reads), host.READ_C)
    for f in ("score", "sub_score", "mis_num", "mapq", "n_sj", "n_rep", "best"):
        rc[f] = reads[f]
    pc = np.zeros(len(rep), host.REPORT_C)
    for f in ("pos", "aln_score", "flag", "paired_idx", "sj_type", "bdir"):
        pc[f] = rep[f]
    pc["chr"] = np.where(rep["chr"] < 0, 0xFFFF, rep["chr"])
    owner = np.repeat(np.arange(len(reads)), reads["n_rep"])
    first_op = cig[np.minimum(rep["cigar_off"], max(len(cig) - 1, 0))] if len(cig) else np.zeros(len(rep), np.uint32)
    plain = (rep["n_cigar"] == 1) & (first_op == (rl[owner].astype(np.uint32) << 4))
    pc["n_cigar"] = np.where(plain, host.CIGAR_FULL_MATCH, rep["n_cigar"])
    stored = [cig[int(o):int(o) + int(k)] for o, k, p in zip(rep["cigar_off"], rep["n_cigar"], plain) if not p and k]
    cc = np.concatenate(stored) if stored else np.zeros(0, np.uint32)
    assert 0 < plain.sum() < len(rep) and len(cc) < len(cig)
    r2, p2, c2 = host.expand_compact(rc, pc, cc, rl)

    class R:
        pass
    res = R(); res.reads, res.reports, res.cigar, res.sj = r2, p2, c2, sj
    common.assert_same(res, (reads, rep, cig, sj))


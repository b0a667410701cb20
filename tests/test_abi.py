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

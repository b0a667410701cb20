"""ctypes binding of oracle/liboracle.so -- the CHECKER.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_HARNESS = os.path.join(ORACLE_DIR, "_ref", "ref_harness")
REF_INDEXER = os.path.join(ORACLE_DIR, "_ref", "bwt_index")
ORACLE_CLI = os.path.join(ORACLE_DIR, "dart_oracle")

READ_OUT = np.dtype([("score", "<i4"), ("sub_score", "<i4"), ("mis_num", "<i4"), ("mapq", "<i4"), ("n_rep", "<i4"),
                     ("best", "<i4"), ("rep_off", "<i4"), ("sj_off", "<i4"), ("n_sj", "<i4")])
REPORT_OUT = np.dtype([("aln_score", "<i4"), ("sj_type", "<i4"), ("flag", "<i4"), ("paired_idx", "<i4"), ("chr", "<i4"),
                       ("bdir", "<i4"), ("pos", "<i8"), ("cigar_off", "<u4"), ("n_cigar", "<u4")])
SJ_OUT = np.dtype([("g1", "<i8"), ("g2", "<i8"), ("type", "<i4"), ("read_idx", "<i4")])
COUNTER_KEYS = ["n_occ_blocks", "n_lf", "n_sa", "n_search", "n_2occ4", "n_nw", "nw_cells", "n_reseed", "reseed_window", "ref_bases"]


class OrcParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("max_gaps", "max_dup", "max_intron", "min_intron", "max_mismatch", "multi_hit", "all_sj", "paired")]


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ORACLE_DIR, "dart_oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)


class Oracle:
    def __init__(self, prefix: str):
        build()
        self.lib = C.CDLL(LIB)
        self.lib.orc_index_load.restype = C.c_void_p
        self.lib.orc_index_load.argtypes = [C.c_char_p]
        self.lib.orc_index_free.argtypes = [C.c_void_p]
        vp = C.c_void_p
        self.lib.orc_map_batch.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]
        self.lib.orc_nw.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        self.lib.orc_seeds.argtypes = [vp, vp, C.c_char_p, C.c_int, vp, vp, vp, C.c_int]
        self.ix = self.lib.orc_index_load(prefix.encode())
        if not self.ix:
            raise RuntimeError("orc_index_load failed for " + prefix)

    def close(self):
        if self.ix:
            self.lib.orc_index_free(self.ix)
            self.ix = None

    @staticmethod
    def params(**kw) -> OrcParams:
        p = OrcParams(5, 100, 500000, 5, 0, 0, 0, 0)
        for k, v in kw.items():
            setattr(p, k, int(v))
        return p

    def map_batch(self, params: OrcParams, seq_off, rlen, flat, threads: int = 4):
        n = len(rlen)
        a, b, c = np.ascontiguousarray(seq_off, np.uint32), np.ascontiguousarray(rlen, np.uint16), np.ascontiguousarray(flat, np.uint8)
        caps = (C.c_size_t * 3)(n * 64 + 1024, n * 256 + 4096, n * 8 + 64)
        used = (C.c_size_t * 3)()
        reads = np.zeros(n, READ_OUT); rep = np.zeros(caps[0], REPORT_OUT); cig = np.zeros(caps[1], np.uint32); sj = np.zeros(caps[2], SJ_OUT)
        ctr = (C.c_uint64 * 10)()
        rc = self.lib.orc_map_batch(self.ix, C.byref(params), n, a.ctypes.data, b.ctypes.data, c.ctypes.data, reads.ctypes.data,
                                    rep.ctypes.data, cig.ctypes.data, sj.ctypes.data, caps, used, threads, ctr)
        if rc:
            raise RuntimeError("orc_map_batch capacity")
        self.counters = {k: int(ctr[i]) for i, k in enumerate(COUNTER_KEYS)}
        return reads, rep[:used[0]], cig[:used[1]], sj[:used[2]]

    def nw(self, s1: bytes, s2: bytes):
        cap = len(s1) + len(s2) + 8
        o1 = C.create_string_buffer(cap); o2 = C.create_string_buffer(cap)
        ln = self.lib.orc_nw(s1, s2, o1, o2, cap)
        return o1.raw[:ln], o2.raw[:ln]

    def seeds(self, params: OrcParams, seq: bytes, cap: int = 4096):
        rp = np.zeros(cap, np.int32); sl = np.zeros(cap, np.int32); gp = np.zeros(cap, np.int64)
        n = self.lib.orc_seeds(self.ix, C.byref(params), seq, len(seq), rp.ctypes.data, sl.ctypes.data, gp.ctypes.data, cap)
        return rp[:n], sl[:n], gp[:n]

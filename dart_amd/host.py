"""ctypes binding of libdartgpu.so (include/dartgpu.h) plus the index loader.

Python here is plumbing for tests and bench.py; the product's host program is the C++ `dart`
command line (dart_amd/csrc/host/).  There is no CPU fallback: if libdartgpu.so is missing or no
HIP device is present, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DARTGPU_LIB") or os.path.join(_HERE, "libdartgpu.so")      # (DARTGPU_LIB: another build of the library, for A/B runs)


class IndexView(C.Structure):
    _fields_ = [("bwt", C.c_void_p), ("bwt_words", C.c_uint64), ("primary", C.c_uint64), ("L2", C.c_uint64 * 5),
                ("seq_len", C.c_uint64), ("sa", C.c_void_p), ("n_sa", C.c_uint64), ("sa_intv", C.c_int32),
                ("pac", C.c_void_p), ("l_pac", C.c_int64), ("n_chr", C.c_int32), ("chr_off", C.c_void_p),
                ("chr_len", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("max_gaps", "max_dup", "max_intron", "min_intron", "max_mismatch",
                                          "multi_hit", "all_sj", "paired")]


READ_OUT = np.dtype([("score", "<i4"), ("sub_score", "<i4"), ("mis_num", "<i4"), ("mapq", "<i4"), ("n_rep", "<i4"),
                     ("best", "<i4"), ("rep_off", "<i4"), ("sj_off", "<i4"), ("n_sj", "<i4")])
REPORT_OUT = np.dtype([("aln_score", "<i4"), ("sj_type", "<i4"), ("flag", "<i4"), ("paired_idx", "<i4"), ("chr", "<i4"),
                       ("bdir", "<i4"), ("pos", "<i8"), ("cigar_off", "<u4"), ("n_cigar", "<u4")])
SJ_OUT = np.dtype([("g1", "<i8"), ("g2", "<i8"), ("type", "<i4"), ("read_idx", "<i4")])
READ_C = np.dtype([("score", "<u2"), ("sub_score", "<u2"), ("mis_num", "<u2"), ("mapq", "u1"), ("n_sj", "u1"), ("n_rep", "<u2"), ("best", "<u2")])
REPORT_C = np.dtype([("pos", "<i4"), ("aln_score", "<u2"), ("flag", "<u2"), ("paired_idx", "<i2"), ("chr", "<u2"), ("n_cigar", "u1"),
                     ("sj_type", "i1"), ("bdir", "u1"), ("pad", "u1")])
CIGAR_FULL_MATCH = 255
assert READ_OUT.itemsize == 36 and REPORT_OUT.itemsize == 40 and SJ_OUT.itemsize == 24 and READ_C.itemsize == 12 and REPORT_C.itemsize == 16


def expand_compact(reads_c, reports_c, cigar_c, rlen):
    """dg_read_c / dg_report_c arrays + the stored CIGAR ops + the read lengths -> (reads, reports, cigar ops) in the full record
    dtypes: rep_off / sj_off as running sums; a report marked DG_CIGAR_FULL_MATCH gets its "<length of its read>M" back; the stored ops
    lie in two regions (include/dartgpu.h): first, report by report, those of the reports with pad bit 0 clear, then those of the
    reports with it set -- a report's ops start at the running sum of the stored counts inside its region.  The reference expansion
    of the compact layout."""
    n = len(reads_c)
    r = np.zeros(n, READ_OUT)
    for f in ("score", "sub_score", "mis_num", "mapq", "n_rep", "best", "n_sj"):
        r[f] = reads_c[f]
    nrep = reads_c["n_rep"].astype(np.int64)
    r["rep_off"] = np.cumsum(nrep) - nrep
    nsj = reads_c["n_sj"].astype(np.int64)
    r["sj_off"] = np.cumsum(nsj) - nsj
    m = len(reports_c)
    assert m == int(nrep.sum()), "the reports do not add up to the reads' n_rep"
    p = np.zeros(m, REPORT_OUT)
    for f in ("aln_score", "sj_type", "flag", "paired_idx", "bdir", "pos"):
        p[f] = reports_c[f]
    p["chr"] = np.where(reports_c["chr"] == 0xFFFF, -1, reports_c["chr"].astype(np.int32))
    plain = reports_c["n_cigar"] == CIGAR_FULL_MATCH
    second = (reports_c["pad"] & 1) != 0
    stored = np.where(plain, 0, reports_c["n_cigar"]).astype(np.int64)
    full = np.where(plain, 1, reports_c["n_cigar"]).astype(np.int64)
    p["n_cigar"] = full
    off_full = np.cumsum(full) - full
    p["cigar_off"] = off_full
    cig = np.zeros(int(full.sum()), np.uint32)
    owner = np.repeat(np.arange(n), nrep)                         # the read of every report
    cig[off_full[plain]] = np.asarray(rlen, np.uint32)[owner[plain]] << 4
    # where a report's stored ops start: the running sum inside its region; the second region begins behind the whole first one
    s1 = np.where(second, 0, stored); s2 = np.where(second, stored, 0)
    src = np.where(second, int(s1.sum()) + np.cumsum(s2) - s2, np.cumsum(s1) - s1)
    sel = stored > 0
    if sel.any():
        st = stored[sel]
        k = np.arange(int(st.sum())) - np.repeat(np.cumsum(st) - st, st)
        cig[np.repeat(off_full[sel], st) + k] = np.asarray(cigar_c, np.uint32)[np.repeat(src[sel], st) + k]
    return r, p, cig


class IndexFiles(C.Structure):
    _fields_ = [("bwt_path", C.c_char_p), ("sa_path", C.c_char_p), ("pac_path", C.c_char_p), ("l_pac", C.c_int64), ("n_chr", C.c_int32),
                ("chr_off", C.c_void_p), ("chr_len", C.c_void_p), ("expected_reads", C.c_uint64)]


INIT_ASYNC_AIDS = 1


class Index:
    """The BWA-format index files as the reference loads them (bwt_index.cpp:15-35,102-121,229-251).  The headers and the chromosome
    table are read at once; the three big arrays only when something asks for them (dg_init's host-array view, tests): dg_init_files
    takes them from the files straight to HBM."""

    def __init__(self, prefix: str):
        self.prefix = prefix
        hdr = np.fromfile(prefix + ".bwt", dtype=np.uint64, count=5)
        self.primary = int(hdr[0])
        self.L2 = [0] + [int(x) for x in hdr[1:5]]
        self.seq_len = self.L2[4]
        sa_hdr = np.fromfile(prefix + ".sa", dtype=np.uint64, count=7)
        self.sa_intv = int(sa_hdr[5])
        self.n_sa = (self.seq_len + self.sa_intv) // self.sa_intv
        self._bwt = self._sa = self._pac = None
        with open(prefix + ".ann") as f:
            first = f.readline().split()
            self.l_pac, n = int(first[0]), int(first[1])
            self.names, lens = [], []
            for _ in range(n):
                self.names.append(f.readline().split()[1])
                lens.append(int(f.readline().split()[1]))
        self.chr_len = np.asarray(lens, dtype=np.int64)
        self.chr_off = np.concatenate([[0], np.cumsum(self.chr_len)[:-1]]).astype(np.int64)

    @property
    def bwt(self):
        if self._bwt is None:
            self._bwt = np.fromfile(self.prefix + ".bwt", dtype=np.uint32, offset=40)
        return self._bwt

    @property
    def sa(self):
        if self._sa is None:
            sa = np.zeros(self.n_sa, dtype=np.uint64)
            sa[0] = np.uint64(0xFFFFFFFFFFFFFFFF)
            body = np.fromfile(self.prefix + ".sa", dtype=np.uint64, offset=56, count=self.n_sa - 1)
            sa[1:1 + len(body)] = body
            self._sa = sa
        return self._sa

    @property
    def pac(self):
        if self._pac is None:
            self._pac = np.fromfile(self.prefix + ".pac", dtype=np.uint8)
        return self._pac

    def files(self, expected_reads: int = 0) -> IndexFiles:
        f = IndexFiles()
        f.expected_reads = int(expected_reads)
        self._paths = [(self.prefix + e).encode() for e in (".bwt", ".sa", ".pac")]
        f.bwt_path, f.sa_path, f.pac_path = self._paths
        f.l_pac = self.l_pac; f.n_chr = len(self.names)
        f.chr_off = self.chr_off.ctypes.data; f.chr_len = self.chr_len.ctypes.data
        return f

    def view(self) -> IndexView:
        v = IndexView()
        v.bwt = self.bwt.ctypes.data; v.bwt_words = self.bwt.size; v.primary = self.primary
        for i in range(5):
            v.L2[i] = self.L2[i]
        v.seq_len = self.seq_len; v.sa = self.sa.ctypes.data; v.n_sa = self.n_sa; v.sa_intv = self.sa_intv
        v.pac = self.pac.ctypes.data; v.l_pac = self.l_pac; v.n_chr = len(self.names)
        v.chr_off = self.chr_off.ctypes.data; v.chr_len = self.chr_len.ctypes.data
        return v


def pack_reads(seqs):
    """list of bytes / 2-D uint8 array -> (seq_off u32, rlen u16, flat uint8)."""
    if isinstance(seqs, np.ndarray) and seqs.ndim == 2:
        n, l = seqs.shape
        return (np.arange(n, dtype=np.uint32) * np.uint32(l)), np.full(n, l, dtype=np.uint16), np.ascontiguousarray(seqs).reshape(-1)
    lens = np.asarray([len(s) for s in seqs], dtype=np.uint16)
    off = np.concatenate([[0], np.cumsum(lens.astype(np.int64))[:-1]]).astype(np.uint32) if len(seqs) else np.zeros(0, np.uint32)
    flat = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if len(seqs) else np.zeros(0, np.uint8)
    return off, lens, flat


def interleave_pairs(m1: np.ndarray, m2: np.ndarray) -> np.ndarray:
    """mate 1 rows / mate 2 rows -> [2n, rlen] with mate 2 reverse-complemented the way the
    reference's loader does (GetData.cpp:157-162: anything but ACGT/acgt becomes 'N')."""
    comp = np.full(256, ord("N"), dtype=np.uint8)
    for a, b in zip(b"ACGTacgt", b"TGCATGCA"):
        comp[a] = b
    out = np.empty((2 * m1.shape[0], m1.shape[1]), dtype=np.uint8)
    out[0::2] = m1
    out[1::2] = comp[m2[:, ::-1]]
    return out


def _load_lib():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libdartgpu.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    lib.dg_init.restype = C.c_void_p
    lib.dg_init.argtypes = [C.POINTER(IndexView), C.POINTER(Params), C.c_int, C.POINTER(C.c_int)]
    lib.dg_init_files.restype = C.c_void_p
    lib.dg_init_files.argtypes = [C.POINTER(IndexFiles), C.POINTER(Params), C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.dg_index_wait.argtypes = [C.c_void_p]
    lib.dg_init_report.restype = C.c_char_p
    lib.dg_init_report.argtypes = [C.c_void_p]
    lib.dg_device_count.restype = C.c_int
    lib.dg_last_error.restype = C.c_char_p
    lib.dg_last_error.argtypes = [C.c_void_p]
    lib.dg_destroy.argtypes = [C.c_void_p]
    lib.dg_clone.restype = C.c_void_p
    lib.dg_clone.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.dg_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
    lib.dg_params_default.argtypes = [C.POINTER(Params)]
    vp = C.c_void_p
    lib.dg_map_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.dg_batch_upload.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.dg_batch_upload_packed.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_size_t]
    lib.dg_map_batch_packed.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_size_t, vp, vp, vp, vp, vp, vp]
    lib.dg_batch_download_compact.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.dg_map_batch_compact.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_size_t, vp, vp, vp, vp, vp, vp]
    lib.dg_host_alloc.restype = C.c_void_p
    lib.dg_host_alloc.argtypes = [C.c_size_t]
    lib.dg_host_free.argtypes = [vp]
    lib.dg_batch_run.argtypes = [vp, vp]
    lib.dg_batch_download.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.dg_batch_device_ptrs.argtypes = [vp, vp]
    lib.dg_batch_device_ptrs_compact.argtypes = [vp, vp]
    if hasattr(lib, "dg_batch_device_records_compact"):      # (absent from older builds of the library loaded through DARTGPU_LIB for A/B runs)
        lib.dg_batch_device_records_compact.argtypes = [vp, vp, vp]
    lib.dg_last_timings.argtypes = [vp, vp, vp, C.c_int]
    lib.dg_last_counters.argtypes = [vp, vp, C.c_int]
    lib.dg_probe_seeds.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.dg_probe_nw.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t]
    lib.dg_probe_nw_mode.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t]
    return lib


def pack_reads_2bit(arr: np.ndarray):
    """[n, rlen] uint8 ASCII (A/C/G/T/N upper case only) -> (words u32 [n, ceil(rlen/16)], nlist u32) for dg_map_batch_packed:
    16 bases per word, first base in the top bits; N stored as A and listed as flat index i * 16 * W + pos."""
    n, l = arr.shape
    W = (l + 15) // 16
    code = np.full(256, 255, np.uint8)
    for ch, v in zip(b"ACGTN", (0, 1, 2, 3, 0)):
        code[ch] = v
    c = code[arr]
    if (c == 255).any():
        raise ValueError("packed reads hold A, C, G, T, N only")
    padded = np.zeros((n, 16 * W), np.uint32)
    padded[:, :l] = c
    sh = (30 - 2 * np.arange(16, dtype=np.uint32))
    words = (padded.reshape(n, W, 16) << sh).sum(axis=2, dtype=np.uint64).astype(np.uint32)
    ri, pi = np.nonzero(arr == ord("N"))
    nlist = (ri.astype(np.uint64) * (16 * W) + pi.astype(np.uint64)).astype(np.uint32)
    return np.ascontiguousarray(words), nlist


class PinnedArray:
    """numpy view of page-locked host memory (dg_host_alloc): copies to/from it are DMA transfers"""

    def __init__(self, lib, shape, dtype):
        self.lib = lib
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        self.ptr = lib.dg_host_alloc(max(n * dt.itemsize, 64))
        if not self.ptr:
            raise MemoryError("dg_host_alloc failed")
        buf = (C.c_char * (n * dt.itemsize)).from_address(self.ptr)
        self.a = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)

    def free(self):
        if self.ptr:
            self.a = None
            self.lib.dg_host_free(self.ptr)
            self.ptr = None


def default_params(**kw) -> Params:
    p = Params(5, 100, 500000, 5, 0, 0, 0, 0)
    for k, v in kw.items():
        setattr(p, k, int(v))
    return p


class BatchResult:
    def __init__(self, reads, reports, cigar, sj):
        self.reads, self.reports, self.cigar, self.sj = reads, reports, cigar, sj

    def cigar_string(self, rep_index: int) -> str:
        r = self.reports[rep_index]
        ops = self.cigar[r["cigar_off"]: r["cigar_off"] + r["n_cigar"]]
        return "".join("%d%s" % (o >> 4, "MIDNS"[o & 15]) for o in ops)


class DartGPU:
    """dg_ctx wrapper: DART's per-read mapping path on one MI355X."""

    def __init__(self, index: Index, params: Params | None = None, device: int = 0, from_files: bool | None = None, async_aids: bool = False, expected_reads: int = 0):
        """from_files (default: unless DART_INIT_VIEW=1): dg_init_files -- the index files go straight to HBM; False: dg_init with the
        host-array view.  async_aids: the look-up aids are built in the background (dg_index_wait / wait_index joins them)."""
        self.lib = _load_lib()
        self.index = index
        self.params = params or default_params()
        st = C.c_int(0)
        if from_files is None:
            from_files = os.environ.get("DART_INIT_VIEW") != "1"
        if from_files:
            f = index.files(expected_reads)      # (expected_reads: 0 = unknown -> full aids; a short job gets lean ones)
            self.ctx = self.lib.dg_init_files(C.byref(f), C.byref(self.params), device, INIT_ASYNC_AIDS if async_aids else 0, C.byref(st))
        else:
            v = index.view()
            self.ctx = self.lib.dg_init(C.byref(v), C.byref(self.params), device, C.byref(st))
        if not self.ctx:
            raise RuntimeError("dg_init failed (%d): %s" % (st.value, (self.lib.dg_last_error(None) or b"").decode()))

    def wait_index(self):
        self._chk(self.lib.dg_index_wait(self.ctx), "dg_index_wait")

    def init_report(self) -> str:
        return (self.lib.dg_init_report(self.ctx) or b"").decode()

    def clone(self):
        """A second context sharing this one's index on the same device (dg_clone): for a second batch in flight."""
        other = object.__new__(DartGPU)
        other.lib, other.index, other.params, other._parent = self.lib, self.index, self.params, self
        st = C.c_int(0)
        other.ctx = self.lib.dg_clone(self.ctx, C.byref(st))
        if not other.ctx:
            raise RuntimeError("dg_clone failed (%d): %s" % (st.value, (self.lib.dg_last_error(None) or b"").decode()))
        self._clones = getattr(self, "_clones", []) + [other]
        return other

    def close(self):
        for o in getattr(self, "_clones", []):
            o.close()
        self._clones = []
        if getattr(self, "ctx", None):
            self.lib.dg_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, (self.lib.dg_last_error(self.ctx) or b"").decode()))

    def set_params(self, params: Params):
        self.params = params
        self._chk(self.lib.dg_set_params(self.ctx, C.byref(params)), "dg_set_params")

    def upload(self, seq_off, rlen, flat):
        self._n = len(rlen)
        self._rlen_of_batch = np.asarray(rlen, np.uint16).copy()
        self._keep = (np.ascontiguousarray(seq_off, np.uint32), np.ascontiguousarray(rlen, np.uint16), np.ascontiguousarray(flat, np.uint8))
        a, b, c = self._keep
        self._chk(self.lib.dg_batch_upload(self.ctx, self._n, a.ctypes.data, b.ctypes.data, c.ctypes.data), "dg_batch_upload")

    def run(self):
        used = (C.c_size_t * 3)()
        self._chk(self.lib.dg_batch_run(self.ctx, used), "dg_batch_run")
        self._used = [int(x) for x in used]
        return self._used

    def download(self) -> BatchResult:
        u = self._used
        reads = np.zeros(self._n, dtype=READ_OUT)
        reports = np.zeros(max(u[0], 1), dtype=REPORT_OUT)
        cigar = np.zeros(max(u[1], 1), dtype=np.uint32)
        sj = np.zeros(max(u[2], 1), dtype=SJ_OUT)
        caps = (C.c_size_t * 3)(len(reports), len(cigar), len(sj))
        self._chk(self.lib.dg_batch_download(self.ctx, reads.ctypes.data, reports.ctypes.data, cigar.ctypes.data, sj.ctypes.data, caps), "dg_batch_download")
        return BatchResult(reads, reports[:u[0]], cigar[:u[1]], sj[:u[2]])

    def map_batch(self, seq_off, rlen, flat) -> BatchResult:
        self.upload(seq_off, rlen, flat)
        self.run()
        return self.download()

    def download_compact(self) -> BatchResult:
        """the records through the compact types (dg_batch_download_compact), expanded to the full dtypes"""
        u = self._used
        reads = np.zeros(self._n, dtype=READ_C); reports = np.zeros(max(u[0], 1), dtype=REPORT_C)
        cigar = np.zeros(max(u[1], 1), dtype=np.uint32); sj = np.zeros(max(u[2], 1), dtype=SJ_OUT)
        caps = (C.c_size_t * 3)(len(reports), len(cigar), len(sj))
        n_ops = C.c_size_t(0)
        self._chk(self.lib.dg_batch_download_compact(self.ctx, reads.ctypes.data, reports.ctypes.data, cigar.ctypes.data, sj.ctypes.data, caps, C.byref(n_ops)), "dg_batch_download_compact")
        self.last_compact_ops = int(n_ops.value)
        r, p, cg = expand_compact(reads, reports[:u[0]], cigar[:int(n_ops.value)], self._rlen_of_batch)
        return BatchResult(r, p, cg, sj[:u[2]])

    def upload_packed(self, words, nlist, rlen_all, rlen=None):
        self._n = int(words.shape[0])
        self._rlen_of_batch = np.full(self._n, rlen_all, np.uint16) if rlen is None else np.asarray(rlen, np.uint16).copy()
        w = np.ascontiguousarray(words, np.uint32); nl = np.ascontiguousarray(nlist, np.uint32)
        rl = None if rlen is None else np.ascontiguousarray(rlen, np.uint16)
        self._keep = (w, nl, rl)
        self._chk(self.lib.dg_batch_upload_packed(self.ctx, self._n, int(rlen_all), None if rl is None else rl.ctypes.data, int(w.shape[1]) if w.ndim == 2 else 1,
                                                  w.ctypes.data, nl.ctypes.data if len(nl) else None, len(nl)), "dg_batch_upload_packed")

    def map_batch_packed(self, words, nlist, rlen_all, rlen=None) -> BatchResult:
        self.upload_packed(words, nlist, rlen_all, rlen)
        self.run()
        return self.download()

    def map_batch_compact(self, words, nlist, rlen_all, rlen=None) -> BatchResult:
        """dg_map_batch_compact in one call: packed reads in, compact records out (built inside the run), expanded to the full dtypes;
        arrays sized by a guess and grown on DG_ERR_CAPACITY the way a C caller would"""
        n = int(words.shape[0])
        w = np.ascontiguousarray(words, np.uint32); nl = np.ascontiguousarray(nlist, np.uint32)
        rl = None if rlen is None else np.ascontiguousarray(rlen, np.uint16)
        self._n = n
        self._rlen_of_batch = np.full(n, rlen_all, np.uint16) if rl is None else rl.copy()
        cap = [n + n // 8 + 64, n // 4 + 64, 64]
        while True:
            reads = np.zeros(max(n, 1), dtype=READ_C); reports = np.zeros(cap[0], dtype=REPORT_C)
            cigar = np.zeros(cap[1], dtype=np.uint32); sj = np.zeros(cap[2], dtype=SJ_OUT)
            caps = (C.c_size_t * 3)(*cap); used = (C.c_size_t * 3)()
            rc = self.lib.dg_map_batch_compact(self.ctx, n, None, None if rl is None else rl.ctypes.data, None, int(rlen_all), int(w.shape[1]) if w.ndim == 2 else 1,
                                               w.ctypes.data, nl.ctypes.data if len(nl) else None, len(nl),
                                               reads.ctypes.data, reports.ctypes.data, cigar.ctypes.data, sj.ctypes.data, caps, used)
            u = [int(x) for x in used]
            if rc == -4:                                   # DG_ERR_CAPACITY: `used` holds the need (of what is known so far)
                cap = [max(cap[0], u[0] + 16), max(2 * cap[1], u[1] + 16), max(cap[2], u[2] + 16)]
                continue
            self._chk(rc, "dg_map_batch_compact")
            self._used = u
            r, p, cg = expand_compact(reads[:n], reports[:u[0]], cigar[:u[1]], self._rlen_of_batch)
            return BatchResult(r, p, cg, sj[:u[2]])

    def pinned(self, shape, dtype) -> "PinnedArray":
        return PinnedArray(self.lib, shape, dtype)

    def device_reads_tensor(self, compact: bool = False):
        """torch uint8 view [n_reads, 36] (or [n_reads, 16] of the compact type) of the per-read records in HBM (for RCCL collectives)."""
        import torch
        ptrs = (C.c_void_p * 4)()
        if compact:
            self._chk(self.lib.dg_batch_device_ptrs_compact(self.ctx, ptrs), "dg_batch_device_ptrs_compact")
        else:
            self._chk(self.lib.dg_batch_device_ptrs(self.ctx, ptrs), "dg_batch_device_ptrs")

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (self._n, READ_C.itemsize if compact else READ_OUT.itemsize), "typestr": "|u1", "data": (int(ptrs[0]), False), "version": 2}
        return torch.as_tensor(v, device="cuda")

    def device_records_compact(self):
        """the four compact record arrays of the last compact run as torch uint8 views of HBM (dg_batch_device_records_compact):
        [dg_read_c bytes, dg_report_c bytes, stored CIGAR ops bytes, dg_sj_out bytes] -- for the RCCL gather to the writing rank"""
        import torch
        ptrs = (C.c_void_p * 4)(); counts = (C.c_size_t * 4)()
        self._chk(self.lib.dg_batch_device_records_compact(self.ctx, ptrs, counts), "dg_batch_device_records_compact")
        out = []
        for k, item in enumerate((READ_C.itemsize, REPORT_C.itemsize, 4, SJ_OUT.itemsize)):
            nbytes = int(counts[k]) * item

            class _View:
                pass
            v = _View()
            if nbytes == 0 or not ptrs[k]:
                out.append(torch.empty(0, dtype=torch.uint8, device="cuda"))
                continue
            v.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptrs[k]), False), "version": 2}
            out.append(torch.as_tensor(v, device="cuda"))
        return out

    def timings(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = self.lib.dg_last_timings(self.ctx, names, ms, 16)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def counters(self):
        out = (C.c_uint64 * 64)()
        n = self.lib.dg_last_counters(self.ctx, out, 64)
        keys = ["steps", "occ_blocks", "lf_steps", "sa_lookups", "seeds", "candidates", "nw_calls", "nw_cells", "reseed_calls", "reseed_window",
                "steps_executed", "occ_blocks_executed", "ktab_lookups", "lf_steps_executed", "direct_extensions", "k_seed_max_trips_per_read", "k_seed_wave_trips_max", "k_seed_wave_trips_sum", "k_reseed_trips", "k_reseed_wave_ticks_100mhz",
                "seedq_trips_begin", "seedq_trips_step", "seedq_trips_compare", "seedq_trips_locate", "seedq_trips_refill",
                "seedq_slots_begin", "seedq_slots_step", "seedq_slots_compare", "seedq_slots_locate", "seedq_slots_refill", "seedq_phases",
                "wave_ticks_k_seed_qf", "wave_ticks_k_seed_heavy", "wave_ticks_k_chain_heavy", "wave_ticks_k_pair", "wave_ticks_k_report", "general_path_units", "wave_chained_units", "batch_runs", "reruns_capacity_total", "reruns_scan_total",
                "k_reseed_windows_scanned_again_whole", "k_reseed_items", "k_reseed_chunks_through_pool"]
        return {keys[i]: int(out[i]) for i in range(min(n, len(keys)))}

    def probe_seeds(self, seq_off, rlen, flat):
        n = len(rlen)
        a, b, c = np.ascontiguousarray(seq_off, np.uint32), np.ascontiguousarray(rlen, np.uint16), np.ascontiguousarray(flat, np.uint8)
        cap = max(1024, n * 64)
        while True:
            so = np.zeros(n + 1, np.uint32); rp = np.zeros(cap, np.int32); sl = np.zeros(cap, np.int32); gp = np.zeros(cap, np.int64)
            used = C.c_size_t(0)
            rc = self.lib.dg_probe_seeds(self.ctx, n, a.ctypes.data, b.ctypes.data, c.ctypes.data, so.ctypes.data, rp.ctypes.data,
                                         sl.ctypes.data, gp.ctypes.data, cap, C.byref(used))
            if rc == -4:
                cap = int(used.value) + 16
                continue
            self._chk(rc, "dg_probe_seeds")
            u = int(used.value)
            return so, rp[:u], sl[:u], gp[:u]

    def probe_nw(self, pairs, mode: int = 0):
        """nw_alignment of (a, b) byte pairs through form `mode` of the kernels (dg_probe_nw_mode)."""
        n = len(pairs)
        a_off = np.zeros(n + 1, np.uint32); b_off = np.zeros(n + 1, np.uint32)
        for i, (x, y) in enumerate(pairs):
            a_off[i + 1] = a_off[i] + len(x); b_off[i + 1] = b_off[i] + len(y)
        a = np.frombuffer(b"".join(p[0] for p in pairs) + b"\0", dtype=np.uint8).copy()
        b = np.frombuffer(b"".join(p[1] for p in pairs) + b"\0", dtype=np.uint8).copy()
        cap = int(a_off[n]) + int(b_off[n]) + 16
        out_off = np.zeros(n + 1, np.uint32); out_len = np.zeros(n + 1, np.uint32)
        oa = np.zeros(cap, np.uint8); ob = np.zeros(cap, np.uint8)
        self._chk(self.lib.dg_probe_nw_mode(self.ctx, mode, n, a_off.ctypes.data, b_off.ctypes.data, a.ctypes.data, b.ctypes.data, out_off.ctypes.data,
                                            out_len.ctypes.data, oa.ctypes.data, ob.ctypes.data, cap), "dg_probe_nw_mode")
        res = []
        for i in range(n):
            o, l = int(out_off[i]), int(out_len[i])
            res.append((oa[o:o + l].tobytes(), ob[o:o + l].tobytes()))
        return res

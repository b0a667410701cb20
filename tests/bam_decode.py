"""Test helper: a from-the-specification BAM reader (BGZF blocks, header, records -> SAM text lines), used to check what the host
program's `-bo` writes (dart_amd/csrc/host/bam_writer.h) against SAM text.  Independent of the writer: nothing is shared with it."""
import struct, zlib


def bgzf_blocks(raw: bytes):
    """[(uncompressed bytes, compressed block size)]; checks the gzip / 'BC' framing, CRC32 and ISIZE of every block"""
    out, p = [], 0
    while p < len(raw):
        assert raw[p:p + 4] == b"\x1f\x8b\x08\x04", "not a BGZF block at %d" % p
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        assert xlen == 6 and raw[p + 12:p + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        data = zlib.decompress(raw[p + 18:p + bsize - 8], -15)
        crc, isize = struct.unpack_from("<II", raw, p + bsize - 8)
        assert crc == (zlib.crc32(data) & 0xFFFFFFFF) and isize == len(data) and isize <= 0xFF00
        out.append((data, bsize))
        p += bsize
    return out


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def decode(raw: bytes):
    """-> (header text, [(name, length)], [SAM line without newline], [bin of every record])"""
    blocks = bgzf_blocks(raw)
    assert blocks and blocks[-1][0] == b"" and blocks[-1][1] == 28, "no BGZF end-of-file block"
    data = b"".join(b for b, _ in blocks)
    assert data[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]; p += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, p)[0]; p += 4
        name = data[p:p + l_name - 1].decode(); assert data[p + l_name - 1] == 0; p += l_name
        refs.append((name, struct.unpack_from("<i", data, p)[0])); p += 4
    header_end = p
    lines, bins = [], []
    while p < len(data):
        bs = struct.unpack_from("<i", data, p)[0]; p += 4
        rec = data[p:p + bs]; p += bs
        refid, pos, l_rn, mapq, bin_, n_cig, flag, l_seq, mrefid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", rec, 0)
        q = 32
        name = rec[q:q + l_rn - 1].decode(); assert rec[q + l_rn - 1] == 0; q += l_rn
        cig = struct.unpack_from("<%dI" % n_cig, rec, q); q += 4 * n_cig
        cigar = "".join("%d%s" % (c >> 4, "MIDNSHP=X"[c & 15]) for c in cig) or "*"
        sb = rec[q:q + (l_seq + 1) // 2]; q += (l_seq + 1) // 2
        seq = "".join("=ACMGRSVTWYHKDBN"[(sb[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq)) or "*"
        qb = rec[q:q + l_seq]; q += l_seq
        qual = "*" if l_seq == 0 or all(x == 0xFF for x in qb) else "".join(chr(x + 33) for x in qb)
        tags = []
        while q < len(rec):
            tag = rec[q:q + 2].decode(); ty = chr(rec[q + 2]); q += 3
            if ty == "A": tags.append("%s:A:%s" % (tag, chr(rec[q]))); q += 1
            elif ty in "cCsSiI":
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[ty]
                v = struct.unpack_from(fmt, rec, q)[0]; q += struct.calcsize(fmt)
                lo, hi = {"c": (-128, -1), "C": (0, 255), "s": (-32768, -129), "S": (256, 65535), "i": (-2 ** 31, -32769), "I": (65536, 2 ** 32 - 1)}[ty]
                assert lo <= v <= hi, "integer tag %s=%d not in the smallest type that holds it (%s)" % (tag, v, ty)
                tags.append("%s:i:%d" % (tag, v))
            elif ty == "Z":
                e = rec.index(b"\0", q); tags.append("%s:Z:%s" % (tag, rec[q:e].decode())); q = e + 1
            else:
                raise AssertionError("tag type %s" % ty)
        rname = refs[refid][0] if refid >= 0 else "*"
        rnext = "*" if mrefid < 0 else ("=" if mrefid == refid else refs[mrefid][0])
        lines.append("\t".join([name, str(flag), rname, str(pos + 1), str(mapq), cigar, rnext, str(mpos + 1), str(tlen), seq, qual] + tags))
        rlen = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8))
        assert bin_ == reg2bin(pos, pos + (rlen if rlen > 0 else 1)), "bin of %s" % name
        bins.append(bin_)
    # the header ends its BGZF block: no record shares a block with it
    acc = 0
    for b, _ in blocks:
        acc += len(b)
        if acc >= header_end:
            assert acc == header_end, "the header does not end on a block boundary"
            break
    return text, refs, lines, bins

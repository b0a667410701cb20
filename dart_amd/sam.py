"""SAM text and junctions.tab from the flat result records -- the host-side mirror of
OutputPairedAlignments / OutputSingledAlignments (Mapping.cpp:208-369), the SAM header
(Mapping.cpp:741-751), UpdateGlobalSJMap + OutputSpliceJunctions (Mapping.cpp:567-577,683-716)
and the statistics block (Mapping.cpp:812-822).  Python twin of dart_amd/csrc/host/sam_writer.cpp;
used by the tests and bench.py to turn records into the reference's bytes.
"""
from __future__ import annotations

import numpy as np

_COMP = {ord(a): b for a, b in zip("ACGTacgt", "TGCATGCA")}
_XS_A = ["", " XS:A:+", " XS:A:-"]


def revcomp_str(s: str) -> str:   # GetComplementarySeq, tools.cpp:3-29
    return "".join(_COMP.get(ord(c), "N") for c in reversed(s))


def sam_header(names, lengths) -> str:
    out = ["@PG\tID:Dart\tPN:Dart\tVN:1.4.6\n"]
    for n, l in zip(names, lengths):
        out.append("@SQ\tSN:%s\tLN:%d\n" % (n, int(l)))
    return "".join(out)


def cigar_str(cigar, rep) -> str:
    o, n = int(rep["cigar_off"]), int(rep["n_cigar"])
    return "".join("%d%s" % (int(x) >> 4, "MIDNS"[int(x) & 15]) for x in cigar[o:o + n])


class Stats:
    def __init__(self):
        self.total = self.unique = self.unmapped = self.paired = 0


def format_records(headers, seqs, quals, reads, reports, cigar, chr_names, paired: bool, multi_hit: bool = False,
                   unique_only: bool = False, fastq: bool = True, stats: Stats | None = None) -> str:
    """seqs/quals as stored by the loader (mate 2 reverse-complemented / reversed). Returns SAM body."""
    out = []
    n = len(reads)
    st = stats or Stats()
    st.total += n
    is_pair_mode = paired and n % 2 == 0
    for k in range(n):
        r = reads[k]
        mate2 = is_pair_mode and (k & 1)
        rp = reports[int(r["rep_off"]): int(r["rep_off"]) + int(r["n_rep"])]
        seq = seqs[k]
        q = quals[k] if fastq else "*"
        if r["score"] == 0:
            st.unmapped += 1
            out.append("%s\t%d\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\tAS:i:0\tXS:i:0\n" % (headers[k], rp[0]["flag"], seq, q))
            continue
        if unique_only and not r["mapq"] > 3:
            continue
        if r["mapq"] == 50:
            st.unique += 1
        if is_pair_mode:
            m = reads[k ^ 1]
            mp = reports[int(m["rep_off"]): int(m["rep_off"]) + int(m["n_rep"])]
        alt = altq = None
        for j in range(int(r["best"]), int(r["n_rep"])):
            pr = rp[j]
            show = (pr["aln_score"] > 0) if is_pair_mode else (pr["aln_score"] == r["score"])
            if show:
                if pr["sj_type"] == -1:
                    xs = 0
                elif pr["sj_type"] in (0, 2):
                    xs = 2 if mate2 else 1
                else:
                    xs = 1 if mate2 else 2
                use_alt = (pr["bdir"] == 1) if mate2 else (pr["bdir"] == 0)
                if use_alt and alt is None:
                    alt = revcomp_str(seq)
                    altq = q[::-1] if fastq else "*"
                s_out = alt if use_alt else seq
                q_out = (altq if use_alt else q) if fastq else "*"
                cg = cigar_str(cigar, pr)
                pj = int(pr["paired_idx"])
                if is_pair_mode and pj != -1 and mp[pj]["aln_score"] > 0:
                    a, b = (mp[pj], pr) if mate2 else (pr, mp[pj])     # a = mate 1's report, b = mate 2's
                    l1 = len(seqs[k ^ 1]) if mate2 else len(seq)
                    l2 = len(seq) if mate2 else len(seqs[k ^ 1])
                    dist = int(np.int32(int(b["pos"]) - int(a["pos"]) + (l2 if a["bdir"] else -l1)))
                    if mate2:
                        dist = -dist
                    elif j == int(r["best"]):
                        st.paired += 2
                    out.append("%s\t%d\t%s\t%d\t%d\t%s\t=\t%d\t%d\t%s\t%s\tNM:i:%d\tAS:i:%d\tXS:i:%d%s\n" % (
                        headers[k], pr["flag"], chr_names[int(pr["chr"])], pr["pos"], r["mapq"], cg, mp[pj]["pos"], dist,
                        s_out, q_out, r["mis_num"], r["score"], r["sub_score"], _XS_A[xs]))
                else:
                    out.append("%s\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s\tNM:i:%d\tAS:i:%d\tXS:i:%d%s\n" % (
                        headers[k], pr["flag"], chr_names[int(pr["chr"])], pr["pos"], r["mapq"], cg, s_out, q_out,
                        r["mis_num"], r["score"], r["sub_score"], _XS_A[xs]))
                if not is_pair_mode and not multi_hit:
                    break
            if is_pair_mode and not multi_hit:
                break
    return "".join(out)


def junction_table(sj, chr_names, chr_off, chr_len, l_pac: int) -> str:
    """std::map<(g1,g2)> order; chromosome by ChrLocMap.lower_bound(g1) (AbsLoc2ChrLoc :683-695)."""
    if len(sj) == 0:
        return ""
    keys = {}
    for g1, g2 in zip(sj["g1"].tolist(), sj["g2"].tolist()):
        keys[(g1, g2)] = keys.get((g1, g2), 0) + 1
    n = len(chr_names)
    loc = [int(chr_off[i] + chr_len[i] - 1) for i in range(n)] + [int(2 * l_pac - chr_off[i] - 1) for i in reversed(range(n))]
    who = list(range(n)) + list(reversed(range(n)))
    out = []
    for (g1, g2) in sorted(keys):
        lo = int(np.searchsorted(loc, g1, side="left"))
        if lo >= 2 * n:
            continue
        c = who[lo]
        out.append("%s\t%d\t%d\t%d\n" % (chr_names[c], g1 + 1 - int(chr_off[c]), g2 + 1 - int(chr_off[c]), keys[(g1, g2)]))
    return "".join(out)

"""Accuracy scorers beside parity -- what the reference ships as Evaluation/{eva, SJ_Eva, FluxEva}:

  eva      sensitivity + average sequence identity of a SAM file against the genome  (GeneralEvaluation.cpp:28-140)
  SJ_Eva   how many predicted junctions lie within 5 bp of a true one                (SJ_Evaluation.cpp:94-116)
  FluxEva  how many alignments start where the read was simulated from               (FluxEvaluation.cpp:26-80: `flux_eva`, on
           Flux-style read names "chr:left-rightW..."; `mapping_accuracy` asks the same question of a batch's record arrays
           with the synthetic generator's own truth)

    python -m dart_amd.evaluate eva OUT.sam GENOME.fa
    python -m dart_amd.evaluate sj  junctions.tab TRUE_JUNCTIONS.txt
    python -m dart_amd.evaluate flux OUT.sam

`mapping_accuracy` scores the record arrays of a batch directly (bench.py's `accuracy`).  Host-side tools: no GPU involved.
"""
from __future__ import annotations

import re
import sys

import numpy as np

_CIG = re.compile(r"(\d+)([MIDNSHP=X])")


def alignment_identity(cigar: str, pos0: int, qseq: str, ref: str):
    """(identical columns, alignment columns) of one record: the pairwise alignment rebuilt from the CIGAR as eva does
    (GeneralEvaluation.cpp:28-77): M/=/X consume both, I the read (gap in the genome), D the genome (gap in the read), S the
    read only, N the genome only; an op that would run past the read or the chromosome ends the walk."""
    rpos, gpos, same, cols = 0, pos0, 0, 0
    rlen, glen = len(qseq), len(ref)
    for num, op in _CIG.findall(cigar):
        n = int(num)
        if op in "MIS=X" and rpos + n > rlen:
            break
        if op in "MD=X" and gpos + n > glen:
            break
        if op == "I":
            rpos += n; cols += n
        elif op == "D":
            gpos += n; cols += n
        elif op == "S":
            rpos += n
        elif op == "N":
            gpos += n
        elif op != "H":
            a = np.frombuffer(qseq[rpos:rpos + n].upper().encode(), np.uint8)
            b = np.frombuffer(ref[gpos:gpos + n].encode(), np.uint8)
            same += int((a == b).sum()); cols += n
            rpos += n; gpos += n
    return same, cols


def eva(sam_lines, chroms: dict):
    """eva's two numbers.  sam_lines: iterable of SAM text lines; chroms: name -> sequence (upper case).
    Per query name only the first two records count (GeneralEvaluation.cpp:120-125); a record is aligned when it has a CIGAR,
    a position and a known chromosome; identity is accumulated per record in thousandths, truncated, as the reference does."""
    total = aligned = 0
    idy_sum = 0
    prev, hits = None, 0
    for line in sam_lines:
        if not line or line[0] == "@":
            continue
        f = line.rstrip("\n").split("\t")
        if f[0] != prev:
            prev, hits = f[0], 1
        else:
            hits += 1
            if hits > 2:
                continue
        total += 1
        pos0 = int(f[3]) - 1
        if f[5] == "*" or pos0 < 0 or f[2] not in chroms:
            continue
        aligned += 1
        same, cols = alignment_identity(f[5], pos0, f[9], chroms[f[2]])
        if cols > 0:
            idy_sum += 1000 * same // cols
    return {"records": total, "aligned": aligned, "sensitivity": aligned / total if total else 0.0,
            "avg_identity": idy_sum / aligned / 1000.0 if aligned else 0.0}


def sj_eva(predicted, truth, slack: int = 5):
    """SJ_Eva: predicted / truth = iterables of (chr, start, end); a prediction is right when both ends lie within `slack` - 1 bp
    of the same true junction (abs difference < 5, SJ_Evaluation.cpp:106).  Also the recall the reference does not print."""
    by_chr = {}
    for c, s, e in truth:
        by_chr.setdefault(c, []).append((int(s), int(e)))
    arr = {c: np.asarray(sorted(v), np.int64) for c, v in by_chr.items()}
    pred = [(c, int(s), int(e)) for c, s, e in predicted]
    right = 0
    found = {c: np.zeros(len(a), bool) for c, a in arr.items()}
    for c, s, e in pred:
        a = arr.get(c)
        if a is None:
            continue
        lo = np.searchsorted(a[:, 0], s - slack + 1, side="left"); hi = np.searchsorted(a[:, 0], s + slack - 1, side="right")
        ok = np.nonzero(np.abs(a[lo:hi, 1] - e) < slack)[0]
        if len(ok):
            right += 1
            found[c][lo + ok] = True
    n_true = sum(len(a) for a in arr.values())
    return {"predicted": len(pred), "annotated": right, "precision": right / len(pred) if pred else 0.0,
            "true_junctions": n_true, "recall": (sum(int(f.sum()) for f in found.values()) / n_true) if n_true else 0.0}


def _atoi(t: str) -> int:
    """C atoi: leading blanks, an optional sign, then digits; anything else ends the number (no digits: 0)"""
    m = re.match(r"\s*([+-]?\d+)", t)
    return int(m.group(1)) if m else 0


def flux_eva(sam_lines):
    """FluxEva (FluxEvaluation.cpp:26-80).  A read's name says where the Flux simulator took it from: "<chr>:<left>-<right>W..." (the text
    between the first ':' and the first '-' and from there to the first 'W', read with atoi).  Per name only the first two records
    count; a record without CIGAR is "empty", one with MAPQ 0 is left out, the rest are right when the chromosome is the name's and POS
    lies in [left, right].  Returns the program's three numbers: right, scored (= total - empty - MAPQ 0) and the accuracy in per cent,
    rounded as it prints it ((int)(1000 * (right / scored + 0.0005)) / 10)."""
    total = right = low = empty = 0
    prev, hits = None, 0
    for line in sam_lines:
        line = line.rstrip("\n")
        if line == "":
            break                                       # (the program stops at the first empty line)
        if line[0] == "@":
            continue
        f = line.split()
        header, p_chr, gpos, mapq, cigar = f[0], f[2], _atoi(f[3]), _atoi(f[4]), f[5]
        # IdentifyGenomicRegion, FluxEvaluation.cpp:10-24: std::string::substr(pos, len) with int arithmetic -- a character that is not
        # there gives -1, and a negative length wraps to "the rest of the string"
        sub = lambda t, pos, n: t[pos:] if n < 0 else t[pos:pos + n]
        p1, p2, p3 = header.find(":"), header.find("-"), header.find("W")
        r_chr = sub(header, 0, p1)
        left = _atoi(sub(header, p1 + 1, p2 - p1 - 1))
        right_pos = _atoi(sub(header, p2 + 1, p3 - p2 + 1))
        if header != prev:
            prev, hits = header, 1
        else:
            hits += 1
        if hits > 2:
            continue
        total += 1
        if cigar == "*":
            empty += 1
        elif mapq == 0:
            low += 1
        elif p_chr == r_chr and left <= gpos <= right_pos:
            right += 1
    scored = total - empty - low
    acc = int(1000 * (right / scored + 0.0005)) / 10.0 if total > 0 and scored > 0 else 0.0
    return {"right": right, "scored": scored, "accuracy_percent": acc, "records": total, "empty": empty, "mapq0": low}


def mapping_accuracy(res, truth, tolerance: int = 10):
    """FluxEva's question on a batch's records: of the plain fragments (no planted indel or intron: their truth is exact), how
    many reads are mapped, and how many best alignments start within `tolerance` bp of where the read was sampled from
    ((POS - leading soft clip) vs the true leftmost position).  res: dart_amd.host.BatchResult of interleaved pairs."""
    rd, rp, cg = res.reads, res.reports, res.cigar
    best = rp[np.clip(rd["rep_off"] + rd["best"], 0, len(rp) - 1)]
    npair = len(rd) // 2
    tp = np.stack([truth["pos1"][:npair], truth["pos2"][:npair]], 1).reshape(-1)
    tc = np.repeat(truth["chr"][:npair], 2)
    plain = np.repeat(truth["plain"][:npair], 2)
    mapped = rd["score"] > 0
    first_op = cg[np.clip(best["cigar_off"], 0, max(len(cg) - 1, 0))] if len(cg) else np.zeros(len(best), np.uint32)
    lead_s = np.where((best["n_cigar"] > 0) & ((first_op & 15) == 4), first_op >> 4, 0).astype(np.int64)     # leading soft clip
    ok = mapped & (best["chr"] == tc) & (np.abs(best["pos"] - lead_s - tp) <= tolerance)
    conf = plain & (rd["mapq"] > 0)                                # FluxEva leaves MAPQ 0 out
    return {"reads": int(plain.sum()), "mapped_frac": round(float(mapped[plain].mean()), 5), "correct_frac": round(float(ok[plain].mean()), 5),
            "correct_frac_mapq_gt0": round(float(ok[conf].mean()), 5) if conf.any() else None, "tolerance_bp": tolerance,
            "note": "plain fragments of batch 0; (POS - leading soft clip) of the best report vs the position the read was sampled from; the rest are reads "
                    "from planted repeat families placed at another copy"}


def _read_fasta(path):
    chroms, name, parts = {}, None, []
    for line in open(path):
        if line.startswith(">"):
            if name is not None:
                chroms[name] = "".join(parts).upper()
            name, parts = line[1:].split()[0], []
        else:
            parts.append(line.strip())
    if name is not None:
        chroms[name] = "".join(parts).upper()
    return chroms


def main(argv):
    if len(argv) == 3 and argv[0] == "eva":
        r = eva(open(argv[1]), _read_fasta(argv[2]))
        print("sensitivity = %d / %d = %.3f, AvgSeqIdy = %.3f" % (r["aligned"], r["records"], r["sensitivity"], r["avg_identity"]))
        return 0
    if len(argv) == 3 and argv[0] == "sj":
        tab = lambda p: [tuple(l.split()[:3]) for l in open(p) if l.strip()]
        r = sj_eva(tab(argv[1]), tab(argv[2]))
        print("%d of %d predicted junctions annotated (%.3f); %d true junctions, recall %.3f" % (r["annotated"], r["predicted"], r["precision"], r["true_junctions"], r["recall"]))
        return 0
    if len(argv) == 2 and argv[0] == "flux":
        r = flux_eva(open(argv[1]))
        print("Acc = %d / %d = %.2f" % (r["right"], r["scored"], r["accuracy_percent"]))
        return 0
    print(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

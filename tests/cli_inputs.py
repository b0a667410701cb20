"""Input files for the command-line parity tests (FASTQ, FASTA, .gz, interlaced, odd counts, lower case)."""
import gzip, os
import numpy as np
import common
from dart_amd import synth

VARIANTS = [
    (["-f", "q1.fq", "-f2", "q2.fq", "-mis", "5"], "paired fastq"),
    (["-f", "fa1.fa", "-f2", "fa2.fa", "-mis", "5"], "paired fasta"),
    (["-f", "q1.fq.gz", "-f2", "q2.fq.gz", "-mis", "5", "-all_sj"], "paired fastq.gz"),
    (["-f", "fa1.fa.gz", "-mis", "3"], "single fasta.gz"),
    (["-f", "inter.fq", "-p", "-mis", "5"], "interlaced -p"),
    (["-f", "q1.fq", "-mis", "2", "-unique"], "single, odd read count"),
    (["-f", "q1.fq", "q2.fq", "-mis", "2"], "two single-end libraries"),
    (["-f", "q1.fq", "-p"], "-p with an odd number of reads"),
]


def make(workdir):
    c = common.build_case("pe101_spliced", workdir)
    d = os.path.join(workdir, "cli")
    if os.path.exists(os.path.join(d, "inter.fq")):
        return c, d
    os.makedirs(d, exist_ok=True)
    m1, m2 = synth.make_reads(c["genome"], 3001, rlen=101, seed=77, spliced_frac=0.3)
    m1[5, 10] = ord("a"); m1[6, 50] = ord("n"); m2[7, 3] = ord("R")
    synth.write_fasta_reads(os.path.join(d, "fa1.fa"), m1, 1); synth.write_fasta_reads(os.path.join(d, "fa2.fa"), m2, 2)
    synth.write_fastq(os.path.join(d, "q1.fq"), m1, 1); synth.write_fastq(os.path.join(d, "q2.fq"), m2, 2)
    for f in ("q1.fq", "q2.fq", "fa1.fa"):
        open(os.path.join(d, f + ".gz"), "wb").write(gzip.compress(open(os.path.join(d, f), "rb").read()))
    a = open(os.path.join(d, "q1.fq"), "rb").read().split(b"\n"); b = open(os.path.join(d, "q2.fq"), "rb").read().split(b"\n")
    with open(os.path.join(d, "inter.fq"), "wb") as f:
        for i in range(0, len(a) - 1, 4):
            f.write(b"\n".join(a[i:i + 4]) + b"\n"); f.write(b"\n".join(b[i:i + 4]) + b"\n")
    return c, d

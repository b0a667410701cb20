"""CPU suite, only where oracle/_ref exists (this container): the oracle's command line against
the reference's own object code, byte for byte, on input-format variants and a seeded fuzz."""
import os, subprocess
import pytest
import common, oracle_py, cli_inputs
from dart_amd import synth

pytestmark = pytest.mark.skipif(not os.path.exists(oracle_py.REF_HARNESS), reason="oracle/_ref not built (no /root/reference)")


def run_both(d, prefix, flags):
    rr = subprocess.run([oracle_py.REF_HARNESS, "map", "-i", prefix] + flags + ["-o", "ref.sam", "-j", "ref.j"], cwd=d, stdout=subprocess.PIPE, check=True)
    ro = subprocess.run([oracle_py.ORACLE_CLI, "-i", prefix] + flags + ["-o", "orc.sam", "-j", "orc.j", "-t", "3"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    # the statistics block of stdout (Mapping.cpp:812-822): the harness prints it from the reference's own counters
    assert common.stats_block(rr.stdout) == common.stats_block(ro.stdout) != "", (rr.stdout[-600:], ro.stdout[-600:])
    a, b = open(os.path.join(d, "ref.sam")).read(), open(os.path.join(d, "orc.sam")).read()
    assert a == b, common.first_diff(b, a)
    assert open(os.path.join(d, "ref.j")).read() == open(os.path.join(d, "orc.j")).read()


@pytest.mark.parametrize("flags,label", cli_inputs.VARIANTS, ids=[v[1] for v in cli_inputs.VARIANTS])
def test_oracle_cli_matches_reference_on_input_variants(flags, label, workdir):
    oracle_py.build()
    c, d = cli_inputs.make(workdir)
    run_both(d, c["prefix"], flags)


def test_oracle_matches_reference_on_fresh_fuzz(workdir):
    """a genome and reads that are in no fixture: repeats-heavy, short reads, 3 % errors"""
    oracle_py.build()
    d = os.path.join(workdir, "fuzz"); os.makedirs(d, exist_ok=True)
    g = synth.make_genome([700000, 300000], seed=91, repeat_scale=80.0, n_introns=200)
    g.write_fasta(os.path.join(d, "g.fa"))
    subprocess.run([oracle_py.REF_INDEXER, "g.fa", "g"], cwd=d, stdout=subprocess.DEVNULL, check=True)
    m1, m2 = synth.make_reads(g, 6000, rlen=76, seed=92, spliced_frac=0.25, sub_rate=0.03, indel_frac=0.05, n_frac=0.02)
    synth.write_fastq(os.path.join(d, "a.fq"), m1, 1); synth.write_fastq(os.path.join(d, "b.fq"), m2, 2)
    for flags in ([], ["-mis", "5"], ["-mis", "4", "-unique", "-max_dup", "500"]):
        run_both(d, os.path.join(d, "g"), ["-f", "a.fq", "-f2", "b.fq"] + flags)


def test_oracle_matches_reference_on_reads_with_odd_characters(workdir):
    """fresh genome, paired and single-end reads with a literal '-' (a gap to AddNewCigarElements), lower case, N and IUPAC letters planted at 1 % of the positions,
    -mis 12 and 30: the reads that reach the reference's string code in ways the ACGTN fixtures do not"""
    import numpy as np
    oracle_py.build()
    d = os.path.join(workdir, "fuzz_odd"); os.makedirs(d, exist_ok=True)
    g = synth.make_genome([500000, 250000], seed=191, repeat_scale=40.0, n_introns=300)
    g.write_fasta(os.path.join(d, "g.fa"))
    subprocess.run([oracle_py.REF_INDEXER, "g.fa", "g"], cwd=d, stdout=subprocess.DEVNULL, check=True)
    rng = np.random.default_rng(192)
    m1, m2 = synth.make_reads(g, 5000, rlen=125, seed=193, spliced_frac=0.4, sub_rate=0.01, indel_frac=0.2, n_frac=0.0)
    def odd(m):
        return np.where(rng.random(m.shape) < 0.01, rng.choice(np.frombuffer(b"---acgtRYn", np.uint8), size=m.shape), m).astype(np.uint8)
    synth.write_fastq(os.path.join(d, "a.fq"), odd(m1), 1); synth.write_fastq(os.path.join(d, "b.fq"), odd(m2), 2)
    for flags in (["-f", "a.fq", "-f2", "b.fq", "-mis", "12"], ["-f", "a.fq", "-mis", "30"], ["-f", "a.fq", "b.fq", "-mis", "12", "-m"]):
        run_both(d, os.path.join(d, "g"), flags)

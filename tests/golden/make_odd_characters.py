"""Regenerates tests/golden/odd_characters.mis12.sam.gz / .junctions.tab: what the REFERENCE's object code (oracle/_ref/ref_harness, built by `make -C oracle ref` from
/root/reference) writes for tests/common.py::odd_character_reads -- single-end reads with a literal '-', lower case, N and IUPAC letters -- over the pe101_spliced case's
genome (indexed by the reference's indexer), -mis 12.  Runs only where /root/reference exists; the outputs are committed.

  python tests/golden/make_odd_characters.py
"""
import gzip, hashlib, json, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(HERE))
import common
from dart_amd import synth

spec = common.MANIFEST["cases"]["pe101_spliced"]
g = synth.make_genome(spec["lengths"], seed=spec["gseed"], repeat_scale=spec["rscale"], n_introns=spec["nintr"])
d = tempfile.mkdtemp()
g.write_fasta(os.path.join(d, "g.fa"))
subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "bwt_index"), "g.fa", "g"], cwd=d, stdout=subprocess.DEVNULL)
seqs = common.odd_character_reads(g)
common.write_se_fastq(os.path.join(d, "odd.fq"), seqs)
subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "ref_harness"), "map", "-i", "g", "-f", "odd.fq", "-mis", "12", "-o", "ref.sam", "-j", "ref.j"], cwd=d, stdout=subprocess.DEVNULL)
sam = open(os.path.join(d, "ref.sam"), "rb").read()
with gzip.GzipFile(os.path.join(HERE, "odd_characters.mis12.sam.gz"), "wb", mtime=0) as f:
    f.write(sam)
open(os.path.join(HERE, "odd_characters.mis12.junctions.tab"), "w").write(open(os.path.join(d, "ref.j")).read())
meta = {"reads": len(seqs), "reads_sha256": hashlib.sha256(b"\n".join(seqs)).hexdigest(), "sam_sha256": hashlib.sha256(sam).hexdigest(),
        "reads_with_a_dash": sum(1 for s in seqs if b"-" in s), "mapped": sum(1 for l in sam.decode().splitlines() if l and l[0] != "@" and l.split("\t")[2] != "*")}
json.dump(meta, open(os.path.join(HERE, "odd_characters.json"), "w"), indent=1)
print(json.dumps(meta, indent=1))

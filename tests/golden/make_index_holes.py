"""Regenerates tests/golden/index_holes.json: the digests of the five files the REFERENCE indexer (oracle/_ref/bwt_index, built by
`make -C oracle ref` from /root/reference) writes for tests/common.py::write_holes_fasta's FASTA -- ambiguous bases, holes, header
comments.  Runs only where /root/reference exists; the digests are committed.

  python tests/golden/make_index_holes.py
"""
import hashlib, json, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(HERE))
import common

d = tempfile.mkdtemp()
fa = os.path.join(d, "holes.fa")
common.write_holes_fasta(fa)
subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "bwt_index"), fa, os.path.join(d, "holes")], stdout=subprocess.DEVNULL)
out = {"fasta_sha256": hashlib.sha256(open(fa, "rb").read()).hexdigest(),
       "index_sha256": {ext: hashlib.sha256(open(os.path.join(d, "holes." + ext), "rb").read()).hexdigest() for ext in ("bwt", "sa", "pac", "ann", "amb")},
       "ann": open(os.path.join(d, "holes.ann")).read(), "amb": open(os.path.join(d, "holes.amb")).read()}
json.dump(out, open(os.path.join(HERE, "index_holes.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

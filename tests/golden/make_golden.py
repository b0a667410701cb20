"""Regenerates tests/golden/* from the reference's own object code (oracle/_ref, built by
`make -C oracle ref` from /root/reference).  Runs only where /root/reference exists; the outputs
(data: inputs' digests and expected outputs) are committed, this script documents how.

  python tests/golden/make_golden.py
"""
import gzip, hashlib, json, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
from dart_amd import synth

REF = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
IDX = os.path.join(ROOT, "oracle", "_ref", "bwt_index")

# name -> genome spec, read spec, flag sets
CASES = {
    "pe101_spliced": dict(lengths=[300000, 200000], gseed=20, rscale=20.0, nintr=200, npairs=1500, rlen=101, rseed=7,
                          spliced=0.3, paired=True, sub=0.01, flags=[[], ["-mis", "5"], ["-mis", "5", "-all_sj", "-max_dup", "1000"]]),
    "se100": dict(lengths=[400000], gseed=23, rscale=40.0, nintr=100, npairs=2000, rlen=100, rseed=10,
                  spliced=0.1, paired=False, sub=0.01, flags=[[], ["-mis", "3", "-unique"]]),
    "pe151_spliced": dict(lengths=[250000, 150000, 100000], gseed=22, rscale=30.0, nintr=300, npairs=1000, rlen=151, rseed=9,
                          spliced=0.3, paired=True, sub=0.005, flags=[["-mis", "5"], ["-mis", "2", "-min_intron", "10", "-max_intron", "200000"]]),
}

# chr20-sized genomes whose reference-built index files are pinned by digest (the files themselves are ~100 MB)
BIG_INDEX = {
    "chr20_planted": dict(lengths=[64444167], gseed=20, names=["chr20"], model="planted"),       # = bench.py --genome chr20
    "chr20_human": dict(lengths=[64444167], gseed=20, names=["chr20"], model="human"),           # = bench.py --genome chr20 --genome-model human
}

def stats_block(stdout_bytes):
    """the '\t# of ...' lines of a run's stdout, with without the junction file's name (same helper as tests/common.py)"""
    import re
    lines = [l for l in stdout_bytes.decode("latin1").replace("\r", "\n").split("\n") if l.startswith("\t# of")]
    return "".join(re.sub(r" \(file: .*\)$", "", l) + "\n" for l in lines)

def case_inputs(spec, d):
    g = synth.make_genome(spec["lengths"], seed=spec["gseed"], repeat_scale=spec["rscale"], n_introns=spec["nintr"])
    m1, m2 = synth.make_reads(g, spec["npairs"], rlen=spec["rlen"], seed=spec["rseed"], spliced_frac=spec["spliced"],
                              paired=spec["paired"], sub_rate=spec["sub"], indel_frac=0.05, n_frac=0.01)
    return g, m1, m2

def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        if a is not None:
            h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()

def main():
    manifest = {}
    d = tempfile.mkdtemp()
    for name, spec in CASES.items():
        g, m1, m2 = case_inputs(spec, d)
        fa = os.path.join(d, name + ".fa"); g.write_fasta(fa)
        subprocess.check_call([IDX, fa, os.path.join(d, name)], stdout=subprocess.DEVNULL)
        synth.write_fastq(os.path.join(d, name + "_1.fq"), m1, 1)
        files = ["-f", os.path.join(d, name + "_1.fq")]
        if spec["paired"]:
            synth.write_fastq(os.path.join(d, name + "_2.fq"), m2, 2); files += ["-f2", os.path.join(d, name + "_2.fq")]
        entry = {"inputs_sha256": digest(g.codes, m1, m2), "index_sha256": {}, "runs": []}
        for ext in ("bwt", "sa", "pac", "ann", "amb"):
            entry["index_sha256"][ext] = hashlib.sha256(open(os.path.join(d, name + "." + ext), "rb").read()).hexdigest()
        for k, fl in enumerate(spec["flags"]):
            sam = os.path.join(d, "o.sam"); junc = os.path.join(d, "o.junc"); dump = os.path.join(d, "o.dump")
            ref_stdout = subprocess.run([REF, "map", "-i", os.path.join(d, name)] + files + ["-o", sam, "-j", junc, "-dump", dump] + fl, stdout=subprocess.PIPE, check=True).stdout
            base = "%s.run%d" % (name, k)
            # the statistics block the reference prints at the end of a run (Mapping.cpp:812-822; the harness prints it from the reference's
            # own counters): the "# of" lines, without the junction file's name
            with open(os.path.join(HERE, base + ".stats.txt"), "w") as f: f.write(stats_block(ref_stdout))
            with gzip.GzipFile(os.path.join(HERE, base + ".sam.gz"), "wb", mtime=0) as f: f.write(open(sam, "rb").read())
            with open(os.path.join(HERE, base + ".junctions.tab"), "wb") as f: f.write(open(junc, "rb").read())
            if k == 0:   # per-stage dump (seeds, candidates, final seeds) of the first run
                with gzip.GzipFile(os.path.join(HERE, base + ".stages.gz"), "wb", mtime=0) as f: f.write(open(dump, "rb").read())
            entry["runs"].append({"flags": fl, "base": base})
        manifest[name] = entry
    # nw_alignment known answers: the 8 from SURVEY 8a + 1500 random pairs
    rng = np.random.default_rng(5)
    pairs = [("ACGT", "AGT"), ("ACGTACGTAC", "ACGTTACGTAC"), ("AAAAACCCCC", "AAAAAGGGGGCCCCC"), ("ACGTNACGT", "ACGTAACGT"),
             ("ACGTACGT", "ACGACGT"), ("A", "ACGTACGT"), ("ACGTACGTACGTACGT", "ACGTACGTTTACGTACGT"), ("GATTACA", "GCATGCT")]
    alpha = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for i in range(1500):
        m = int(rng.integers(1, 70)); a = alpha[rng.choice(5, m, p=[.24, .24, .24, .24, .04])]
        if rng.random() < 0.7:
            b = a.copy()
            mut = rng.random(len(b)) < 0.12
            b[mut] = alpha[rng.integers(0, 4, int(mut.sum()))]
            cut = sorted(rng.integers(0, len(b) + 1, 2)); 
            if rng.random() < 0.5: b = np.concatenate([b[:cut[0]], b[cut[1]:]])
            else: b = np.concatenate([b[:cut[0]], alpha[rng.integers(0, 4, int(rng.integers(1, 9)))], b[cut[0]:]])
        else:
            b = alpha[rng.integers(0, 4, int(rng.integers(1, 100)))]
        if len(b) == 0: b = alpha[:1]
        pairs.append((a.tobytes().decode(), b.tobytes().decode()))
    inp = "".join("%s %s\n" % p for p in pairs)
    out = subprocess.run([REF, "nw"], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    with gzip.GzipFile(os.path.join(HERE, "nw_known_answers.tsv.gz"), "wb", mtime=0) as f:
        for (a, b), line in zip(pairs, out):
            o1, o2 = line.split()
            f.write(("%s\t%s\t%s\t%s\n" % (a, b, o1, o2)).encode())
    # larger nw_alignment answers with the genome side in ACGT only (what RefSequence holds): these reach the wave-wide forms of
    # the kernels (8-lane groups up to 64 columns, the whole-wave anti-diagonal form beyond, several 64-column blocks, traceback
    # words of more than 16 rows); own generator so the first file's bytes do not change
    rng2 = np.random.default_rng(77)
    big = []
    for i in range(420):
        m = int(rng2.integers(1, 260)) if i % 3 else int(rng2.integers(1, 40))
        a = alpha[rng2.choice(5, m, p=[.245, .245, .245, .245, .02])]
        b = a.copy(); b[b == ord("N")] = ord("A")
        mut = rng2.random(len(b)) < 0.08
        b[mut] = alpha[rng2.integers(0, 4, int(mut.sum()))]
        for _ in range(int(rng2.integers(0, 3))):
            c0 = int(rng2.integers(0, len(b) + 1))
            if rng2.random() < 0.5: b = np.concatenate([b[:c0], b[c0 + int(rng2.integers(1, 12)):]])
            else: b = np.concatenate([b[:c0], alpha[rng2.integers(0, 4, int(rng2.integers(1, 40)))], b[c0:]])
        if i % 7 == 0: b = alpha[rng2.integers(0, 4, int(rng2.integers(1, 330)))]
        if len(b) == 0: b = alpha[:1]
        big.append((a.tobytes().decode(), b.tobytes().decode()))
    inp = "".join("%s %s\n" % p for p in big)
    out = subprocess.run([REF, "nw"], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    with gzip.GzipFile(os.path.join(HERE, "nw_known_answers_large.tsv.gz"), "wb", mtime=0) as f:
        for (a, b), line in zip(big, out):
            o1, o2 = line.split()
            f.write(("%s\t%s\t%s\t%s\n" % (a, b, o1, o2)).encode())
    # BWT_Search known answers on the first case's index: every start position of 60 reads
    name = "pe101_spliced"; spec = CASES[name]; g, m1, m2 = case_inputs(spec, d)
    reads = "".join(m1[i].tobytes().decode() + "\n" for i in range(0, 60))
    out = subprocess.run([REF, "search", "-i", os.path.join(d, name)], input=reads.encode(), stdout=subprocess.PIPE, check=True).stdout.decode()
    out = out[out.index("\n") + 1:] if out.startswith("Load") else out
    with gzip.GzipFile(os.path.join(HERE, "bwt_search_known_answers.txt.gz"), "wb", mtime=0) as f:
        f.write(out.encode())
    # the reference indexer's files for genomes of the bench's size class (64 444 167 bp, one chromosome: BASELINE configs[1]) -- the large
    # paths of the GPU index builder (tests/test_gpu_index.py::test_gpu_index_builder_matches_reference_indexer_at_chr20_size) are pinned on
    # these, not only the <= 500 kb cases above.  ~40 s of the reference's bwt_index per genome.
    big = {}
    for bname, kw in BIG_INDEX.items():
        g = synth.make_genome(kw["lengths"], seed=kw["gseed"], names=kw["names"], model=kw["model"])
        fa = os.path.join(d, bname + ".fa"); g.write_fasta(fa)
        subprocess.check_call([IDX, fa, os.path.join(d, bname)], stdout=subprocess.DEVNULL)
        big[bname] = {"spec": kw, "codes_sha256": digest(g.codes),
                      "index_sha256": {ext: hashlib.sha256(open(os.path.join(d, bname + "." + ext), "rb").read()).hexdigest() for ext in ("bwt", "sa", "pac", "ann", "amb")}}
        for ext in ("fa", "bwt", "sa", "pac", "ann", "amb"):
            os.remove(os.path.join(d, bname + "." + ext))
    json.dump({"cases": CASES, "manifest": manifest, "big_index": big}, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("golden written:", sorted(os.listdir(HERE)))

if __name__ == "__main__":
    main()

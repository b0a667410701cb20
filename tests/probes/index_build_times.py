"""Measurement probe: builds the GRCh38-sized index once on the GPU and prints the builder's phase log (seconds since the
build began and seconds inside dg_sort_pairs at every line) plus the digests of the four files, so that two builder versions can
be compared for time AND bytes.  usage: python tests/probes/index_build_times.py [total_bp] [model]   (64444167 = the chr20-sized genome of the golden digests; DART_INDEX_CLI=1 adds the `dart index` leg)"""
import hashlib, os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from dart_amd import index_build, synth

total = int(sys.argv[1]) if len(sys.argv) > 1 else 0
model = sys.argv[2] if len(sys.argv) > 2 else "planted"
names, lengths = (bench.GRCH38_NAMES, bench.GRCH38) if total == 0 else (["chr20"], [total]) if total == 64444167 else (["chrA", "chrB"], [total - total // 3, total // 3])
t = time.time()
g = synth.make_genome(lengths, seed=20, names=names, model=model)
print("genome %d bp (%s) generated in %.1f s" % (sum(lengths), model, time.time() - t), flush=True)
d = tempfile.mkdtemp(dir="/dev/shm")
prefix = os.path.join(d, "g")
t = time.time()
info = index_build.build_index_from_genome(g, prefix, log=lambda m: print(m, flush=True))
print("index built in %.1f s: %s" % (time.time() - t, info), flush=True)
def digests(prefix):
    out = {}
    for ext in (".pac", ".ann", ".amb", ".bwt", ".sa"):
        h = hashlib.sha256()
        with open(prefix + ext, "rb") as f:
            while True:
                b = f.read(1 << 24)
                if not b: break
                h.update(b)
        out[ext] = h.hexdigest()
        print("%s %d bytes sha256 %s" % (ext, os.path.getsize(prefix + ext), out[ext]), flush=True)
        os.unlink(prefix + ext)
    return out

want = digests(prefix)
if os.environ.get("DART_INDEX_CLI") == "1":          # the same genome as FASTA through `dart index ref.fa prefix`, process start to exit
    import subprocess
    fa = os.path.join(d, "g.fa")
    t = time.time(); g.write_fasta(fa); print("FASTA written in %.1f s (%d bytes)" % (time.time() - t, os.path.getsize(fa)), flush=True)
    t = time.time()
    r = subprocess.run([os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "dart_amd", "dart"), "index", fa, os.path.join(d, "cli")], capture_output=True, text=True)
    print(r.stdout + r.stderr, flush=True)
    print("dart index: rc %d, %.1f s from process start to exit" % (r.returncode, time.time() - t), flush=True)
    os.unlink(fa)
    got = digests(os.path.join(d, "cli"))
    print("dart index files identical to the library call's:", got == want, flush=True)
os.rmdir(d)

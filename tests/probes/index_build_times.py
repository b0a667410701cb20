"""Measurement probe: builds the GRCh38-sized index once on the GPU and prints the builder's phase log (seconds since the
build began and seconds inside dg_sort_pairs at every line) plus the digests of the four files, so that two builder versions can
be compared for time AND bytes.  usage: python tests/probes/index_build_times.py [total_bp] [model]"""
import hashlib, os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from dart_amd import index_build, synth

total = int(sys.argv[1]) if len(sys.argv) > 1 else 0
model = sys.argv[2] if len(sys.argv) > 2 else "planted"
names, lengths = (bench.GRCH38_NAMES, bench.GRCH38) if total == 0 else (["chrA", "chrB"], [total - total // 3, total // 3])
t = time.time()
g = synth.make_genome(lengths, seed=20, names=names, model=model)
print("genome %d bp (%s) generated in %.1f s" % (sum(lengths), model, time.time() - t), flush=True)
d = tempfile.mkdtemp(dir="/dev/shm")
prefix = os.path.join(d, "g")
t = time.time()
info = index_build.build_index_from_genome(g, prefix, log=lambda m: print(m, flush=True))
print("index built in %.1f s: %s" % (time.time() - t, info), flush=True)
for ext in (".pac", ".ann", ".amb", ".bwt", ".sa"):
    h = hashlib.sha256()
    with open(prefix + ext, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b: break
            h.update(b)
    print("%s %d bytes sha256 %s" % (ext, os.path.getsize(prefix + ext), h.hexdigest()), flush=True)
    os.unlink(prefix + ext)
os.rmdir(d)

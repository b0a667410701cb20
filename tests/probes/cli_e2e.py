"""Usage: python profiles/probes/cli_e2e.py [pairs] [chr20|grch38|<bp>]
End-to-end timing of the product's `dart` command line (FASTQ on /tmp -> SAM on /tmp) on the bench workload, and a
byte comparison of its SAM / junctions with the oracle's command line.  Usage: python profiles/probes/cli_e2e.py [pairs]"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from dart_amd import synth
import oracle_py
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
label, gnames, glens = bench.genome_spec(sys.argv[2] if len(sys.argv) > 2 else "chr20")
print(label + ")")
prefix, g = bench.prepare_index("/tmp/dart_bench_cache", (gnames, glens), 0, lambda: None)
m1, m2 = synth.make_reads(g, pairs, rlen=101, seed=1000, sub_rate=0.01, indel_frac=0.02, n_frac=0.002)
import shutil
need = pairs * 2 * 560                     # FASTQ in + SAM out
d = "/dev/shm/cli_e2e" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > need * 1.2 else "/tmp/cli_e2e"
os.makedirs(d, exist_ok=True)
print("files under", d)
synth.write_fastq_fast(os.path.join(d, "1.fq"), m1, 1); synth.write_fastq_fast(os.path.join(d, "2.fq"), m2, 2)
dart = os.path.join(ROOT, "dart_amd", "dart")
variants = [{"DART_INFLIGHT": "2", "DART_STREAMING": "1"}, {"DART_INFLIGHT": "2"}, {"DART_INFLIGHT": "2", "DART_WRITE": "mmap"}, {"DART_INFLIGHT": "2", "DART_PINNED": "1"}, {"DART_INFLIGHT": "2", "DART_BATCH": "500000"}]
if os.environ.get("CLI_E2E_VARIANTS"):          # "A=1,B=2;C=3": one run per ;-separated set
    variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in os.environ["CLI_E2E_VARIANTS"].split(";")]
for env_extra in variants:
    env = dict(os.environ, DART_TIMING="1", **env_extra)
    for f_ in ("gpu.sam", "gpu.j"):                 # (truncating a multi-GB tmpfs file at open is not part of the job)
        if os.path.exists(os.path.join(d, f_)): os.remove(os.path.join(d, f_))
    t = time.time()
    r = subprocess.run([dart, "-i", prefix, "-f", "1.fq", "-f2", "2.fq", "-o", "gpu.sam", "-j", "gpu.j", "-t", "16", "-mis", "5"], cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    dt = time.time() - t
    print(env_extra, "dart wall %.2f s = %.2f M reads/s (incl. index load + dg_init)" % (dt, 2 * pairs / dt / 1e6), r.stderr.decode().strip().splitlines()[-1:])
if os.environ.get("CLI_E2E_BAM", "1") == "1":
  t = time.time()
  r = subprocess.run([dart, "-i", prefix, "-f", "1.fq", "-f2", "2.fq", "-bo", "gpu.bam", "-j", "gpu.j", "-t", "16", "-mis", "5"], cwd=d, env=dict(os.environ, DART_TIMING="1", DART_INFLIGHT="2"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
  dt = time.time() - t
  print("-bo (BAM, 16 compression threads): dart wall %.2f s = %.2f M reads/s, %d bytes" % (dt, 2 * pairs / dt / 1e6, os.path.getsize(os.path.join(d, "gpu.bam"))), r.stderr.decode().strip().splitlines()[-1:])
if os.environ.get("CLI_E2E_COMPARE", "1") == "1":
    oracle_py.build()
    t = time.time()
    subprocess.run([oracle_py.ORACLE_CLI, "-i", prefix, "-f", "1.fq", "-f2", "2.fq", "-o", "orc.sam", "-j", "orc.j", "-t", "16", "-mis", "5"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    print("oracle CLI wall %.2f s" % (time.time() - t))
    same = subprocess.run(["cmp", "gpu.sam", "orc.sam"], cwd=d).returncode == 0 and subprocess.run(["cmp", "gpu.j", "orc.j"], cwd=d).returncode == 0
    print("SAM + junctions byte-identical to the oracle CLI:", same, os.path.getsize(os.path.join(d, "gpu.sam")), "bytes")

"""Builds a synthetic genome of the given size (24 chromosomes with GRCh38 proportions when >= 2 Gbp), indexes it on the GPU
with the memory-lean sorter, and runs bench.py's workload on it.  Usage: python profiles/probes/big_index.py GENOME_BP [steps]"""
import os, sys, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
glen = int(float(sys.argv[1]))
steps = sys.argv[2] if len(sys.argv) > 2 else "16"
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
env = dict(os.environ)
if os.environ.get("FORCE_BUCKETED") == "1":
    env["DART_SA_BUCKETED"] = "1"
t = time.time()
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--genome", str(glen), "--steps", steps, "--warmup", "4", "--cpu-sample-pairs", "32000"],
                   env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
open(os.path.join(ROOT, "gpurun_out", "bench_big_%d.err" % glen), "w").write(r.stderr.decode())
print(r.stderr.decode()[-3000:])
line = r.stdout.decode().strip().splitlines()[-1] if r.stdout.strip() else "{}"
d = json.loads(line)
print("wall %.1f s" % (time.time() - t), "returncode", r.returncode)
print({k: d.get(k) for k in ("value", "ms_per_step", "cpu_baseline")})
print(d.get("kernels_ms_one_batch_in_flight"))
open(os.path.join(ROOT, "gpurun_out", "bench_big_%d.json" % glen), "w").write(line + "\n")

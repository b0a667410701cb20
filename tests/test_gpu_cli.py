"""GPU suite: the product's `dart` command line (C++ host over the C ABI) against the golden SAM
of the reference and against the oracle's command line on every input-format variant."""
import os, subprocess
import pytest
import common, oracle_py, cli_inputs
from dart_amd import synth

pytestmark = pytest.mark.gpu
DART = os.path.join(common.ROOT, "dart_amd", "dart")


def test_dart_cli_reproduces_golden_sam(workdir):
    import __graft_entry__ as ge
    ge.build()
    for name in sorted(common.MANIFEST["cases"]):
        c = common.build_case(name, workdir)
        d = os.path.join(workdir, "gold_" + name); os.makedirs(d, exist_ok=True)
        synth.write_fastq(os.path.join(d, "1.fq"), c["m1"], 1)
        files = ["-f", "1.fq"]
        if c["spec"]["paired"]:
            synth.write_fastq(os.path.join(d, "2.fq"), c["m2"], 2); files += ["-f2", "2.fq"]
        for run in c["runs"]:
            subprocess.run([DART, "-i", c["prefix"]] + files + ["-o", "o.sam", "-j", "o.j", "-t", "4"] + run["flags"], cwd=d, stdout=subprocess.DEVNULL, check=True)
            got, want = open(os.path.join(d, "o.sam")).read(), common.golden_sam(run["base"])
            assert got == want, common.first_diff(got, want)
            assert open(os.path.join(d, "o.j")).read() == common.golden_junctions(run["base"])


@pytest.mark.parametrize("flags,label", cli_inputs.VARIANTS, ids=[v[1] for v in cli_inputs.VARIANTS])
def test_dart_cli_matches_oracle_cli_on_input_variants(flags, label, workdir):
    oracle_py.build()
    c, d = cli_inputs.make(workdir)
    env = dict(os.environ, DART_BATCH="5000")      # several batches, so batch seams are exercised too
    subprocess.run([DART, "-i", c["prefix"]] + flags + ["-o", "gpu.sam", "-j", "gpu.j"], cwd=d, stdout=subprocess.DEVNULL, check=True, env=env)
    subprocess.run([oracle_py.ORACLE_CLI, "-i", c["prefix"]] + flags + ["-o", "orc.sam", "-j", "orc.j"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    a, b = open(os.path.join(d, "orc.sam")).read(), open(os.path.join(d, "gpu.sam")).read()
    assert a == b, common.first_diff(b, a)
    assert open(os.path.join(d, "orc.j")).read() == open(os.path.join(d, "gpu.j")).read()


def test_dart_cli_error_behaviour(workdir):
    r = subprocess.run([DART, "-intron", "5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Error! Unknow parameter: -intron" in r.stderr
    r = subprocess.run([DART, "-i", "nowhere", "-f", __file__], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Please specify a valid reference index" in r.stderr
    r = subprocess.run([DART, "-v"], stdout=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout.startswith(b"DART v1.4.6")


def test_bench_two_ranks_through_the_self_launcher(workdir):
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts two ranks itself (torch.distributed.run), each maps
    its own distinct batches host to host, the per-read records are gathered to rank 0 inside the timed region, and ONE JSON line
    comes back with n_gpus = 2.  On a one-GPU box the ranks share the card and talk over gloo (DART_BENCH_REHEARSE=1): the launch,
    threading, ordering and gather logic of the N>1 path, not its speed."""
    import json, subprocess, sys
    env = dict(os.environ, DART_BENCH_REHEARSE="1", DART_BENCH_CACHE=os.path.join(workdir, "bench2_cache"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--genome", "3000000", "--pairs", "20000", "--batches", "3", "--steps", "2",
                        "--warmup", "1", "--inflight", "2", "--cpu-sample-pairs", "4000"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "gather" in line["config"]["parallelism"] and "host-to-host" in line["config"]["workload"]
    assert line["cpu_baseline"]["gpu_records_identical_on_sample"] is True
    assert line["roofline"]["frac"] > 0 and line["accuracy"]["correct_frac"] > 0.9

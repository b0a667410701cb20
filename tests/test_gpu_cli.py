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
            r = subprocess.run([DART, "-i", c["prefix"]] + files + ["-o", "o.sam", "-j", "o.j", "-t", "4"] + run["flags"], cwd=d, stdout=subprocess.PIPE, check=True)
            got, want = open(os.path.join(d, "o.sam")).read(), common.golden_sam(run["base"])
            assert got == want, common.first_diff(got, want)
            assert open(os.path.join(d, "o.j")).read() == common.golden_junctions(run["base"])
            # the statistics block of the run's stdout (Mapping.cpp:812-822) against the reference's own
            assert common.stats_block(r.stdout) == common.golden_stats(run["base"]), (run["base"], r.stdout[-600:])


def test_dart_cli_reads_with_odd_characters_reproduce_the_reference_sam(workdir):
    """Single-end reads with a literal '-', lower case, N and IUPAC letters, -mis 12, through both host pipelines: the SAM and the junctions the reference's object code wrote
    for them (tests/golden/odd_characters.*; half of the reads hold a dash -- what the string forms of dg_report.h are kept for)."""
    import gzip, json, hashlib
    c = common.build_case("pe101_spliced", workdir)
    meta = json.load(open(os.path.join(common.GOLDEN, "odd_characters.json")))
    seqs = common.odd_character_reads(c["genome"])
    assert hashlib.sha256(b"\n".join(seqs)).hexdigest() == meta["reads_sha256"], "the read generator drifted from the golden inputs"
    d = os.path.join(workdir, "odd_characters_gpu"); os.makedirs(d, exist_ok=True)
    common.write_se_fastq(os.path.join(d, "odd.fq"), seqs)
    want = gzip.open(os.path.join(common.GOLDEN, "odd_characters.mis12.sam.gz"), "rt").read()
    for env in (dict(os.environ, DART_BATCH="1000"), dict(os.environ, DART_BATCH="1000", DART_STREAMING="1")):
        subprocess.run([DART, "-i", c["prefix"], "-f", "odd.fq", "-mis", "12", "-o", "o.sam", "-j", "o.j", "-t", "4"], cwd=d, stdout=subprocess.DEVNULL, check=True, env=env)
        got = open(os.path.join(d, "o.sam")).read()
        assert got == want, common.first_diff(got, want)
        assert open(os.path.join(d, "o.j")).read() == open(os.path.join(common.GOLDEN, "odd_characters.mis12.junctions.tab")).read()


def test_dart_cli_bam_output(workdir):
    """`-bo`: the BAM that `dart` writes (host/bam_writer.h; the reference: htslib's sam_parse1 + sam_write1 per SAM line), decoded by
    the specification-based reader of the tests and compared with the reference-generated golden SAM; several batches, so the BGZF
    stream crosses batch seams.  (" XS:A:+", which the reference joins with a blank, does not survive its own BAM output either.)"""
    import re
    import bam_decode
    for name in sorted(common.MANIFEST["cases"]):
        c = common.build_case(name, workdir)
        d = os.path.join(workdir, "bam_" + name); os.makedirs(d, exist_ok=True)
        synth.write_fastq(os.path.join(d, "1.fq"), c["m1"], 1)
        files = ["-f", "1.fq"]
        if c["spec"]["paired"]:
            synth.write_fastq(os.path.join(d, "2.fq"), c["m2"], 2); files += ["-f2", "2.fq"]
        run = c["runs"][0]
        subprocess.run([DART, "-i", c["prefix"]] + files + ["-bo", "o.bam", "-j", "o.j", "-t", "3"] + run["flags"], cwd=d, stdout=subprocess.DEVNULL, check=True,
                       env=dict(os.environ, DART_BATCH="4000"))
        hdr, refs, lines, bins = bam_decode.decode(open(os.path.join(d, "o.bam"), "rb").read())
        want = common.golden_sam(run["base"]).splitlines()
        assert hdr == "".join(l + "\n" for l in want if l.startswith("@"))
        body = [re.sub(r" XS:A:[+-]$", "", l) for l in want if not l.startswith("@")]
        assert len(lines) == len(body)
        for a, b in zip(lines, body):
            assert a == b
        assert open(os.path.join(d, "o.j")).read() == common.golden_junctions(run["base"])


@pytest.mark.parametrize("host_path", ["parallel", "streaming"])
@pytest.mark.parametrize("flags,label", cli_inputs.VARIANTS, ids=[v[1] for v in cli_inputs.VARIANTS])
def test_dart_cli_matches_oracle_cli_on_input_variants(flags, label, host_path, workdir):
    """both host pipelines of `dart` (the parallel one takes plain FASTQ: fast_fastq.h; the streaming one everything, and all of it
    with DART_STREAMING=1) against the oracle's command line, byte for byte"""
    oracle_py.build()
    c, d = cli_inputs.make(workdir)
    env = dict(os.environ, DART_BATCH="5000")      # several batches, so batch seams are exercised too
    if host_path == "streaming":
        env["DART_STREAMING"] = "1"
    rg = subprocess.run([DART, "-i", c["prefix"]] + flags + ["-o", "gpu.sam", "-j", "gpu.j", "-t", "3"], cwd=d, stdout=subprocess.PIPE, check=True, env=env)
    ro = subprocess.run([oracle_py.ORACLE_CLI, "-i", c["prefix"]] + flags + ["-o", "orc.sam", "-j", "orc.j"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    a, b = open(os.path.join(d, "orc.sam")).read(), open(os.path.join(d, "gpu.sam")).read()
    assert a == b, common.first_diff(b, a)
    assert open(os.path.join(d, "orc.j")).read() == open(os.path.join(d, "gpu.j")).read()
    assert common.stats_block(rg.stdout) == common.stats_block(ro.stdout) != "", (rg.stdout[-600:], ro.stdout[-600:])


@pytest.mark.parametrize("host_path", ["parallel", "streaming"])
def test_dart_cli_awkward_fastq(host_path, workdir):
    """FASTQ the readers must agree on with the reference's GetNextEntry (GetData.cpp:77-132): headers with blanks / slashes / tabs and
    repeated '@', quality lines that start with '@' or '+', qualities longer than the read (shorter ones are undefined behaviour in
    the reference: GetData.cpp:158-159 copies rlen bytes out of a shorter string), lower-case and IUPAC bases, a last
    line without newline, 4001+ reads so the 4000-read chunk rule decides the unpaired tail of an odd -p file, and -- second file --
    a record without bases in the middle (the stream ends there)"""
    oracle_py.build()
    c, d0 = cli_inputs.make(workdir)
    d = os.path.join(workdir, "awkward"); os.makedirs(d, exist_ok=True)
    m1, m2 = synth.make_reads(c["genome"], 4603, rlen=101, seed=78, spliced_frac=0.2)
    def rec(i, seq, tag):
        s = seq.tobytes().decode()
        if i % 7 == 1: s = s[:40].lower() + s[40:]
        if i % 11 == 2: s = s[:10] + "R" + s[11:]
        q = "".join(chr(33 + (i * 7 + k * 3) % 41) for k in range(len(s)))
        if i % 5 == 0: q = "@" + q[1:]
        if i % 13 == 3: q = "+" + q[1:]
        if i % 19 == 5: q = q + "IIII"
        h = ["@r%d/%s" % (i, tag), "@@r%d extra words" % i, "@r%d\tx" % i, "@r%d" % i][i % 4]
        return "%s\n%s\n+%s\n%s\n" % (h, s, "" if i % 3 else h[1:], q)
    with open(os.path.join(d, "a1.fq"), "w") as f:
        f.write("".join(rec(i, m1[i], "1") for i in range(4603))[:-1])            # no newline at the end
    with open(os.path.join(d, "a2.fq"), "w") as f:
        f.write("".join(rec(i, m2[i], "2") for i in range(4603)))
    with open(os.path.join(d, "inter_odd.fq"), "w") as f:
        f.write("".join(rec(i, m1[i], "1") + (rec(i, m2[i], "2") if i < 4302 else "") for i in range(4303)))
    with open(os.path.join(d, "stop.fq"), "w") as f:
        f.write("".join(rec(i, m1[i], "1") for i in range(700)) + "@empty\n\n+\n\n" + "".join(rec(i, m1[i], "1") for i in range(700, 900)))
    env = dict(os.environ, DART_BATCH="4000")
    if host_path == "streaming":
        env["DART_STREAMING"] = "1"
    for flags in (["-f", "a1.fq", "-f2", "a2.fq", "-mis", "5"], ["-f", "inter_odd.fq", "-p", "-mis", "5"], ["-f", "stop.fq", "-mis", "3"], ["-f", "a1.fq", "a2.fq", "-mis", "2", "-m"]):
        subprocess.run([DART, "-i", c["prefix"]] + flags + ["-o", "gpu.sam", "-j", "gpu.j", "-t", "5"], cwd=d, stdout=subprocess.DEVNULL, check=True, env=env)
        subprocess.run([oracle_py.ORACLE_CLI, "-i", c["prefix"]] + flags + ["-o", "orc.sam", "-j", "orc.j"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        a, b = open(os.path.join(d, "orc.sam"), "rb").read(), open(os.path.join(d, "gpu.sam"), "rb").read()
        assert a == b, (flags, common.first_diff(b.decode("latin1"), a.decode("latin1")))
        assert open(os.path.join(d, "orc.j")).read() == open(os.path.join(d, "gpu.j")).read()


@pytest.mark.parametrize("host_path", ["parallel", "streaming"])
def test_dart_cli_three_contexts_in_flight_keep_input_order(host_path, workdir):
    """DART_GPUS=1 DART_INFLIGHT=3 (three contexts on the device, each with its own host thread, batches finishing out of order) and small
    batches: the SAM is still the oracle command line's, byte for byte -- the ordered writer puts batch k behind batch k-1 whichever
    context finished first (the reference's only multi-worker guarantee is its OutputLock, Mapping.cpp:644-664: ours is stronger, input order).
    DART_GPUS=n (n devices in one process) adds devices to the same pool of contexts: ordering is the same mechanism."""
    oracle_py.build()
    c, d = cli_inputs.make(workdir)
    env = dict(os.environ, DART_BATCH="4000", DART_GPUS="1", DART_INFLIGHT="3")
    if host_path == "streaming":
        env["DART_STREAMING"] = "1"
    flags = ["-f", "q1.fq", "q1.fq", "q1.fq", "-f2", "q2.fq", "q2.fq", "q2.fq", "-mis", "5"]          # three libraries: 9 000 pairs, several batches per library
    subprocess.run([DART, "-i", c["prefix"]] + flags + ["-o", "gpu3.sam", "-j", "gpu3.j", "-t", "4"], cwd=d, stdout=subprocess.DEVNULL, check=True, env=env)
    subprocess.run([oracle_py.ORACLE_CLI, "-i", c["prefix"]] + flags + ["-o", "orc3.sam", "-j", "orc3.j"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    a, b_ = open(os.path.join(d, "orc3.sam")).read(), open(os.path.join(d, "gpu3.sam")).read()
    assert a == b_, common.first_diff(b_, a)
    assert open(os.path.join(d, "orc3.j")).read() == open(os.path.join(d, "gpu3.j")).read()


@pytest.mark.parametrize("host_path", ["parallel", "streaming"])
def test_dart_cli_two_devices_in_one_pool_keep_input_order(host_path, workdir):
    """`dart` drives every device of the node from one pool of mapping threads (one root context + clones per device, one ordered writer:
    Mapping.cpp:644-664,792-793).  A one-GPU box has one device, so that path never ran: DART_SAME_DEVICE_TIMES=2 opens
    device 0 twice -- two index replicas, two roots, 2 x 2 contexts in flight, small batches finishing out of order -- and the SAM and the
    junctions must still be the oracle command line's, byte for byte."""
    oracle_py.build()
    c, d = cli_inputs.make(workdir)
    env = dict(os.environ, DART_BATCH="3000", DART_SAME_DEVICE_TIMES="2", DART_INFLIGHT="2")
    if host_path == "streaming":
        env["DART_STREAMING"] = "1"
    flags = ["-f", "q1.fq", "q1.fq", "q1.fq", "-f2", "q2.fq", "q2.fq", "q2.fq", "-mis", "5"]
    subprocess.run([DART, "-i", c["prefix"]] + flags + ["-o", "gpu2d.sam", "-j", "gpu2d.j", "-t", "4"], cwd=d, stdout=subprocess.DEVNULL, check=True, env=env)
    subprocess.run([oracle_py.ORACLE_CLI, "-i", c["prefix"]] + flags + ["-o", "orc2d.sam", "-j", "orc2d.j"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    a, b_ = open(os.path.join(d, "orc2d.sam")).read(), open(os.path.join(d, "gpu2d.sam")).read()
    assert a == b_, common.first_diff(b_, a)
    assert open(os.path.join(d, "orc2d.j")).read() == open(os.path.join(d, "gpu2d.j")).read()


def test_dart_index_writes_the_reference_indexers_files(workdir):
    """`dart index ref.fa prefix` (main.cpp:125-127 -> bwa_idx_build): FASTA with ambiguous bases, holes and header comments (.gz too) -> the five
    files are the bytes the REFERENCE's bwt_index wrote (tests/golden/index_holes.json); then the golden genomes written as FASTA -> the digests
    of tests/golden/manifest.json; and `dart -i` maps with an index `dart index` built."""
    import gzip, json
    gold = json.load(open(os.path.join(common.GOLDEN, "index_holes.json")))
    fa = os.path.join(workdir, "holes_gpu.fa")
    common.write_holes_fasta(fa)
    with open(fa, "rb") as f, gzip.open(fa + ".gz", "wb") as g:
        g.write(f.read())
    for src, tag in ((fa, "plain"), (fa + ".gz", "gz")):
        prefix = os.path.join(workdir, "holes_gpu_" + tag)
        r = subprocess.run([DART, "index", src, prefix], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout
        for ext, want in gold["index_sha256"].items():
            assert common.sha(prefix + "." + ext) == want, (tag, ext, r.stdout)
    for name in sorted(common.MANIFEST["cases"]):
        c = common.build_case(name, workdir)
        fa = os.path.join(workdir, "cli_index_%s.fa" % name)
        c["genome"].write_fasta(fa)
        prefix = os.path.join(workdir, "cli_index_" + name)
        subprocess.run([DART, "index", fa, prefix], check=True, stdout=subprocess.DEVNULL)
        for ext, want in common.MANIFEST["manifest"][name]["index_sha256"].items():
            assert common.sha(prefix + "." + ext) == want, (name, ext)
    # the last one, used: same SAM as with the test's own index
    d = os.path.join(workdir, "cli_index_map"); os.makedirs(d, exist_ok=True)
    synth.write_fastq(os.path.join(d, "1.fq"), c["m1"], 1)
    files = ["-f", "1.fq"]
    if c["spec"]["paired"]:
        synth.write_fastq(os.path.join(d, "2.fq"), c["m2"], 2); files += ["-f2", "2.fq"]
    run = c["runs"][0]
    subprocess.run([DART, "-i", prefix] + files + ["-o", "o.sam", "-j", "o.j"] + run["flags"], cwd=d, stdout=subprocess.DEVNULL, check=True)
    assert open(os.path.join(d, "o.sam")).read() == common.golden_sam(run["base"])


def test_gpu_index_builder_fasta_path_matches_reference_indexer(workdir):
    """dart_amd/index_build.py from the FASTA with ambiguous bases, on the GPU (both drivers of the kernels)"""
    import json
    from dart_amd import index_build
    gold = json.load(open(os.path.join(common.GOLDEN, "index_holes.json")))
    fa = os.path.join(workdir, "holes_gpu_py.fa")
    common.write_holes_fasta(fa)
    for driver in ("", "python"):
        os.environ["DART_INDEX_DRIVER"] = driver
        try:
            prefix = os.path.join(workdir, "holes_gpu_py_" + (driver or "native"))
            index_build.build_index_from_fasta(fa, prefix, device="cuda")
        finally:
            del os.environ["DART_INDEX_DRIVER"]
        for ext, want in gold["index_sha256"].items():
            assert common.sha(prefix + "." + ext) == want, (driver, ext)


def test_dart_cli_error_behaviour(workdir):
    r = subprocess.run([DART, "-intron", "5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Error! Unknow parameter: -intron" in r.stderr
    r = subprocess.run([DART, "-i", "nowhere", "-f", __file__], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"Please specify a valid reference index" in r.stderr
    r = subprocess.run([DART, "-v"], stdout=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout.startswith(b"DART v1.4.6")


def test_rccl_single_rank_carries_the_record_arrays(workdir):
    """backend "nccl" (= RCCL) with ONE rank -- all a one-GPU box allows: the library's HBM record arrays go through the all_gather and the
    batched point-to-point calls of the multi-GPU gather (dart_amd/dist.py) and arrive intact (tests/rccl_smoke.py, a child process with its
    own process group)."""
    import subprocess, sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(common.HERE, "rccl_smoke.py"), workdir], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "rccl smoke ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_bench_two_ranks_through_the_self_launcher(workdir):
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts two ranks itself (torch.distributed.run), each maps
    its own distinct batches host to host, the per-read records are gathered to rank 0 inside the timed region, and ONE JSON line
    comes back with n_gpus = 2.  On a one-GPU box the ranks share the card and talk over gloo (DART_BENCH_REHEARSE=1): the launch,
    threading, ordering and gather logic of the N>1 path, not its speed."""
    import json, subprocess, sys
    env = dict(os.environ, DART_BENCH_REHEARSE="1", DART_BENCH_CACHE=os.path.join(workdir, "bench2_cache"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--genome", "3000000", "--pairs", "20000", "--batches", "3", "--steps", "2", "--weak",
                        "--warmup", "1", "--inflight", "2", "--cpu-sample-pairs", "4000", "--verify-gather"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "gather" in line["config"]["parallelism"] and "host-to-host" in line["config"]["workload"]
    assert line["scaling"] == "weak" and line["gather"]["mode"] == "full" and line["gather"]["bytes_received_by_rank0_total"] > 0
    assert line["gather"]["verified_against_single_rank_mapping"] is True          # rank 0's gathered record set == its own mapping of the same reads
    assert line["value_with_writer_download"] > 0                                  # the second pass: rank 0 also brings the gathered records to host memory
    assert line["cpu_baseline"]["gpu_records_identical_on_sample"] is True and line["cpu_baseline"]["value_t1"] > 0
    assert line["roofline"]["frac"] > 0 and line["accuracy"]["correct_frac"] > 0.9


def test_bench_two_ranks_strong_scaling_one_job_sharded(workdir):
    """BASELINE configs[3] as written -- ONE job sharded over the ranks (`--total-pairs`, "scaling": "strong"), here 70 001 pairs over two ranks
    (35 000 + 35 001: a partial last batch and ranks with different batch counts), the full SAM-order gather inside the timed region, and rank 0's
    gathered records compared with its own single-rank mapping of the same reads.  gloo on one GPU (DART_BENCH_REHEARSE=1): logic, not speed."""
    import json, subprocess, sys
    env = dict(os.environ, DART_BENCH_REHEARSE="1", DART_BENCH_CACHE=os.path.join(workdir, "bench2_cache"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "2", "--genome", "3000000", "--pairs", "20000", "--total-pairs", "70001", "--steps", "2",
                        "--warmup", "1", "--inflight", "2", "--no-cpu-baseline", "--verify-gather"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and "70001 pairs" in line["config"]["parallelism"]
    assert abs(line["value"] * 1e6 * line["ms_per_step"] * 1e-3 / (2 * 70001) - 1) < 1e-3  # value = the whole job's reads over the step time
    assert line["gather"]["verified_against_single_rank_mapping"] is True
    assert line["value_weak_scaling"] > 0 and line["phases_s"]["total"] > 0          # the weak-scaling rate beside it (every rank's own whole batches), and where the run's time went

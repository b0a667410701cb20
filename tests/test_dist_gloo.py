"""CPU suite: the N>1 path (shard -> map -> gather to rank 0) with world_size 2 over gloo.  The
per-rank mapping is played by the oracle here (no GPU in this container); what is under test is the
sharding and the record gather/rebase that bench.py and the multi-GPU driver use."""
import os, subprocess, sys, textwrap
import numpy as np
import common


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import common, oracle_py
    from dart_amd import host, dist as ddist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = common.build_case("pe101_spliced", {workdir!r})
    orc = oracle_py.Oracle(c["prefix"])
    n_pairs = c["reads"].shape[0] // 2
    lo, hi = ddist.shard_bounds(n_pairs, world, rank)
    so, rl, flat = host.pack_reads(c["reads"][2 * lo: 2 * hi])
    p = orc.params(paired=1, max_mismatch=5)
    local = orc.map_batch(p, so, rl, flat, threads=2)
    got = ddist.gather_records(*local)
    if rank == 0:
        so, rl, flat = host.pack_reads(c["reads"])
        want = orc.map_batch(p, so, rl, flat, threads=2)
        for g, w in zip(got, want):
            assert np.array_equal(g, w), "gathered records differ from the single-rank run"
        print("GATHER_OK", len(got[0]), len(got[1]))
    dist.destroy_process_group()
''')


def test_two_rank_shard_and_gather(workdir):
    common.build_case("pe101_spliced", workdir)       # build the index once, before the ranks start
    script = os.path.join(workdir, "gloo_worker.py")
    open(script, "w").write(WORKER.format(root=common.ROOT, workdir=workdir))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and "GATHER_OK" in out, out[-3000:]


def test_shard_bounds_cover_everything():
    from dart_amd import dist as ddist
    for n in (0, 1, 7, 1000, 1001):
        for w in (1, 2, 3, 8):
            cover = []
            for r in range(w):
                lo, hi = ddist.shard_bounds(n, w, r)
                cover += list(range(lo, hi))
            assert cover == list(range(n))


def test_bench_refuses_a_rank_count_that_is_not_gpus(monkeypatch):
    """`bench.py --gpus N` must run N ranks: under a launcher with another world size it stops (before touching any GPU)"""
    import subprocess, sys
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py"), "--gpus", "3"], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "--gpus 3 but WORLD_SIZE is 2" in r.stderr


def test_bench_self_launch_command(monkeypatch):
    """without a launcher, `bench.py --gpus N` starts N ranks through torch.distributed.run on 127.0.0.1 and returns their code"""
    import bench, subprocess, argparse
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 0)
    rc = bench.self_launch(argparse.Namespace(gpus=4), ["--gpus", "4", "--steps", "3"])
    cmd = seen["cmd"]
    assert rc == 0 and cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")

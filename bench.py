#!/usr/bin/env python3
"""bench.py -- DART's per-read mapping hot path on N MI355X (one process per GPU), host to host.

Workload at N=1: BASELINE.json configs[2], "Full GRCh38 index, 10 M paired-end 2x101 bp, 1 MI355X": a GRCh38-SIZED synthetic
genome (real GRCh38 is unobtainable offline) and, per step, TEN DISTINCT batches of 1 M pairs (10 M pairs).
A step is timed from the first batch handed to the library in page-locked HOST buffers to the last result record back in page-locked
HOST buffers ("first batch submitted -> last result record on host"): H2D of the reads, every kernel of the path,
D2H of the four record arrays, several batches in flight on separate contexts so copies overlap kernels.  `value` is that rate.
`config.input` says which entry point carried the reads (packed 2 bit/base + N list, or ASCII); the other one and the
device-resident rate (reads already in HBM, records left in HBM: the kernels alone) are reported beside it, measured after
the timed region.

`python bench.py --gpus N` starts N ranks itself (torch.distributed.run, one process per GPU, RCCL) when it is not already
running under a launcher.  For N>1 every rank maps its own 10 M distinct pairs per step (weak scaling: reads shard, the index is
replicated, no data-path collective in the mapping itself) and the per-read records of every batch are gathered to rank 0
over RCCL inside the timed region (the SAM-order gather of north_star; --no-gather switches it off).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` (dominant kernel, live HIP-event time vs
algorithmic bytes) and `cpu_baseline` (the oracle timed on a bounded sample of the same reads on the host cores).
"""
import argparse
import json
import os
import subprocess
import sys
import time

# the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); a mapping context uses two streams, so the
# batches in flight shared queues and waited for each other (profiles/probes/hwq_sweep.sh).  Read at HIP initialisation: must be
# set before torch / libdartgpu touch the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
T_PROCESS_START = time.time()
sys.path.insert(0, ROOT)

CHR20_LEN = 64444167


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_sources_sha256() -> str:
    """fingerprint of the device code (dart_amd/csrc/*.h, *.hip): ties the committed PMC passes (profiles/traffic.json) to the kernels that
    are really running -- the GPU box has no .git, so a commit id is not available there"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dart_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def host_cores() -> int:
    """cores this process may really use: the cgroup CPU quota when there is one (the GPU boxes expose 256 hardware threads but
    grant a 16-core share per GPU), else the affinity mask"""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0]); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // p))
        except Exception:
            pass
    if os.environ.get("DART_CPU_THREADS"):
        n = int(os.environ["DART_CPU_THREADS"])
    return n


# GRCh38 primary-assembly chromosome lengths (chr1..22, X, Y): the metric is quoted "vs GRCh38"; the real sequence cannot be
# fetched (no network), so the genome is synthetic with these sizes
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
          135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983,
          50818468, 156040895, 57227415]
GRCH38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]


def genome_spec(arg):
    """'grch38' | 'chr20' | <bp> -> (label, names, lengths)"""
    a = str(arg).lower()
    if a == "grch38":
        return "GRCh38-sized synthetic genome (24 chromosomes, %d bp" % sum(GRCH38), GRCH38_NAMES, GRCH38
    n = CHR20_LEN if a == "chr20" else int(float(a))
    return "%s synthetic genome (%d bp" % ("chr20-sized" if n == CHR20_LEN else "single-chromosome", n), ["chr20"], [n]


PREP_S = {}                                       # rank 0: seconds spent making the synthetic genome and building its index in this run


def prepare_index(cache_dir, genome, rank, barrier, n_introns=0, repeat_scale=1.0, model="planted"):
    """genome: a length in bp (one chromosome) or (names, lengths).  Rank 0 generates and indexes it once per box."""
    import numpy as np
    import torch
    from dart_amd import synth, index_build
    names, lengths = (["chr20"], [int(genome)]) if not isinstance(genome, (tuple, list)) else genome
    total = sum(lengths)
    prefix = os.path.join(cache_dir, "g%d" % total + ("c%d" % len(lengths) if len(lengths) > 1 else "") + ("_i%d" % n_introns if n_introns else "") +
                          ("_r%g" % repeat_scale if repeat_scale != 1.0 else "") + ("_%s" % model if model != "planted" else ""))
    done = prefix + ".done"
    if rank == 0 and not os.path.exists(done):
        os.makedirs(cache_dir, exist_ok=True)
        t = time.time()
        g = synth.make_genome(lengths, seed=20, repeat_scale=repeat_scale, n_introns=n_introns, names=names, model=model)
        np.save(prefix + ".codes.npy", g.codes)
        np.save(prefix + ".introns.npy", g.introns)
        PREP_S["genome_generated_s"] = round(time.time() - t, 1)
        log("[bench] genome %d bp generated in %.1f s" % (total, time.time() - t))
        t = time.time()
        index_build.build_index_from_genome(g, prefix, log=log)
        torch.cuda.empty_cache()
        PREP_S["index_build_s"] = round(time.time() - t, 1)      # di_build_files (libdartindex.so): the five files of this genome, written
        log("[bench] index built in %.1f s" % (time.time() - t))
        open(done, "w").write("ok")
    barrier()
    codes = np.load(prefix + ".codes.npy")
    ipath = prefix + ".introns.npy"
    g = synth.Genome(list(names), list(lengths), codes, np.load(ipath) if os.path.exists(ipath) else np.zeros((0, 3), np.int64))
    return prefix, g


def self_launch(args, argv):
    """--gpus N without a launcher around us: start N ranks (one per GPU) and hand back their exit code.  Runs before anything
    in this process has touched the GPU (a process that has must never exec or be replaced)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    log("[bench] --gpus %d without a launcher: starting %s" % (args.gpus, " ".join(cmd)))
    return subprocess.call(cmd)


def with_human_like_leg(args, argv):
    """The default run on one GPU: this process stays off the GPU and runs two children one after the other, each alone on it -- the measurement
    itself (`--as-child`: everything this file did before round 4) and, if the time budget allows, the same timed region on the human-like genome
    (`--genome-model human`: about half of the genome in repeat classes, what a real human index looks like to the seeding stage).  A child that
    ran inside the first one's process found that process's sixteen hardware queues still mapped and lost 8 %.  Prints the first child's line
    with `value_human_like` / `human_like` added."""
    me = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, me] + argv + ["--as-child"], stdout=subprocess.PIPE)
    out = r.stdout.decode().strip().splitlines()
    if r.returncode != 0 or not out:
        sys.stdout.write(r.stdout.decode())
        return r.returncode or 1
    line = json.loads(out[-1])
    used = time.time() - T_PROCESS_START
    if used > args.human_like_budget:
        line["human_like"] = {"skipped": "%.0f s used by the legs before it, budget %.0f s (--human-like-budget)" % (used, args.human_like_budget)}
    else:
        try:
            t = time.time()
            h = subprocess.run([sys.executable, me, "--as-child", "--genome-model", "human", "--no-secondary", "--no-cpu-baseline", "--sustained-s", "12", "--steps", str(args.steps), "--warmup", str(args.warmup),
                                "--pairs", str(args.pairs), "--batches", str(args.batches), "--inflight", str(args.inflight), "--mis", str(args.mis), "--cache", args.cache],
                               stdout=subprocess.PIPE, timeout=400)
            hl = json.loads(h.stdout.decode().strip().splitlines()[-1])
            line["value_human_like"] = hl["value"]
            line["human_like"] = {
                "value": hl["value"], "unit": hl["unit"], "ms_per_step": hl["ms_per_step"], "steps": hl["steps"], "warmup": hl["warmup"], "workload": hl["config"]["workload"],
                "kernels_ms": hl["kernels_ms"], "kernels_ms_one_batch_in_flight": hl["kernels_ms_one_batch_in_flight"], "sustained": hl.get("sustained"),
                "roofline": {k: hl["roofline"].get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms_standalone", "frac_of_random_line_ceiling", "seeding_mixed_ceiling", "fingerprint", "stages")},
                "counters_per_launch": {k: hl["counters_per_launch"].get(k) for k in ("steps", "steps_executed", "occ_blocks_executed", "seeds", "candidates", "nw_calls", "reseed_calls", "general_path_units", "wave_chained_units")},
                "what": "`bench.py --genome-model human --no-secondary --no-cpu-baseline` with this run's steps / warm-up / batches, a second child process after the first one has left the GPU "
                        "(its own genome, index and batches: %.0f s)" % (time.time() - t)}
        except Exception as e:
            line["human_like"] = {"skipped": "the child run failed: " + repr(e)}
            log("[bench] human-like leg failed:", repr(e))
    print(json.dumps(line), flush=True)
    return 0


def run_cli(dart_exe, prefix, g, label, seed_pairs, args, cpu_pairs=0, gz_pairs=0):
    """`dart` as a child process: FASTQ files on tmpfs -> SAM + junctions on tmpfs, process start to exit (HIP start-up, index files -> HBM, FASTQ
    parsing, mapping, SAM formatting, writing), best of two.  seed_pairs: [(seed, pairs)] -- the reads are synth.make_reads(g, pairs, seed) one
    after the other.  cpu_pairs: the CPU command line (the oracle's, all host cores) on the head of the same files, end to end, and its wall
    extrapolated to the whole job (index load + reads / its mapping rate).  gz_pairs: the head of the files gzipped, through `dart` again."""
    import shutil, tempfile, re
    import numpy as np
    from dart_amd import synth
    t = time.time()
    total_pairs = sum(p_ for _, p_ in seed_pairs)
    need = total_pairs * 2 * 560
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 1.3 * need else args.cache
    d = tempfile.mkdtemp(prefix="dart_cli_", dir=base)
    try:
        from concurrent.futures import ThreadPoolExecutor
        firsts = [sum(p_ for _, p_ in seed_pairs[:j]) for j in range(len(seed_pairs))]
        for nm in ("1.fq", "2.fq"):
            open(os.path.join(d, nm), "wb").close()
        heads = {}

        def one_chunk(j):                                   # (records have a fixed length: every chunk goes to its own place of the two files)
            seed, pairs = seed_pairs[j]
            m1, m2 = synth.make_reads(g, pairs, rlen=101, seed=seed, sub_rate=args.sub_rate, indel_frac=args.indel_frac, n_frac=0.002)
            synth.write_fastq_fast(os.path.join(d, "1.fq"), m1, 1, first_id=firsts[j], in_place=True); synth.write_fastq_fast(os.path.join(d, "2.fq"), m2, 2, first_id=firsts[j], in_place=True)
            if j == 0:
                heads[0] = (m1[:max(cpu_pairs, gz_pairs)].copy(), m2[:max(cpu_pairs, gz_pairs)].copy())
        with ThreadPoolExecutor(max_workers=min(4, len(seed_pairs))) as ex:
            list(ex.map(one_chunk, range(len(seed_pairs))))
        head = heads.get(0)
        log("[bench] command-line run: %d pairs of FASTQ under %s prepared in %.1f s" % (total_pairs, d, time.time() - t))
        cores = host_cores()
        cmd = lambda f1, f2, out: [dart_exe, "-i", prefix, "-f", f1, "-f2", f2, "-o", out, "-j", out + ".j", "-t", str(cores), "-mis", str(args.mis)]

        def timed_dart(f1, f2, out, reps, extra_env=None):
            best = None
            for _ in range(reps):
                for f_ in (out, out + ".j"):                            # (truncating a multi-GB tmpfs file at open is not part of the job)
                    if os.path.exists(os.path.join(d, f_)):
                        os.remove(os.path.join(d, f_))
                t0 = time.perf_counter()
                r = subprocess.run(cmd(f1, f2, out), cwd=d, env=dict(os.environ, DART_TIMING="1", DART_INFLIGHT="2", **(extra_env or {})), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                dt = time.perf_counter() - t0
                if r.returncode != 0:
                    raise RuntimeError("dart exited with %d: %s" % (r.returncode, r.stderr.decode()[-300:]))
                tl = [l for l in r.stderr.decode().splitlines() if l.startswith("[dart timing]")]
                if best is None or dt < best[0]:
                    best = (dt, tl[-1] if tl else None, os.path.getsize(os.path.join(d, out)))
            return best
        best = timed_dart("1.fq", "2.fq", "out.sam", 2)
        res = {"value": round(2 * total_pairs / best[0] / 1e6, 3), "unit": "M reads/s", "wall_s": round(best[0], 3), "sam_bytes": best[2],
               "what": "`dart -i <%s> -f 1.fq -f2 2.fq -o out.sam -j out.j -t %d -mis %d` as a child process, %d pairs 2x101, files on %s: process start, HIP start-up, index files -> HBM, "
                       "FASTQ parsing, mapping, SAM formatting and writing; best of 2" % (label.split(" (")[0], cores, args.mis, total_pairs, base),
               "stages": best[1]}
        m_ = re.search(r"start-up ([0-9.]+) s", best[1] or "")
        if m_:
            res["startup_s"] = float(m_.group(1))
        if getattr(args, "cli_full_parity", False):
            # The SAM text and the junctions of the WHOLE job against the CPU command line (the oracle's, all host cores) on the same files (the default leg
            # compares the first --cli-cpu-pairs only): ~100 s of CPU for 10 M pairs, so it is opt-in (--cli-full-parity); the two SAM files are compared by digest.
            import hashlib
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_py
            oracle_py.build()

            def digest(path):
                h = hashlib.sha256()
                with open(path, "rb") as f:
                    for blk in iter(lambda: f.read(1 << 24), b""):
                        h.update(blk)
                return h.hexdigest(), os.path.getsize(path)
            t0 = time.perf_counter()
            r = subprocess.run([oracle_py.ORACLE_CLI, "-i", prefix, "-f", "1.fq", "-f2", "2.fq", "-o", "cpu_full.sam", "-j", "cpu_full.j", "-t", str(cores), "-mis", str(args.mis)], cwd=d,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            dt_full = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError("the CPU command line (whole job) exited with %d: %s" % (r.returncode, r.stderr.decode()[-300:]))
            a_, b_ = digest(os.path.join(d, "out.sam")), digest(os.path.join(d, "cpu_full.sam"))
            ja, jb = digest(os.path.join(d, "out.sam.j")), digest(os.path.join(d, "cpu_full.j"))
            res["whole_job_parity"] = {"pairs": total_pairs, "sam_identical": a_ == b_, "junctions_identical": ja == jb, "sam_bytes": a_[1], "sam_sha256": a_[0][:16],
                                       "cpu_command_line_wall_s": round(dt_full, 1), "speedup_whole_job_measured": round(dt_full / best[0], 1),
                                       "what": "oracle/dart_oracle -t %d on the same 1.fq / 2.fq, process start to exit; SHA-256 of the two SAM files and of the two junction files" % cores}
            log("[bench] whole-job parity of the command line:", res["whole_job_parity"])
            os.remove(os.path.join(d, "cpu_full.sam"))
        if cpu_pairs > 0 and head is not None:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_py
            oracle_py.build()
            n_c = min(cpu_pairs, len(head[0]))
            synth.write_fastq_fast(os.path.join(d, "c1.fq"), head[0][:n_c], 1); synth.write_fastq_fast(os.path.join(d, "c2.fq"), head[1][:n_c], 2)
            t0 = time.perf_counter()
            r = subprocess.run([oracle_py.ORACLE_CLI, "-i", prefix, "-f", "c1.fq", "-f2", "c2.fq", "-o", "cpu.sam", "-j", "cpu.j", "-t", str(cores), "-mis", str(args.mis)], cwd=d,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError("the CPU command line exited with %d: %s" % (r.returncode, r.stderr.decode()[-300:]))
            m_ = re.search(r"index load ([0-9.]+) s, mapping phase ([0-9.]+) s", r.stderr.decode())
            load_s, map_s = (float(m_.group(1)), float(m_.group(2))) if m_ else (0.0, dt)
            other_s = max(0.0, dt - load_s - map_s)                      # parsing, formatting, writing: grows with the job like the mapping phase
            whole = load_s + (map_s + other_s) * total_pairs / n_c
            # the same reads through `dart`: the SAM must be the CPU command line's, byte for byte
            timed_dart("c1.fq", "c2.fq", "gpu_head.sam", 1)
            same = open(os.path.join(d, "cpu.sam"), "rb").read() == open(os.path.join(d, "gpu_head.sam"), "rb").read() and \
                open(os.path.join(d, "cpu.j"), "rb").read() == open(os.path.join(d, "gpu_head.sam.j"), "rb").read()
            res["cpu_command_line"] = {"kind": "port", "cores": cores, "pairs": n_c, "wall_s": round(dt, 3), "index_load_s": round(load_s, 3), "mapping_phase_s": round(map_s, 3),
                                       "value": round(2 * n_c / dt / 1e6, 4), "unit": "M reads/s",
                                       "wall_s_extrapolated_to_the_whole_job": round(whole, 1),
                                       "what": "oracle/dart_oracle (the CPU restatement's command line) on the first %d pairs of the same files, %d threads, process start to exit; "
                                               "extrapolation to %d pairs = index load + (the rest) x %d / %d" % (n_c, cores, total_pairs, total_pairs, n_c),
                                       "sam_and_junctions_identical_to_dart_on_these_pairs": bool(same)}
            res["speedup_vs_cpu_command_line_whole_job"] = round(whole / best[0], 1)
            # The reference's OWN object code on this box's host CPU (oracle/_ref/ref_harness: the reference's translation units compiled where they lie, linked under
            # our driver of its ReadMapping loop; the built binary travels with the repo, the sources do not): one thread -- the harness has no -t --, the first
            # ref_pairs pairs of the same files, its mapping loop timed by itself (FASTQ parsing and SAM text included, as in the reference's ReadMapping; index load apart).
            ref_exe = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
            n_r = min(n_c, int(os.environ.get("DART_BENCH_REF_PAIRS", "100000")))
            if os.path.exists(ref_exe) and n_r > 0:
                try:
                    synth.write_fastq_fast(os.path.join(d, "r1.fq"), head[0][:n_r], 1); synth.write_fastq_fast(os.path.join(d, "r2.fq"), head[1][:n_r], 2)
                    t0 = time.perf_counter()
                    rr = subprocess.run([ref_exe, "map", "-i", prefix, "-f", "r1.fq", "-f2", "r2.fq", "-o", "ref.sam", "-j", "ref.j", "-mis", str(args.mis)], cwd=d,
                                        stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
                    dtr = time.perf_counter() - t0
                    mr = re.search(r"index load ([0-9.]+) s, mapping phase ([0-9.]+) s, (\d+) reads", rr.stderr.decode())
                    if rr.returncode == 0 and mr:
                        # its SAM against the port's on the same pairs (the head of cpu.sam: header + the records of the first n_r pairs are a prefix, reads are independent)
                        ref_sam = open(os.path.join(d, "ref.sam"), "rb").read()
                        same_ref = open(os.path.join(d, "cpu.sam"), "rb").read(len(ref_sam)) == ref_sam
                        r_map = float(mr.group(2))
                        res["cpu_reference_object_code"] = {"kind": "reference", "cores": 1, "pairs": n_r, "wall_s": round(dtr, 3), "index_load_s": float(mr.group(1)), "mapping_phase_s": r_map,
                                                            "value": round(2 * n_r / r_map / 1e6, 5), "unit": "M reads/s (mapping phase, one thread)",
                                                            "port_mapping_phase_per_thread": round(2 * n_c / map_s / cores / 1e6, 5),
                                                            "sam_identical_to_the_port_on_these_pairs": bool(same_ref),
                                                            "what": "oracle/_ref/ref_harness = the reference's own translation units (bwt_search, AlignmentCandidates, nw_alignment, tools, KmerAnalysis, Mapping, "
                                                                    "GetData, bwt_index, BWT_Index/*) compiled in the builder's container and linked under a driver of ReadMapping's loop; first %d pairs of "
                                                                    "the command-line run's files, -mis %d, one thread; value = reads / its mapping loop's wall (parsing and SAM text included)" % (n_r, args.mis)}
                    else:
                        log("[bench] reference harness: rc %d, %s" % (rr.returncode, rr.stderr.decode()[-300:]))
                except Exception as e:
                    log("[bench] reference harness leg failed:", repr(e))
        if gz_pairs > 0 and head is not None:
            import gzip
            n_z = min(gz_pairs, total_pairs)
            rec_len = os.path.getsize(os.path.join(d, "1.fq")) // total_pairs            # (write_fastq_fast: every record has the same length)

            def gz_head(k_):                                   # the first n_z records of mate file k_, plain and gzipped (level 1, as sequencers' pipelines write them)
                src, nm = ("1.fq", "z1.fq") if k_ == 0 else ("2.fq", "z2.fq")
                with open(os.path.join(d, src), "rb") as fi, open(os.path.join(d, nm), "wb") as fp, gzip.open(os.path.join(d, nm + ".gz"), "wb", compresslevel=1) as fo:
                    left = n_z * rec_len
                    while left > 0:
                        blk = fi.read(min(left, 1 << 24)); left -= len(blk)
                        fp.write(blk); fo.write(blk)
            with ThreadPoolExecutor(max_workers=2) as ex:
                list(ex.map(gz_head, (0, 1)))
            bz = min((timed_dart("z1.fq.gz", "z2.fq.gz", "gz.sam", 1) for _ in range(2)), key=lambda b_: b_[0])
            bs = timed_dart("z1.fq.gz", "z2.fq.gz", "gz_stream.sam", 1, extra_env={"DART_GZ_STREAM": "1"})
            bp = timed_dart("z1.fq", "z2.fq", "plain.sam", 1)
            same = open(os.path.join(d, "gz.sam"), "rb").read() == open(os.path.join(d, "plain.sam"), "rb").read() == open(os.path.join(d, "gz_stream.sam"), "rb").read()
            res["gz_input"] = {"pairs": n_z, "wall_s": round(bz[0], 3), "value": round(2 * n_z / bz[0] / 1e6, 3), "unit": "M reads/s", "stages": bz[1],
                               "wall_s_streaming_reader": round(bs[0], 3), "value_streaming_reader": round(2 * n_z / bs[0] / 1e6, 3),
                               "wall_s_same_reads_plain_fastq": round(bp[0], 3),
                               "sam_identical_to_plain_fastq_input": bool(same),
                               "what": "the first %d pairs as 1.fq.gz / 2.fq.gz through the same command line (GetData.cpp:181-247), best of 2: the files are inflated whole "
                                       "(the system's libdeflate, fast_fastq.h) and go through the parallel pipeline; `streaming_reader` = DART_GZ_STREAM=1, zlib's gzread one thread per "
                                       "mate file (what a file too large to inflate whole, or one the two readers of the reference would see differently, takes)" % n_z}
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


class Batch:
    """one distinct batch of pairs: ASCII and packed forms in page-locked host memory"""

    def __init__(self, lib, g, pairs, rlen, seed, sub_rate, indel_frac, spliced, want_truth=False, rows=None):
        """rows = (a, b): only pairs [a, b) of the batch that `seed` generates (a rank's share of a global batch in --total-pairs mode;
        the batch is a pure function of (seed, pairs), so the union over ranks is the same reads whatever the number of ranks)"""
        import numpy as np
        from dart_amd import synth, host
        r = synth.make_reads(g, pairs, rlen=rlen, seed=seed, sub_rate=sub_rate, indel_frac=indel_frac, n_frac=0.002, spliced_frac=spliced, return_truth=want_truth)
        m1, m2 = r[0], r[1]
        self.truth = r[2] if want_truth else None
        self.seed, self.rows = seed, rows
        if rows is not None:
            m1, m2 = m1[rows[0]:rows[1]], m2[rows[0]:rows[1]]
            if self.truth is not None:
                self.truth = {k: v[rows[0]:rows[1]] for k, v in self.truth.items()}
        arr = host.interleave_pairs(m1, m2)
        so, rl, flat = host.pack_reads(arr)
        words, nlist = host.pack_reads_2bit(arr)
        self.n = len(rl); self.rlen = rlen; self.W2 = int(words.shape[1]); self.n_n = len(nlist)
        self.so = host.PinnedArray(lib, (self.n,), np.uint32); self.so.a[:] = so
        self.rl = host.PinnedArray(lib, (self.n,), np.uint16); self.rl.a[:] = rl
        self.seq = host.PinnedArray(lib, (len(flat) + 64,), np.uint8); self.seq.a[:len(flat)] = flat
        self.words = host.PinnedArray(lib, words.shape, np.uint32); self.words.a[:] = words
        self.nlist = host.PinnedArray(lib, (max(self.n_n, 1),), np.uint32); self.nlist.a[:self.n_n] = nlist
        self.bases = int(rl.sum())
        self.bytes_ascii = len(flat) + 6 * self.n
        self.bytes_packed = words.nbytes + 4 * self.n_n


class Worker:
    """one context = one batch in flight: its own device buffers, its own page-locked output arrays, driven by its own thread"""

    def __init__(self, gpu, n, mode, records):
        import ctypes as C
        import numpy as np
        from dart_amd import host
        self.gpu, self.mode, self.records, self.C = gpu, mode, records, C
        self.n_cap = n
        self.used = (C.c_size_t * 3)()
        self.alloc(int(n * 1.3) + 1024, 4 * n + 4096, n + 1024)
        self.kern = {}; self.n_runs = 0; self.last_records = records

    def alloc(self, reports, ops, tuples):
        """the page-locked output arrays (the compact types use the front of the full ones'); grown when a batch needs more (a repeat-rich genome
        has more reports per read)"""
        import numpy as np
        from dart_amd import host
        gpu, n, C = self.gpu, self.n_cap, self.C
        self.caps = (C.c_size_t * 3)(reports, ops, tuples)
        self.o_r = gpu.pinned((n,), host.READ_OUT); self.o_p = gpu.pinned((self.caps[0],), host.REPORT_OUT)
        self.o_c = gpu.pinned((self.caps[1],), np.uint32); self.o_s = gpu.pinned((self.caps[2],), host.SJ_OUT)
        self.c_r = self.o_r.a.view(np.uint8)[:n * host.READ_C.itemsize].view(host.READ_C); self.c_p = self.o_p.a.view(np.uint8)[:self.caps[0] * host.REPORT_C.itemsize].view(host.REPORT_C)

    def map(self, b, mode=None, records=None):
        g, lib = self.gpu, self.gpu.lib
        mode = mode or self.mode
        records = records or self.records
        if mode != "resident":
            self.last_records = records
        self.last_empty = b.n == 0
        self.last_n = b.n
        if b.n == 0:                                   # (a rank whose share of the step has fewer batches than another's: it still takes part in the gather)
            for k in range(3):
                self.used[k] = 0
            return
        if records == "compact" and mode in ("packed", "ascii"):
            pk = mode == "packed"
            rc = lib.dg_map_batch_compact(g.ctx, b.n, None if pk else b.so.a.ctypes.data, None if pk else b.rl.a.ctypes.data, None if pk else b.seq.a.ctypes.data,
                                          b.rlen, b.W2, b.words.a.ctypes.data if pk else None, (b.nlist.a.ctypes.data if b.n_n else None) if pk else None, b.n_n if pk else 0,
                                          self.c_r.ctypes.data, self.c_p.ctypes.data, self.o_c.a.ctypes.data, self.o_s.a.ctypes.data, self.caps, self.used)
        elif mode == "packed":
            rc = lib.dg_map_batch_packed(g.ctx, b.n, b.rlen, None, b.W2, b.words.a.ctypes.data, b.nlist.a.ctypes.data if b.n_n else None, b.n_n,
                                         self.o_r.a.ctypes.data, self.o_p.a.ctypes.data, self.o_c.a.ctypes.data, self.o_s.a.ctypes.data, self.caps, self.used)
        elif mode == "ascii":
            rc = lib.dg_map_batch(g.ctx, b.n, b.so.a.ctypes.data, b.rl.a.ctypes.data, b.seq.a.ctypes.data,
                                  self.o_r.a.ctypes.data, self.o_p.a.ctypes.data, self.o_c.a.ctypes.data, self.o_s.a.ctypes.data, self.caps, self.used)
        else:                                          # "resident": the batch uploaded last, records left in HBM
            rc = lib.dg_batch_run(g.ctx, self.used)
        if rc == -4 and not getattr(self, "_grown", False):      # DG_ERR_CAPACITY: `used` holds the need; grow the arrays once and map the batch again
            u = [int(x) for x in self.used]
            self.alloc(max(int(self.caps[0]), int(u[0] * 1.25) + 1024), max(int(self.caps[1]), int(u[1] * 1.5) + 4096, 2 * int(self.caps[1])), max(int(self.caps[2]), int(u[2] * 1.5) + 1024))
            self._grown = True
            try:
                return self.map(b, mode, records)
            finally:
                self._grown = False
        g._chk(rc, "dg_map_batch(%s)" % mode)
        g._n = b.n; g._used = [int(x) for x in self.used]
        self.last_rlen = b.rlen
        for name, ms in g.timings():
            self.kern[name] = self.kern.get(name, 0.0) + ms
        self.n_runs += 1

    def result(self):
        import numpy as np
        from dart_amd import host
        u = [int(x) for x in self.used]
        if self.last_records == "compact":
            r, p, cg = host.expand_compact(self.c_r[:self.last_n].copy(), self.c_p[:u[0]].copy(), self.o_c.a[:u[1]].copy(), np.full(self.last_n, self.last_rlen, np.uint16))
            return host.BatchResult(r, p, cg, self.o_s.a[:u[2]].copy())
        return host.BatchResult(self.o_r.a[:self.last_n].copy(), self.o_p.a[:u[0]].copy(), self.o_c.a[:u[1]].copy(), self.o_s.a[:u[2]].copy())

    def out_bytes(self, n):
        u = [int(x) for x in self.used]
        return (12 * n + 16 * u[0] if self.last_records == "compact" else 36 * n + 40 * u[0]) + 4 * u[1] + 24 * u[2]


RATE_LOG = [] if os.environ.get("DART_BENCH_RATE_LOG") else None      # (items done, perf_counter) marks of the timed region: the rate over time of a long run
RATE_EVERY = int(os.environ.get("DART_BENCH_RATE_EVERY", "1000"))
T_PHASE = []          # (name, wall clock) marks of the run's phases: rank 0 prints what each took, and the line carries it (the driver gives a run 600 s)


def phase(name):
    T_PHASE.append((name, time.time()))


def main():
    global RATE_LOG, RATE_EVERY
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30, help="timed steps (one step = --batches distinct batches); 30 steps = about one second: filling and draining the\n                    pipeline of batches in flight costs a fixed ~20 ms, which a 6-step run (0.2 s) shows as -8 %%")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=int(os.environ.get("DART_BENCH_PAIRS", "1000000")), help="pairs per batch")
    ap.add_argument("--batches", type=int, default=int(os.environ.get("DART_BENCH_BATCHES", "10")), help="distinct batches per step and GPU (10 x 1 M pairs = BASELINE configs[2])")
    ap.add_argument("--genome", default=os.environ.get("DART_BENCH_GENOME", "grch38"),
                    help="grch38 (default: 24 chromosomes with GRCh38 sizes, 3.09 Gbp) | chr20 | <bp> (one chromosome)")
    ap.add_argument("--genome-model", choices=["planted", "human"], default="planted",
                    help="planted (default, SURVEY 8d): i.i.d. + planted repeat families, ~18 %% of the genome; human: about half of the genome in human-like repeat "
                         "classes (SINE/LINE/older interspersed families, segmental duplications, satellites, microsatellites: dart_amd/synth.py)")
    ap.add_argument("--repeat-scale", type=float, default=1.0, help="scales the planted repeat families of the synthetic genome (1.0 = SURVEY 8d's: ~18 %% of the genome)")
    ap.add_argument("--mis", type=int, default=5, help="-mis N (MaxMismatch); the reference default is 0, see DESIGN.md")
    ap.add_argument("--input", choices=["packed", "ascii"], default="packed", help="entry point that carries the reads in the timed region")
    ap.add_argument("--records", choices=["compact", "full"], default="compact", help="record types that carry the results in the timed region (include/dartgpu.h: 16 + 20 bytes, or 36 + 40)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=150000)
    ap.add_argument("--rlen", type=int, default=101)
    ap.add_argument("--spliced", type=float, default=0.0, help="fraction of reads spanning a planted intron (BASELINE config 5 shape: --rlen 151 --spliced 0.3 --introns 20000)")
    ap.add_argument("--introns", type=int, default=0, help="introns planted in the synthetic genome")
    ap.add_argument("--max-intron", type=int, default=500000)
    ap.add_argument("--sub-rate", type=float, default=0.01, help="per-base substitution rate of the synthetic reads (experiments only; the bench line is quoted at the default)")
    ap.add_argument("--indel-frac", type=float, default=0.02, help="fraction of reads carrying one short indel (experiments only)")
    ap.add_argument("--gather", choices=["full", "reads", "none"], default="full",
                    help="N>1 only, inside the timed region: full = every rank's compact records of every batch -- per-read records, reports, stored CIGAR ops, "
                         "junction tuples -- go HBM -> HBM to rank 0 over RCCL, sizes first (the SAM-order gather of north_star: the reference has ONE ordered writer); "
                         "reads = the per-read records only (round 2); none")
    ap.add_argument("--no-gather", action="store_true", help="= --gather none")
    ap.add_argument("--total-pairs", type=int, default=0,
                    help="STRONG scaling (BASELINE configs[3]: 50 M pairs sharded over 8 GPUs): a step = this many pairs in all, sharded over the ranks in contiguous, "
                         "balanced pair ranges (dart_amd/dist.py); the reads are the same whatever the number of ranks.  Default 0 = weak scaling: --batches per rank")
    ap.add_argument("--weak", action="store_true", help="N>1: weak scaling (every rank maps its own --batches per step) as `value`; the default for N>1 is BASELINE configs[3], "
                                                       "one 50 M-pair job per step sharded over the ranks (strong scaling), with the weak rate beside it as `value_weak_scaling`")
    ap.add_argument("--verify-gather", action="store_true",
                    help="N>1, after the timed region: rank 0 expands what it gathered from every rank for one step and compares it, batch by batch, with its own "
                         "single-rank mapping of the same reads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cli-big-pairs", type=int, default=10000000,
                    help="after the timed region: `dart` end to end on BASELINE configs[2] itself -- this many 2x101 pairs (the timed region's reads, as FASTQ files) against the "
                         "GRCh38-sized index, process start to exit; 0 = skip (skipped too when the workload is not the default one)")
    ap.add_argument("--cli-full-parity", action="store_true", help="the command-line leg also maps the WHOLE job with the CPU command line (~100 s on 16 cores for 10 M pairs) and compares "
                                                                   "the two SAM files and junction files by digest")
    ap.add_argument("--cli-cpu-pairs", type=int, default=400000, help="pairs of the same files the CPU command line (oracle/dart_oracle, all host cores) maps beside it, end to end")
    ap.add_argument("--cli-gz-pairs", type=int, default=2000000, help="pairs of the same files, gzipped, through `dart` (what users feed DART: GetData.cpp:181-247); 0 = skip")
    ap.add_argument("--cli-pairs", type=int, default=0,
                    help="after the timed region: the product's `dart` command line end to end (FASTQ files -> SAM file, process start, index load and dg_init included) on "
                         "this many 2x101 pairs against the chr20-sized genome; 0 = skip (the default since round 4: --cli-big-pairs measures the same on the headline index, "
                         "and the run's time goes to the human-like leg instead)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary rates (other entry point, device-resident)")
    ap.add_argument("--repeats", type=int, default=-1, help="more runs of the same K steps after the timed region (value_repeats); default: 2, with --no-secondary 0")
    ap.add_argument("--sustained-s", type=float, default=-1.0,
                    help="one GPU, secondary rates on: after the timed region, the same steps for about this many seconds (`value_sustained`, with the rate of each quarter "
                         "of the run: a 0.4 s timed region says nothing about the rate a long job settles at); 0 = skip; default: 20 s, 0 with --no-secondary")
    ap.add_argument("--human-like-budget", type=float, default=330.0,
                    help="the default workload on one GPU with the CPU legs on: after the measurement, the same timed region on the human-like genome (--genome-model human) as a "
                         "second child process, if the run has used fewer seconds than this so far (the leg costs ~110 s: genome, index, batches); 0 = never")
    ap.add_argument("--as-child", action="store_true", help="(set by this file) the measurement itself, in a process of its own: see with_human_like_leg")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("DART_BENCH_INFLIGHT", "12")),
                    help="batches in flight per GPU: contexts sharing one index, one host thread each (dg_clone)")
    ap.add_argument("--cache", default=os.environ.get("DART_BENCH_CACHE", "/tmp/dart_bench_cache"))
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    T_PHASE.append(("start", time.time()))
    if args.gpus > 1 and args.total_pairs == 0 and not args.weak:
        args.total_pairs = 50000000              # BASELINE configs[3]: "Full GRCh38, 50 M paired-end 2x101 bp sharded across 8xMI355X" -- the N>1 line IS that job
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    if (world == 1 and "RANK" not in os.environ and not args.as_child and args.human_like_budget > 0 and not args.no_cpu_baseline and not args.no_secondary and
            args.genome == "grch38" and args.genome_model == "planted" and not args.spliced and args.rlen == 101 and args.repeat_scale == 1):
        sys.exit(with_human_like_leg(args, sys.argv[1:]))
    if args.gpus != world:
        log("[bench] --gpus %d but WORLD_SIZE is %d: start it as `python bench.py --gpus N` or under a launcher with N ranks" % (args.gpus, world))
        sys.exit(2)

    import threading
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    import torch
    from dart_amd import synth, host

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    rehearse = world > 1 and os.environ.get("DART_BENCH_REHEARSE") == "1"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # DART_BENCH_REHEARSE=1: several ranks with gloo, on ONE GPU or on none of their own (a 1-GPU box cannot run RCCL across
        # ranks); exercises the launch / threading / ordering / gather logic of the N>1 path, not its performance
        if rehearse:
            local = 0
        torch.cuda.set_device(local)
        dist.init_process_group("gloo" if rehearse else "nccl", rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

    label, gnames, glens = genome_spec(args.genome)
    try:
        prefix, g = prepare_index(args.cache, (gnames, glens), rank, barrier, args.introns, args.repeat_scale, args.genome_model)
    except Exception as e:                         # e.g. not enough memory for the 6.2 G-symbol suffix sort: say so, use chr20
        if world > 1 or len(glens) == 1:
            raise
        log("[bench] %s could not be indexed here (%r): falling back to the chr20-sized genome" % (label, e))
        torch.cuda.empty_cache()
        label, gnames, glens = genome_spec("chr20")
        label = "FALLBACK (GRCh38-sized index build failed) " + label
        prefix, g = prepare_index(args.cache, (gnames, glens), rank, barrier, args.introns, args.repeat_scale, args.genome_model)
    phase("genome + index (rank 0 builds, the others wait)")
    # ---- the step's distinct batches, in page-locked host memory (before the library context is created: the index builder has just released
    #      > 100 GB of HBM, which the driver clears in the background; allocations that need those pages wait for it -- seconds of hipMalloc) ----
    lib = host._load_lib()
    t = time.time()
    strong = args.total_pairs > 0
    if not strong:
        specs = [(1000 + 100 * rank + j, None) for j in range(max(1, args.batches))]
    else:
        # this rank's contiguous share [lo, hi) of the step's pairs, cut at the boundaries of the global batches (global batch j = seed 1000 + j)
        from dart_amd import dist as ddist
        lo, hi = ddist.shard_bounds_balanced(args.total_pairs, world, rank)
        specs = []
        p = lo
        while p < hi:
            j = p // args.pairs
            e = min(hi, (j + 1) * args.pairs, args.total_pairs)
            specs.append((1000 + j, (p - j * args.pairs, e - j * args.pairs)))
            p = e
        n_local = len(specs)
        if dist is not None:                           # every rank takes part in the same number of gathers: pad with empty batches
            tt = torch.tensor([n_local], dtype=torch.int64, device="cpu" if rehearse else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            specs += [(0, (0, 0))] * (int(tt.item()) - n_local)
    nb = len(specs)
    with ThreadPoolExecutor(max_workers=min(4, nb)) as ex:
        batches = list(ex.map(lambda js: Batch(lib, g, args.pairs if js[1] is None else min(args.pairs, args.total_pairs - (js[0] - 1000) * args.pairs) if js[1][1] else 1,
                                               args.rlen, js[0], args.sub_rate, args.indel_frac, args.spliced, want_truth=(js is specs[0]), rows=js[1]), specs))
    n_reads = max(b.n for b in batches)
    reads_per_step = 2 * args.total_pairs if strong else 2 * args.pairs * nb * world          # whole job, all ranks
    if rank == 0:
        log("[bench] %d distinct batches of %d pairs generated in %.1f s" % (nb, args.pairs, time.time() - t))

    phase("this rank's batches")
    ix = host.Index(prefix)
    params = host.default_params(paired=1, max_mismatch=args.mis, max_intron=args.max_intron)
    t = time.time()
    gpu = host.DartGPU(ix, params, device=local)           # dg_init_files: the index files straight to HBM, full look-up aids (the job size is not given)
    init_s = time.time() - t
    torch.cuda.synchronize()
    init_report = gpu.init_report()
    if rank == 0:
        log("[bench] dg_init_files %.2f s: %s" % (time.time() - t, gpu.init_report()))

    # `inflight` contexts share the index; each is driven by its own host thread, the way the reference runs ReadMapping in -t
    # threads.  Work item i = batch i % nb on context i % inflight; a step = nb items.
    n_ctx = max(1, min(args.inflight, nb * max(1, args.steps)))
    workers = [Worker(gpu if k == 0 else gpu.clone(), n_reads, args.input, args.records) for k in range(n_ctx)]
    for k, w in enumerate(workers):                # sizes every context's device buffers before any counted step (distinct batches)
        w.map(batches[k % nb])
        w.kern = {}; w.n_runs = 0
    gather_mode = "none" if (dist is None or args.no_gather) else args.gather
    if gather_mode == "full" and args.records != "compact":
        gather_mode = "reads"
    do_gather = gather_mode != "none"
    gather_buf = None
    gather_dev = "cpu" if rehearse else "cuda"
    recv_bufs = None                                # rank 0: [source rank][array] byte buffers in HBM for what the other ranks send
    if gather_mode == "full" and rank == 0:
        cap = [max(int(workers[0].caps[0]), 3 * n_reads), max(int(workers[0].caps[1]), 8 * n_reads), max(int(workers[0].caps[2]), n_reads)]      # (room for a repeat-rich genome: HBM is not scarce)
        recv_bufs = [None] + [[torch.empty(n_reads * 12, dtype=torch.uint8, device=gather_dev), torch.empty(int(cap[0]) * 16, dtype=torch.uint8, device=gather_dev),
                               torch.empty(int(cap[1]) * 4, dtype=torch.uint8, device=gather_dev), torch.empty(int(cap[2]) * 24, dtype=torch.uint8, device=gather_dev)] for _ in range(1, world)]
    gathered = None                                 # --verify-gather: rank 0 keeps what arrived, per (source rank, batch)
    gather_bytes = [0]
    # the writer's next step (Mapping.cpp:644-664: ONE ordered writer): rank 0 brings what it gathered from HBM to page-locked host memory, batch by
    # batch in input order -- switched on for a second timed pass after the contract's (`value_with_writer_download`): it is the host link of ONE GPU
    # that carries all ranks' records then
    writer_dl = [False]
    host_bufs = None
    if recv_bufs is not None:
        host_bufs = [None] + [[torch.empty(b_.numel(), dtype=torch.uint8, pin_memory=not rehearse) for b_ in recv_bufs[r]] for r in range(1, world)]

    def gather(w, item=None):
        nonlocal gather_buf
        if gather_mode == "full":
            from dart_amd import dist as ddist
            parts = [torch.empty(0, dtype=torch.uint8, device=gather_dev)] * 4 if w.last_empty else w.gpu.device_records_compact()
            if rehearse:
                parts = [p_.cpu() for p_ in parts]
            counts = ddist.gather_compact_to_rank0(parts, recv_bufs, world, rank)
            if not rehearse:
                torch.cuda.current_stream().synchronize()   # the context's next run overwrites these records
            if rank == 0:
                gather_bytes[0] += int(counts[1:].sum())
                if writer_dl[0]:
                    for r in range(1, world):
                        for k in range(4):
                            nb_ = int(counts[r, k])
                            if nb_:
                                host_bufs[r][k][:nb_].copy_(recv_bufs[r][k][:nb_], non_blocking=True)
                    if not rehearse:
                        torch.cuda.current_stream().synchronize()
                if gathered is not None and item is not None:
                    for r in range(world):
                        src = parts if r == 0 else [recv_bufs[r][k][:int(counts[r, k])] for k in range(4)]
                        gathered[(r, item % nb)] = [x.cpu().numpy().copy() for x in src]
            return
        local_t = w.gpu.device_reads_tensor(compact=(w.last_records == "compact"))      # 12 (or 36) bytes per read, straight from HBM
        if rehearse:
            local_t = local_t.cpu()
        if gather_buf is None:
            gather_buf = {}
        key = tuple(local_t.shape)
        if rank == 0 and key not in gather_buf:
            gather_buf[key] = [torch.empty_like(local_t) for _ in range(world)]
        dist.gather(local_t, gather_buf[key] if rank == 0 else None, dst=0)
        if not rehearse:
            torch.cuda.current_stream().synchronize()       # the context's next run overwrites these records

    stagger_s = float(os.environ.get("DART_BENCH_STAGGER_MS", "1.0")) * 1e-3

    def run_items(n_items, mode, ws=workers, first_item=0, bl=None, static=False):
        """n_items work items over the contexts ws, each driven by its own host thread.  A context takes the NEXT item when it is free (the reference's threads
        pull the next chunk under a lock, Mapping.cpp:598-603: GetNextChunk) -- rounds 1-4 dealt item i to context i mod n, and a long run then lasted as long as
        its slowest context (the contexts of a process do not progress at one rate: profiles/r05/f_sustained_*); static=True keeps the deal (every context exactly
        its share: the passes that must leave a batch in every context).  The main thread takes the items in order: gathers (N > 1) happen in item order."""
        bl_ = batches if bl is None else bl
        owner = [-1] * n_items                              # which context mapped item i
        cv = threading.Condition()
        n_done = [0]                                        # items 0 .. n_done-1 are not all done; done_flag marks single items
        done_flag = bytearray(n_items)
        free = [threading.Semaphore(0) for _ in ws]        # its records have been gathered, the next item may start
        next_item = [0]
        errs = []

        def take(k, mine):
            if static:
                i = k + mine * len(ws)
                return i if i < n_items else -1
            with cv:
                i = next_item[0]
                if i >= n_items:
                    return -1
                next_item[0] = i + 1
                return i

        def work(k):
            try:
                # the contexts do not all upload at once: the link serves one 56 MB upload in ~1 ms, twelve together in ~12 ms during which
                # the GPU would have nothing to do; context k hands its first batch over k x stagger later (a streaming host is in this state anyway)
                if stagger_s > 0 and k:
                    time.sleep(k * stagger_s)
                mine = 0
                while not errs:
                    i = take(k, mine)
                    if i < 0:
                        break
                    mine += 1
                    ws[k].map(bl_[(first_item + i) % len(bl_)], mode)
                    with cv:
                        owner[i] = k; done_flag[i] = 1
                        cv.notify_all()
                    if do_gather:
                        free[k].acquire()
            except Exception as e:                          # surface the failure instead of hanging the main thread
                with cv:
                    errs.append(e)
                    cv.notify_all()
        th = [threading.Thread(target=work, args=(k,)) for k in range(len(ws))]
        for t_ in th:
            t_.start()
        try:
            for i in range(n_items):
                with cv:
                    while not done_flag[i] and not errs:
                        cv.wait()
                if errs:
                    break
                if RATE_LOG is not None and (i + 1) % RATE_EVERY == 0:
                    RATE_LOG.append((i + 1, time.perf_counter()))       # DART_BENCH_RATE_LOG: when the (i + 1)-th item of this call was back (profiles/probes/sustained.sh)
                if do_gather:
                    gather(ws[owner[i]], first_item + i)
                    free[owner[i]].release()
        except BaseException as e:                          # (a failing gather must not leave the context threads waiting for their turn for ever)
            with cv:
                errs.append(e)
        finally:
            for k in range(len(ws)):
                for _ in range(n_items):
                    free[k].release()
        for t_ in th:
            t_.join()
        if errs:
            raise errs[0]

    def timed(n_items, mode, ws=workers, bl=None):
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_items(n_items, mode, ws, 0, bl)
        barrier(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearse else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    phase("library start-up + first batch of every context")
    run_items(args.warmup * nb, args.input)
    for w in workers:
        w.kern = {}; w.n_runs = 0
    phase("warm-up steps")
    if RATE_LOG is not None:
        del RATE_LOG[:]
    elapsed = timed(args.steps * nb, args.input)
    phase("timed region")
    if RATE_LOG is not None and rank == 0:
        t_wall0 = time.time() - (time.perf_counter() - RATE_LOG[0][1]) if RATE_LOG else 0
        with open(os.environ["DART_BENCH_RATE_LOG"], "w") as f:
            prev = None
            for n_done, t_ in RATE_LOG:
                if prev is not None:
                    f.write("%.3f items %d rate_M_reads_per_s %.1f\n" % (t_wall0 + (t_ - RATE_LOG[0][1]), n_done, (n_done - prev[0]) * reads_per_step / nb / (t_ - prev[1]) / 1e6))
                prev = (n_done, t_)
        del RATE_LOG[:]
    runs = sum(w.n_runs for w in workers)
    kern = {}
    for w in workers:                                       # per-worker sums, merged after the threads have joined
        for k, v in w.kern.items():
            kern[k] = kern.get(k, 0.0) + v
    kern = {k: v / max(runs, 1) for k, v in kern.items()}
    # the spread of the figure: two more runs of the same K steps (the line's `value` is the first, the contract's)
    repeats = [elapsed] + [timed(args.steps * nb, args.input) for _ in range(args.repeats if args.repeats >= 0 else (0 if args.no_secondary else 2))]
    sustained = None
    if args.sustained_s < 0:
        args.sustained_s = 0.0 if args.no_secondary else 20.0
    if world == 1 and args.sustained_s > 0:
        keep = (RATE_LOG, RATE_EVERY)
        n_sus = max(args.steps, int(args.sustained_s / (elapsed / args.steps)))
        RATE_LOG, RATE_EVERY = [], max(1, n_sus * nb // 8)
        t_s0 = time.perf_counter()
        e_s = timed(n_sus * nb, args.input)
        marks = [(0, t_s0)] + RATE_LOG
        RATE_LOG, RATE_EVERY = keep
        sustained = {"value": round(reads_per_step * n_sus / e_s / 1e6, 2), "steps": n_sus, "seconds": round(e_s, 1),
                     "rate_by_eighth_of_the_run": [round((marks[i][0] - marks[i - 1][0]) * reads_per_step / nb / (marks[i][1] - marks[i - 1][1]) / 1e6, 1) for i in range(1, len(marks))]}
        phase("sustained pass")
    counters = workers[0].gpu.counters()
    # batches a context had to run again since it was created, over all contexts: capacities that grew (expected while the first batches
    # size the buffers), scans that did not complete (dg_scan.h: should be 0)
    for key in ("reruns_capacity_total", "reruns_scan_total"):
        counters[key] = sum(w.gpu.counters().get(key, 0) for w in workers)
    if counters["reruns_scan_total"]:
        # The batch was mapped again and its records are the right ones (the timed region contains the lost time), so the line below stands;
        # but a look-back that gives up is a defect to look into: the line carries the count, DART_BENCH_STRICT=1 (probes) makes it fatal.
        log("[bench] WARNING: %d batch(es) were run again because a device-side scan gave up (dg_scan.h): %s" % (counters["reruns_scan_total"], [w.gpu.lib.dg_last_error(w.gpu.ctx) for w in workers]))
        if os.environ.get("DART_BENCH_STRICT") == "1":
            sys.exit(3)

    # ---- which resource do the batches in flight fill?  The kernels' waves report the time they were resident (summed over the waves, last batch of
    #      context 0 inside the timed region); against the batch period that gives the share of the GPU's wave slots, of its vector register file
    #      (registers as the hardware allocates them: next_free_vgpr rounded up to 8) and of its LDS that each kernel holds on average ----
    period_ms = elapsed / (args.steps * nb) * 1e3
    tick_ms = 1e-5
    # (registers as allocated = next_free_vgpr rounded up to 8, LDS per wave: profiles/probes/kres.py on this round's build)
    res = {"k_seed_qf": (136, 51200 / 4), "k_seed_heavy": (232, 0), "k_chain_heavy": (136, 12952), "k_pair": (176, 77992 / 4), "k_report": (232, 16128), "k_reseed": (136, 15744)}
    n_cu_ = torch.cuda.get_device_properties(local).multi_processor_count
    occupancy = {"batch_period_ms": round(period_ms, 3), "kernels": {}}
    tot = [0.0, 0.0, 0.0]
    for k_, (vg, lds_per_wave) in res.items():
        ticks = counters.get("wave_ticks_" + k_, counters.get("k_reseed_wave_ticks_100mhz", 0) if k_ == "k_reseed" else 0)
        wave_ms = ticks * tick_ms
        sh = [wave_ms / (n_cu_ * 32 * period_ms), wave_ms * vg / (n_cu_ * 4 * 512 * period_ms), wave_ms * lds_per_wave / (n_cu_ * 163840 * period_ms)]
        occupancy["kernels"][k_] = {"wave_ms_per_batch": round(wave_ms, 1), "share_of_wave_slots": round(sh[0], 4), "share_of_vector_registers": round(sh[1], 4), "share_of_lds": round(sh[2], 4)}
        tot = [a + b for a, b in zip(tot, sh)]
    occupancy["sum_over_these_kernels"] = {"share_of_wave_slots": round(tot[0], 4), "share_of_vector_registers": round(tot[1], 4), "share_of_lds": round(tot[2], 4)}
    occupancy["note"] = ("wave-resident time per batch / (CUs x 32 wave slots | 4 SIMDs x 512 registers | 160 KB LDS) x the batch period; the batch is the last one context 0 mapped "
                         "inside the timed region (%d batches in flight)" % len(workers))

    # ---- secondary rates, outside the timed region (fewer items) ----
    secondary = {}
    if gather_mode == "full" and not args.no_secondary:
        writer_dl[0] = True
        e_dl = timed(args.steps * nb, args.input)
        writer_dl[0] = False
        secondary["value_with_writer_download"] = round(reads_per_step * args.steps / e_dl / 1e6, 3)
    if strong and world > 1 and not args.no_secondary:
        # the weak-scaling rate beside the strong one: every rank maps ITS OWN whole batches of the job again (as many as the rank with the fewest has, at most
        # --batches), the gathers as in the timed region; whole-job reads of that pass / its max-over-ranks time
        own = [b_ for b_ in batches if b_.n == 2 * args.pairs][:max(1, args.batches)]
        tt = torch.tensor([len(own)], dtype=torch.int64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        nbw = int(tt.item())
        if nbw > 0:
            e_w = timed(args.steps * nbw, args.input, workers, own[:nbw])
            secondary["value_weak_scaling"] = round(2 * args.pairs * nbw * world * args.steps / e_w / 1e6, 3)
            secondary["weak_scaling_pass"] = "every rank maps %d whole batches of its share of the job per step (the rank with the fewest sets the number), %d steps, gathers as in the timed region" % (nbw, args.steps)
    phase("secondary passes with the gather")
    gather_timed = do_gather
    do_gather = False                                      # (the secondary rates are per-GPU diagnostics of other entry points: no gather)
    if not args.no_secondary:
        items = min(args.steps, 3) * nb
        other = "ascii" if args.input == "packed" else "packed"
        secondary["value_%s_input" % other] = round(reads_per_step / nb * items / timed(items, other) / 1e6, 3)
        other_rec = "full" if args.records == "compact" else "compact"
        for w in workers:
            w.records = other_rec
        secondary["value_%s_records" % other_rec] = round(reads_per_step / nb * items / timed(items, args.input) / 1e6, 3)
        for w in workers:
            w.records = args.records
        run_items(len(workers), "resident", static=True)   # (every context holds the batch it mapped last)
        secondary["value_device_resident"] = round(reads_per_step / nb * items / timed(items, "resident") / 1e6, 3)
    # the same item with ONE batch in flight, for per-kernel durations without other batches' kernels sharing the GPU
    w0 = workers[0]; w0.kern = {}; w0.n_runs = 0
    if args.no_secondary:
        w0.map(batches[0], args.input)                     # (resident runs need an uploaded batch)
        w0.kern = {}; w0.n_runs = 0
    run_items(2, "resident", [w0])
    iso = {k: v / 2 for k, v in w0.kern.items()}
    do_gather = gather_timed
    barrier(); torch.cuda.synchronize()

    # ---- --verify-gather: one more step, untimed; rank 0 keeps what the gather delivered and compares it with its own mapping of the same reads ----
    gather_verified = None
    if args.verify_gather and gather_mode == "full":
        if rank == 0:
            gathered = {}
        run_items(nb, args.input)
        if rank == 0:
            from dart_amd import dist as ddist
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import common
            ok = True
            for r in range(world):
                if strong:
                    lo, hi = ddist.shard_bounds_balanced(args.total_pairs, world, r)
                    sp = []
                    p_ = lo
                    while p_ < hi:
                        j = p_ // args.pairs
                        e = min(hi, (j + 1) * args.pairs, args.total_pairs)
                        sp.append((1000 + j, (p_ - j * args.pairs, e - j * args.pairs), min(args.pairs, args.total_pairs - j * args.pairs)))
                        p_ = e
                else:
                    sp = [(1000 + 100 * r + j, None, args.pairs) for j in range(nb)]
                for i, (seed, rows, n_gen) in enumerate(sp):
                    rb_, pb_, cb_, sb_ = gathered[(r, i)]
                    b = Batch(gpu.lib, g, n_gen, args.rlen, seed, args.sub_rate, args.indel_frac, args.spliced, rows=rows)
                    workers[0].map(b, args.input); want = workers[0].result()
                    rc_ = rb_.view(host.READ_C); pc_ = pb_.view(host.REPORT_C)
                    reads_x, rep_x, cig_x = host.expand_compact(rc_, pc_, cb_.view(np.uint32), np.full(len(rc_), args.rlen, np.uint16))
                    got = host.BatchResult(reads_x, rep_x, cig_x, sb_.view(host.SJ_OUT))
                    try:
                        common.assert_same(got, (want.reads, want.reports, want.cigar, want.sj))
                    except AssertionError as e_:
                        ok = False
                        log("[bench] gathered records of rank %d batch %d differ from the single-rank mapping: %s" % (r, i, str(e_)[:300]))
            gather_verified = ok
            gathered = None

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = reads_per_step * args.steps / elapsed / 1e6

    # ---- roofline of the dominant kernel: algorithmic bytes (reference algorithm + layout) of one launch / its mean
    #      HIP-event duration in the timed region ----
    alg = {
        "k_seed": 64 * counters["occ_blocks"] + batches[0].bases + 16 * n_reads,
        "k_locate": 64 * counters["lf_steps"] + 8 * counters["sa_lookups"] + 8 * counters["seeds"],
    }
    for d_ in (kern, iso):                                   # (the re-seeding kernels run on the second stream, beside k_report)
        if "k_reseed(overlapped)" in d_:
            d_["k_reseed"] = d_.pop("k_reseed(overlapped)")
    alg["k_reseed"] = counters["reseed_window"] // 4 + 64 * counters["reseed_calls"]      # window bases at 2 bit/base, streamed once (B_ref)
    stage_kernels = [k for k in ("k_seed", "k_locate", "k_pair", "k_report", "k_chain_heavy", "k_reseed") if k in kern]
    per_read_B = (alg["k_seed"] + alg["k_locate"]) / n_reads
    # PMC traffic: only from passes taken with THESE kernel sources on THIS workload (profiles/run_profile.sh stores the
    # fingerprint of the bench line it profiled in profiles/traffic.json); anything else says nothing about this run
    fingerprint = {"csrc_sha256": kernel_sources_sha256(),
                   "workload": "genome=%s model=%s pairs=%d rlen=%d spliced=%g introns=%d repeat_scale=%g mis=%d sub=%g indel=%g" %
                               (args.genome, args.genome_model, args.pairs, args.rlen, args.spliced, args.introns, args.repeat_scale, args.mis, args.sub_rate, args.indel_frac)}
    tj = {}
    for tname in ("traffic.json", "traffic_human.json", "traffic_spliced.json"):         # (the default workload's passes; the human-like genome's; the spliced 2x151 shape's)
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tpath):
            try:
                cand = json.load(open(tpath))
                if cand.get("_fingerprint") == fingerprint:
                    tj = cand
                    break
            except Exception:
                pass
    # The dominant kernel of an HBM roofline: the one that moves the most bytes through the memory side (PMC; the seeding stage: half of a
    # batch's traffic) -- without a matching PMC profile, the one that runs longest alone.  (The longest stand-alone stage alone is a coin
    # flip by now: seeding and the general-path report are both ~1.0 ms.)  `stages` carries the same figures for every stage either way.
    ms_of = lambda k: iso.get(k) or kern[k]
    dom = max(stage_kernels, key=(lambda k: tj.get(k, 0)) if any(tj.get(k) for k in stage_kernels) else ms_of)
    traffic = tj.get(dom)
    stages = {k: {"ms_standalone": round(ms_of(k), 4), "ms_in_timed_region": round(kern[k], 4), "traffic": tj.get(k),
                  "GBps": round(tj[k] / (ms_of(k) * 1e-3) / 1e9, 1) if tj.get(k) else None,
                  "frac": round(tj[k] / (ms_of(k) * 1e-3) / 8e12, 5) if tj.get(k) else None,
                  "frac_of_random_line_ceiling": round(tj[k] / (ms_of(k) * 1e-3) / 2.58e12, 4) if tj.get(k) else None} for k in stage_kernels}
    dom_bytes = alg.get(dom)
    if dom_bytes is None:      # pair / report kernels: their own input and output (seeds in, records out, read bases compared)
        dom_bytes = 8 * counters["seeds"] + 90 * n_reads
    own = {
        # Occ blocks + 16-byte table entries + per located search one 8-byte SA entry and ~2 text windows of 20 bytes
        # + the read's 2-bit/mask words in + 16-byte hits out
        "k_seed": 64 * counters.get("occ_blocks_executed", 0) + 16 * counters.get("ktab_lookups", 0) + 48 * counters.get("direct_extensions", 0)
                  + 56 * n_reads + 16 * counters["seeds"],
        "k_locate": 64 * counters.get("lf_steps_executed", 0) + 8 * counters["sa_lookups"] + 8 * counters["seeds"],
    }
    own_bytes = int(own.get(dom, dom_bytes))
    ms_alone = ms_of(dom)
    # `achieved` is a HARDWARE statement: the bytes the dominant kernel really moves through the memory side (PMC FETCH_SIZE + WRITE_SIZE of one
    # launch; without a matching profile: the bytes it requests, from the live counters -- a lower bound) over the launch's stand-alone
    # duration, against 8 TB/s.  The rate on the REFERENCE algorithm's bytes stays beside it as `algorithmic_*`: the prefix
    # table, the full SA and the direct text comparison remove most of those bytes, so that rate is a work-elimination factor, not a
    # fraction of any hardware limit.
    moved = traffic if traffic else own_bytes
    achieved = moved / (ms_alone * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                "achieved_basis": ("PMC FETCH_SIZE + WRITE_SIZE bytes of one launch (profiles/traffic.json: same kernel sources, same workload)" if traffic else
                                   "bytes the kernel requests per launch, from its live work counters (no PMC pass matches these kernel sources and this workload): a lower bound of the traffic")
                                  + " / the launch's stand-alone HIP-event duration measured in this run",
                "kernel_ms_standalone": round(ms_alone, 4), "kernel_ms_in_timed_region": round(kern[dom], 4),
                "own_requested_bytes_per_launch": own_bytes,
                # what the memory system allows for THIS access pattern, measured with known bytes (profiles/probes/fetch_calib.hip,
                # profiles/r04/e_fetch_size_calibration_random_accesses.txt): random 64-byte accesses over the 3.1 GB Occ array run at 40.3 G lines/s
                # = 2.58 TB/s from 4 waves per CU on (16-byte accesses, which round 2's probe used: 49 G/s); a random 128-byte access is ONE fabric
                # request but runs at half that rate (21-22 G/s): the bytes, not the requests, are what the memory side limits.  The same probe
                # calibrates the counter: FETCH_SIZE is exact for random accesses of up to 64 bytes (one 64-byte request each, also for 16 bytes)
                # and reports half the bytes of 128-byte and wider ones (MI355X_MICROARCH.md's note on wide streams) -- the seeding kernels' accesses
                # are 8 to 64 bytes wide, so their FETCH_SIZE needs no correction
                "random_64B_line_ceiling_GBps": 2580.0,
                "frac_of_random_line_ceiling": round(achieved / 2580.0, 4),
                # ... per class of access (profiles/probes/footprint_sweep.sh, profiles/r04/y_random_line_rate_vs_footprint.txt): a random look-up made with ONE load
                # instruction (16 bytes of the prefix table, 8 of the suffix array) runs at 46-49 G/s whatever the footprint (3 to 118 GB); a random 64-byte
                # access is FOUR 16-byte loads, each translated on its own once the footprint is beyond the translation caches' reach: 39 G/s over the 3.1 GB
                # Occ array (UTCL1 misses 31 %), 21 G/s over 8 GB, 19 G/s over 16-200 GB (99.5 %) -- ~76 G load instructions/s is what the translation path
                # takes.  The seeding kernels make their wide accesses only in the 3.1 GB Occ array and the 0.8 GB text, so what the memory side allows them is
                "seeding_mixed_ceiling": (lambda one, wide: (lambda ms_min: {
                    "single_load_lookups": int(one), "occ_blocks_and_text_windows": int(wide), "single_load_rate_G_per_s": 46.0, "wide_rate_G_per_s": 39.3, "streamed_bytes": int(max(0, moved - 64 * (one + wide))),
                    "ms_min": round(ms_min, 4), "frac_of_mixed_ceiling": round(ms_min / ms_alone, 4),
                    "what": "single-load look-ups = prefix-table entries + suffix-array entries read by the seeding kernels, wide = Occ blocks + text windows of the direct comparisons "
                            "(1.5 lines each), the rest of the launch's PMC bytes streamed at 8 TB/s; ms_min = single / 46 G/s + wide / 39.3 G/s + streamed / 8 TB/s"})(
                        (one / 46.0e9 + wide / 39.3e9 + max(0, moved - 64 * (one + wide)) / 8e12) * 1e3))(
                    counters.get("ktab_lookups", 0) + counters.get("seedq_slots_locate", 0), counters.get("occ_blocks_executed", 0) + 1.5 * counters.get("direct_extensions", 0)) if dom == "k_seed" else None,
                "algorithmic_bytes_per_launch": int(dom_bytes),
                "algorithmic_GBps_standalone": round(dom_bytes / (ms_alone * 1e-3) / 1e9, 1),
                "algorithmic_GBps_in_timed_region": round(dom_bytes / (kern[dom] * 1e-3) / 1e9, 1),
                "work_elimination": round(dom_bytes / moved, 2),
                "fm_bytes_per_read": round(per_read_B, 1),
                # SURVEY 8d's whole-job form: reads/s x algorithmic bytes per read (this GPU's share of `value`)
                "whole_job_algorithmic_GBps_per_gpu": round(value / world * 1e6 * per_read_B / 1e9, 1),
                "fingerprint": fingerprint,
                "stages": stages,
                "note": "one launch = one batch of %d reads.  achieved / frac = real bytes over the stand-alone duration (see achieved_basis); algorithmic_* = what the "
                        "reference's algorithm and layout would fetch for the same reads (SURVEY 8d: 64 B per Occ block of bwt_2occ4, per LF step, 8 B per SA entry), "
                        "kept as a work-elimination factor" % n_reads}

    # ---- CPU baseline: the oracle ("port") on a bounded sample of the same reads, all host cores ----
    cpu = None
    b0 = batches[0]
    so = b0.so.a; rl = b0.rl.a; flat = b0.seq.a
    res0 = None
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_py
        cores = host_cores()
        # enough work per thread that the sample is not dominated by thread start-up: >= 2000 pairs per core
        ns = min(max(args.cpu_sample_pairs, 2000 * cores), args.pairs) * 2
        orc = oracle_py.Oracle(prefix)
        t = time.perf_counter()
        o_reads, o_rep, o_cig, o_sj = orc.map_batch(orc.params(paired=1, max_mismatch=args.mis, max_intron=args.max_intron), so[:ns], rl[:ns], flat, threads=cores)
        dt = time.perf_counter() - t
        # parity on the sample (outside every timed region): the host records of batch 0 through the timed entry point, and of a
        # second distinct batch's head through the other context, against the oracle
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_parity import cigars_of
        def same_as_oracle(res, o_reads, o_rep, o_cig, nsub):
            ok = bool(np.array_equal(o_reads[["score", "sub_score", "mis_num", "mapq", "n_rep", "best", "n_sj"]], res.reads[:nsub][["score", "sub_score", "mis_num", "mapq", "n_rep", "best", "n_sj"]]))
            nrep = int(o_reads["rep_off"][-1] + o_reads["n_rep"][-1])
            f = ["aln_score", "sj_type", "flag", "paired_idx", "chr", "bdir", "pos", "n_cigar"]
            ok = ok and bool(np.array_equal(o_rep[f], res.reports[:nrep][f]))
            return ok and bool(np.array_equal(cigars_of(o_rep, o_cig), cigars_of(res.reports[:nrep], res.cigar)))
        workers[0].map(b0, args.input); res0 = workers[0].result()
        same = same_as_oracle(res0, o_reads, o_rep, o_cig, ns)
        if nb > 1 and len(workers) > 1:
            b1 = batches[1]; n1 = min(ns, 100000)
            o1 = orc.map_batch(orc.params(paired=1, max_mismatch=args.mis, max_intron=args.max_intron), b1.so.a[:n1], b1.rl.a[:n1], b1.seq.a, threads=cores)
            workers[1].map(b1, args.input)
            same = same and same_as_oracle(workers[1].result(), o1[0], o1[1], o1[2], n1)
        # one core beside all cores (SURVEY 8d "plus -t 1"), on a smaller head of the same sample
        n1t = min(ns, 2 * max(2000, args.cpu_sample_pairs // 8))
        t = time.perf_counter()
        orc.map_batch(orc.params(paired=1, max_mismatch=args.mis, max_intron=args.max_intron), so[:n1t], rl[:n1t], flat, threads=1)
        dt1 = time.perf_counter() - t
        cpu = {"value": round(ns / dt / 1e6, 5), "unit": "M reads/s", "cores": cores, "kind": "port",
               "sample": "first %d pairs of batch 0, oracle/dart_oracle.c with %d threads, %.1f s wall" % (ns // 2, cores, dt),
               "value_t1": round(n1t / dt1 / 1e6, 5), "sample_t1": "first %d pairs of batch 0, 1 thread, %.1f s wall" % (n1t // 2, dt1),
               "gpu_records_identical_on_sample": same,
               # the CPU baseline is the PORT (the reference cannot travel to the GPU box).  How its speed relates to the reference's own object code
               # was measured where both exist, the builder's container: oracle/dart_oracle -t 1 against oracle/_ref/ref_harness, mapping phase taken
               # from the difference of a 4 k-pair and a 24 k-pair run of 2x101 -mis 5 on a 40 Mbp human-like genome: 18.6 k against 28.7 k reads/s
               # per core (round 1, 500 kb genome: 0.7).  value / port_vs_reference = what reference `dart` would show on these cores.
               "port_vs_reference": 0.65, "value_reference_equivalent": round(ns / dt / 1e6 / 0.65, 5)}
        log("[bench] oracle counters on sample:", orc.counters)

    # accuracy beside parity (the reference's Evaluation/eva idea): of the plain fragments (no planted indel or
    # intron), how many reads' best alignment starts within 10 bp of where the read was taken from?  Outside every timed region.
    accuracy = None
    try:
        from dart_amd import evaluate
        if res0 is None:
            workers[0].map(b0, args.input); res0 = workers[0].result()
        accuracy = evaluate.mapping_accuracy(res0, b0.truth, tolerance=10)
    except Exception as e:
        log("[bench] accuracy not computed:", repr(e))

    # ---- the drop-in product end to end (SURVEY 8d "plus end-to-end wall"): `dart -i IDX -f 1.fq -f2 2.fq -o out.sam` as a child process ----
    cli = cli_big = None
    dart_exe = os.path.join(ROOT, "dart_amd", "dart")
    if not args.no_cpu_baseline and os.path.exists(dart_exe):
        # the library context of the timed region is closed first: `dart` is measured as a user runs it, alone on the GPU
        try:
            gpu.close()                                    # (closes its clones too)
        except Exception as e:
            log("[bench] closing the timed region's contexts:", repr(e))
        if args.cli_big_pairs > 0 and args.genome == "grch38" and args.genome_model == "planted" and not args.spliced and args.rlen == 101:
            try:
                # BASELINE configs[2] through the command line: the SAME reads as the timed region's batches (seeds 1000 ...), 1 M pairs per seed
                cli_big = run_cli(dart_exe, prefix, g, label, [(1000 + j, min(1000000, args.cli_big_pairs - j * 1000000)) for j in range((args.cli_big_pairs + 999999) // 1000000)],
                                  args, cpu_pairs=args.cli_cpu_pairs, gz_pairs=args.cli_gz_pairs)
            except Exception as e:
                log("[bench] command-line run (GRCh38-sized) failed:", repr(e))
        if args.cli_pairs > 0:
            try:
                cprefix, cg = prepare_index(args.cache, (["chr20"], [CHR20_LEN]), 0, lambda: None)
                cli = run_cli(dart_exe, cprefix, cg, "chr20-sized synthetic genome", [(1000, args.cli_pairs)], args, cpu_pairs=0, gz_pairs=0)
            except Exception as e:
                log("[bench] command-line run failed:", repr(e))

    in_bytes = b0.bytes_packed if args.input == "packed" else b0.bytes_ascii
    line = {
        "metric": "M paired-end reads/sec (2x%d bp vs GRCh38-sized index), host to host; records bit-identical to CPU dart" % args.rlen,
        "value": round(value, 4), "unit": "M reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "u64/u8 integer", "data": "synthetic",
        "value_repeats": [round(reads_per_step * args.steps / e / 1e6, 2) for e in repeats],
        "config": {"workload": label + ", %s), %d M DISTINCT pairs 2x%d bp per GPU and step (%d batches of %d pairs), %s-mis %d, host-to-host"
                               % ("i.i.d. + planted repeat families x%g: SURVEY 8d" % args.repeat_scale if args.genome_model == "planted" else
                                  "human-like repeat content, about half of the genome: SINE/LINE/older interspersed families, segmental duplications, satellites, microsatellites", round(nb * args.pairs / 1e6), args.rlen, nb, args.pairs,
                                  ("%.0f %% of the reads spliced over %d planted introns, -max_intron %d, " % (100 * args.spliced, args.introns, args.max_intron)) if args.spliced else "", args.mis),
                   "input": ("packed reads (2 bit/base + N list, dg_map_batch_packed): %.1f MB per batch" if args.input == "packed" else "ASCII reads (dg_map_batch): %.1f MB per batch") % (in_bytes / 1e6),
                   "output": ("%s + CIGAR ops + junction tuples into page-locked host arrays: %%.1f MB per batch" %
                              ("compact records (dg_read_c 12 B + dg_report_c 16 B, no CIGAR for plain full-length matches, lossless)" if args.records == "compact" else "dg_read_out 36 B + dg_report_out 40 B")) % (workers[0].out_bytes(n_reads) / 1e6),
                   "host_link": "57 GB/s per direction, full duplex for the copy engines (profiles/probes/duplex_probe.hip, profiles/r03/f_duplex_probe_and_download_cost.txt)",
                   "timed_region": "first batch handed over in host memory -> last record back in host memory (H2D + all kernels + D2H, %d batches in flight)" % len(workers),
                   "pairs_per_gpu_per_step": nb * args.pairs, "read_len": args.rlen, "spliced_fraction": args.spliced, "batches_in_flight_per_gpu": len(workers),
                   "synthetic_genome_repeat_content": ("planted repeat families cover ~18 %% of the genome at --repeat-scale 1 (real human DNA: ~50 %%; `--genome-model human` runs that: "
                                                       "profiles/r03/ holds its line beside this one)") if args.genome_model == "planted" else
                                                      "human-like: ~50 %% of the genome in repeat classes with human-like copy numbers and divergences (dart_amd/synth.py::_make_genome_human)",
                   "parallelism": ("reads sharded x%d (%s), index replicated" % (world, "one job of %d pairs per step cut into contiguous balanced pair ranges" % args.total_pairs if strong else "every rank maps its own batches")) +
                                  ({"full": ", SAM-order gather inside the timed region: every rank's compact records of every batch (per-read records, reports, stored CIGAR ops, junction tuples) HBM -> HBM to rank 0 over RCCL, sizes first",
                                    "reads": ", RCCL gather of the per-read records (%d B per read) to rank 0 inside the timed region" % (12 if args.records == "compact" else 36),
                                    "none": ", no data-path collective"}[gather_mode])},
        "kernels_ms": {k: round(v, 4) for k, v in kern.items()},
        "kernels_ms_one_batch_in_flight": {k: round(v, 4) for k, v in iso.items()},
        "counters_per_launch": counters,
        "library_start_up": {"dg_init_files_s": round(init_s, 3), "split": init_report},
        "gpu_occupancy_in_flight": occupancy,
        "roofline": roofline,
        "cpu_baseline": cpu,
        "accuracy": accuracy,
    }
    if cli_big and cpu and "cpu_reference_object_code" in cli_big:
        # the reference's own object code, timed on THIS box (one thread), beside the port: the ratio replaces the builder's-container figure
        ro = cli_big["cpu_reference_object_code"]
        cpu["reference_object_code_one_thread"] = ro
        if ro["value"] > 0 and cpu.get("value_t1"):
            # one thread against one thread (the port's figure inside a 16-thread run is lower per thread: shared memory bandwidth)
            cpu["port_vs_reference_measured_here"] = round(cpu["value_t1"] / ro["value"], 3)
            cpu["value_reference_equivalent_measured_here"] = round(cpu["value"] / cpu["port_vs_reference_measured_here"], 5)
    if cli_big:
        line["value_cli_end_to_end_grch38"] = cli_big["value"]
        line["cli_end_to_end_grch38"] = cli_big
    if cli:
        line["value_cli_end_to_end"] = cli["value"]
        line["cli_end_to_end"] = cli
    phase("secondary rates, one batch in flight, CPU legs")
    line["phases_s"] = {T_PHASE[i][0]: round(T_PHASE[i][1] - T_PHASE[i - 1][1], 1) for i in range(1, len(T_PHASE))}
    line["phases_s"]["total"] = round(T_PHASE[-1][1] - T_PHASE[0][1], 1)
    if PREP_S:
        line["phases_s"]["of the first phase"] = dict(PREP_S)
    log("[bench] phases (s):", line["phases_s"])
    if gather_mode == "full":
        line["gather"] = {"mode": "full", "bytes_received_by_rank0_total": gather_bytes[0], "verified_against_single_rank_mapping": gather_verified}
    if sustained:
        line["value_sustained"] = sustained["value"]
        line["sustained"] = sustained
    line.update(secondary)
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

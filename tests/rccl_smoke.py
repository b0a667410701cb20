"""Run as a child process by tests/test_gpu_cli.py::test_rccl_single_rank_carries_the_record_arrays (a process group of its own).
One rank, backend "nccl" (= RCCL): the library's compact record arrays, as HBM pointers wrapped in torch tensors, go through the calls the
multi-GPU gather makes -- all_gather of the byte counts, then batch_isend_irecv of exactly the bytes used (here: to itself) -- and what
arrives must be what the library's own download gives.  A one-GPU box cannot run two RCCL ranks (one rank per device), so this is as far
as the collective path can be exercised there: RCCL is loaded, initialised on the device and moves the library's buffers."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist
from dart_amd import synth, index_build, host, dist as ddist

work = sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
g = synth.make_genome([400000, 200000], seed=33, repeat_scale=20.0, n_introns=100)
prefix = os.path.join(work, "rccl_idx")
index_build.build_index_from_genome(g, prefix)
m1, m2 = synth.make_reads(g, 20000, rlen=101, seed=34, spliced_frac=0.2, indel_frac=0.05, n_frac=0.01)
arr = host.interleave_pairs(m1, m2)
words, nlist = host.pack_reads_2bit(arr)
gpu = host.DartGPU(host.Index(prefix), host.default_params(paired=1, max_mismatch=5))
want = gpu.map_batch_compact(words, nlist, 101)                       # (leaves the compact records in HBM)
parts = gpu.device_records_compact()
assert all(p.is_cuda for p in parts) and parts[0].numel() == 12 * len(arr)
# 1. the gather's own code path with one rank: all_gather of the sizes over RCCL
counts = ddist.gather_compact_to_rank0(parts, [None], 1, 0)
assert [int(x) for x in counts[0]] == [int(p.numel()) for p in parts], counts
# 2. the point-to-point leg: exactly the bytes used, sent and received in one batch (to itself)
recv = [torch.zeros(int(p.numel()) + 64, dtype=torch.uint8, device="cuda") for p in parts]
ops = []
for k in range(4):
    if parts[k].numel():
        ops.append(dist.P2POp(dist.irecv, recv[k][:parts[k].numel()], 0))
        ops.append(dist.P2POp(dist.isend, parts[k], 0))
for req in dist.batch_isend_irecv(ops):
    req.wait()
torch.cuda.synchronize()
# 3. what arrived = the records the library's download gives for the same batch
rc = recv[0][:parts[0].numel()].cpu().numpy().view(host.READ_C); pc = recv[1][:parts[1].numel()].cpu().numpy().view(host.REPORT_C)
cg = recv[2][:parts[2].numel()].cpu().numpy().view(np.uint32); sj = recv[3][:parts[3].numel()].cpu().numpy().view(host.SJ_OUT)
r, p, c = host.expand_compact(rc, pc, cg, np.full(len(rc), 101, np.uint16))
import common
common.assert_same(host.BatchResult(r, p, c, sj), (want.reads, want.reports, want.cigar, want.sj))
t = torch.ones(1 << 20, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
assert float(t[0]) == 1.0
dist.destroy_process_group()
gpu.close()
print("rccl smoke ok: %d reads, %d bytes through RCCL" % (len(rc), sum(int(p_.numel()) for p_ in parts)))

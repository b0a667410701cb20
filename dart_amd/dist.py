"""Multi-GPU plumbing: reads shard across ranks with no data-path collective; the only exchange is
the final SAM-order gather of the result records to rank 0 (BASELINE.json north_star).  One
process per GPU; torch.distributed backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in CPU tests.

Record volume is ~36 B/read + ~40 B/report + CIGAR ops, i.e. ~0.2 GB per 2 M reads -- far below
one xGMI link (~153 GB/s), so a plain gather (point-to-point sends to rank 0) is the right
collective; no ring/tree tuning applies.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_pairs: int, world: int, rank: int):
    """Contiguous pair ranges per rank (pairs are never split); input order = rank order."""
    per = (n_pairs + world - 1) // world
    lo = min(n_pairs, rank * per)
    return lo, min(n_pairs, lo + per)


def shard_bounds_balanced(n_pairs: int, world: int, rank: int):
    """Contiguous pair ranges whose sizes differ by at most one pair (50 M pairs over 8 ranks: 6.25 M each); input order = rank order."""
    return n_pairs * rank // world, n_pairs * (rank + 1) // world


def gather_compact_to_rank0(parts, recv_bufs, world: int, rank: int):
    """The SAM-order gather of one batch per rank (north_star; Mapping.cpp:644-664 has ONE ordered writer): every rank hands its four
    compact record arrays (uint8 tensors, in HBM under RCCL) to rank 0, sizes first, then point-to-point sends of exactly the bytes
    used -- xGMI is point-to-point, and 30 bytes per read are far below one link.  recv_bufs[r][k] (rank 0 only): byte buffers that
    hold rank r's arrays.  Returns on rank 0 the byte counts [world][4]; rank 0's own arrays stay where they are (parts)."""
    dev = parts[0].device
    sizes = torch.tensor([int(p.numel()) for p in parts], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    ops = []
    if rank == 0:
        counts = torch.stack(all_sizes).cpu().numpy()
        for r in range(1, world):
            for k in range(4):
                nb = int(counts[r, k])
                if nb:
                    if nb > recv_bufs[r][k].numel():
                        raise RuntimeError("gather buffer for rank %d array %d too small: %d > %d bytes" % (r, k, nb, recv_bufs[r][k].numel()))
                    ops.append(dist.P2POp(dist.irecv, recv_bufs[r][k][:nb], r))
    else:
        counts = None
        for k in range(4):
            if parts[k].numel():
                ops.append(dist.P2POp(dist.isend, parts[k], 0))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return counts


def gather_records(reads: np.ndarray, reports: np.ndarray, cigar: np.ndarray, sj: np.ndarray, device=None):
    """Gathers one rank's records to rank 0 and rebases the offsets so that rank 0 holds one record set
    in global read order.  Returns (reads, reports, cigar, sj) on rank 0, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device or ("cuda" if dist.get_backend() == "nccl" else "cpu")
    sizes = torch.tensor([len(reads), len(reports), len(cigar), len(sj)], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = torch.stack(all_sizes).cpu().numpy()
    out = []
    for k, arr in enumerate((reads, reports, cigar, sj)):
        mx = int(all_sizes[:, k].max())
        item = arr.dtype.itemsize
        buf = torch.zeros(max(mx, 1) * item, dtype=torch.uint8, device=dev)
        if len(arr):
            buf[: len(arr) * item] = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1)).to(dev)
        recv = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, recv, dst=0)
        if rank == 0:
            parts = [recv[r].cpu().numpy()[: int(all_sizes[r, k]) * item].view(arr.dtype) for r in range(world)]
            out.append(parts)
    if rank != 0:
        return None
    r_parts, p_parts, c_parts, s_parts = out
    read_base = rep_base = cig_base = sj_base = 0
    for r in range(world):
        rp, pp, sp = r_parts[r].copy(), p_parts[r].copy(), s_parts[r].copy()
        rp["rep_off"] += rep_base; rp["sj_off"] += sj_base
        pp["cigar_off"] += cig_base
        sp["read_idx"] += read_base
        r_parts[r], p_parts[r], s_parts[r] = rp, pp, sp
        read_base += len(rp); rep_base += len(pp); cig_base += len(c_parts[r]); sj_base += len(sp)
    return (np.concatenate(r_parts), np.concatenate(p_parts), np.concatenate(c_parts), np.concatenate(s_parts))

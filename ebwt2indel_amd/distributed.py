"""Multi-GPU sharding of the phase-4 scan: one process per GPU (torch.distributed; backend "nccl" is RCCL
on ROCm), BWT + flag vectors replicated, scan range partitioned, one gather of the emitted cluster records
to rank 0 (north_star: "scan range partitioned across the 8 GPUs ... RCCL gather over xGMI of the emitted
cluster records").  The reference has no counterpart (single process, ebwt2InDel.cpp:1584-1674)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice of closing positions for `rank`; cuts fall on 64-position flag words."""
    def cut(i):
        if i <= 0:
            return 0
        if i >= world:
            return n
        return min(n, ((n * i // world) + 63) // 64 * 64)
    return cut(rank), cut(rank + 1)


def gather_packed(packed, group=None, device=None, dst: int = 0) -> Optional[List[bytes]]:
    """Variable-length gather of one byte string per rank to `dst` (None elsewhere).
    Sizes travel in an all_gather, payloads in a padded gather; tensors live on `device`
    (a CUDA/HIP device for the nccl backend, CPU for gloo)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device("cpu") if device is None else device
    size = torch.tensor([len(packed)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if len(packed):
        src = torch.frombuffer(packed, dtype=torch.uint8)      # bytes / memoryview (e.g. Calls.blob()): no extra host copy
        buf[: len(packed)].copy_(src, non_blocking=False)
    outs = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(buf, outs, dst=dst, group=group)
    if rank != dst:
        return None
    return [bytes(o[:s].cpu().numpy().tobytes()) for o, s in zip(outs, sizes)]


def write_gathered(parts: List[bytes], params, two_samples: bool):
    """Rank 0: shard call sets in rank order -> one .snp text with a running cluster_nr (ebwt2InDel.cpp:52)."""
    from . import api

    text, cluster_nr, events = b"", 1, 0
    total = 0
    for part in parts:
        calls = api.Calls.unpack(part)
        total += len(calls)
        t, cluster_nr, events = calls.format_snp(params, two_samples, cluster_nr, events)
        text += t
    return text, cluster_nr, events, total

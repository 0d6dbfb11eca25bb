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


_bufs = {}


def _buffers(world: int, cap: int, dev, root: bool, slot: int = 0):
    """Staging tensors reused across steps: send buffer on `dev`; on the root a [world, cap] receive buffer on `dev`
    and a page-locked host mirror.  `slot` selects one of several independent sets (gathers in flight)."""
    import torch

    key = (world, str(dev), root, slot)
    cur = _bufs.get(key)
    if cur is None or cur["cap"] < cap:
        cur = {"cap": cap, "send": torch.empty(cap, dtype=torch.uint8, device=dev)}
        if root:
            cur["recv"] = torch.empty(world * cap, dtype=torch.uint8, device=dev)
            cur["host"] = torch.empty(world * cap, dtype=torch.uint8, pin_memory=True) if dev.type == "cuda" else cur["recv"]
        _bufs[key] = cur
    return cur


class _DeviceBytes:
    """`length` bytes of device memory at `ptr`, for torch.as_tensor (no copy)."""

    def __init__(self, ptr: int, length: int):
        self.__cuda_array_interface__ = {"shape": (length,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}


class PendingGather:
    """A gather in flight (gather_device_async): result() waits for it and returns what gather_packed returns."""

    def __init__(self, event, views, keep, gathered=None):
        self._event, self._views, self.keep = event, views, keep
        self.gathered = gathered   # timing event: every rank's image is resident in the root's HBM (before the root's PCIe download)

    def result(self):
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        return self._views


_slot = [0]


def gather_device_async(ptr: int, length: int, device, group=None, dst: int = 0, keep=None) -> PendingGather:
    """gather_packed for a call set that is still in HBM (ScanJob.device_blob), without blocking the host on the
    payload: the image goes GPU -> GPU over xGMI without touching this rank's host memory; only the root downloads,
    once, into a page-locked buffer.  Only the exchange of the sizes is synchronous; the payload gather and the root's
    download are stream-ordered behind it, so a caller that pipelines its steps runs them under the next step's
    kernels (two sets of staging buffers alternate).  `keep`: object that owns the device image (the ScanJob); it is
    handed back as PendingGather.keep and must stay alive until result() returned."""
    import torch
    import torch.distributed as dist

    dev = torch.device(device)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    src = torch.as_tensor(_DeviceBytes(ptr, length), device=dev) if length else torch.empty(0, dtype=torch.uint8, device=dev)
    sizes_t = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes_t, torch.tensor([length], dtype=torch.int64, device=dev), group=group)
    sizes = [int(x) for x in sizes_t.tolist()]
    cap = (max(max(sizes), 1) + 0xFFFF) & ~0xFFFF          # same on every rank
    _slot[0] ^= 1
    buf = _buffers(world, cap + cap // 4, dev, rank == dst, slot=_slot[0])
    send = buf["send"][:cap]
    if length:
        send[:length].copy_(src)
    outs = [buf["recv"][r * cap:(r + 1) * cap] for r in range(world)] if rank == dst else None
    work = dist.gather(send, outs, dst=dst, group=group, async_op=True)
    work.wait()                                           # orders the current stream behind the collective, not the host
    gathered = torch.cuda.Event(enable_timing=True)
    gathered.record(torch.cuda.current_stream(dev))
    views = None
    if rank == dst:
        host = buf["host"]
        for r in range(world):
            host[r * cap: r * cap + sizes[r]].copy_(outs[r][: sizes[r]], non_blocking=True)
        base = host.numpy()
        views = [memoryview(base[r * cap: r * cap + sizes[r]]) for r in range(world)]
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    return PendingGather(ev, views, keep, gathered)


def gather_device(ptr: int, length: int, device, group=None, dst: int = 0) -> Optional[List[memoryview]]:
    """Blocking form of gather_device_async."""
    return gather_device_async(ptr, length, device, group=group, dst=dst).result()


def gather_packed(packed, group=None, device=None, dst: int = 0) -> Optional[List[memoryview]]:
    """Variable-length gather of one byte string (bytes / memoryview, e.g. Calls.blob(), or a uint8 tensor on `device`) per rank to `dst`
    (None elsewhere).  One small all_gather tells every rank all sizes (so all derive the same padded length), one
    gather moves the payloads; tensors live on `device` (the rank's GPU for nccl = RCCL over xGMI, CPU for gloo).
    On the root the result is a list of views into a reused page-locked host buffer: valid until the next call."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device("cpu") if device is None else torch.device(device)
    import torch as _t
    on_device = isinstance(packed, _t.Tensor)
    n = int(packed.numel()) if on_device else len(packed)
    sizes_t = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes_t, torch.tensor([n], dtype=torch.int64, device=dev), group=group)
    sizes = [int(x) for x in sizes_t.tolist()]
    cap = (max(max(sizes), 1) + 0xFFFF) & ~0xFFFF          # same on every rank
    buf = _buffers(world, cap + cap // 4, dev, rank == dst)  # grow with some slack, slice to the agreed length
    send = buf["send"][:cap]
    if n:
        send[:n].copy_(packed if on_device else torch.frombuffer(packed, dtype=torch.uint8))
    outs = [buf["recv"][r * cap:(r + 1) * cap] for r in range(world)] if rank == dst else None
    dist.gather(send, outs, dst=dst, group=group)
    if rank != dst:
        if on_device:
            torch.cuda.current_stream(dev).synchronize()  # the source image may be released once this returns
        return None
    host = buf["host"]
    if dev.type == "cuda":
        for r in range(world):
            host[r * cap: r * cap + sizes[r]].copy_(outs[r][: sizes[r]], non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
    base = host.numpy()
    return [memoryview(base[r * cap: r * cap + sizes[r]]) for r in range(world)]


class NodeArena:
    """The ranks of ONE node hand their records to rank 0 through host memory they all map (round 5): every rank downloads its
    own wire image over its own PCIe link into its slot of a shared-memory segment, rank 0 reads the slots -- no rank's link
    carries another rank's records (the RCCL gather to GPU 0, gather_device_async, puts every record of the job on GPU 0's link:
    35 MB per step at N = 8, longer than a rank's kernels).  No collective on the data path: a slot carries its own sequence
    number and length, rank 0 polls them, and a counter of consumed steps holds the ranks back from a slot rank 0 still reads.

        layout   [consumed u64, 56 B pad] then per rank `depth` slots of [seq u64, len u64, 48 B pad, image (cap bytes)]

    Set up collectively (one broadcast of the segment's name, one barrier); `cap` must hold the largest image of a step
    (EbError otherwise: the caller falls back to the gather).  Page-locked per rank (eb_host_register) when the ranks have
    GPUs; plain shared memory otherwise (CPU rehearsal with byte strings)."""

    HDR = 64

    def __init__(self, cap: int, depth: int = 4, group=None, pin: bool = True):
        import mmap
        import os
        import torch.distributed as dist

        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.group, self.depth = group, depth
        self.cap = ((int(cap) + self.HDR + 4095) & ~4095) - self.HDR     # a slot (header + image) is a whole number of pages: every
        self.slot = self.HDR + self.cap                                    # rank page-locks pages no other rank writes
        self.size = 4096 + self.world * depth * self.slot
        name = [None]
        if self.rank == 0:
            # /dev/shm when it has the room (a container may give it 64 MB: writing past that is a SIGBUS, not an error), else a file
            # in the temporary directory -- a shared mapping of any file is shared memory; its pages are page-locked by the ranks
            import tempfile
            where = "/dev/shm"
            try:
                st = os.statvfs(where)
                if st.f_bavail * st.f_frsize < self.size + (self.size >> 3):
                    where = tempfile.gettempdir()
            except OSError:
                where = tempfile.gettempdir()
            name[0] = os.path.join(where, f"ebwt2indel_arena_{os.getpid()}_{id(self) & 0xFFFFFF:x}")
            fd = os.open(name[0], os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            os.posix_fallocate(fd, 0, self.size)           # (the blocks exist: no SIGBUS later)
        if self.world > 1:
            dist.broadcast_object_list(name, src=0, group=group)
        if self.rank != 0:
            fd = os.open(name[0], os.O_RDWR)
        self._mm = mmap.mmap(fd, self.size)
        os.close(fd)
        if self.world > 1:
            dist.barrier(group=group)                 # everyone has it mapped
        if self.rank == 0:
            os.unlink(name[0])                        # the mappings keep it alive; nothing is left behind whatever happens later
        self._np = np.frombuffer(self._mm, dtype=np.uint8)
        self._u64 = self._np[: self.size // 8 * 8].view(np.uint64)
        self._base = self._np.ctypes.data
        self._pinned = None
        if pin:
            from . import api
            mine = self._base + 4096 + self.rank * depth * self.slot
            if api.lib().eb_host_register(mine, depth * self.slot) == 0:
                self._pinned = mine
            else:
                # the download then goes through the runtime's own staging (slower, same bytes): a node whose limits on locked
                # memory refuse the registration still runs
                import sys
                print(f"[ebwt2indel] rank {self.rank}: the arena's slots could not be page-locked ({api.lib().eb_last_error().decode()}); downloads into it are staged",
                      file=sys.stderr, flush=True)
        if self.rank == 0:
            self._u64[0] = 0
        self.step = 0                                 # next step this rank publishes
        self.taken = 0                                # rank 0: steps collected so far

    def _slot_off(self, rank: int, step: int) -> int:
        return 4096 + (rank * self.depth + step % self.depth) * self.slot

    def _wait_free(self, step: int):
        import time
        while step >= int(self._u64[0]) + self.depth:   # rank 0 still reads the slot's previous occupant
            time.sleep(0)

    def publish_job(self, job) -> int:
        """This rank's share of the next step: the job's compact wire image, downloaded straight into the slot."""
        step = self.step
        self._wait_free(step)
        off = self._slot_off(self.rank, step)
        n = job.wire_to_host(self._base + off + self.HDR, self.cap)
        self._u64[off // 8 + 1] = n
        self._u64[off // 8] = step + 1                # (x86: stores are seen in order; the image was complete before wire_to_host returned)
        self.step += 1
        return n

    def publish_bytes(self, data) -> int:
        """The same with a byte string (a call set already in host memory; CPU rehearsal)."""
        step = self.step
        self._wait_free(step)
        off = self._slot_off(self.rank, step)
        n = len(data)
        if n > self.cap:
            raise ValueError(f"the image takes {n} bytes, a slot of the arena holds {self.cap}")
        self._np[off + self.HDR: off + self.HDR + n] = np.frombuffer(data, dtype=np.uint8)
        self._u64[off // 8 + 1] = n
        self._u64[off // 8] = step + 1
        self.step += 1
        return n

    def collect(self) -> List[memoryview]:
        """Rank 0: the images of every rank for the next step, in rank order, as views into the arena -- valid until release()."""
        import time
        step = self.taken
        views = []
        for r in range(self.world):
            off = self._slot_off(r, step)
            while int(self._u64[off // 8]) != step + 1:
                time.sleep(0)
            n = int(self._u64[off // 8 + 1])
            views.append(memoryview(self._np[off + self.HDR: off + self.HDR + n]))
        self.taken += 1
        return views

    def release(self):
        """Rank 0 is done with every step collected so far: their slots may be written again."""
        self._u64[0] = self.taken

    def close(self):
        if self._pinned is not None:
            from . import api
            api.lib().eb_host_unregister(self._pinned)
            self._pinned = None
        self._u64 = self._np = None
        try:
            self._mm.close()
        except BufferError:
            pass                                       # a caller still holds a view: the mapping goes with the process


class _DeviceWords:
    """`words` int64 words of device memory at `ptr`, for torch.as_tensor (no copy)."""

    def __init__(self, ptr: int, words: int):
        self.__cuda_array_interface__ = {"shape": (words,), "typestr": "<i8", "data": (ptr, False), "version": 3, "strides": None}


_NAV_COUNTERS = ("leaves", "nodes", "lcp_values", "da_values", "minima", "rank_queries", "batches")
_NAV_SUMS = ("lcp_pos_sum", "lcp_pos_sq_sum")      # modulo 2^64: they travel as four 16-bit limbs each, so that no
                                                   # backend ever adds two int64 values that overflow (signed overflow is
                                                   # undefined on the gloo path and unspecified for RCCL)


def _limbs(v: int) -> List[int]:
    v &= (1 << 64) - 1
    return [(v >> (16 * k)) & 0xFFFF for k in range(4)]


def _from_limbs(l) -> int:
    """Sum over the ranks of each limb (< world * 2^16, exact in int64) -> the sum of the values modulo 2^64."""
    return sum(int(x) << (16 * k) for k, x in enumerate(l)) & ((1 << 64) - 1)


def navigate_sharded(bwt1, params, bwt2=None, group=None, device=None):
    """Phases 2-3 split over the ranks: each rank walks its share of the Weiner-link subtrees
    (eb_flags_build_shard), then ONE all-reduce per bit vector combines them -- every bit has a single writer over all
    ranks, so integer SUM == OR (RCCL has no bitwise OR).  `device`: the rank's GPU for nccl (vectors are reduced in
    place in HBM over xGMI); CPU for gloo (vectors staged through host memory: rehearsal only).  Every rank returns
    complete Flags; stats hold the job totals (seconds: slowest rank)."""
    import torch
    import torch.distributed as dist
    from . import api

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    import os
    fl = api.Flags.build_shard(bwt1, bwt2, params.K, params.k_right, rank, world)
    if world == 1 and not (dist.is_initialized() and os.environ.get("EB_FORCE_COLLECTIVES") == "1"):   # the latter: rehearse the collectives with one rank
        fl.finalize()
        return fl
    st = fl.stats
    dev = torch.device("cpu") if device is None else torch.device(device)
    # SUM stands in for OR only while every bit has ONE writer over all ranks; a doubly-set bit would carry into its
    # neighbour and corrupt phase 4 silently.  popcount(a + b) = popcount(a) + popcount(b) - carries, so the ones of the
    # reduced vectors must equal the sum of the ranks' ones -- checked on every run (four small device reductions).
    ones = torch.tensor(fl.popcounts(), dtype=torch.int64, device=dev)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM, group=group)
    if dev.type == "cuda":
        ptrs, words = fl.device_vectors()
        for ptr in ptrs:
            if ptr:
                dist.all_reduce(torch.as_tensor(_DeviceWords(ptr, words), device=dev), op=dist.ReduceOp.SUM, group=group)
        torch.cuda.current_stream(dev).synchronize()
    else:
        vecs = fl.export()
        for v in vecs:
            if v is not None:
                dist.all_reduce(torch.from_numpy(v.view(np.int64)), op=dist.ReduceOp.SUM, group=group)
        n = bwt1.n + (bwt2.n if bwt2 is not None else 0)
        fl.close()
        fl = api.Flags.from_shard_sum(n, *vecs, device=bwt1.device)
        fl.stats = st
        got = [int(np.unpackbits(v.view(np.uint8)).sum()) if v is not None else 0 for v in vecs]
    if dev.type == "cuda":
        got = fl.popcounts()
    if got != [int(v) for v in ones.tolist()]:
        raise RuntimeError(f"sharded traversal: a bit of the flag vectors was set by more than one rank (ones after the reduction {got}, "
                           f"sum of the ranks' ones {ones.tolist()}; order thrK, thrR, minima, da)")
    if dev.type == "cuda":
        fl.finalize()
    cnt = torch.tensor([int(getattr(st, k)) for k in _NAV_COUNTERS] + [x for k in _NAV_SUMS for x in _limbs(int(getattr(st, k)))], dtype=torch.int64, device=dev)
    sec = torch.tensor([st.seconds_leaves, st.seconds_nodes, st.seconds_traversal], dtype=torch.float64, device=dev)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(sec, op=dist.ReduceOp.MAX, group=group)
    cnt = cnt.tolist()
    for k, v in zip(_NAV_COUNTERS, cnt):
        setattr(st, k, int(v))
    for i, k in enumerate(_NAV_SUMS):
        setattr(st, k, _from_limbs(cnt[len(_NAV_COUNTERS) + 4 * i: len(_NAV_COUNTERS) + 4 * i + 4]))
    if st.n > 0 and not api.lib().eb_nav_covers_every_slot_once(st):
        # the shards collected the complement of the LCP thresholds, which is exact only if every position has one writer
        raise RuntimeError(f"sharded traversal wrote {st.lcp_values} of {st.n} LCP slots (or some of them twice): the input is not the eBWT of a read collection "
                           f"(the single-GPU path handles such inputs the reference's way)")
    st.seconds_leaves, st.seconds_nodes, st.seconds_traversal = float(sec[0]), float(sec[1]), float(sec[2])
    return fl


def write_gathered(parts: List[bytes], params, two_samples: bool):
    """Rank 0: shard call sets in rank order -> one .snp text with a running cluster_nr (ebwt2InDel.cpp:52)."""
    from . import api

    text, cluster_nr, events = b"", 1, 0
    total = 0
    for part in parts:
        calls = api.Calls.unpack(bytes(part) if not isinstance(part, (bytes, bytearray)) else part)
        total += len(calls)
        t, cluster_nr, events = calls.format_snp(params, two_samples, cluster_nr, events)
        text += t
    return text, cluster_nr, events, total

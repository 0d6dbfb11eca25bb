"""ebwt2InDel over the GPUs of one node: one process per GPU, same flags as the executable.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        -m ebwt2indel_amd.run_distributed -1 reads.ebwt -o out.snp [-2 other.ebwt | -d reads.da] [-L -R -k -g -v -m -c -q -t]

Every rank loads the eBWT(s) onto its GPU; phases 2-3 are split by Weiner-link subtrees and combined with one all-reduce
per bit vector (RCCL over xGMI); phase 4 is split by closing position; every rank downloads its compact call-set image over
its own PCIe link into a host arena all ranks map (ebd.NodeArena; EB_DIST_GATHER=rccl: the images travel GPU -> GPU to rank 0
instead, which downloads all of them), and rank 0 runs the event logic and writes the .snp -- byte for byte the file of the
single-GPU executable (tests/test_gpu_distributed.py).  Backend: nccl (= RCCL); EB_DIST_BACKEND=gloo stages the vectors through
host memory (rehearsal with several ranks on one GPU).  The reference has no counterpart (one process, ebwt2InDel.cpp:1677-1823)."""
from __future__ import annotations

import argparse
import os
import sys
import time


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="ebwt2indel_amd.run_distributed", add_help=True)
    ap.add_argument("-1", dest="in1", required=True)
    ap.add_argument("-2", dest="in2")
    ap.add_argument("-d", dest="da")
    ap.add_argument("-o", dest="out", required=True)
    for flag in ("-L", "-R", "-k", "-g", "-v", "-m", "-c", "-q", "-t"):
        ap.add_argument(flag, type=int, default=0)
    a = ap.parse_args(argv)
    if a.in2 and a.da:
        print("Error: Document array (-d) can only be used with one input BWT file (-1)")   # ebwt2InDel.cpp:1760-1765
        return 1

    import torch
    import torch.distributed as dist

    from . import api
    from . import distributed as ebd

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count()
    backend = os.environ.get("EB_DIST_BACKEND", "nccl")
    if ndev < 1:
        print("Error: no GPU visible (the hot path has no CPU fallback)")
        return 1
    if backend == "nccl" and world > ndev:
        # RCCL needs one GPU per rank; wrapping LOCAL_RANK would put two ranks on one device
        print(f"Error: {world} ranks but {ndev} GPU(s): one GPU per rank is required with nccl (EB_DIST_BACKEND=gloo lets ranks share devices)")
        return 1
    local = int(os.environ.get("LOCAL_RANK", "0")) % ndev
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    torch.cuda.set_device(local)
    if backend == "nccl":
        # RCCL prints a version banner on stdout while the group comes up: keep stdout clean for callers
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        finally:
            os.dup2(keep, 1)
            os.close(keep)
        comm_dev = torch.device(f"cuda:{local}")
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
        comm_dev = torch.device("cpu")
    try:
        args = []
        for flag in ("-L", "-R", "-k", "-g", "-v", "-m", "-c", "-q", "-t"):
            v = getattr(a, flag[1:])
            if v:
                args += [flag, str(v)]
        p = api.make_params(args)
        two = bool(a.in2) or bool(a.da)
        t0 = time.time()
        b1 = api.DnaBwt(a.in1, term=p.term, device=local)
        b2 = api.DnaBwt(a.in2, term=p.term, device=local) if a.in2 else None
        t1 = time.time()
        fl = ebd.navigate_sharded(b1, p, b2, device=comm_dev)
        if a.da:
            with open(a.da, "rb") as fh:
                fl.set_da(fh.read())
        t2 = time.time()
        n = b1.n + (b2.n if b2 is not None else 0)
        begin, end = ebd.shard_range(n, rank, world)
        job = api.scan_clusters_begin(b1, fl, p, bwt2=b2, begin=begin, end=end, device_only=True, wire=True)
        job.wait()
        ptr, length = job.device_wire()
        if os.environ.get("EB_DIST_GATHER", "arena") == "rccl" and comm_dev.type == "cuda":
            parts = ebd.gather_device(ptr, length, comm_dev)         # rounds 2-4: GPU -> GPU to rank 0, which downloads every image
        else:
            # every rank downloads its own image over its own PCIe link into a host arena all ranks map; rank 0 reads the slots
            t = torch.tensor([length], dtype=torch.int64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            arena = ebd.NodeArena(cap=int(t.item()) + 4096, depth=1)
            arena.publish_job(job)
            parts = [bytes(v) for v in arena.collect()] if rank == 0 else None
            dist.barrier()
            arena.close()
        job.finish()
        t3 = time.time()
        if rank == 0:
            text, cluster_nr, events, total = ebd.write_gathered(parts, p, two)
            with open(a.out, "wb") as fh:
                fh.write(text)
            st = fl.stats
            print(f"ebwt2InDel on {world} GPU(s): n={n}, {st.nodes} suffix-tree nodes, {total} clusters -> {text.count(b'>')} records in {a.out}")
            print(f"Timings (s): load+index {t1 - t0:.3f}, phases 2-3 {t2 - t1:.3f}, phase 4 + gather {t3 - t2:.3f}, write {time.time() - t3:.3f}")
        dist.barrier()
    finally:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

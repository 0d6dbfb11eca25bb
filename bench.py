#!/usr/bin/env python3
"""bench.py -- phase-4 hot path of ebwt2InDel on MI355X: BWT positions scanned per second.

One step = one full pass of the cluster scan + left/right context extraction over the whole BWT
(all ranks together: the BWT and flag vectors are replicated per GPU, the scan range is partitioned,
the emitted cluster records of every rank are delivered to rank 0's host memory -- each rank's share over
its own PCIe link into a host arena; EB_BENCH_GATHER=rccl: the RCCL gather over xGMI of rounds 2-4).  Inputs are resident in HBM before
the timed region.  Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      dominant kernel of the step (k_scan): 0.75 algorithmic bytes per BWT position
                (3-bit symbol + 2 LCP-threshold bits + 1 LCP-minimum bit) vs 8 TB/s HBM,
  rank_roofline rank4 micro-benchmark: 64 algorithmic bytes per parallel_rank query,
  cpu_baseline  the CPU oracle (oracle/ebwt_oracle.c, a port of the reference algorithm) timed on
                this host on a bounded sample of the same workload family.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (Rounds 2-4 set GPU_MAX_HW_QUEUES=16 here.  Round 5 leaves the HIP runtime's default of 4 hardware queues: the library creates an
# index's scan / context / walk / copy streams in a row, so they sit on different queues, and every dispatch is faster with 4 queues than
# with 8 or 16 -- profiles/r05_hw_queues.txt.  A value in the environment is respected and echoed in config.env.)

import numpy as np  # noqa: E402


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def launch_ranks(n_gpus: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (torch.distributed.run, one process per GPU, RCCL)
    as CHILD processes with this command line, relay rank 0's JSON line and return the launcher's exit code.  Nothing
    here initialises HIP (a process that did must not be replaced or forked into ranks)."""
    import socket
    import subprocess

    backend = os.environ.get("EB_BENCH_BACKEND", "nccl")
    import torch
    ndev = torch.cuda.device_count()          # counts devices without creating a HIP context
    if backend == "nccl" and ndev < n_gpus:
        log(f"[bench] --gpus {n_gpus} needs {n_gpus} GPUs with RCCL, this node shows {ndev}; "
            f"(EB_BENCH_BACKEND=gloo rehearses the N-rank path with ranks sharing devices)")
        return 2
    if ndev < 1:
        log("[bench] no GPU visible: the hot path has no CPU fallback")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    log(f"[bench] launching {n_gpus} ranks ({backend}): {' '.join(cmd)}")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    for l in r.stdout.splitlines():
        if not l.startswith("{"):
            log(l)
    if r.returncode != 0 or len(lines) != 1:
        log(f"[bench] the {n_gpus}-rank run failed (exit {r.returncode}, {len(lines)} JSON lines)")
        return r.returncode or 1
    print(lines[0], flush=True)
    return 0


def _child_line(extra, args, timeout=600):
    """One run of this very script in a child process (after the parent's own line was measured): its JSON line, or {"error": ...}."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + extra + ["--read-len", str(args.read_len), "--coverage", str(args.coverage), "--steps", str(min(args.steps, 10)),
                                                                "--warmup", str(min(args.warmup, 3)), "--cpu-sample-reads", "0", "--rank-queries", str(min(args.rank_queries, 1 << 20)), "--no-other-modes"]
    t0 = time.time()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if os.environ.get("EB_BENCH_TRACE") == "1":
            log(f"[bench] child {' '.join(extra)}, stderr:\n" + "\n".join("    " + ln[:300] for ln in r.stderr.splitlines() if ln.startswith("[")))
        if r.returncode != 0 or not lines:
            return {"error": (r.stderr or r.stdout)[-400:], "returncode": r.returncode}
        d = json.loads(lines[-1])
        d["process_seconds"] = round(time.time() - t0, 1)
        return d
    except Exception as e:      # noqa: BLE001 -- the headline line must not depend on these
        return {"error": repr(e)[-400:]}


def _digest(d):
    if "error" in d:
        return d
    return {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
            "step_roofline_frac": d["step_roofline"]["frac"], "algorithmic_bytes_per_position": d["roofline"]["algorithmic_bytes_per_position"],
            "scan_ms": d["phase4_device_ms"]["scan"], "extend_ms": d["phase4_device_ms"]["extend"], "right_ms": d["phase4_device_ms"]["right"],
            "scan_reruns": d["phase4_device_ms"]["scan_reruns"], "batch_tiles": d["phase4_device_ms"]["batch_tiles"],
            "alone": d["phase4_device_ms"]["alone"], "one_shot_ms": d["one_shot"]["ms"] if d.get("one_shot") else None,
            "ms_per_step_without_fl_table": d["fl_table"]["ms_per_step_without_it"] if d.get("fl_table") else None,
            "traversal_seconds": d["phases23_gpu"]["seconds"], "nodes": d["phases23_gpu"]["nodes"], "leaves": d["phases23_gpu"]["leaves"],
            **({"document_array_packed_from_hbm_seconds": d["phases23_gpu"]["document_array_packed_from_hbm_seconds"]} if "document_array_packed_from_hbm_seconds" in d["phases23_gpu"] else {}),
            "snp_records": d.get("snp_records"), "snp_sha256": d.get("snp_sha256"), "snp_verified": d.get("snp_verified"), "process_seconds": d.get("process_seconds")}


def other_modes(args) -> dict:
    """Modes 2 and 3 (BASELINE configs[2], and configs[3] on this one GPU) through this very script, one child process each, so that
    the driver's default run records them too: the same pipelined steps, fewer of them, no CPU leg; a digest of each child's line."""
    out = {"what": "short runs of this script with --mode 2 (two BWTs, --reads as two collections) and --mode 3 (merged BWT + document array, twice "
                   "--reads: BASELINE configs[3] on ONE GPU), child processes started after the line above was measured; digest of their lines"}
    for mode, reads in ((2, args.reads), (3, 2 * args.reads)):
        out[f"mode{mode}"] = _digest(_child_line(["--mode", str(mode), "--reads", str(reads)], args))
    return out


def realistic_reads(args) -> dict:
    """The metric's workload with the reference's documented kind of input (README.md:88-100): reads with 1 % substitution errors, and
    half as many such reads plus their reverse complements -- child processes as in other_modes, each held to the full-size oracle record
    of its input (profiles/r05_fullsize_config2err.json / config2rc.json) when --reads is the default."""
    return {"what": "short runs of this script with --error-rate-ppm 10000, and with --reads / 2 --error-rate-ppm 10000 --reverse-complements (the same number of reads in the "
                    "collection); child processes started after the line above was measured; digest of their lines",
            "errors_1pct": _digest(_child_line(["--reads", str(args.reads), "--error-rate-ppm", "10000"], args)),
            "errors_1pct_with_reverse_complements": _digest(_child_line(["--reads", str(args.reads // 2), "--error-rate-ppm", "10000", "--reverse-complements"], args))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads in the synthetic sample (configs[1]: 50 M)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--mode", type=int, default=1, choices=(1, 2, 3),
                    help="1: one collection (BASELINE configs[1], the metric's workload); 2: the reads as TWO collections of --reads / 2 each, two BWTs "
                         "(configs[2] at the default --reads); 3: the same two collections merged + document array (configs[3] with --reads 100000000)")
    ap.add_argument("--error-rate-ppm", type=int, default=0, help="substitution errors per million bases in the reads (the reference's documented input "
                    "is simulated reads with an error profile, README.md:88-100); 0: error-free reads, the BASELINE workload")
    ap.add_argument("--reverse-complements", action="store_true", help="add the reverse complement of every read to the collection (README.md:90); "
                    "with --reads 25000000 the collection has the BASELINE's 50 M reads")
    ap.add_argument("--rank-queries", type=int, default=1 << 28)   # SURVEY.md §8(d) metric 2: 2^28 uniform random positions
    ap.add_argument("--cpu-sample-reads", type=int, default=400_000, help="reads in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-dram-reads", type=int, default=7_000_000, help="reads in the DRAM-resident CPU-baseline sample (n = 151 x reads; 0 = skip)")
    ap.add_argument("--rehearse-rank-of", type=int, default=0, metavar="W",
                    help="one-GPU rehearsal of ONE rank's share of a W-GPU run: scan range W//2 of W, collectives with a single rank "
                         "(needs EB_BENCH_FORCE_DIST=1); the JSON line is labelled and its value counts the share only")
    ap.add_argument("--no-other-modes", action="store_true", help="default run (mode 1, one GPU): do not append the short runs of modes 2 and 3 "
                    "(child processes of this script: configs[2], and configs[3] on this one GPU) and of reads with errors / reverse complements to the line "
                    "as `other_modes` / `realistic_reads`")
    ap.add_argument("--no-pipeline", action="store_true", help="finish every step before the next begins (profiling: under rocprofv3 the "
                    "overlapped record download runs as a blit kernel and slows the scan it overlaps)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if os.environ.get("EB_SCAN_DEBUG"):
        # the shipped library has no work-skipping switches (they exist only in the -DEB_SCAN_EXPERIMENTS profiling build);
        # a bench run must not even look as if it could have been hollowed out
        raise SystemExit("EB_SCAN_DEBUG is set: refusing to benchmark (unset it; the switch belongs to the profiling build)")

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            # Started bare with --gpus N: this process becomes the launcher of N ranks (one per GPU) and never touches a
            # GPU itself -- counting devices does not initialise HIP; the ranks are fresh child processes.
            sys.exit(launch_ranks(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus and os.environ.get("EB_BENCH_FORCE_DIST") != "1":
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks: refusing to mislabel the run")

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # EB_BENCH_BACKEND=gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks (ranks share devices,
    # records travel as CPU tensors); the real multi-GPU run uses nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("EB_BENCH_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPU(s): one GPU per rank is required with nccl")
    local = local % ndev
    use_dist = world > 1 or os.environ.get("EB_BENCH_FORCE_DIST") == "1"   # the latter: exercise the collectives with one rank
    if os.environ.get("EB_BENCH_FORCE_DIST") == "1":
        os.environ["EB_FORCE_COLLECTIVES"] = "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when the first communicator comes up; stdout must carry exactly one
        # JSON line, so fd 1 points at stderr until the communicator exists.
        sys.stdout.flush()
        saved_out = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                torch.cuda.set_device(local)
                dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local}"))
                dist.barrier()
                warm = torch.zeros(1, device=f"cuda:{local}")
                dist.all_reduce(warm)
                torch.cuda.synchronize()
            else:
                dist.init_process_group(backend=backend)
                dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_out, 1)
            os.close(saved_out)
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    comm_dev = dev if backend == "nccl" else torch.device("cpu")

    from ebwt2indel_amd import api, sim
    from ebwt2indel_amd import distributed as ebd

    # ---------------- synthetic workload (SURVEY.md 8d recipe), identical on every rank:
    # random genome, second haplotype with a SNP / 500 bp and a 1-3 bp indel / 2 kb, error-free reads alternating
    # between the haplotypes.  The eBWT is built on this rank's GPU (ebwt2indel_amd/tools/ebwt_gen_slabs.hip) and never
    # leaves HBM: generator output -> eb_index_create_from_device.
    genome_len = int(args.reads * args.read_len / args.coverage)
    threads = max(1, (os.cpu_count() or 8) // max(1, world))
    realistic = args.error_rate_ppm > 0 or args.reverse_complements
    if args.mode != 1 and realistic:
        raise SystemExit("--error-rate-ppm / --reverse-complements: mode 1 only")
    dev_ebwt2 = None
    t0 = time.time()
    if args.mode == 1:
        g = sim.random_genome(genome_len, args.seed)
        h2 = sim.mutate_haplotype(g, args.seed)
        text, offs = sim.haplotype_text([g, h2])
        starts = sim.sample_read_starts([g, h2], offs, args.reads, args.read_len, args.seed)
        del g, h2
    else:
        # two individuals, each homozygous for one haplotype of the same diploid genome, --reads / 2 reads each (the recipe of
        # tests/_fullsize.py config3 / config4, whose full-size oracle records the .snp below is held to)
        text, s0, s1 = sim.two_sample_text(args.reads // 2, args.read_len, args.coverage)
    t_host = time.time() - t0
    t0 = time.time()
    # (the slab-by-slab generator since round 3: 1.2 s instead of 6 s at this size and a tenth of the temporaries; the same bytes --
    # tests/test_gpu_fullsize.py holds both generators to the recorded SHA-256 of this input; EB_BENCH_ONE_PASS_GENERATOR=1: the old one)
    build = sim.build_ebwt_gpu if os.environ.get("EB_BENCH_ONE_PASS_GENERATOR") == "1" else sim.build_ebwt_gpu_slabbed
    if realistic:
        # reads with substitution errors (and their reverse complements) are substrings of no genome: the reads themselves, one behind
        # the other, are the generator's text (sim.error_sample_gpu; tests/test_gpu_generator.py holds it to the host builder)
        d_reads, d_starts = sim.error_sample_gpu(text, starts, args.read_len, args.seed, args.error_rate_ppm, rc=args.reverse_complements, device=local)
        dev_ebwt = sim.build_ebwt_gpu_slabbed(int(d_reads.data_ptr()), int(d_starts.data_ptr()), args.read_len, n_reads=int(d_starts.numel()), text_len=int(d_reads.numel()), device=local)
        del d_reads, d_starts
        torch.cuda.empty_cache()
    elif args.mode == 1:
        dev_ebwt = build(text, starts, args.read_len, device=local)
    elif args.mode == 2:
        dev_ebwt = sim.build_ebwt_gpu_slabbed(text, s0, args.read_len, device=local)
        dev_ebwt2 = sim.build_ebwt_gpu_slabbed(text, s1, args.read_len, device=local)
    else:
        dev_ebwt = sim.build_ebwt_gpu_slabbed(text, np.concatenate([s0, s1]), args.read_len, n_first=len(s0), want_da=True, device=local)
    n = dev_ebwt.n + (dev_ebwt2.n if dev_ebwt2 is not None else 0)
    t_gen = time.time() - t0
    del text
    if args.mode == 1:
        del starts
        workload = (f"mode 1, {args.reads} x {args.read_len} bp reads, {args.coverage:g}x diploid genome {genome_len} bp, "
                    + (f"{args.error_rate_ppm / 1e4:g} % substitution errors, " if args.error_rate_ppm else "")
                    + (f"+ the reverse complement of every read ({2 * args.reads} reads in all), " if args.reverse_complements else "")
                    + f"n={n} BWT positions (BASELINE.json configs[1] when reads=50000000" + (" and the reads are error-free)" if realistic else ")"))
    else:
        del s0, s1
        workload = (f"mode {args.mode}, two collections of {args.reads // 2} x {args.read_len} bp reads (two individuals, each homozygous for one haplotype of a "
                    f"{genome_len} bp diploid genome), " + ("two BWTs" if args.mode == 2 else "one merged BWT + document array")
                    + f", n={n} BWT positions (BASELINE.json " + ("configs[2] when reads=50000000)" if args.mode == 2 else "configs[3] when reads=100000000, on the GPUs of this run)"))
    if rank == 0:
        log(f"[bench] workload: {workload}; haplotypes+read starts {t_host:.1f}s on host, eBWT built on GPU in {t_gen:.1f}s")

    # The generator has just given ~100 GB of temporaries back, and the driver wipes freed HBM in the background: the next large
    # hipMalloc of the process waits for that wipe ("HBM allocation" -- up to 2.3 s of the traversal's
    # "wall_seconds" in round 2 and early round 3, none of it work of the path, and nothing a run on a file would see).  One
    # allocation made HERE takes the wait; it is kept until the traversal is over so that its own release costs nothing there.
    t1 = time.time()
    wipe_absorber = torch.empty(8 << 30, dtype=torch.uint8, device=f"cuda:{local}")
    torch.cuda.synchronize()
    t_wipe = time.time() - t1

    # ---------------- resident state: index + flag vectors on this GPU (phases 1-3, outside the timed region)
    p = api.make_params()
    t1 = time.time()
    bwt = api.DnaBwt(None, device_ptr=dev_ebwt.ptr, n=dev_ebwt.n, device=local)
    bwt2 = api.DnaBwt(None, device_ptr=dev_ebwt2.ptr, n=dev_ebwt2.n, device=local) if dev_ebwt2 is not None else None
    t_index = time.time() - t1
    # a few thousand symbols for the rank spot check, then the ASCII text leaves HBM
    probe_len = min(bwt.n, 1 << 20)
    head = np.empty(probe_len, dtype=np.uint8)
    sim._hip_copy_to_host(head, dev_ebwt.ptr, probe_len)
    # The 7.55 GB ASCII text is released only AFTER phases 2-3: a large hipFree is followed by a driver-side wipe and the next
    # large hipMalloc of the process waits for it ("HBM allocation") -- round 2 freed it here and the
    # traversal's first allocations then sat behind the wipe for ~2 s of "phases 2-3" wall time that was no work of the path.
    # EB_BENCH_FREE_EARLY=1: the round-2 order.
    free_early = os.environ.get("EB_BENCH_FREE_EARLY") == "1"
    if free_early and args.mode == 1:
        dev_ebwt.free()
    t1 = time.time()
    # N > 1: the traversal itself is split over the ranks (Weiner-link subtrees) and the bit vectors are combined by
    # one all-reduce each; N = 1: the plain build.
    flags = (ebd.navigate_sharded(bwt, p, bwt2, device=comm_dev) if use_dist else
             api.navigate_two_bwts(bwt, bwt2, p) if bwt2 is not None else api.navigate_one_bwt(bwt, p))
    if use_dist:
        dist.barrier()
    t_nav = time.time() - t1
    t_da = None
    if args.mode == 3:
        # the document array is the generator's, still in HBM: packed where it lies (a run of the tool reads its -d file: eb_flags_set_da_file)
        t1 = time.time()
        flags.set_da_device(dev_ebwt.da_ptr, dev_ebwt.n)
        t_da = time.time() - t1
    if not (free_early and args.mode == 1):
        dev_ebwt.free()
    if dev_ebwt2 is not None:
        dev_ebwt2.free()
    del wipe_absorber
    ns = flags.stats
    if rank == 0:
        log(f"[bench] index {t_index*1e3:.1f} ms ({bwt.device_bytes/1e9:.2f} GB); phases 2-3 on GPU {t_nav:.3f}s: {ns.nodes} nodes, {ns.leaves} leaves, "
            f"{ns.rank_queries} rank queries in {ns.batches} launches; traversals {ns.seconds_traversal:.3f}s -> "
            f"{ns.rank_queries/max(ns.seconds_traversal,1e-9)/1e9:.2f} GQ/s in situ (the wall time also covers HBM allocation -- right after the generator "
            f"freed ~100 GB the driver's wipe of that memory can stall a hipMalloc for seconds -- and, for N > 1, the all-reduce of the bit vectors)")
    assert ns.lcp_values == n, "phases 2-3 did not cover every LCP slot"

    begin, end = ebd.shard_range(n, rank, world)
    if args.rehearse_rank_of > 1:
        assert world == 1 and use_dist, "--rehearse-rank-of: one rank with EB_BENCH_FORCE_DIST=1"
        begin, end = ebd.shard_range(n, args.rehearse_rank_of // 2, args.rehearse_rank_of)

    # One step = one phase-4 pass over the whole BWT + hand-over of the cluster records to rank 0's host memory.
    # Steps are software-pipelined, as a caller with a queue of work would run them (run_steps below): passes are begun
    # (eb_scan_clusters_begin) ahead of the one whose records are waited for / gathered, so the PCIe download and the
    # RCCL gather of step i overlap the kernels of the steps behind it.  Exactly `steps` passes start and finish between
    # the two barriers.
    # How the ranks' records reach rank 0's host memory (N > 1, or one rank with EB_BENCH_FORCE_DIST=1):
    #   arena (default, round 5): every rank downloads its own compact wire image over its OWN PCIe link into its slot of a
    #         host arena all ranks of the node map (ebd.NodeArena); rank 0 reads the slots.  No collective on the data path.
    #   rccl  (EB_BENCH_GATHER=rccl; rounds 2-4): sizes by all_gather, images GPU -> GPU over xGMI to rank 0, which downloads
    #         all of them over its one link (35 MB per step at N = 8: longer than a rank's kernels).
    #   host  (EB_BENCH_GATHER=host, and gloo without GPUs of their own): every rank downloads its call set, gather of the blobs.
    gather_mode = os.environ.get("EB_BENCH_GATHER", "arena") if use_dist else "none"
    if os.environ.get("EB_BENCH_HOST_GATHER", "0") == "1":
        gather_mode = "host"
    if gather_mode == "rccl" and comm_dev.type != "cuda":
        gather_mode = "host"
    device_gather = gather_mode == "rccl"
    arena_mode = gather_mode == "arena"
    on_device = device_gather or arena_mode          # the call set stays in HBM, the job packs its wire image

    def begin_step():
        return api.scan_clusters_begin(bwt, flags, p, bwt2=bwt2, begin=begin, end=end, device_only=on_device, wire=on_device, pipelined=not args.no_pipeline)

    arena = None
    if arena_mode:
        # the arena's slots are sized from one pass: twice the largest image over the ranks
        j0 = begin_step()
        j0.wait()
        _, len0 = j0.device_wire()
        j0.finish()
        t = torch.tensor([len0], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        arena = ebd.NodeArena(cap=int(1.5 * t.item()) + (1 << 20), depth=4)

    # collect(job) hands the job's records over and may leave the hand-over in flight; resolve() completes it.
    wait_us = []

    def collect(job):
        if arena_mode:
            t0 = time.perf_counter()
            arena.publish_job(job)               # waits for the job's last kernel, then one DMA over this rank's link
            wait_us.append((time.perf_counter() - t0) * 1e6)
            return job
        if device_gather:
            t0 = time.perf_counter()
            job.wait()
            wait_us.append((time.perf_counter() - t0) * 1e6)
            ptr, length = job.device_wire()      # compact image: rank 0's PCIe link carries every record of the job
            return ebd.gather_device_async(ptr, length, comm_dev, keep=job)
        calls = job.finish()                                             # records now in page-locked host memory
        parts = ebd.gather_packed(calls.blob(), device=comm_dev) if use_dist else None
        return calls, parts

    gathered_events = []   # device_gather: one event per step, "all ranks' images resident in rank 0's HBM"

    def resolve(pending):
        if arena_mode:
            parts = None
            if rank == 0:
                parts = arena.collect()          # every rank's image of this step is in host memory
                arena.release()                  # (nothing reads them before the next slot of that rank is written: the writer below runs after the last step)
            return pending.finish(), parts
        if device_gather:
            parts = pending.result()                                     # rank 0: records of all ranks in page-locked host memory
            if pending.gathered is not None:
                gathered_events.append(pending.gathered)
            return pending.keep.finish(), parts                          # releases the job's device image
        return pending

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    step_done = []     # completion time of every timed step (its records in rank 0's host memory)

    def run_steps(k, record=None):
        last = None

        def done(pending):
            nonlocal last
            last = resolve(pending)
            if record is not None:
                record(last[0].stats)
                step_done.append(time.perf_counter())

        if args.no_pipeline:
            for _ in range(k):
                done(collect(begin_step()))
            return last
        # A caller with a queue of work runs the two halves of a pass on two host threads: this one begins passes
        # (eb_scan_clusters_begin enqueues the whole pass -- scan, ordering, context kernels, wire image -- without waiting for the
        # device), the helper takes each begun pass, waits for its records and hands them over (PCIe download into the host
        # arena / the library's page-locked blob, or size exchange + RCCL gather + rank 0's download) and resolves the previous
        # pass's hand-over.  At most five passes are in flight (one being begun, one being handed over, one awaiting its
        # resolution, two of slack).  The device work of consecutive passes overlaps as the library orders it (eb_scan_job.cpp);
        # the hand-over of pass i runs under the kernels of the passes behind it.  Exactly k passes start and finish inside this call.
        import queue
        import threading
        sys.setswitchinterval(2e-5)   # the two threads hand the interpreter lock over within microseconds, not CPython's default 5 ms
        begun_q = queue.Queue()
        in_flight = threading.Semaphore(int(os.environ.get("EB_BENCH_IN_FLIGHT", "5")))   # passes begun and not yet handed over
        failure = []

        def helper():
            try:
                torch.cuda.set_device(dev)
                pending = None
                while True:
                    job = begun_q.get()
                    if job is None:
                        break
                    started = (collect_t if trace else collect)(job)
                    if pending is not None:
                        done(pending)
                        in_flight.release()
                    pending = started
                if pending is not None:
                    done(pending)
                    in_flight.release()
            except BaseException as e:           # noqa: BLE001 -- reported by the main thread
                failure.append(e)
                for _ in range(k + 5):
                    in_flight.release()

        trace = os.environ.get("EB_BENCH_TRACE") == "1"   # host-side timing of the two threads (stderr)
        t_begin, t_collect = [], []
        if trace:
            _collect, _begin = collect, begin_step

            def collect_t(job):
                t0 = time.perf_counter()
                r = _collect(job)
                t_collect.append((t0, time.perf_counter()))
                return r

            def begin_t():
                t0 = time.perf_counter()
                r = _begin()
                t_begin.append((t0, time.perf_counter()))
                return r
        th = threading.Thread(target=helper, name="hand-over")
        th.start()
        try:
            for _ in range(k):
                in_flight.acquire()
                if failure:
                    break
                begun_q.put((begin_t if trace else begin_step)())
        finally:
            begun_q.put(None)
            th.join()
        if failure:
            raise failure[0]
        if trace and rank == 0 and len(t_begin) <= 24:
            z = t_begin[0][0]
            log("[bench] host trace, ms since the first begin: begin calls " + " ".join(f"{(a - z) * 1e3:.2f}-{(b - z) * 1e3:.2f}" for a, b in t_begin))
            log("[bench] host trace, ms since the first begin: hand-overs   " + " ".join(f"{(a - z) * 1e3:.2f}-{(b - z) * 1e3:.2f}" for a, b in t_collect)
                + f"; all done {(time.perf_counter() - z) * 1e3:.2f}")
        if trace and len(t_begin) > 8 and rank == 0:
            bd = np.array([y - x for x, y in t_begin]) * 1e3
            cd = np.array([y - x for x, y in t_collect]) * 1e3
            log("[bench] host trace: begin calls longer than twice the median (step: ms): " + ", ".join(f"{i}: {v:.2f}" for i, v in enumerate(bd) if v > 2 * np.median(bd))
                + f" (median {np.median(bd):.2f}); hand-overs: " + ", ".join(f"{i}: {v:.2f}" for i, v in enumerate(cd) if v > 2 * np.median(cd)) + f" (median {np.median(cd):.2f})")
            b = np.array(t_begin[4:]) * 1e6
            c = np.array(t_collect[4:]) * 1e6
            log(f"[bench] host trace, us: begin call {np.mean(b[:, 1] - b[:, 0]):.0f}, between begin calls {np.mean(b[1:, 0] - b[:-1, 1]):.0f}, "
                f"begin period {np.mean(np.diff(b[:, 0])):.0f}; hand-over call {np.mean(c[:, 1] - c[:, 0]):.0f} (of which waiting for the job "
                f"{np.mean(wait_us[-len(c):]) if wait_us else 0:.0f}), period {np.mean(np.diff(c[:, 0])):.0f}")
        return last

    # ---------------- the FL table (round 4): an index that is scanned again and again keeps one 16-byte chunk per 16 occurrences
    # of a letter, so that a step of the right-context walk is one 16-byte read (ebwt2indel.h, eb_index_build_psi).  Phase 4 builds
    # it by itself with the third job on an index -- a one-shot run of the tool never pays for it -- so the first two passes below run
    # without it and are reported as such; then it is built here, timed.  It is part of the resident index the timed region runs on,
    # like the select directory and the scan planes.  EB_PSI=0: no table (the round-3 walk).
    fl_table = one_shot = None
    if rank == 0 or use_dist:
        # the FIRST pass on the index, as a one-shot run of the tool gets it: scan planes built, buffers allocated, blob pinned
        t1 = time.perf_counter()
        c_first = api.scan_clusters(bwt, flags, p, bwt2=bwt2, begin=begin, end=end)
        one_shot_ms = (time.perf_counter() - t1) * 1e3
        st_first = c_first.stats
        del c_first
        st0 = api.scan_clusters(bwt, flags, p, bwt2=bwt2, begin=begin, end=end).stats
        # the same pipelined steps WITHOUT the table (a short timed region of its own, before the table exists; EB_PSI=0 keeps the
        # library from building it with the third job): what the step costs a caller whose index does not get one -- too large for
        # the spare HBM, or scanned once -- and the figure to compare with rounds 1-3
        ms_without = None
        if os.environ.get("EB_PSI") != "0" and not args.no_pipeline:
            psi_env = os.environ.get("EB_PSI")
            os.environ["EB_PSI"] = "0"
            api.config_reload()                  # (the library reads its switches once)
            try:
                run_steps(3)
                sync()
                t1 = time.perf_counter()
                run_steps(10)
                sync()
                ms_without = (time.perf_counter() - t1) / 10 * 1e3
            finally:
                if psi_env is None:
                    del os.environ["EB_PSI"]
                else:
                    os.environ["EB_PSI"] = psi_env
                api.config_reload()
            gathered_events.clear()
        before = {"scan_ms": round(st0.ms_scan / max(1, int(st0.scan_launches)), 4), "extend_ms": round(st0.ms_extend, 4), "right_ms": round(st0.ms_right, 4),
                  "context_span_ms": round(st0.ms_context, 4), "requests_right": int(st0.requests_right), "pass_ms_first_launch_to_records_on_host": round(st0.ms_total, 4),
                  "what": "the second pass on the index, run by itself: no table yet (what a one-shot run of the tool gets)"}
        t1 = time.perf_counter()
        if os.environ.get("EB_PSI") != "0":
            try:
                bwt.build_psi()
            except Exception as e:      # refused (memory, size): the directory select serves
                log(f"[bench] FL table not built: {e}")
        t_psi = time.perf_counter() - t1
        have, psi_bytes, psi_void = bwt.psi()
        one_shot = {"ms": round(one_shot_ms, 3), "device_ms_first_launch_to_records_on_host": round(st_first.ms_total, 4), "scan_ms": round(st_first.ms_scan, 4),
                    "extend_ms": round(st_first.ms_extend, 4), "right_ms": round(st_first.ms_right, 4), "scan_reruns": int(st_first.scan_reruns),
                    "what": "the FIRST phase-4 pass on the index, wall clock of the call: what a one-shot run of the tool pays -- scan planes built if the traversal "
                            "has not left them behind, device temporaries and the page-locked blob allocated, no FL table"}
        fl_table = {"present": have, "bytes": int(psi_bytes), "void_chunks": int(psi_void), "build_ms": round(t_psi * 1e3, 2) if have else None,
                    "pass_before_it": before,
                    "ms_per_step_without_it": round(ms_without, 4) if ms_without is not None else None,
                    "what": "FL table of a repeatedly scanned index: one 16-byte chunk per 16 occurrences of a letter (38-bit position, 15-bit bitmap of the occurrences that do not sit right behind their predecessor, their distances in 75 bits); "
                            "FL = one 16-byte read by one lane instead of a directory read + a block probe by four; built once, outside the timed region"}
        if rank == 0:
            log(f"[bench] FL table: {fl_table['present']}, {psi_bytes/1e9:.2f} GB, {psi_void} void chunks, built in {t_psi*1e3:.1f} ms")

    if args.warmup > 0:
        run_steps(args.warmup)
    gathered_events.clear()
    scan_ms, ext_ms, right_ms, tot_ms, ctx_ms, req_ext, req_right, req_tab, reruns, batch_tiles = [], [], [], [], [], [], [], [], [], []
    scan_launches = 1

    def record(st):
        nonlocal scan_launches
        scan_ms.append(st.ms_scan)
        scan_launches = max(1, int(st.scan_launches))
        ext_ms.append(st.ms_extend)
        right_ms.append(st.ms_right)
        tot_ms.append(st.ms_total)
        ctx_ms.append(st.ms_context)
        req_ext.append(int(st.requests_extend))
        req_right.append(int(st.requests_right))
        req_tab.append(int(st.requests_right_table))
        reruns.append(int(st.scan_reruns))
        batch_tiles.append(int(st.batch_tiles))

    sync()
    if os.environ.get("EB_BENCH_TRACE") == "1":
        log("[bench] timed region starts")
    t0 = time.perf_counter()
    calls, parts = run_steps(args.steps, record)
    sync()
    elapsed = time.perf_counter() - t0
    if arena_mode and parts is not None:
        parts = [bytes(v) for v in parts]    # (views into the arena)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    if os.environ.get("EB_BENCH_TRACE") == "1" and rank == 0:
        log("[bench] per pass, ms (HIP events): scan " + " ".join(f"{v:.2f}" for v in scan_ms) + " | extend " + " ".join(f"{v:.2f}" for v in ext_ms)
            + " | right " + " ".join(f"{v:.2f}" for v in right_ms) + " | begin..end " + " ".join(f"{v:.2f}" for v in tot_ms))
    # The same steps under the other delivery contract: a step counts as delivered when every rank's records are
    # resident in rank 0's HBM -- rank 0's PCIe download (35 MB per step at N = 8: ~0.65 ms, more than a rank's kernels) then runs
    # under the following steps and is no longer in the step.  Cadence of the "gathered" events on rank 0's stream.
    ms_per_step_hbm = None
    if len(gathered_events) > 2:
        ms_per_step_hbm = gathered_events[0].elapsed_time(gathered_events[-1]) / (len(gathered_events) - 1)
    # spread over the timed steps: intervals between consecutive completions (steps are pipelined, so a step's own duration is
    # not observable from outside; in steady state the interval IS the step time); the first interval holds the pipeline fill
    # (single intervals are no use: the hand-over thread often resolves two passes back to back; windows of four steps are)
    td = np.array([t0] + step_done)
    W = 4 if len(td) > 8 else 1
    win = (td[W:] - td[:-W])[1:] / W * 1e3 if len(td) > W + 1 else np.diff(td) * 1e3
    step_spread = {"ms_per_step_min": round(float(win.min()), 4), "ms_per_step_median": round(float(np.median(win)), 4),
                   "ms_per_step_max": round(float(win.max()), 4), "window_steps": W,
                   **({"windows": [round(float(v), 3) for v in win[::W]]} if os.environ.get("EB_BENCH_TRACE_WINDOWS") == "1" else {}),
                   "what": "time per step over sliding windows of `window_steps` consecutive step completions on this rank (the window holding the pipeline fill excluded)"} if len(win) else None
    value = (end - begin if args.rehearse_rank_of > 1 else n) * args.steps / elapsed

    # ---------------- dominant kernel of the step: k_scan (HIP events on the stream it is launched on, inside the library).
    # A step launches it once per sub-shard; ms_scan is the sum of those launches, so the per-launch average covers
    # shard/launches positions.
    shard_positions = end - begin
    scan_sum_ms = float(np.mean(scan_ms))
    scan_avg_ms = scan_sum_ms / scan_launches
    positions_per_launch = shard_positions / scan_launches
    ALG_B = 0.75 if args.mode == 1 else 0.875   # 3-bit symbol + 2 LCP-threshold bits + 1 minimum bit (+ 1 document bit, modes 2 and 3)
    scan_gbs = ALG_B * positions_per_launch / (scan_avg_ms * 1e-3) / 1e9 if scan_avg_ms > 0 else 0.0

    # ---------------- the same kernels with the chip to themselves.  In the timed region the scan of pass i+1 shares every CU with
    # the context kernels of pass i (eb_scan.hip, "the scan beside the previous job's context kernels"): each kernel then takes
    # longer than alone while the passes take less.  `roofline` / `context_roofline` report the timed region, as the contract
    # says; their "alone" entries come from a few passes run one at a time right here (same inputs, same process, HIP events on
    # the kernels' streams), which is the figure to compare with earlier rounds and with the kernel's own roofline.
    alone = None
    if rank == 0 or use_dist:
        a_scan, a_ext, a_right, a_ctx, a_tot = [], [], [], [], []
        for k in range(7):
            st1 = api.scan_clusters(bwt, flags, p, bwt2=bwt2, begin=begin, end=end).stats
            if k >= 2:
                a_scan.append(st1.ms_scan / max(1, int(st1.scan_launches))); a_ext.append(st1.ms_extend); a_right.append(st1.ms_right)
                a_ctx.append(st1.ms_context); a_tot.append(st1.ms_total)
        alone = {"passes": len(a_scan), "scan_ms": round(float(np.mean(a_scan)), 4), "extend_ms": round(float(np.mean(a_ext)), 4), "right_ms": round(float(np.mean(a_right)), 4),
                 "context_span_ms": round(float(np.mean(a_ctx)), 4), "pass_ms_first_launch_to_records_on_host": round(float(np.mean(a_tot)), 4)}

    # ---------------- rank micro-benchmark (rank 0): uniform random positions, 64 algorithmic bytes per query
    rank_info = None
    out = {}
    if rank == 0:
        m = args.rank_queries
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        d_pos = torch.randint(0, bwt.n + 1, (m,), dtype=torch.int64, device=dev, generator=g)
        d_out = torch.empty((m, 4), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):
            bwt.rank_device(d_pos.data_ptr(), m, d_out.data_ptr())
        ms = [bwt.rank_device(d_pos.data_ptr(), m, d_out.data_ptr()) for _ in range(5)]
        rank_ms = float(np.mean(ms))
        gqs = m / (rank_ms * 1e-3) / 1e9
        # spot-check against the definition on the first 2^20 symbols (host copy)
        chk = np.arange(0, probe_len + 1, max(1, probe_len // 4096), dtype=np.uint64)
        got = bwt.parallel_rank(chk)
        pref = np.stack([np.concatenate([[0], np.cumsum(head == c)]) for c in b"ACGT"], axis=1)
        assert np.array_equal(got, pref[chk.astype(np.int64)]), "rank kernel disagrees with the definition"
        # Beside SURVEY 8(d)'s accounting (64 B per query against the 8 TB/s peak): what the memory system allows.  Every block read is
        # served as a 128-byte line (PMC, profiles/r02_pmc_traffic.json), and the chip streams 6.29 TB/s at best (MI355X guide,
        # float4 copy; scripts/stream_probe.hip: 6.2-6.3): frac_of_line_bound counts the lines alone (<= 49.1 GQ/s), and
        # frac_of_traffic_bound everything a query moves through HBM: the line + its 8-byte position + its 32-byte result = 168 B
        # (<= 37.4 GQ/s for this ABI, whatever the kernel).
        STREAM = 6290.0
        rank_info = {"bound": "hbm", "achieved": round(64 * gqs, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(64 * gqs / 8000.0, 4),
                     "traffic": None, "gq_per_s": round(gqs, 3), "queries": m, "ms": round(rank_ms, 4),
                     "index_bytes": int(bwt.device_bytes),
                     "frac_of_line_bound": round(128 * gqs / STREAM, 4), "frac_of_traffic_bound": round(168 * gqs / STREAM, 4),
                     "line_bound_gq_per_s": round(STREAM / 128, 2), "traffic_bound_gq_per_s": round(STREAM / 168, 2), "measured_stream_gb_per_s": STREAM}
        # the same kernel on the query pattern the traversal issues: the six boundaries of a deep suffix-tree node lie within
        # a few dozen rows, so they share blocks (nodes in random order; SURVEY.md 8(d) metric 2 (ii) in isolation)
        m_nodes = min(1 << 24, m // 6)
        base_pos = torch.randint(0, max(bwt.n - 64, 1), (m_nodes,), dtype=torch.int64, device=dev, generator=g)
        node_pos = (base_pos[:, None] + torch.tensor([0, 5, 11, 17, 23, 30], dtype=torch.int64, device=dev)[None, :]).reshape(-1).contiguous()
        torch.cuda.synchronize()
        bwt.rank_device(node_pos.data_ptr(), m_nodes * 6, d_out.data_ptr())
        node_ms = float(np.mean([bwt.rank_device(node_pos.data_ptr(), m_nodes * 6, d_out.data_ptr()) for _ in range(3)]))
        node_gqs = m_nodes * 6 / (node_ms * 1e-3) / 1e9
        rank_info["node_pattern"] = {"what": "six queries within 30 rows per node, nodes uniformly random", "nodes": m_nodes, "ms": round(node_ms, 4),
                                     "gq_per_s": round(node_gqs, 2), "achieved": round(64 * node_gqs, 1), "frac": round(64 * node_gqs / 8000.0, 4)}
        del d_pos, d_out, base_pos, node_pos

        # ---------------- output of the timed path -> .snp (host formatting, outside the metric)
        if parts is None:
            parts = [bytes(calls.blob())]
        text, cluster_nr, events, total_recs = ebd.write_gathered(parts, p, args.mode != 1)
        out["snp_sha256"] = hashlib.sha256(text).hexdigest()
        out["snp_records"] = text.count(b">")
        out["clusters_emitted"] = total_recs
        # The default workload's .snp is known: scripts/fullsize_check.py config2 walked this very eBWT (7.55 G positions) through the
        # CPU oracle on the GPU box's host threads -- flag vectors and .snp identical to the GPU's -- and left the SHA-256
        # under profiles/.  Every default run is held to it (any number of GPUs: the shards' call sets concatenate to the same text).
        out["snp_verified"] = None
        record = ("r03_fullsize_config3.json" if args.mode == 2 else "r03_fullsize_config4.json" if args.mode == 3 else
                  "r03_fullsize_config2.json" if not realistic else
                  "r05_fullsize_config2err.json" if (args.error_rate_ppm, args.reverse_complements) == (10_000, False) else
                  "r05_fullsize_config2rc.json" if (args.error_rate_ppm, args.reverse_complements) == (10_000, True) else "none")
        try:
            with open(os.path.join(ROOT, "profiles", record)) as fh:
                known = json.load(fh)
        except OSError:
            known = None
        default_input = (args.read_len, args.coverage, args.seed) == (150, 30.0, 1)
        if args.rehearse_rank_of <= 1 and known is not None and default_input and known["input"]["mode"] == args.mode and known["input"]["reads_per_collection"] == (args.reads if args.mode == 1 else args.reads // 2) and sum(known["input"]["n"]) == int(n):
            if out["snp_sha256"] != known["oracle_snp_sha256"] or out["snp_records"] != known["snp_records"]:
                raise SystemExit(f"[bench] .snp of the timed path differs from the full-size CPU oracle run (profiles/{record}): "
                                 f"{out['snp_sha256']} vs {known['oracle_snp_sha256']}")
            out["snp_verified"] = (f"sha256 equals the CPU oracle's whole-run .snp of this very input at full size (profiles/{record}; "
                                   "tests/test_gpu_fullsize.py also holds the input's SHA-256 and every bit vector to that record)")

    # ---------------- CPU baseline on a bounded sample of the same workload family (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample_reads > 0:
        from tests import _oracle
        orc = _oracle.Oracle()
        s_reads = args.cpu_sample_reads
        s_genome = int(s_reads * args.read_len / args.coverage)
        from concurrent.futures import ThreadPoolExecutor
        po = _oracle.Params.from_args([])
        se2 = s_da = sidx2 = None
        if args.mode == 1:
            sr, _, _ = sim.diploid_sample(s_genome, s_reads, args.read_len, seed=args.seed + 1000)
            se = sim.build_ebwt(sr, threads=threads)
            del sr
        else:
            r0, r1 = sim.two_samples(s_genome, s_reads // 2, args.read_len, seed=args.seed + 1000)
            if args.mode == 2:
                se, se2 = sim.build_ebwt(r0, threads=threads), sim.build_ebwt(r1, threads=threads)
            else:
                se, s_da = sim.build_merged_ebwt(r0, r1, threads=threads)
            del r0, r1
        # single thread, the reference's algorithm: index, phases 2-3, phase 4
        n_s = len(se) + (len(se2) if se2 is not None else 0)
        sidx = orc.index(se)                        # keep the handles alive across the calls
        if se2 is not None:
            sidx2 = orc.index(se2)
        t0 = time.perf_counter()
        of = _oracle.OracleFlags(orc, sidx, sidx2, po.K, po.k_right, s_da)
        t_nav_cpu = time.perf_counter() - t0
        t0 = time.perf_counter()
        res = of.scan(po, 0, n_s)
        t_scan_cpu = time.perf_counter() - t0
        want = res["text"]
        got = (api.run_one_dataset(se, p, device=local) if args.mode == 1 else api.run_two_datasets(se, se2, p, device=local) if args.mode == 2
               else api.run_two_datasets_da(se, s_da, p, device=local))
        assert got == want, "GPU .snp differs from the CPU oracle on the baseline sample"
        rt = orc.lib.orc_rank_bench(sidx.h, 20_000_000, 7, None)
        # the same scan sharded over the host's cores (the oracle's scan is re-entrant; ctypes releases the GIL)
        cores = max(1, min(os.cpu_count() or 1, 64))
        cuts = [ebd.shard_range(n_s, i, cores) for i in range(cores)]
        with ThreadPoolExecutor(cores) as ex:
            t0 = time.perf_counter()
            shards = list(ex.map(lambda be: of.scan(po, be[0], be[1])["events"], cuts))
            t_scan_all = time.perf_counter() - t0
        assert sum(shards) == res["events"], "sharded CPU scan disagrees with the whole-range scan"
        # the unmodified reference itself (oracle/_ref, built from the upstream sources in the build container) on the same
        # file: end to end only -- it prints no phase timings -- and its .snp must equal the GPU's byte for byte
        reference = None
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "ebwt2InDel_ref")
        if os.path.exists(ref_bin):
            import subprocess
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                pth, outp = os.path.join(td, "sample.ebwt"), os.path.join(td, "ref.snp")
                se.tofile(pth)
                more = []
                if se2 is not None:
                    se2.tofile(pth + "2")
                    more = ["-2", pth + "2"]
                elif s_da is not None:
                    np.asarray(s_da).tofile(pth + ".da")
                    more = ["-d", pth + ".da"]
                t0 = time.perf_counter()
                rr = subprocess.run([ref_bin, "-1", pth] + more + ["-o", outp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
                t_ref = time.perf_counter() - t0
                with open(outp, "rb") as fh:
                    ref_snp = fh.read()
            assert rr.returncode == 0 and ref_snp == got, "GPU .snp differs from the reference binary's on the baseline sample"
            reference = {"binary": "oracle/_ref/ebwt2InDel_ref (g++ -Ofast build of the unmodified upstream sources)", "cores": 1,
                         "e2e_seconds": round(t_ref, 2), "e2e_positions_per_s": round(n_s / t_ref, 1), "snp_identical_to_gpu": True}
        # ---- the same on a DRAM-resident sample (>= 1 G positions: index 0.5 GB + flag vectors 0.4 GB, far beyond the host's L3):
        # the 60 M-position sample above lives in cache and flatters the CPU (36-47 M positions/s for the
        # phase-4 loop once the index is DRAM-resident).  The sample is built by the GPU generator (the host suffix sorter
        # would take minutes), the oracle's index and traversal run on all host threads, its scan sharded over them; ONE
        # thread then scans a 128 M-position sub-range of that index: the single-core figure `value` reports.
        dram = None
        if args.cpu_dram_reads > 0 and args.mode == 1:
            from tests import _fullsize as fsz
            b_reads = args.cpu_dram_reads
            b_glen = int(b_reads * args.read_len / args.coverage)
            bg = sim.random_genome(b_glen, args.seed + 2000)
            bh = sim.mutate_haplotype(bg, args.seed + 2000)
            btext, boffs = sim.haplotype_text([bg, bh])
            bstarts = sim.sample_read_starts([bg, bh], boffs, b_reads, args.read_len, args.seed + 2000)
            del bg, bh
            bdev = sim.build_ebwt_gpu(btext, bstarts, args.read_len, device=local)
            del btext, bstarts
            bhost, _ = bdev.to_host()
            bb = api.DnaBwt(None, device_ptr=bdev.ptr, n=bdev.n, device=local)
            bdev.free()
            bfl = api.navigate_one_bwt(bb, p)
            b_got = api.scan_clusters(bb, bfl, p).format_snp(p, False)[0]
            del bfl, bb
            n_b = len(bhost)
            t0 = time.perf_counter()
            bidx = orc.index(bhost, threads=cores)
            t_bidx = time.perf_counter() - t0
            del bhost
            t0 = time.perf_counter()
            bof = _oracle.OracleFlags(orc, bidx, None, po.K, po.k_right, threads=cores)
            t_bnav = time.perf_counter() - t0
            bcuts = [ebd.shard_range(n_b, i, cores * 4) for i in range(cores * 4)]
            with ThreadPoolExecutor(cores) as ex:
                t0 = time.perf_counter()
                bparts = list(ex.map(lambda be: bof.scan(po, be[0], be[1]), bcuts))
                t_bscan = time.perf_counter() - t0
            b_want = fsz.renumber([r["text"] for r in bparts], [r["cluster_nr"] - 1 for r in bparts])
            assert b_want == b_got, "GPU .snp differs from the CPU oracle on the DRAM-resident baseline sample"
            sub = (n_b // 2 // 64 * 64, min(n_b, n_b // 2 // 64 * 64 + (1 << 27)))
            t0 = time.perf_counter()
            bof.scan(po, sub[0], sub[1])
            t_b1 = time.perf_counter() - t0
            brt = orc.lib.orc_rank_bench(bidx.h, 20_000_000, 7, None)
            dram = {"sample": f"{b_reads} x {args.read_len} bp reads (n={n_b}), same generator, built on the GPU; oracle index + flag vectors {n_b * 7 // 8 / 1e9:.2f} GB",
                    "single_thread_phase4_positions_per_s": round((sub[1] - sub[0]) / t_b1, 1), "single_thread_subrange": [int(sub[0]), int(sub[1])],
                    "single_thread_seconds": round(t_b1, 2), "rank_queries_per_s": round(20_000_000 / brt, 1),
                    "all_cores": {"cores": cores, "index_seconds": round(t_bidx, 2), "phases23_seconds": round(t_bnav, 2), "phase4_seconds": round(t_bscan, 3),
                                  "phase4_positions_per_s": round(n_b / t_bscan, 1), "e2e_positions_per_s": round(n_b / (t_bnav + t_bscan), 1)},
                    "snp_identical_to_gpu": True, "snp_records": b_got.count(b">")}
            del bof, bidx
        cpu = {"value": round(n_s / t_scan_cpu, 1), "unit": "positions/s", "cores": 1, "kind": "port",
               "sample": f"{s_reads} x {args.read_len} bp reads (n={n_s}), same generator; phase-4 scan only ({t_scan_cpu:.3f}s); "
                         f"phases 2-3 took {t_nav_cpu:.2f}s",
               "e2e_positions_per_s": round(n_s / (t_nav_cpu + t_scan_cpu), 1),
               "rank_queries_per_s": round(20_000_000 / rt, 1), "snp_identical_to_gpu": True,
               "all_cores": {"value": round(n_s / t_scan_all, 1), "unit": "positions/s", "cores": cores,
                             "what": "the same phase-4 scan, range sharded over host threads"},
               "reference": reference}
        if dram is not None:
            # the headline figure is the DRAM-resident one; the cache-resident sample keeps its own keys
            cpu["cache_resident_value"] = cpu["value"]
            cpu["cache_resident_sample"] = cpu["sample"]
            cpu["value"] = dram["single_thread_phase4_positions_per_s"]
            cpu["sample"] = (f"phase-4 scan of a {sub[1] - sub[0]}-position sub-range by ONE thread ({dram['single_thread_seconds']} s) inside a DRAM-resident sample: " + dram["sample"]
                             + f"; the 60 M-position (cache-resident) sample of round 2 is under cache_resident_* / e2e_* / reference")
            cpu["dram_resident"] = dram

    # HBM traffic of the kernels from the PMC passes committed under profiles/ (counters cannot be collected from inside this
    # process): bytes per launch from the request-size split -- TCC_EA0_RDREQ_128B x 128 + TCC_EA0_RDREQ_64B x 64 + WRITE_SIZE
    # (scripts/profile_round.sh, scripts/pmc_summary.py) -- divided by the launch time measured HERE; used only when the
    # profile was taken on the same launch size.  GB/s, the unit of `achieved`.
    traffic = ctx_traffic = rank_traffic = None
    try:
        pmc_path = next(q for q in (os.path.join(ROOT, "profiles", f"r{r:02d}_pmc_traffic.json") for r in (5, 4, 3, 2)) if os.path.exists(q))
        with open(pmc_path) as fh:
            pmc = json.load(fh)
        same = args.mode == 1 and abs(pmc["workload_positions"] - positions_per_launch) <= 0.01 * positions_per_launch
        if same and scan_avg_ms > 0:
            traffic = round(pmc["kernels"]["k_scan<1>"]["hbm_bytes_per_launch_by_request_size"] / (scan_avg_ms * 1e-3) / 1e9, 2)
        if same and ctx_ms and np.mean(ctx_ms) > 0:
            b = pmc["kernels"]["k_extend"]["hbm_bytes_per_launch_by_request_size"] + pmc["kernels"]["k_right"]["hbm_bytes_per_launch_by_request_size"]
            ctx_traffic = round(b / (float(np.mean(ctx_ms)) * 1e-3) / 1e9, 2)
        if rank_info is not None and abs(pmc["workload_positions"] - n) <= 0.01 * n and pmc["rank_kernel"]["queries"] == rank_info["queries"]:
            rank_traffic = round(pmc["rank_kernel"]["hbm_bytes_per_launch_by_request_size"] / (rank_info["ms"] * 1e-3) / 1e9, 2)
            rank_info["traffic"] = rank_traffic
            rank_info["traffic_note"] = ("every fabric read of this kernel is a 128-byte request (PMC: TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ): a query moves "
                                         "128 B (line of its 64-byte block) + 8 B (position) + 32 B (result) = 168 B of HBM traffic, so 64 B/query at "
                                         "the chip's ~6 TB/s streaming limit caps at ~36 GQ/s = 0.29 of this roofline for uniformly random queries")
    except Exception:
        pass

    # algorithmic bytes the context kernels ask for: a 64-byte block per k_extend request; per k_right request a 16-byte chunk of the
    # FL table, or 64 bytes (directory line / block) without it
    # (the walk's requests are counted by kind on the device: a void chunk of the table costs its 16-byte read AND the directory
    # select's 64-byte requests -- round-4 advisor finding)
    ctx_bytes = float(np.mean(req_ext)) * 64 + float(np.mean(req_tab)) * 16 + (float(np.mean(req_right)) - float(np.mean(req_tab))) * 64
    if fl_table and fl_table.get("present") and fl_table.get("ms_per_step_without_it") and fl_table["ms_per_step_without_it"] > ms_per_step:
        fl_table["break_even_passes"] = round(fl_table["build_ms"] / (fl_table["ms_per_step_without_it"] - ms_per_step), 1)
        fl_table["break_even_note"] = "passes after which building the table has paid for itself: build_ms / (ms_per_step_without_it - ms_per_step)"
    if rank == 0:
        line = {
            **({"rehearsal": f"ONE rank's share of a {args.rehearse_rank_of}-GPU run on one GPU (positions [{begin}, {end}), collectives with a single rank): "
                             "value and step_roofline count this share only; not a benchmark result"} if args.rehearse_rank_of > 1 else {}),
            "metric": "BWT positions scanned/sec (whole node)", "value": round(value, 1), "unit": "positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "step_spread": step_spread,
            **({"ms_per_step_records_in_rank0_hbm": round(ms_per_step_hbm, 4),
                "ms_per_step_note": "ms_per_step: records of all ranks in rank 0's HOST memory (the contract `value` is quoted on); ms_per_step_records_in_rank0_hbm: "
                                    "the same steps counted as delivered once the gathered call sets are resident in rank 0's HBM (cadence of the end of the gather on "
                                    "rank 0's stream; its PCIe download then overlaps the following steps)"} if ms_per_step_hbm is not None else {}),
            "config": {"workload": workload, "mode": args.mode, "options": "-L31 -R30 -k16 -g10 -v2 -m3 -c20 -q0",
                       # every path-selecting switch of the library / this script that is set in the environment (none: the defaults ran)
                       "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("EB_", "EBGEN_")) or k in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY")},
                       "records_to_rank0": {"none": "one rank: the library downloads the call set into page-locked host memory", "arena": "every rank downloads its own compact wire image over its own PCIe link into a host arena all ranks map; rank 0 reads the slots",
                                            "rccl": "sizes by all_gather, wire images GPU -> GPU over xGMI (RCCL gather), rank 0 downloads all of them", "host": "every rank downloads its call set; gather of the blobs"}[gather_mode],
                       "parallelism": f"scan range split over {world} GPU(s), BWT+flags replicated, cluster records of all ranks delivered to rank 0's host memory; " + ("steps back to back" if args.no_pipeline else "steps pipelined (one host thread begins passes, a second hands their records over; at most five in flight)")},
            "roofline": {"bound": "hbm", "achieved": round(scan_gbs, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(scan_gbs / 8000.0, 5), "traffic": traffic, "kernel": "k_scan",
                         "avg_launch_ms": round(scan_avg_ms, 4), "launches_per_step": scan_launches, "positions_per_launch": int(positions_per_launch),
                         "algorithmic_bytes_per_position": ALG_B,
                         "measured": "timed region" + ("" if args.no_pipeline else ": the launch shares every CU with the previous pass's context kernels (the scan holds 8 workgroup slots per CU "
                                                       "with the FL table, 4 without), so it lasts longer than alone -- see `alone`"),
                         **({"alone": {"avg_launch_ms": alone["scan_ms"], "achieved": round(ALG_B * positions_per_launch / (alone["scan_ms"] * 1e-3) / 1e9, 2),
                                       "frac": round(ALG_B * positions_per_launch / (alone["scan_ms"] * 1e-3) / 1e9 / 8000.0, 5),
                                       "what": f"the same launch with the chip to itself: {alone['passes']} passes run one at a time after the timed region"}} if alone else {})},
            # the whole step against the scan's roofline (SURVEY.md 8(d) metric 1: 0.75 B x positions / step time): what `value` is worth
            "step_roofline": {"bound": "hbm", "achieved": round(ALG_B * value / 1e9, 2), "peak": 8000.0 * world, "unit": "GB/s",
                              "frac": round(ALG_B * value / 1e9 / (8000.0 * world), 5),
                              "what": f"{ALG_B} algorithmic bytes x all positions / ms_per_step, against the HBM peak of the GPUs used: scan + "
                                      "survivor ordering + both context kernels + hand-over of the records"},
            # the kernels that dominate the step: k_extend, then k_right (a large call set runs them one behind the other; a small one
            # side by side on two streams), dependent random reads.  Requests
            # are counted on the device where they are issued (eb_scan_stats.requests_*), 64 bytes each, over first start .. last end.
            "context_roofline": {"bound": "hbm", "achieved": round(ctx_bytes / (max(np.mean(ctx_ms), 1e-9) * 1e-3) / 1e9, 2),
                                 "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(ctx_bytes / (max(np.mean(ctx_ms), 1e-9) * 1e-3) / 1e9 / 8000.0, 5),
                                 "traffic": ctx_traffic, "kernel": "k_extend, then k_right", "ms": round(float(np.mean(ctx_ms)), 4),
                                 "requests_extend": int(np.mean(req_ext)), "requests_right": int(np.mean(req_right)), "requests_right_table_chunks": int(np.mean(req_tab)),
                                 "bytes_per_request": {"k_extend": 64, "k_right": {"table_chunk": 16, "directory_read_or_block": 64}}, "requests_per_s": round((np.mean(req_ext) + np.mean(req_right)) / (max(np.mean(ctx_ms), 1e-9) * 1e-3), 1),
                                 "what": "rank 0's shard, timed region (the next pass's scan shares the chip while these run); k_extend: one request per 64-byte block of a rank "
                                         "pair (1-2 per LF step); k_right: one per FL-table chunk (without the table: one per select-directory read + one per block probed); the chip serves ~47 G independent random 64-byte "
                                         "reads per second, each as a 128-byte line (profiles/r02_rank_probe.txt, r02_rank_probe_cache_policy.txt, r02_pmc_traffic.json)",
                                 **({"alone": {"ms": alone["context_span_ms"],
                                               "achieved": round(ctx_bytes / (alone["context_span_ms"] * 1e-3) / 1e9, 2),
                                               "frac": round(ctx_bytes / (alone["context_span_ms"] * 1e-3) / 1e9 / 8000.0, 5),
                                               "requests_per_s": round((np.mean(req_ext) + np.mean(req_right)) / (alone["context_span_ms"] * 1e-3), 1)}} if alone else {})},
            "rank_roofline": rank_info,
            "one_shot": one_shot,
            "fl_table": fl_table,
            "cpu_baseline": cpu,
            "phase4_device_ms": {"scan": round(scan_sum_ms, 4), "extend": round(float(np.mean(ext_ms)), 4),
                                 "right": round(float(np.mean(right_ms)), 4), "context_span": round(float(np.mean(ctx_ms)), 4),
                                 "total": round(float(np.mean(tot_ms)), 4), "what": "per kernel in the timed region (passes overlap: kernels share the chip)",
                                 "scan_reruns": int(sum(reruns)), "batch_tiles": int(batch_tiles[-1]) if batch_tiles else None,
                                 "alone": alone},
            # seconds: the two traversals (leaves and nodes side by side) + packing of the bit vectors, first launch to last result;
            # wall_seconds: the whole call, i.e. also HBM allocation (which, right behind the generator's 100 GB of frees, can wait
            # seconds for the driver's wipe) and, for N > 1, the all-reduce of the vectors
            "phases23_gpu": {"seconds": round(float(ns.seconds_traversal), 4), "wall_seconds": round(t_nav, 4), "waited_before_for_the_generators_freed_hbm_seconds": round(t_wipe, 3), "traversal_shards": world if use_dist else 1,
                             **({"document_array_packed_from_hbm_seconds": round(t_da, 4)} if t_da is not None else {}),
                             "seconds_leaves": round(float(ns.seconds_leaves), 4), "seconds_nodes": round(float(ns.seconds_nodes), 4),
                             "rank_gq_per_s": round(ns.rank_queries / max(ns.seconds_traversal, 1e-9) / 1e9, 2), "nodes": int(ns.nodes), "leaves": int(ns.leaves), "rank_queries": int(ns.rank_queries),
                             "launches": int(ns.batches),
                             # end to end = the WHOLE phases 2-3 call (allocations, and for N > 1 the all-reduce + popcount check + finalize) + one phase-4 pass;
                             # the traversal-only rate has its own key (round-2 advisor finding: the two had been mixed)
                             "e2e_positions_per_s": round(n / (t_nav + ms_per_step * 1e-3), 1),
                             "traversal_only_positions_per_s": round(n / max(float(ns.seconds_traversal), 1e-9), 1)},
        }
        line.update(out)
        if args.mode == 1 and world == 1 and not use_dist and not args.no_other_modes and not realistic and args.rehearse_rank_of <= 1:
            line["other_modes"] = other_modes(args)
            line["realistic_reads"] = realistic_reads(args)
        print(json.dumps(line), flush=True)
    if arena is not None:
        arena.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

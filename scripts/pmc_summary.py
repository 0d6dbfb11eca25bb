"""rocprofv3 counter tables (scripts/profile_round.sh) -> profiles/<tag>_pmc_traffic.json: mean FETCH_SIZE / WRITE_SIZE per
launch of the product's kernels, with the gfx950 correction the MI355X guide prescribes for wide coalesced reads."""
import csv, glob, json, os, re, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
# run on the GPU box by scripts/profile_round.sh (the per-dispatch counter tables are too large to travel back):
# the summary goes next to them, under gpurun_out/, and is copied to profiles/ by hand
dst_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles")
KERNELS = {"k_scan<1>": r"k_scan<1",   # k_scan<1, false> / <1, true> since round 3 (second parameter: tile loads one tile ahead)
            "k_extend": r"k_extend", "k_right": r"k_right", "k_nodes1c": r"k_nodes1c", "k_nodes1p": r"k_nodes1p", "k_leaves1c": r"k_leaves1c",
           "k_leaves<1>": r"k_leaves<1>", "k_rank4_batch": r"k_rank4_batch", "k_long_runs": r"k_long_runs"}


def table(sub, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "")
                for key, pat in KERNELS.items():
                    if re.search(re.escape(pat), name):
                        acc[key].append(float(row["Counter_Value"]))
                        break
    return acc


fetch, write = table("fetch", "FETCH_SIZE"), table("write", "WRITE_SIZE")
n = 7_550_000_000
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/profile_round.sh) on "
                 "`python3 bench.py --cpu-sample-reads 0 --no-pipeline --steps 5 --warmup 1`, MI355X",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB per launch (mean over launches). gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE "
                "reports 1/2 of the bytes of a wide coalesced streaming read of 16 B per lane; other widths are uncalibrated. k_scan streams "
                "two arrays: the scan planes with 16 B per lane (n/4 bytes, reported at 1/2) and the fused flag words with 8 B per lane "
                "(n/8 bytes, taken as reported): corrected = FETCH_SIZE + n/8 + WRITE_SIZE. That estimate lands within a few percent of the "
                "bytes the two streams hold (streamed_bytes_per_launch), i.e. the kernel reads its inputs once; the 8-B-per-lane width is "
                "uncalibrated, so the last few percent are not significant. The random 64-byte gathers of the other kernels are reported raw.",
       "workload_positions": n, "kernels": {}}
for key in KERNELS:
    if key not in fetch and key not in write:
        continue
    f = fetch.get(key, [])
    w = write.get(key, [])
    if key == "k_scan<1>":   # the first launches of a run are warm-up; all launches process the same range
        pass
    e = {"fetch_kib": sum(f) / len(f) if f else None, "write_kib": sum(w) / len(w) if w else None, "launches_sampled": max(len(f), len(w))}
    if key == "k_scan<1>" and f and w:
        e["hbm_bytes_per_launch_corrected"] = int(e["fetch_kib"] * 1024 + n / 8 + e["write_kib"] * 1024)
        e["streamed_bytes_per_launch"] = int(0.375 * n)   # what the kernel's two input streams hold: planes n/4 + flags n/8
        e["algorithmic_bytes_per_launch"] = int(0.75 * n) # 6 bits per position, whatever the layout
    out["kernels"][key] = e
# ---- the rank kernel alone (scripts/rank_only.py): memory-side read requests of the L2, all sizes and 32-byte ones
def one(sub, counter, kernel="k_rank4_batch"):
    vals = []
    for path in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter and kernel in row.get("Kernel_Name", ""):
                    vals.append(float(row["Counter_Value"]))
    return vals


rank = {}
for sub, counters in (("rdreq", ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum")), ("rdreq2", ("TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")), ("tcc", ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum")),
                      ("rfetch", ("FETCH_SIZE",)), ("rwrite", ("WRITE_SIZE",))):
    for cn in counters:
        v = one(sub, cn)
        if v:
            rank[cn] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
if rank:
    q = 1 << 28
    out["rank_kernel"] = {"what": "k_rank4_batch alone, 2^28 uniformly random queries on a 7.55 G-symbol index (scripts/rank_only.py); per launch",
                          "queries": q, "counters": rank}
    if "TCC_EA0_RDREQ_sum" in rank:
        r = rank["TCC_EA0_RDREQ_sum"]["per_launch_mean"]
        r32 = rank.get("TCC_EA0_RDREQ_32B_sum", {}).get("per_launch_mean", 0.0)
        out["rank_kernel"]["read_requests_per_query"] = r / q
        out["rank_kernel"]["read_requests_32B_per_query"] = r32 / q
        out["rank_kernel"]["reading"] = ("a query reads one random 64-byte block (4 x 16 B by a quad) plus its share of the coalesced position stream "
                                         "(8 B); RDREQ counts the L2's requests to the fabric, RDREQ_32B those of 32 bytes: (RDREQ - RDREQ_32B) of 64 B")
# request sizes of the bench's kernels (streaming scan against the random-access context kernels)
sizes = {}
for kern in ("k_scan<1>", "k_extend", "k_right"):
    e = {}
    for cn in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"):
        v = one("rdreq_bench", cn, KERNELS.get(kern, kern))
        if v:
            e[cn] = sum(v) / len(v)
    if e:
        sizes[kern] = e
if sizes:
    out["read_request_sizes_per_launch"] = sizes
def derive(out):
    """HBM bytes per launch from the request-size split (exact: every request is counted with its size), where it was taken."""
    sizes = out.get("read_request_sizes_per_launch", {})
    for kern, e in sizes.items():
        k = out["kernels"].get(kern)
        if not k:
            continue
        rd = e.get("TCC_EA0_RDREQ_128B_sum", 0.0) * 128 + e.get("TCC_EA0_RDREQ_64B_sum", 0.0) * 64
        wr = (k.get("write_kib") or 0.0) * 1024
        k["hbm_read_bytes_per_launch_by_request_size"] = int(rd)
        k["hbm_bytes_per_launch_by_request_size"] = int(rd + wr)
    rk = out.get("rank_kernel")
    if rk and "TCC_EA0_RDREQ_128B_sum" in rk["counters"]:
        c = rk["counters"]
        rd = c["TCC_EA0_RDREQ_128B_sum"]["per_launch_mean"] * 128 + c.get("TCC_EA0_RDREQ_64B_sum", {"per_launch_mean": 0})["per_launch_mean"] * 64
        wr = c.get("WRITE_SIZE", {"per_launch_mean": 0})["per_launch_mean"] * 1024
        rk["hbm_bytes_per_launch_by_request_size"] = int(rd + wr)
        rk["hbm_bytes_per_query"] = (rd + wr) / rk["queries"]
        rk["finding"] = ("every fabric read request of the kernel is a 128-byte one (TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ, _64B and _32B ~ 0): a random "
                         "64-byte block read moves a whole 128-byte L2 line, whatever cache policy the load carries (profiles/r02_rank_probe_cache_policy.txt)")
    out["units_note"] = ("*_by_request_size: TCC_EA0_RDREQ_128B x 128 B + TCC_EA0_RDREQ_64B x 64 B (+ WRITE_SIZE); FETCH_SIZE = TCC_EA0_RDREQ x 64 B undercounts "
                         "128-byte requests by half (MI355X guide), hence the round-1 style 'corrected' figure kept beside it")
    return out


dst = os.path.join(dst_dir, f"{tag}_pmc_traffic.json")
if len(sys.argv) > 3 and sys.argv[3] == "--derive-only":   # re-derive the byte figures of an existing summary
    out = json.load(open(dst))
out = derive(out)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"].get("k_scan<1>"), indent=1))

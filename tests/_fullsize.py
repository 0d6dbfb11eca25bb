"""The BASELINE configurations at FULL SIZE, shared by the `-m gpu` tests (tests/test_gpu_fullsize.py) and the script
that records what they are held to (scripts/fullsize_check.py).

  config2  BASELINE configs[1]: mode 1, 50 M x 150 bp, 30x diploid, n = 7.55 G        (bench.py's own input, seed 1)
  config3  BASELINE configs[2]: mode 2, two collections of 25 M x 150 bp as TWO BWTs (2 x 3.775 G positions)
  config4  BASELINE configs[3]: mode 3, 2 x 50 M x 150 bp merged BWT + document array (n = 15.1 G)

Inputs are regenerated on the GPU from seeds (ebwt2indel_amd/tools/ebwt_gen.hip) and identified by the SHA-256 of their
ASCII bytes.  What a run of the PRODUCT yields -- nothing here touches the oracle -- is a `Result`: traversal counters,
SHA-256 of each bit vector (thrK, thrR, minima, da) and of the whole-run .snp.  profiles/r03_fullsize_<name>.json holds
the same figures computed by the CPU oracle on the GPU box's host threads (whole range, all modes)."""
import hashlib
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READ_LEN = 150
CONFIGS = {
    "config2": dict(mode=1, reads=50_000_000, baseline="configs[1]: Mode 1: 50M x 150 bp reads (~7.5 Gbp BWT) on one MI355X"),
    "config3": dict(mode=2, reads=25_000_000, baseline="configs[2]: Mode 2: two collections 2 x 25M x 150 bp, dual-BWT LF navigation"),
    "config4": dict(mode=3, reads=50_000_000, baseline="configs[3]: Mode 3: merged BWT + document array, 100M x 150 bp"),
}
# round 5: the reference's documented kind of input at configs[1] size -- reads with 1 % substitution errors, and 25 M such reads
# plus their reverse complements (README.md:88-100); error-free forward-only reads were all that rounds 1-4 had above 0.4 M positions
CONFIGS["config2err"] = dict(mode=1, reads=50_000_000, err_per_million=10_000, rc=False,
                             baseline="configs[1] with sequencing errors: Mode 1: 50M x 150 bp reads, 1 % substitution errors (~7.5 Gbp BWT) on one MI355X")
CONFIGS["config2rc"] = dict(mode=1, reads=25_000_000, err_per_million=10_000, rc=True,
                            baseline="configs[1] with sequencing errors and reverse complements: Mode 1: 25M x 150 bp reads, 1 % substitution errors, + their 25M reverse complements (~7.5 Gbp BWT) on one MI355X")
VECTORS = ("thrK", "thrR", "minima", "da")
COUNTERS = ("leaves", "nodes", "minima", "lcp_values", "da_values")


def profile_path(name):
    return os.path.join(ROOT, "profiles", f"{'r05' if 'err_per_million' in CONFIGS[name] else 'r03'}_fullsize_{name}.json")


def sha256(buf) -> str:
    return hashlib.sha256(memoryview(np.ascontiguousarray(buf)).cast("B")).hexdigest()


def sha256_many(bufs):
    """SHA-256 of several large buffers side by side (hashlib releases the GIL on large updates)."""
    with ThreadPoolExecutor(max(1, len(bufs))) as ex:
        return list(ex.map(lambda b: None if b is None else sha256(b), bufs))


class Inputs:
    """ASCII eBWT(s) (+ DA) of a configuration: on the GPU as indexes (the ASCII text is released once packed), and -- when
    keep_host -- as host copies for the oracle."""

    def __init__(self, name, reads=None, keep_host=False, log=print):
        from ebwt2indel_amd import api, sim
        cfg = CONFIGS[name]
        self.name, self.mode = name, cfg["mode"]
        self.reads = reads or cfg["reads"]
        self.hosts, self.da_host, self.bwts, self.sha = [], None, [], {}
        t0 = time.time()
        if self.mode == 1:
            # bench.py's recipe: seed 1
            glen = int(self.reads * READ_LEN / 30.0)
            g = sim.random_genome(glen, 1)
            h2 = sim.mutate_haplotype(g, 1)
            text, offs = sim.haplotype_text([g, h2])
            groups = [(sim.sample_read_starts([g, h2], offs, self.reads, READ_LEN, 1), None, False)]
            if "err_per_million" in cfg:
                # the reads themselves become the generator's text (they are substrings of no genome any more)
                d_reads, d_starts = sim.error_sample_gpu(text, groups[0][0], READ_LEN, 1, cfg["err_per_million"], rc=cfg["rc"])
                self._keep = (d_reads, d_starts)
                text = None
                groups = [((int(d_reads.data_ptr()), int(d_starts.data_ptr()), int(d_starts.numel()), int(d_reads.numel())), None, False)]
        else:
            # scripts/modes23_scale.py's recipe: genome seed 5, starts seed 99; sample 0 from haplotype 1, sample 1 from haplotype 2
            glen = int(2 * self.reads * READ_LEN / 30)
            g = sim.random_genome(glen, 5)
            h2 = sim.mutate_haplotype(g, 5)
            text, offs = sim.haplotype_text([g, h2])
            rng = np.random.Generator(np.random.PCG64(99))
            s0 = (rng.integers(0, len(g) - READ_LEN + 1, size=self.reads, dtype=np.int64) + offs[0]).astype(np.uint64)
            s1 = (rng.integers(0, len(h2) - READ_LEN + 1, size=self.reads, dtype=np.int64) + offs[1]).astype(np.uint64)
            groups = [(s0, None, False), (s1, None, False)] if self.mode == 2 else [(np.concatenate([s0, s1]), self.reads, True)]
        del g, h2
        t_host = time.time() - t0
        t_gen = 0.0
        shas = []
        with ThreadPoolExecutor(2) as ex:            # the host copies are hashed while the GPU goes on
            for starts, n_first, want_da in groups:
                t0 = time.time()
                if isinstance(starts, tuple):
                    dev = sim.build_ebwt_gpu_slabbed(starts[0], starts[1], READ_LEN, n_reads=starts[2], text_len=starts[3])
                    self._keep = None
                    import torch
                    torch.cuda.empty_cache()
                else:
                    dev = sim.build_ebwt_gpu_slabbed(text, starts, READ_LEN, n_first=n_first, want_da=want_da)   # same bytes as the one-pass generator the records were made with (input SHA-256 below)
                t_gen += time.time() - t0
                host, da = dev.to_host()
                shas.append(ex.submit(sha256, host))
                if da is not None:
                    shas.append(ex.submit(sha256, da))
                    self.da_host = da
                self.bwts.append(api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n))
                dev.free()
                if keep_host:
                    self.hosts.append(host)
                del host
            shas = [f.result() for f in shas]
        del text
        self.n = sum(b.n for b in self.bwts)
        if self.mode == 3:
            self.sha = {"ebwt_sha256": [shas[0]], "da_sha256": shas[1]}
        else:
            self.sha = {"ebwt_sha256": shas}
        log(f"[{name}] mode {self.mode}, n = {' + '.join(str(b.n) for b in self.bwts)}; genome + starts {t_host:.1f}s on host, eBWT on the GPU {t_gen:.1f}s")

    def describe(self):
        cfg = CONFIGS[self.name]
        extra = dict(substitution_errors_per_million=cfg["err_per_million"], reverse_complements_added=cfg["rc"]) if "err_per_million" in cfg else {}
        return dict(mode=self.mode, reads_per_collection=self.reads, read_len=READ_LEN, n=[int(b.n) for b in self.bwts], **extra, **self.sha)


class Result:
    """One run of the product on `Inputs`: index(es) -> traversal -> phase 4 -> text."""

    def __init__(self, inp: Inputs, log=print, args=(), flags=None):
        from ebwt2indel_amd import api
        p = api.make_params(list(args))
        self.params = p
        t0 = time.time()
        b1 = inp.bwts[0]
        b2 = inp.bwts[1] if inp.mode == 2 else None
        self.flags = flags if flags is not None else (api.navigate_two_bwts(b1, b2, p) if b2 is not None else api.navigate_one_bwt(b1, p))
        if inp.mode == 3 and flags is None:
            self.flags.set_da(inp.da_host)
        t_nav = time.time() - t0
        t0 = time.time()
        self.calls = api.scan_clusters(b1, self.flags, p, bwt2=b2)
        self.text, self.cluster_nr, self.events = self.calls.format_snp(p, inp.mode != 1)
        t_p4 = time.time() - t0
        ns = self.flags.stats
        self.counters = {k: int(getattr(ns, k)) for k in COUNTERS}
        self.snp_sha256 = hashlib.sha256(self.text).hexdigest()
        self.snp_records = self.text.count(b">")
        self.seconds = dict(traversal=round(float(ns.seconds_traversal), 3), phases23_wall=round(t_nav, 3), phase4_and_text=round(t_p4, 3))
        log(f"[{inp.name}] GPU: phases 2-3 {t_nav:.2f}s (traversals {ns.seconds_traversal:.3f}s, {ns.nodes} nodes, {ns.leaves} leaves), phase 4 + text {t_p4:.2f}s: "
            f"{self.snp_records} records from {len(self.calls)} clusters, sha256 {self.snp_sha256}")

    def vector_shas(self):
        vec = self.flags.export()
        return dict(zip(VECTORS, sha256_many(list(vec)))), vec


def renumber(parts, used):
    """Texts of consecutive scan shards, each numbered from cluster 1 (`used[i]` numbers taken by shard i), -> one text with the
    running cluster_nr of a single pass (ebwt2InDel.cpp:52,1250,1328)."""
    import re
    out, off = [], 0
    pat = re.compile(rb"^>cluster:(\d+)_", re.M)
    for t, u in zip(parts, used):
        if off and t:
            o = off
            t = pat.sub(lambda m: b">cluster:" + str(int(m.group(1)) + o).encode() + b"_", t)
        out.append(t)
        off += u
    return b"".join(out)


def piece(inp: "Inputs", res: "Result", begin: int, end: int):
    """The product's phase 4 over closing positions [begin, end), its clusters numbered from 1."""
    from ebwt2indel_amd import api
    b2 = inp.bwts[1] if inp.mode == 2 else None
    c = api.scan_clusters(inp.bwts[0], res.flags, res.params, bwt2=b2, begin=begin, end=end)
    text, cluster_nr, _ = c.format_snp(res.params, inp.mode != 1)
    st = c.stats
    return dict(text=text, cluster_numbers=cluster_nr - 1, runs_closed=int(st.runs_closed), clusters=int(st.clusters), bases_in_runs=int(st.bases_in_runs))

"""Synthetic-input tooling: read simulator + eBWT/DA builder (no reference counterpart).

The reference ships no data and relies on external eBWT builders
(/root/reference/README.md:38, datasets/simulate.pdf); offline we make our own
inputs.  Recipe follows SURVEY.md section 8(d): uniform random genome, second
haplotype with a SNP every ``snp_every`` bases and a 1-3 base indel every
``indel_every`` bases, error-free fixed-length reads sampled uniformly and
alternating between the two haplotypes.

This module is data tooling only; it never touches the hot path.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_TOOL_SRC = os.path.join(_HERE, "tools", "ebwt_build.cpp")
_TOOL_SO = os.path.join(_BUILD, "libebwt_build.so")

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def build_tool(force: bool = False) -> str:
    """Compile the suffix-sorting eBWT builder (g++ -O3 -fopenmp)."""
    os.makedirs(_BUILD, exist_ok=True)
    if force or not os.path.exists(_TOOL_SO) or os.path.getmtime(_TOOL_SO) < os.path.getmtime(_TOOL_SRC):
        subprocess.check_call(
            ["g++", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", _TOOL_SO, _TOOL_SRC]
        )
    return _TOOL_SO


_lib = None


def _tool():
    global _lib
    if _lib is None:
        so = build_tool()
        _lib = ctypes.CDLL(so)
        _lib.ebwt_build.restype = ctypes.c_int
        _lib.ebwt_build.argtypes = [
            ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint8, ctypes.c_uint64,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
        ]
    return _lib


# --------------------------------------------------------------------------- genomes / reads

def random_genome(length: int, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return BASES[rng.integers(0, 4, size=length, dtype=np.uint8)]


def mutate_haplotype(genome: np.ndarray, seed: int, snp_every: int = 500, indel_every: int = 2000,
                     max_indel: int = 3) -> np.ndarray:
    """Second haplotype: a substitution every ``snp_every`` bases and an insertion or
    deletion of 1..max_indel bases every ``indel_every`` bases (offset so they never coincide)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    g = genome
    pieces: List[np.ndarray] = []
    events: List[Tuple[int, str, int]] = []
    if snp_every:
        for p in range(snp_every // 2, len(g), snp_every):
            events.append((p, "S", 0))
    if indel_every:
        for p in range(indel_every // 2 + snp_every // 4 + 7, len(g), indel_every):
            kind = "I" if rng.integers(0, 2) else "D"
            events.append((p, kind, int(rng.integers(1, max_indel + 1))))
    events.sort()
    cur = 0
    for p, kind, ln in events:
        if p < cur:
            continue
        pieces.append(g[cur:p])
        if kind == "S":
            old = g[p]
            new = BASES[(int(np.where(BASES == old)[0][0]) + int(rng.integers(1, 4))) % 4]
            pieces.append(np.array([new], dtype=np.uint8))
            cur = p + 1
        elif kind == "I":
            pieces.append(BASES[rng.integers(0, 4, size=ln, dtype=np.uint8)])
            cur = p
        else:
            cur = min(len(g), p + ln)
    pieces.append(g[cur:])
    return np.concatenate(pieces)


def sample_reads(haps: Sequence[np.ndarray], n_reads: int, read_len: int, seed: int) -> np.ndarray:
    """n_reads x read_len uint8 matrix; read j comes from haplotype j % len(haps)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xBEAD))
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    idx = np.arange(read_len, dtype=np.int64)
    for h, hap in enumerate(haps):
        rows = np.arange(h, n_reads, len(haps))
        starts = rng.integers(0, len(hap) - read_len + 1, size=len(rows), dtype=np.int64)
        out[rows] = hap[starts[:, None] + idx[None, :]]
    return out


def reads_to_text(reads: Iterable, term: int = 35) -> np.ndarray:
    """Concatenate reads, each followed by the terminator byte."""
    if isinstance(reads, np.ndarray) and reads.ndim == 2:
        n, ln = reads.shape
        t = np.empty((n, ln + 1), dtype=np.uint8)
        t[:, :ln] = reads
        t[:, ln] = term
        return t.reshape(-1)
    parts = []
    tb = bytes([term])
    for r in reads:
        b = bytes(r) if not isinstance(r, (bytes, bytearray)) else bytes(r)
        parts.append(b + tb)
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy()


# --------------------------------------------------------------------------- eBWT construction

def build_ebwt(reads, term: int = 35, threads: int = 0) -> np.ndarray:
    """eBWT (ASCII uint8 array) of one read collection."""
    text = np.ascontiguousarray(reads_to_text(reads, term))
    out = np.empty(len(text), dtype=np.uint8)
    rc = _tool().ebwt_build(text.ctypes.data, len(text), term, len(text), out.ctypes.data, None, threads)
    if rc != 0:
        raise RuntimeError("ebwt_build failed")
    return out


def build_merged_ebwt(reads0, reads1, term: int = 35, threads: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Merged eBWT of two collections plus the ASCII '0'/'1' document array."""
    t0 = reads_to_text(reads0, term)
    t1 = reads_to_text(reads1, term)
    text = np.ascontiguousarray(np.concatenate([t0, t1]))
    out = np.empty(len(text), dtype=np.uint8)
    da = np.empty(len(text), dtype=np.uint8)
    rc = _tool().ebwt_build(text.ctypes.data, len(text), term, len(t0), out.ctypes.data, da.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError("ebwt_build failed")
    return out, da


def build_ebwt_naive(reads, term: int = 35) -> bytes:
    """Pure-Python definition (tiny inputs only); cross-checks the C++ builder in tests."""
    t = chr(term)
    assert term < ord("A")
    rows = []
    for j, r in enumerate(reads):
        s = bytes(r).decode() if not isinstance(r, str) else r
        for i in range(len(s) + 1):
            rows.append((s[i:] + t, j, s[i - 1] if i > 0 else t))
    rows.sort(key=lambda x: (x[0], x[1]))
    return "".join(x[2] for x in rows).encode()


# --------------------------------------------------------------------------- canned workloads

def diploid_sample(genome_len: int, n_reads: int, read_len: int, seed: int, **kw):
    """One individual, two haplotypes (mode 1 input).  Returns (reads, hap1, hap2)."""
    g = random_genome(genome_len, seed)
    h2 = mutate_haplotype(g, seed, **kw)
    return sample_reads([g, h2], n_reads, read_len, seed), g, h2


def two_samples(genome_len: int, n_reads_each: int, read_len: int, seed: int, het_second: bool = False, **kw):
    """Two individuals (mode 2/3 input): sample 0 homozygous for the genome, sample 1
    homozygous for the mutated haplotype (or heterozygous when ``het_second``)."""
    g = random_genome(genome_len, seed)
    h2 = mutate_haplotype(g, seed, **kw)
    r0 = sample_reads([g], n_reads_each, read_len, seed)
    r1 = sample_reads([g, h2] if het_second else [h2], n_reads_each, read_len, seed + 1)
    return r0, r1


# --------------------------------------------------------------------------- GPU generator (benchmark scale)

_GEN_SRC = os.path.join(_HERE, "tools", "ebwt_gen.hip")
_GEN_SLABS_SRC = os.path.join(_HERE, "tools", "ebwt_gen_slabs.hip")   # the slab-by-slab generator (inputs whose keys do not fit one GPU at once)
_GEN_SO = os.path.join(_BUILD, "libebwt_gen.so")
_gen = None


def build_gpu_tool(force: bool = False) -> str:
    """Compile the GPU eBWT generator (hipcc, gfx950)."""
    os.makedirs(_BUILD, exist_ok=True)
    if force or not os.path.exists(_GEN_SO) or os.path.getmtime(_GEN_SO) < max(os.path.getmtime(_GEN_SRC), os.path.getmtime(_GEN_SLABS_SRC)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC",
                               "-Wl,-rpath,/opt/rocm/lib", "-o", _GEN_SO, _GEN_SRC, _GEN_SLABS_SRC])
    return _GEN_SO


def _gpu_tool():
    global _gen
    if _gen is None:
        if not os.path.exists(_GEN_SO):
            raise ImportError(f"{_GEN_SO} missing: run __graft_entry__.build()")
        try:
            import torch  # noqa: F401  (one HIP runtime per process, see _lib.load)
        except Exception:
            pass
        _gen = ctypes.CDLL(_GEN_SO)
        _gen.ebgen_last_error.restype = ctypes.c_char_p
        _gen.ebgen_free.argtypes = [ctypes.c_void_p]
        _gen.ebgen_free.restype = None
        _gen.ebgen_to_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        _gen.ebgen_build.restype = ctypes.c_int
        _gen.ebgen_build.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint8,
                                     ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double)]
        _gen.ebgen_slabs_last_error.restype = ctypes.c_char_p
        _gen.ebgen_build_slabbed.restype = ctypes.c_int
        _gen.ebgen_build_slabbed.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint32,
                                             ctypes.c_uint8, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                             ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p,
                                             ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
        _gen.ebgen_synth_diploid.restype = ctypes.c_int
        _gen.ebgen_synth_diploid.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32,
                                             ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_void_p),
                                             ctypes.POINTER(ctypes.c_uint64)]
    return _gen


def haplotype_text(haps: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenate haplotypes, each followed by its own separator byte (1, 2, ...).  Returns (text, offsets)."""
    assert len(haps) <= 2
    parts, offs, pos = [], [], 0
    for k, hp in enumerate(haps):
        offs.append(pos)
        parts.append(hp)
        parts.append(np.array([k + 1], dtype=np.uint8))
        pos += len(hp) + 1
    return np.ascontiguousarray(np.concatenate(parts)), np.array(offs, dtype=np.int64)


def sample_read_starts(haps: Sequence[np.ndarray], offsets: np.ndarray, n_reads: int, read_len: int, seed: int) -> np.ndarray:
    """Start offsets (into haplotype_text) of n_reads reads; read j comes from haplotype j % len(haps)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xBEAD))
    starts = np.empty(n_reads, dtype=np.uint64)
    for h, hap in enumerate(haps):
        rows = np.arange(h, n_reads, len(haps))
        s = rng.integers(0, len(hap) - read_len + 1, size=len(rows), dtype=np.int64)
        starts[rows] = (s + offsets[h]).astype(np.uint64)
    return starts


def two_sample_text(n_reads_each: int, read_len: int, coverage: float = 30.0, genome_seed: int = 5, starts_seed: int = 99):
    """Two individuals at benchmark scale (modes 2 / 3): sample 0 homozygous for a random genome, sample 1 for its mutated
    haplotype, `n_reads_each` error-free reads each (together `coverage` x of the diploid genome).  Returns (text, starts0, starts1)
    for build_ebwt_gpu_slabbed: two BWTs from the two start lists, or one merged BWT + document array from their concatenation
    with n_first = n_reads_each.  (The recipe of BASELINE configs[2] / configs[3] in tests/_fullsize.py and scripts/modes23_scale.py.)"""
    glen = int(2 * n_reads_each * read_len / coverage)
    g = random_genome(glen, genome_seed)
    h2 = mutate_haplotype(g, genome_seed)
    text, offs = haplotype_text([g, h2])
    rng = np.random.Generator(np.random.PCG64(starts_seed))
    s0 = (rng.integers(0, len(g) - read_len + 1, size=n_reads_each, dtype=np.int64) + offs[0]).astype(np.uint64)
    s1 = (rng.integers(0, len(h2) - read_len + 1, size=n_reads_each, dtype=np.int64) + offs[1]).astype(np.uint64)
    return text, s0, s1


def reads_from_starts(text: np.ndarray, starts: np.ndarray, read_len: int) -> np.ndarray:
    idx = np.arange(read_len, dtype=np.int64)
    return text[starts.astype(np.int64)[:, None] + idx[None, :]]


# --------------------------------------------------------------------------- reads with sequencing errors (+ reverse complements)
# The reference's documented input is simulated reads WITH an error profile and, recommended, their reverse complements
# (/root/reference/README.md:88-100, datasets/simulate.pdf: SimSeq).  Substitution errors here come from a counter-based hash of
# (seed, read, position) -- the same integer arithmetic in numpy and in torch, so the sample made on the GPU at 50 M reads and the
# one made on the host for a small test are the same function of their arguments.

_MASK64 = (1 << 64) - 1


def _mix64_np(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays."""
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def _error_draws_np(seed: int, first_read: int, n_reads: int, read_len: int, per_million: int):
    """(is_error, offset 1..3) for every base of reads first_read .. first_read + n_reads - 1."""
    idx = (np.arange(first_read, first_read + n_reads, dtype=np.uint64)[:, None] * np.uint64(read_len) + np.arange(read_len, dtype=np.uint64)[None, :])
    h = _mix64_np(idx ^ np.uint64((seed * 0x9E3779B97F4A7C15) & _MASK64))
    return (h & np.uint64(0xFFFFF)) < np.uint64(per_million * (1 << 20) // 1_000_000), (((h >> np.uint64(20)) & np.uint64(0xFFFF)) % np.uint64(3) + np.uint64(1)).astype(np.uint8)


def add_substitution_errors(reads: np.ndarray, seed: int, per_million: int = 10_000, first_read: int = 0) -> np.ndarray:
    """reads (n x L, ASCII ACGT) with a base replaced by one of the three others at every position the hash picks (1 % by default)."""
    code = np.zeros(256, dtype=np.uint8)
    code[BASES] = np.arange(4, dtype=np.uint8)
    c = code[reads]
    err, off = _error_draws_np(seed, first_read, reads.shape[0], reads.shape[1], per_million)
    return BASES[np.where(err, (c + off) % 4, c)]


def reverse_complement(reads: np.ndarray) -> np.ndarray:
    code = np.zeros(256, dtype=np.uint8)
    code[BASES] = np.arange(4, dtype=np.uint8)
    return BASES[3 - code[reads[:, ::-1]]]


def error_sample(text: np.ndarray, starts: np.ndarray, read_len: int, seed: int, per_million: int = 10_000, rc: bool = False) -> np.ndarray:
    """Host form (small inputs): the reads text[starts[r] : starts[r] + read_len] with substitution errors, followed -- rc -- by the
    reverse complement of every one of them (read n + r = reverse complement of read r)."""
    reads = add_substitution_errors(reads_from_starts(text, starts, read_len), seed, per_million)
    return np.concatenate([reads, reverse_complement(reads)]) if rc else reads


def error_sample_gpu(text: np.ndarray, starts: np.ndarray, read_len: int, seed: int, per_million: int = 10_000, rc: bool = False, device: int = 0,
                     chunk_reads: int = 1 << 21):
    """The same reads made on the GPU (torch; integer arithmetic only): returns (reads, read_starts) -- a uint8 tensor of all reads one
    behind the other (plus one separator byte) and the uint64 tensor of their start offsets in it: the `text` / `starts` of
    build_ebwt_gpu_slabbed (pass their data_ptr()s; T' = "any text that contains every read")."""
    import torch
    dev = torch.device(f"cuda:{device}")
    n, L = len(starts), read_len
    total = n * (2 if rc else 1)
    t = torch.from_numpy(np.ascontiguousarray(text)).to(dev)
    s = torch.from_numpy(np.ascontiguousarray(starts).astype(np.int64)).to(dev)
    out = torch.empty(total * L + 1, dtype=torch.uint8, device=dev)
    out[-1] = 1                                   # a separator behind the last read, as haplotype_text puts one behind every haplotype
    code = torch.zeros(256, dtype=torch.int64, device=dev)
    code[torch.from_numpy(BASES.astype(np.int64)).to(dev)] = torch.arange(4, dtype=torch.int64, device=dev)
    ascii_of = torch.from_numpy(BASES.copy()).to(dev)
    col = torch.arange(L, dtype=torch.int64, device=dev)

    def i64(v):                                   # a 64-bit pattern as the int64 torch computes in (wraps like uint64)
        v &= _MASK64
        return v - (1 << 64) if v >> 63 else v

    def lsr(x, k):                                # logical shift right of int64
        return (x >> k) & ((1 << (64 - k)) - 1)
    k0, m1, m2 = i64(seed * 0x9E3779B97F4A7C15), i64(0xBF58476D1CE4E5B9), i64(0x94D049BB133111EB)
    thr = per_million * (1 << 20) // 1_000_000
    for a in range(0, n, chunk_reads):
        b = min(n, a + chunk_reads)
        c = code[t[s[a:b, None] + col[None, :]].long()]
        x = (torch.arange(a, b, dtype=torch.int64, device=dev)[:, None] * L + col[None, :]) ^ k0
        x = (x ^ lsr(x, 30)) * m1
        x = (x ^ lsr(x, 27)) * m2
        x = x ^ lsr(x, 31)
        err = (x & 0xFFFFF) < thr
        off = (lsr(x, 20) & 0xFFFF) % 3 + 1
        c = torch.where(err, (c + off) % 4, c)
        out[a * L: b * L] = ascii_of[c].reshape(-1)
        if rc:
            out[(n + a) * L: (n + b) * L] = ascii_of[3 - c.flip(1)].reshape(-1)
        del c, x, err, off
    read_starts = (torch.arange(total, dtype=torch.int64, device=dev) * L)
    torch.cuda.synchronize()
    del t, s
    return out, read_starts


class DeviceEbwt:
    """eBWT (and optional DA) left in GPU memory by the generator; free() releases it."""

    def __init__(self, ptr, da_ptr, n, timings):
        self.ptr, self.da_ptr, self.n, self.timings = ptr, da_ptr, n, timings

    def to_host(self):
        out = np.empty(self.n, dtype=np.uint8)
        _hip_copy_to_host(out, self.ptr, self.n)
        da = None
        if self.da_ptr:
            da = np.empty(self.n, dtype=np.uint8)
            _hip_copy_to_host(da, self.da_ptr, self.n)
        return out, da

    def free(self):
        g = _gpu_tool()
        if self.ptr:
            g.ebgen_free(self.ptr)
            self.ptr = None
        if self.da_ptr:
            g.ebgen_free(self.da_ptr)
            self.da_ptr = None


def _hip_copy_to_host(dst: np.ndarray, src_ptr: int, nbytes: int):
    rc = _gpu_tool().ebgen_to_host(dst.ctypes.data, ctypes.c_void_p(src_ptr), nbytes)
    if rc != 0:
        raise RuntimeError(f"device-to-host copy failed: {_gpu_tool().ebgen_last_error().decode()}")


def build_ebwt_gpu(text: np.ndarray, starts: np.ndarray, read_len: int, term: int = 35, n_first: Optional[int] = None,
                   want_da: bool = False, device: int = 0) -> DeviceEbwt:
    """eBWT of fixed-length reads text[starts[r] : starts[r]+read_len], built on the GPU; result stays on the device."""
    g = _gpu_tool()
    text = np.ascontiguousarray(text, dtype=np.uint8)
    starts = np.ascontiguousarray(starts, dtype=np.uint64)
    d_bwt, d_da, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64(0)
    tm = (ctypes.c_double * 4)()
    rc = g.ebgen_build(text.ctypes.data, len(text), starts.ctypes.data, len(starts), read_len, term,
                       len(starts) if n_first is None else n_first, device, ctypes.byref(d_bwt),
                       ctypes.byref(d_da) if want_da else None, ctypes.byref(n), tm)
    if rc != 0:
        raise RuntimeError(f"ebgen_build failed ({rc}): {g.ebgen_last_error().decode()}")
    return DeviceEbwt(d_bwt.value, d_da.value if want_da else None, n.value, list(tm))


class DeviceSample:
    """A diploid sample made on the GPU (ebgen_synth_diploid): text = haplotype 1, byte 1, haplotype 2, byte 2; free() releases it."""

    def __init__(self, text_ptr, text_len, starts_ptr, n_reads, read_len, genome_len, hap2_len):
        self.text_ptr, self.text_len, self.starts_ptr, self.n_reads, self.read_len = text_ptr, text_len, starts_ptr, n_reads, read_len
        self.genome_len, self.hap2_len = genome_len, hap2_len

    def text_to_host(self):
        out = np.empty(self.text_len, dtype=np.uint8)
        _hip_copy_to_host(out, self.text_ptr, self.text_len)
        return out

    def starts_to_host(self):
        out = np.empty(self.n_reads, dtype=np.uint64)
        _hip_copy_to_host(out, self.starts_ptr, self.n_reads * 8)
        return out

    def free(self):
        g = _gpu_tool()
        for a in ("text_ptr", "starts_ptr"):
            if getattr(self, a):
                g.ebgen_free(getattr(self, a))
                setattr(self, a, None)


def synth_diploid_gpu(genome_len: int, n_reads: int, read_len: int, seed: int, snp_every: int = 500, indel_every: int = 2000, max_indel: int = 3,
                      device: int = 0) -> DeviceSample:
    """The recipe of diploid_sample() evaluated on the GPU with a counter-based generator (5 Gb genomes in a second; another
    random stream than the host recipe's, so the two do not produce the same sample)."""
    g = _gpu_tool()
    d_text, d_starts, m, h2 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = g.ebgen_synth_diploid(genome_len, seed, snp_every, indel_every, max_indel, n_reads, read_len, device, ctypes.byref(d_text), ctypes.byref(m),
                               ctypes.byref(d_starts), ctypes.byref(h2))
    if rc != 0:
        raise RuntimeError(f"ebgen_synth_diploid failed ({rc}): {g.ebgen_slabs_last_error().decode()}")
    return DeviceSample(d_text.value, m.value, d_starts.value, n_reads, read_len, genome_len, h2.value)


def build_ebwt_gpu_slabbed(text, starts, read_len: int, term: int = 35, n_first: Optional[int] = None, want_da: bool = False, device: int = 0,
                           prefix_len: int = -1, group_len: int = -1, n_reads: Optional[int] = None, text_len: Optional[int] = None,
                           lcp_truth: Optional[Tuple[int, int]] = None) -> DeviceEbwt:
    """Like build_ebwt_gpu, through the slab-by-slab generator (tools/ebwt_gen_slabs.hip): nothing of the size of the output but
    the output itself is resident, and the text may have 2^32 symbols or more.  `text` / `starts`: numpy arrays, or device
    pointers (int) with text_len / n_reads.  The returned object carries .info = (P, Pg, max rows of a group, max keys of a slab).
    lcp_truth = (K, k_right): the generator also states the collection's LCP bit vectors by their DEFINITION (common prefixes of
    adjacent read suffixes, from the text's suffix order) -- .truth = device pointers of thrK, thrR, minima ((n + 63) // 64 + 1 words
    each, release with free_truth()), .unknown = rows whose minimum bit it leaves undecided (the edges of slabs and short blocks)."""
    g = _gpu_tool()
    text_dev, starts_dev = isinstance(text, int), isinstance(starts, int)
    if not text_dev:
        text = np.ascontiguousarray(text, dtype=np.uint8)
        text_len = len(text)
    if not starts_dev:
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        n_reads = len(starts)
    d_bwt, d_da, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64(0)
    tm = (ctypes.c_double * 5)()
    info = (ctypes.c_uint64 * 4)()
    truth = (ctypes.c_void_p * 3)()
    unknown = np.zeros(3 * (4 ** 6 + 4 ** 6 // 3 + 2) if lcp_truth else 1, dtype=np.uint64)
    n_unknown = ctypes.c_uint64(0)
    rc = g.ebgen_build_slabbed(ctypes.c_void_p(text) if text_dev else text.ctypes.data, text_len, int(text_dev),
                               ctypes.c_void_p(starts) if starts_dev else starts.ctypes.data, n_reads, int(starts_dev), read_len, term,
                               n_reads if n_first is None else n_first, prefix_len, group_len, device, ctypes.byref(d_bwt),
                               ctypes.byref(d_da) if want_da else None, ctypes.byref(n), tm, info,
                               lcp_truth[0] if lcp_truth else 0, lcp_truth[1] if lcp_truth else 0, truth if lcp_truth else None,
                               unknown.ctypes.data, len(unknown), ctypes.byref(n_unknown))
    if rc != 0:
        raise RuntimeError(f"ebgen_build_slabbed failed ({rc}): {g.ebgen_slabs_last_error().decode()}")
    out = DeviceEbwt(d_bwt.value, d_da.value if want_da else None, n.value, list(tm))
    out.info = tuple(int(v) for v in info)
    out.truth = [truth[i] for i in range(3)] if lcp_truth else None
    out.unknown = np.unique(unknown[:n_unknown.value]) if lcp_truth else None

    def free_truth():
        for ptr in out.truth or []:
            if ptr:
                g.ebgen_free(ctypes.c_void_p(ptr))
        out.truth = None
    out.free_truth = free_truth
    return out


def replicated_ebwt_gpu(n_reads: int, read_len: int, coverage: float, copies: int, seed: int, device: int = 0):
    """eBWT of the collection in which every read of a diploid sample occurs `copies` times in a row (reads j*copies ..
    j*copies + copies - 1 are copies of base read j), as a torch uint8 tensor on `device`.

    Equal suffixes are ordered by read index, so the `copies` instances of a suffix are adjacent and preceded by the
    same symbol: the eBWT is the base collection's with every symbol repeated `copies` times (checked against the
    suffix-sorting builder in tests/test_sim.py).  The suffix tree has the base collection's right-maximal nodes, so the
    CPU reference gets through it in minutes although the text is `copies` times longer -- which is what makes inputs
    past 2^32 positions checkable against the reference binary itself (scripts/ref_past_2p32.py)."""
    import torch

    genome_len = int(n_reads * read_len / coverage)
    g = random_genome(genome_len, seed)
    h2 = mutate_haplotype(g, seed)
    text, offs = haplotype_text([g, h2])
    starts = sample_read_starts([g, h2], offs, n_reads, read_len, seed)
    dev = build_ebwt_gpu(text, starts, read_len, device=device)

    class _Bytes:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}
    base = torch.as_tensor(_Bytes(dev.ptr, dev.n), device=f"cuda:{device}")
    out = base[:, None].expand(-1, copies).reshape(-1) if copies > 1 else base.clone()   # one contiguous copy
    torch.cuda.synchronize()
    del base
    dev.free()
    return out

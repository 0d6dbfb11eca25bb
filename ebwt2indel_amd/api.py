"""Host-side mirror of the reference's interface for the hot path, on top of the C ABI.

Names follow the reference so the parity tests read like its own code would:
``DnaBwt`` ~ ``dna_bwt_t`` (internal/dna_bwt.hpp:24-420) with ``parallel_rank``, ``LF``, ``F``, ``FL``,
``root``, ``first_leaf``; ``navigate_one_bwt`` / ``navigate_two_bwts`` (ebwt2InDel.cpp:555,679);
``run_one_dataset`` / ``run_two_datasets`` / ``run_two_datasets_da`` (:1584,:1344,:1471).
Everything computes on the GPU through libebwt2indel.so; nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import EbError, Params

_L = None


def lib():
    global _L
    if _L is None:
        _L = _lib.load()
    return _L


def _check(rc):
    if rc != 0:
        raise EbError(rc, lib().eb_last_error().decode(errors="replace"))


def device_count() -> int:
    return lib().eb_device_count()


def make_params(args: Sequence[str] = (), **kw) -> Params:
    """Reference CLI options (-L -R -k -g -v -m -c -q -t) -> resolved eb_params (0 means default,
    ebwt2InDel.cpp:1740-1746)."""
    opt = {"-L": "k_left", "-R": "k_right", "-k": "K", "-g": "max_gap", "-v": "max_snvs", "-m": "mcov_out",
           "-c": "complexity", "-q": "max_variants_per_position", "-t": "term"}
    p = Params()
    p.term = 35
    it = iter(args)
    for a in it:
        setattr(p, opt[a], int(next(it)))
    for k, v in kw.items():
        setattr(p, k, v)
    lib().eb_params_resolve(C.byref(p))
    return p


def _u8(a) -> np.ndarray:
    if isinstance(a, (bytes, bytearray, memoryview)):
        return np.frombuffer(a, dtype=np.uint8)
    return np.ascontiguousarray(a, dtype=np.uint8)


class DnaBwt:
    """GPU-resident packed BWT with rank support (dna_bwt_t)."""

    def __init__(self, source, term: int = 35, device: int = 0, device_ptr: Optional[int] = None, n: Optional[int] = None):
        """source: path (str), bytes / uint8 array with the ASCII eBWT, or None with device_ptr+n."""
        self._h = C.c_void_p()
        L = lib()
        if device_ptr is not None:
            _check(L.eb_index_create_from_device(C.c_void_p(device_ptr), n, term, device, C.byref(self._h)))
        elif isinstance(source, str):
            _check(L.eb_index_create_from_file(source.encode(), term, device, C.byref(self._h)))
        else:
            buf = _u8(source)
            _check(L.eb_index_create(buf.ctypes.data, len(buf), term, device, C.byref(self._h)))
        info = _lib.IndexInfo()
        _check(L.eb_index_info(self._h, C.byref(info)))
        self.n = info.n
        self.F_arr = [info.F[i] for i in range(4)]
        self.n_blocks = info.n_blocks
        self.device_bytes = info.device_bytes
        self.term = term
        self.device = device

    def close(self):
        if self._h:
            lib().eb_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def bad_symbol():
        s, p = C.c_uint8(0), C.c_uint64(0)
        lib().eb_index_bad_symbol(C.byref(s), C.byref(p))
        return s.value, p.value

    def size(self) -> int:
        return self.n

    def build_psi(self):
        """FL table for an index that is scanned repeatedly (eb_index_build_psi): FL becomes one 16-byte read."""
        _check(lib().eb_index_build_psi(self._h))

    def psi(self):
        """(present, bytes, void chunks) of the FL table."""
        b, v = C.c_uint64(0), C.c_uint64(0)
        have = lib().eb_index_has_psi(self._h, C.byref(b), C.byref(v))
        return bool(have), b.value, v.value

    # ---- dna_string / dna_bwt primitives, batched
    def parallel_rank(self, pos) -> np.ndarray:
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.zeros((len(pos), 4), dtype=np.uint64)
        _check(lib().eb_rank4_batch(self._h, pos.ctypes.data, len(pos), out.ctypes.data))
        return out

    def __getitem__(self, pos) -> np.ndarray:
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.zeros(len(pos), dtype=np.uint8)
        _check(lib().eb_access_batch(self._h, pos.ctypes.data, len(pos), out.ctypes.data))
        return out

    def LF(self, ranges, sym=None) -> np.ndarray:
        """LF(range_t) -> (m,8) array, or LF(range_t, char) -> (m,2) when ``sym`` is given."""
        lr = np.ascontiguousarray(ranges, dtype=np.uint64).reshape(-1, 2)
        if sym is None:
            out = np.zeros((len(lr), 8), dtype=np.uint64)
            _check(lib().eb_lf_range_batch(self._h, lr.ctypes.data, len(lr), out.ctypes.data))
            return out
        s = np.ascontiguousarray(sym, dtype=np.uint8)
        out = np.zeros((len(lr), 2), dtype=np.uint64)
        _check(lib().eb_lf_range_sym_batch(self._h, lr.ctypes.data, s.ctypes.data, len(lr), out.ctypes.data))
        return out

    def LF_node(self, nodes7) -> np.ndarray:
        nd = np.ascontiguousarray(nodes7, dtype=np.uint64).reshape(-1, 7)
        out = np.zeros((len(nd), 28), dtype=np.uint64)
        _check(lib().eb_lf_node_batch(self._h, nd.ctypes.data, len(nd), out.ctypes.data))
        return out

    def FL(self, pos) -> np.ndarray:
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.zeros(len(pos), dtype=np.uint64)
        _check(lib().eb_fl_batch(self._h, pos.ctypes.data, len(pos), out.ctypes.data))
        return out

    def F(self, i: int) -> int:
        """F column symbol (dna_bwt.hpp:100-110)."""
        F = self.F_arr
        return self.term if i < F[0] else ord("A") if i < F[1] else ord("C") if i < F[2] else ord("G") if i < F[3] else ord("T")

    def root(self):
        return [0, *self.F_arr, self.n, 0]

    def first_leaf(self):
        return [0, self.F_arr[0], 0]

    def rank_device(self, d_pos: int, m: int, d_out: int) -> float:
        """Benchmark form: device pointers in/out, returns the kernel time in ms (HIP events)."""
        ms = C.c_float(0)
        _check(lib().eb_rank4_batch_device(self._h, C.c_void_p(d_pos), m, C.c_void_p(d_out), C.byref(ms)))
        return ms.value


class Flags:
    """LCP_threshold / LCP_minima / DA bitvectors on the GPU (phases 2-3)."""

    def __init__(self, handle, stats=None):
        self._h = handle
        self.stats = stats

    @classmethod
    def build(cls, bwt1: DnaBwt, bwt2: Optional[DnaBwt] = None, K: int = 16, k_right: int = 30) -> "Flags":
        h = C.c_void_p()
        st = _lib.NavStats()
        _check(lib().eb_flags_build(bwt1._h, bwt2._h if bwt2 is not None else None, K, k_right, C.byref(h), C.byref(st)))
        return cls(h, st)

    @classmethod
    def build_shard(cls, bwt1: DnaBwt, bwt2: Optional[DnaBwt], K: int, k_right: int, shard: int, n_shards: int) -> "Flags":
        """This shard's share of phases 2-3 (eb_flags_build_shard): the vectors of all shards must be summed and
        finalize() called before use -- distributed.navigate_sharded does both."""
        h = C.c_void_p()
        st = _lib.NavStats()
        _check(lib().eb_flags_build_shard(bwt1._h, bwt2._h if bwt2 is not None else None, K, k_right, shard, n_shards, C.byref(h), C.byref(st)))
        return cls(h, st)

    def device_vectors(self):
        """([thrK, thrR, minima, da] device pointers, 0 where absent; words per vector)."""
        ptrs = (C.c_void_p * 4)()
        w = C.c_uint64()
        _check(lib().eb_flags_device_vectors(self._h, ptrs, C.byref(w)))
        return [int(p or 0) for p in ptrs], int(w.value)

    def popcounts(self):
        """Ones in thrK, thrR, minima, da (0 where absent), counted on the device."""
        out = (C.c_uint64 * 4)()
        _check(lib().eb_flags_popcounts(self._h, out))
        return [int(v) for v in out]

    def finalize(self):
        _check(lib().eb_flags_finalize(self._h))

    @classmethod
    def from_host(cls, n: int, thrK, thrR, minima, da=None, device: int = 0) -> "Flags":
        h = C.c_void_p()
        arrs = [np.ascontiguousarray(a, dtype=np.uint64) if a is not None else None for a in (thrK, thrR, minima, da)]
        _check(lib().eb_flags_import(n, device, *[a.ctypes.data if a is not None else None for a in arrs], C.byref(h)))
        return cls(h)

    @classmethod
    def from_shard_sum(cls, n: int, thrK_below, thrR_below, minima, da=None, device: int = 0) -> "Flags":
        """The SUM of the shards' exported vectors (shard form: thrK / thrR complemented) -> finalised flags."""
        h = C.c_void_p()
        arrs = [np.ascontiguousarray(a, dtype=np.uint64) if a is not None else None for a in (thrK_below, thrR_below, minima, da)]
        _check(lib().eb_flags_import_shard_sum(n, device, *[a.ctypes.data if a is not None else None for a in arrs], C.byref(h)))
        return cls(h)

    def set_da(self, ascii01):
        d = _u8(ascii01)
        _check(lib().eb_flags_set_da(self._h, d.ctypes.data, len(d)))

    def set_da_file(self, path: str):
        _check(lib().eb_flags_set_da_file(self._h, os.fsencode(path)))

    def set_da_staged(self, staged: "StagedDa"):
        """Hands a document array loaded ahead (StagedDa) to these flags; the staged object is consumed."""
        _check(lib().eb_flags_set_da_staged(self._h, staged._h))
        staged._h = C.c_void_p()            # (consumed on success only)

    def set_da_device(self, d_ptr: int, n: int):
        """ASCII '0'/'1' bytes already in HBM on the flags' device (a generator's output)."""
        _check(lib().eb_flags_set_da_device(self._h, C.c_void_p(d_ptr), n))

    @property
    def words(self) -> int:
        return lib().eb_flags_words(self._h)

    @property
    def has_da(self) -> bool:
        return bool(lib().eb_flags_has_da(self._h))

    def export(self):
        w = self.words
        thrK, thrR, mn = (np.zeros(w, dtype=np.uint64) for _ in range(3))
        da = np.zeros(w, dtype=np.uint64) if self.has_da else None
        _check(lib().eb_flags_export(self._h, thrK.ctypes.data, thrR.ctypes.data, mn.ctypes.data,
                                     da.ctypes.data if da is not None else None))
        return thrK, thrR, mn, da

    def close(self):
        if self._h:
            lib().eb_flags_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StagedDa:
    """A -d file loaded and packed on the GPU ahead of the flags it belongs to (callable on another thread beside the traversal)."""

    def __init__(self, path: str, n_bwt: int, device: int = 0):
        self._h = C.c_void_p()
        _check(lib().eb_da_stage_file(os.fsencode(path), n_bwt, device, C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                lib().eb_da_staged_free(self._h)
        except Exception:
            pass


class Calls:
    """Phase-4 output: one record per surviving cluster, in BWT order."""

    def __init__(self, handle, stats=None):
        self._h = handle
        self.stats = stats

    def view(self):
        v = _lib.CallsView()
        _check(lib().eb_calls_get(self._h, C.byref(v)))
        return v

    def __len__(self):
        return self.view().count

    def arrays(self):
        """(clusters, chains, left words, right words) as raw uint8 copies."""
        v = self.view()
        k, m = v.count, v.n_chains
        lw, rw = (v.k_left + 31) // 32, (v.k_right + 31) // 32

        def grab(ptr, nbytes):
            return np.frombuffer(C.string_at(ptr, nbytes), dtype=np.uint8).copy() if nbytes else np.zeros(0, dtype=np.uint8)
        return (grab(v.clusters, k * C.sizeof(_lib.Cluster)), grab(v.chains, m * C.sizeof(_lib.Chain)),
                grab(v.left, m * lw * 8), grab(v.right, k * rw * 8))

    def blob(self) -> memoryview:
        """Zero-copy view of the call set's wire image; the view keeps this object (and so the memory) alive."""
        p, n = C.c_void_p(), C.c_uint64(0)
        _check(lib().eb_calls_blob(self._h, C.byref(p), C.byref(n)))
        arr = (C.c_uint8 * n.value).from_address(p.value)
        arr._owner = self                     # a ctypes array is the memoryview's exporter: it lives as long as any view of it
        return memoryview(arr)

    def pack(self) -> bytes:
        n = lib().eb_calls_packed_size(self._h)
        buf = (C.c_uint8 * n)()
        _check(lib().eb_calls_pack(self._h, buf, n))
        return bytes(buf)

    @classmethod
    def unpack(cls, data: bytes) -> "Calls":
        h = C.c_void_p()
        b = _u8(data)
        _check(lib().eb_calls_unpack(b.ctypes.data, len(b), C.byref(h)))
        return cls(h)

    def format_snp(self, params: Params, two_samples: bool, cluster_nr: int = 1, events: int = 0):
        """KisSNP2 text of this call set; returns (bytes, next cluster_nr, events)."""
        cn, ev = C.c_uint64(cluster_nr), C.c_uint64(events)
        txt, ln = C.c_void_p(), C.c_uint64(0)
        _check(lib().eb_format_snp(self._h, C.byref(params), int(two_samples), C.byref(cn), C.byref(ev), C.byref(txt), C.byref(ln)))
        # (ctypes.string_at takes the length as a C int: the text of a 151 G-position call set is past 2 GB)
        out = bytes((C.c_char * ln.value).from_address(txt.value)) if ln.value else b""
        lib().eb_free(txt)
        return out, cn.value, ev.value

    def write_snp(self, params: Params, two_samples: bool, path: str, append: bool = False, cluster_nr: int = 1, events: int = 0):
        cn, ev = C.c_uint64(cluster_nr), C.c_uint64(events)
        _check(lib().eb_write_snp(self._h, C.byref(params), int(two_samples), path.encode(), int(append), C.byref(cn), C.byref(ev)))
        return cn.value, ev.value

    def close(self):
        if self._h:
            lib().eb_calls_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def scan_clusters(bwt1: DnaBwt, flags: Flags, params: Params, bwt2: Optional[DnaBwt] = None, begin: int = 0,
                  end: Optional[int] = None) -> Calls:
    """Phase 4 over clusters closing in [begin,end) (ebwt2InDel.cpp:1609-1655 and siblings)."""
    n = bwt1.n + (bwt2.n if bwt2 is not None else 0)
    h = C.c_void_p()
    st = _lib.ScanStats()
    _check(lib().eb_scan_clusters(bwt1._h, bwt2._h if bwt2 is not None else None, flags._h, begin, n if end is None else end,
                                  C.byref(params), C.byref(h), C.byref(st)))
    return Calls(h, st)


class ScanJob:
    """A phase-4 call in flight (eb_scan_clusters_begin); finish() waits for it and returns the Calls."""

    def __init__(self, handle):
        self._h = handle

    def finish(self) -> Calls:
        """Waits, releases the job and returns the Calls (handle None -- only .stats -- for a device-only job)."""
        h, self._h = self._h, None
        out = C.c_void_p()
        st = _lib.ScanStats()
        _check(lib().eb_scan_clusters_end(h, C.byref(out), C.byref(st)))
        return Calls(out if out.value else None, st)

    def wait(self) -> None:
        """Blocks until the job's kernels and copies are complete; the job stays valid."""
        _check(lib().eb_scan_job_wait(self._h))

    def device_blob(self):
        """(device pointer, length) of the call set as one contiguous image in HBM -- the format of Calls.blob().
        Valid until finish()."""
        ptr, ln = C.c_void_p(), C.c_uint64()
        _check(lib().eb_scan_job_device_blob(self._h, C.byref(ptr), C.byref(ln)))
        return int(ptr.value or 0), int(ln.value)

    def __del__(self):
        try:
            if self._h:
                lib().eb_scan_clusters_cancel(self._h)
        except Exception:
            pass


    def device_wire(self):
        """(device pointer, length) of the call set as the COMPACT wire image (eb_scan_job_device_wire): what ranks send
        to rank 0.  Calls.unpack reads it.  Valid until finish()."""
        ptr, ln = C.c_void_p(), C.c_uint64()
        _check(lib().eb_scan_job_device_wire(self._h, C.byref(ptr), C.byref(ln)))
        return int(ptr.value or 0), int(ln.value)


    def wire_to_host(self, dst_ptr: int, cap: int) -> int:
        """Copies the compact wire image into the caller's host memory (page-locked for DMA speed) over this GPU's own PCIe
        link (eb_scan_job_wire_to_host); returns its length."""
        ln = C.c_uint64()
        _check(lib().eb_scan_job_wire_to_host(self._h, C.c_void_p(dst_ptr), cap, C.byref(ln)))
        return int(ln.value)


def config_reload() -> None:
    """The library reads its EB_* switches once, when it is loaded; call this after changing one inside the process."""
    lib().eb_config_reload()


SCAN_DEVICE_ONLY = 1
SCAN_WIRE = 2
SCAN_PIPELINED = 4


def scan_clusters_begin(bwt1: DnaBwt, flags: Flags, params: Params, bwt2: Optional[DnaBwt] = None, begin: int = 0,
                        end: Optional[int] = None, device_only: bool = False, wire: bool = False, pipelined: bool = False) -> ScanJob:
    """Enqueue phase 4 and return at once; the record download overlaps whatever the caller launches next.
    device_only: keep the call set in HBM (ScanJob.device_blob) -- for ranks that forward it with RCCL;
    wire: the job also packs the compact wire image (ScanJob.device_wire) as its last step;
    pipelined: the caller keeps jobs coming (EB_SCAN_PIPELINED): scans share the CUs with the previous job's context kernels."""
    n = bwt1.n + (bwt2.n if bwt2 is not None else 0)
    h = C.c_void_p()
    _check(lib().eb_scan_clusters_begin_ex(bwt1._h, bwt2._h if bwt2 is not None else None, flags._h, begin, n if end is None else end,
                                           C.byref(params), (SCAN_DEVICE_ONLY if device_only else 0) | (SCAN_WIRE if wire else 0) | (SCAN_PIPELINED if pipelined else 0), C.byref(h)))
    return ScanJob(h)


# ---- the three entry points of the reference's main() (ebwt2InDel.cpp:1805-1821), returning .snp bytes

def navigate_one_bwt(bwt: DnaBwt, params: Params) -> Flags:
    return Flags.build(bwt, None, params.K, params.k_right)


def navigate_two_bwts(bwt1: DnaBwt, bwt2: DnaBwt, params: Params) -> Flags:
    return Flags.build(bwt1, bwt2, params.K, params.k_right)


def run_one_dataset(ebwt, params: Optional[Params] = None, device: int = 0) -> bytes:
    p = params or make_params()
    bwt = DnaBwt(ebwt, term=p.term, device=device)
    fl = navigate_one_bwt(bwt, p)
    calls = scan_clusters(bwt, fl, p)
    return calls.format_snp(p, False)[0]


def run_two_datasets(ebwt1, ebwt2, params: Optional[Params] = None, device: int = 0) -> bytes:
    p = params or make_params()
    b1, b2 = DnaBwt(ebwt1, term=p.term, device=device), DnaBwt(ebwt2, term=p.term, device=device)
    fl = navigate_two_bwts(b1, b2, p)
    calls = scan_clusters(b1, fl, p, bwt2=b2)
    return calls.format_snp(p, True)[0]


def run_two_datasets_da(ebwt, da, params: Optional[Params] = None, device: int = 0) -> bytes:
    p = params or make_params()
    bwt = DnaBwt(ebwt, term=p.term, device=device)
    fl = navigate_one_bwt(bwt, p)
    fl.set_da(da)
    calls = scan_clusters(bwt, fl, p)
    return calls.format_snp(p, True)[0]

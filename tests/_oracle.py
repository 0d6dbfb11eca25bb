"""ctypes binding of oracle/_build/liboracle.so -- the CPU restatement used as the checker.

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Builds the library with gcc on first use (no GPU needed).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
ORC_SO = os.path.join(ORC_DIR, "_build", "liboracle.so")
REF_BIN = os.path.join(ORC_DIR, "_ref", "ebwt2InDel_ref")


class Params(C.Structure):
    """Mirror of eb_params (include/ebwt2indel.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "k_left", "k_right", "K", "max_gap", "max_snvs", "mcov_out", "complexity",
        "max_variants_per_position", "term")]

    DEFAULTS = dict(k_left=31, k_right=30, K=16, max_gap=10, max_snvs=2, mcov_out=3, complexity=20,
                    max_variants_per_position=0, term=35)

    @classmethod
    def from_args(cls, args=(), **kw):
        """Parse reference-style CLI options (-L -R -k -g -v -m -c -q -t); 0 means default."""
        opt = {"-L": "k_left", "-R": "k_right", "-k": "K", "-g": "max_gap", "-v": "max_snvs", "-m": "mcov_out",
               "-c": "complexity", "-q": "max_variants_per_position", "-t": "term"}
        vals = dict(cls.DEFAULTS)
        it = iter(args)
        for a in it:
            v = int(next(it))
            name = opt[a]
            if name in ("max_variants_per_position", "term"):
                vals[name] = v
            elif v != 0:
                vals[name] = v
        vals.update(kw)
        return cls(**vals)


class Cluster(C.Structure):
    """Mirror of eb_cluster (include/ebwt2indel.h)."""
    _fields_ = [("begin", C.c_uint64), ("end", C.c_uint64), ("first_chain", C.c_uint32), ("n_slots", C.c_uint16),
                ("right_found", C.c_uint16), ("right_len", C.c_uint32), ("reserved", C.c_uint32)]


class Chain(C.Structure):
    """Mirror of eb_chain (include/ebwt2indel.h)."""
    _fields_ = [("side", C.c_uint8), ("sym", C.c_uint8), ("reserved", C.c_uint16), ("len", C.c_uint32), ("support", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n", "leaves", "nodes", "lcp_values", "da_values", "minima", "clusters",
                                          "bases_in_runs", "runs_closed", "survivors", "chains", "events", "cluster_nr")] + \
               [(n, C.c_double) for n in ("t_load", "t_nav", "t_scan")]


def build_oracle(force=False):
    src = os.path.join(ORC_DIR, "ebwt_oracle.c")
    if force or not os.path.exists(ORC_SO) or os.path.getmtime(ORC_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORC_DIR, "_build/liboracle.so"], stdout=subprocess.DEVNULL)
    return ORC_SO


def _u8(a):
    return np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a, dtype=np.uint8)


class Oracle:
    def __init__(self):
        self.lib = L = C.CDLL(build_oracle())
        vp, u64, u8 = C.c_void_p, C.c_uint64, C.c_uint8
        L.orc_index_from_buffer.restype = vp
        L.orc_index_from_buffer.argtypes = [vp, u64, u8]
        L.orc_index_free.argtypes = [vp]
        L.orc_index_ok.argtypes = [vp]
        L.orc_index_size.restype = u64
        L.orc_index_size.argtypes = [vp]
        L.orc_index_F.argtypes = [vp, vp]
        L.orc_index_bad.argtypes = [vp, vp, vp]
        for f in ("orc_rank4_batch", "orc_access_batch", "orc_lf_range_batch", "orc_fl_batch", "orc_lf_node_batch"):
            getattr(L, f).argtypes = [vp, vp, u64, vp]
        L.orc_lf_range_sym_batch.argtypes = [vp, vp, vp, u64, vp]
        L.orc_navigate_one.restype = vp
        L.orc_navigate_one.argtypes = [vp, C.c_int, C.c_int]
        L.orc_navigate_one_mt.restype = vp
        L.orc_navigate_one_mt.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.orc_navigate_one_sample_check.argtypes = [vp, C.c_int, C.c_int, C.c_int, u64, u64, u64, vp, vp, vp, vp]
        L.orc_index_from_buffer_mt.restype = vp
        L.orc_index_from_buffer_mt.argtypes = [vp, u64, u8, C.c_int]
        L.orc_index_equal.argtypes = [vp, vp]
        L.orc_index_stream_begin.restype = vp
        L.orc_index_stream_begin.argtypes = [u64, u8]
        L.orc_index_stream_append.argtypes = [vp, u64, vp, u64, C.c_int]
        L.orc_index_stream_finish.argtypes = [vp, C.c_int]
        L.orc_navigate_two.restype = vp
        L.orc_navigate_two.argtypes = [vp, vp, C.c_int, C.c_int]
        L.orc_navigate_two_mt.restype = vp
        L.orc_navigate_two_mt.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
        L.orc_flags_set_da.argtypes = [vp, vp, u64]
        L.orc_flags_import.restype = vp
        L.orc_flags_import.argtypes = [u64, vp, vp, vp, vp]
        L.orc_flags_words.restype = u64
        L.orc_flags_words.argtypes = [vp]
        L.orc_flags_export.argtypes = [vp, vp, vp, vp, vp]
        L.orc_flags_counters.argtypes = [vp, vp]
        L.orc_flags_free.argtypes = [vp]
        L.orc_run_mem.restype = vp
        L.orc_run_mem.argtypes = [vp, u64, vp, u64, vp, u64, C.POINTER(Params), C.POINTER(u64), C.POINTER(Stats)]
        L.orc_scan.restype = vp
        L.orc_scan.argtypes = [vp, vp, vp, C.POINTER(Params), u64, u64, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64),
                               C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u64), C.POINTER(u64), vp]
        L.orc_free.argtypes = [vp]
        L.orc_last_hist.argtypes = [vp]
        L.orc_select_check.restype = u64
        L.orc_select_check.argtypes = [vp, u64, u64]
        L.orc_select_check_boundaries.restype = u64
        L.orc_select_check_boundaries.argtypes = [vp, u64, C.POINTER(u64)]
        L.orc_rank_bench.restype = C.c_double
        L.orc_rank_bench.argtypes = [vp, u64, u64, C.POINTER(u64)]

    # ---- whole pipeline
    def run(self, bwt1, bwt2=None, da=None, params=None):
        """Returns (.snp bytes, Stats).  Raises ValueError on a forbidden input symbol."""
        p = params or Params.from_args()
        b1 = _u8(bwt1)
        b2 = _u8(bwt2) if bwt2 is not None else None
        d = _u8(da) if da is not None else None
        n = C.c_uint64(0)
        st = Stats()
        ptr = self.lib.orc_run_mem(b1.ctypes.data, len(b1), b2.ctypes.data if b2 is not None else None,
                                   len(b2) if b2 is not None else 0, d.ctypes.data if d is not None else None,
                                   len(d) if d is not None else 0, C.byref(p), C.byref(n), C.byref(st))
        if not ptr:
            raise ValueError("forbidden character in input")
        out = C.string_at(ptr, n.value)
        self.lib.orc_free(ptr)
        return out, st

    # ---- index + primitives
    def index(self, ascii_bwt, term=35, threads=1):
        """threads > 1: the threaded builder (bit-identical index), for full-size checks."""
        return OracleIndex(self, ascii_bwt, term, threads)


class OracleIndex:
    def __init__(self, orc, ascii_bwt, term=35, threads=1):
        self.o = orc
        self.buf = _u8(ascii_bwt)
        if threads > 1:
            self.h = orc.lib.orc_index_from_buffer_mt(self.buf.ctypes.data, len(self.buf), term, threads)
        else:
            self.h = orc.lib.orc_index_from_buffer(self.buf.ctypes.data, len(self.buf), term)
        self.ok = bool(orc.lib.orc_index_ok(self.h))
        self.n = len(self.buf)
        self.term = term

    @classmethod
    def stream(cls, orc, n, term=35):
        """An index fed in pieces (orc_index_stream_*): append(pos, piece) for every piece, then finish(threads).  For inputs whose
        ASCII form does not fit the host beside the index (151 G symbols come off the GPU a few GB at a time)."""
        self = cls.__new__(cls)
        self.o, self.buf, self.n, self.term, self.ok = orc, None, n, term, False
        self.h = orc.lib.orc_index_stream_begin(n, term)
        assert self.h, "out of host memory for the oracle index"
        return self

    def append(self, pos, piece, threads=1):
        piece = _u8(piece)
        rc = self.o.lib.orc_index_stream_append(self.h, pos, piece.ctypes.data, len(piece), threads)
        assert rc == 0, f"piece [{pos}, {pos + len(piece)}) refused: pieces start on multiples of 128 and only the last may end elsewhere"

    def finish(self, threads=1):
        self.o.lib.orc_index_stream_finish(self.h, threads)
        self.ok = bool(self.o.lib.orc_index_ok(self.h))
        return self

    def __del__(self):
        try:
            self.o.lib.orc_index_free(self.h)
        except Exception:
            pass

    def bad(self):
        s, p = C.c_uint8(0), C.c_uint64(0)
        self.o.lib.orc_index_bad(self.h, C.byref(s), C.byref(p))
        return s.value, p.value

    def F(self):
        f = np.zeros(4, dtype=np.uint64)
        self.o.lib.orc_index_F(self.h, f.ctypes.data)
        return f

    def rank4(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.zeros((len(pos), 4), dtype=np.uint64)
        self.o.lib.orc_rank4_batch(self.h, pos.ctypes.data, len(pos), out.ctypes.data)
        return out

    def access(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.zeros(len(pos), dtype=np.uint8)
        self.o.lib.orc_access_batch(self.h, pos.ctypes.data, len(pos), out.ctypes.data)
        return out

    def lf_range(self, lr):
        lr = np.ascontiguousarray(lr, dtype=np.uint64).reshape(-1, 2)
        out = np.zeros((len(lr), 8), dtype=np.uint64)
        self.o.lib.orc_lf_range_batch(self.h, lr.ctypes.data, len(lr), out.ctypes.data)
        return out

    def lf_range_sym(self, lr, sym):
        lr = np.ascontiguousarray(lr, dtype=np.uint64).reshape(-1, 2)
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        out = np.zeros((len(lr), 2), dtype=np.uint64)
        self.o.lib.orc_lf_range_sym_batch(self.h, lr.ctypes.data, sym.ctypes.data, len(lr), out.ctypes.data)
        return out

    def fl(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.zeros(len(pos), dtype=np.uint64)
        self.o.lib.orc_fl_batch(self.h, pos.ctypes.data, len(pos), out.ctypes.data)
        return out

    def lf_node(self, nodes7):
        nodes7 = np.ascontiguousarray(nodes7, dtype=np.uint64).reshape(-1, 7)
        out = np.zeros((len(nodes7), 28), dtype=np.uint64)
        self.o.lib.orc_lf_node_batch(self.h, nodes7.ctypes.data, len(nodes7), out.ctypes.data)
        return out


def traversal_sample_check(orc, idx, K, k_right, vectors, every, which=0, frontier=1 << 16, threads=1):
    """A sample of the reference's two traversals (every `every`-th subtree below a breadth-first frontier, and the whole top above
    it) held against given thrK / thrR / minima words slot by slot (oracle/ebwt_oracle.c, orc_navigate_one_sample_check): for sizes
    where the whole traversal on the host would take the better part of an hour."""
    v = [np.ascontiguousarray(a, dtype=np.uint64) for a in vectors[:3]]
    assert all(len(a) >= (idx.n + 63) // 64 for a in v)
    out = np.zeros(6, dtype=np.uint64)
    orc.lib.orc_navigate_one_sample_check(idx.h, K, k_right, threads, frontier, every, which, v[0].ctypes.data, v[1].ctypes.data, v[2].ctypes.data, out.ctypes.data)
    return dict(zip(("leaves", "nodes", "lcp_slots", "minima_expected", "mismatches", "first_mismatch_plus_1"), (int(x) for x in out)))


class OracleFlags:
    """Phases 2-3 on the CPU oracle: bitvectors as uint64 word arrays."""

    def __init__(self, orc, idx1, idx2=None, K=16, k_right=30, da=None, vectors=None, threads=1):
        """vectors = (thrK, thrR, minima[, da words]) imports given flag words instead of navigating (scan stress tests).
        threads > 1: the threaded traversal -- same vectors and counters, for full-size checks."""
        self.o = orc
        self.idx1, self.idx2 = idx1, idx2
        L = orc.lib
        if vectors is not None:
            v = [np.ascontiguousarray(a, dtype=np.uint64) if a is not None else None for a in vectors]
            n = idx1.n + (idx2.n if idx2 is not None else 0)
            assert all(a is None or len(a) >= (n + 63) // 64 for a in v)
            self.h = L.orc_flags_import(n, v[0].ctypes.data, v[1].ctypes.data, v[2].ctypes.data,
                                        v[3].ctypes.data if len(v) > 3 and v[3] is not None else None)
            self.words = L.orc_flags_words(self.h)
            self.has_da = len(v) > 3 and v[3] is not None
            return
        if idx2 is not None:
            self.h = L.orc_navigate_two_mt(idx1.h, idx2.h, K, k_right, threads) if threads > 1 else L.orc_navigate_two(idx1.h, idx2.h, K, k_right)
        elif threads > 1:
            self.h = L.orc_navigate_one_mt(idx1.h, K, k_right, threads)
        else:
            self.h = L.orc_navigate_one(idx1.h, K, k_right)
        if da is not None:
            d = _u8(da)
            L.orc_flags_set_da(self.h, d.ctypes.data, len(d))
        self.words = L.orc_flags_words(self.h)
        self.has_da = idx2 is not None or da is not None

    def __del__(self):
        try:
            self.o.lib.orc_flags_free(self.h)
        except Exception:
            pass

    def export(self):
        w = self.words
        thrK, thrR, mn = (np.zeros(w, dtype=np.uint64) for _ in range(3))
        da = np.zeros(w, dtype=np.uint64) if self.has_da else None
        self.o.lib.orc_flags_export(self.h, thrK.ctypes.data, thrR.ctypes.data, mn.ctypes.data,
                                    da.ctypes.data if da is not None else None)
        return thrK, thrR, mn, da

    def counters(self):
        c = np.zeros(6, dtype=np.uint64)
        self.o.lib.orc_flags_counters(self.h, c.ctypes.data)
        return dict(zip(("leaves", "nodes", "lcp_values", "da_values", "minima", "max_stack"), (int(x) for x in c)))

    def scan(self, params, begin, end, cluster_nr=1, capture=False):
        """Phase 4 over clusters whose CLOSING position lies in [begin,end) (the product's ownership rule: a run open at `begin` is rewound to its first row, a run still open at `end` is left to the next shard).  Returns dict(text, cluster_nr, events, stats[, clusters, left, right])."""
        L = self.o.lib
        cn, ev, ln = C.c_uint64(cluster_nr), C.c_uint64(0), C.c_uint64(0)
        recs, chains, left, right = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nrec, nch = C.c_uint64(0), C.c_uint64(0)
        st = np.zeros(5, dtype=np.uint64)
        cap = (lambda x: C.byref(x)) if capture else (lambda x: None)
        ptr = L.orc_scan(self.idx1.h, self.idx2.h if self.idx2 is not None else None, self.h, C.byref(params), begin, end,
                         C.byref(cn), C.byref(ev), C.byref(ln), cap(recs), cap(chains), cap(left), cap(right), cap(nrec), cap(nch),
                         st.ctypes.data)
        text = C.string_at(ptr, ln.value)
        L.orc_free(ptr)
        out = dict(text=text, cluster_nr=cn.value, events=ev.value,
                   stats=dict(zip(("runs_closed", "clusters", "bases_in_runs", "survivors", "chains"), (int(x) for x in st))))
        hist = np.zeros(201, dtype=np.uint64)
        L.orc_last_hist(hist.ctypes.data)
        out["hist"] = hist
        if capture:
            k, m = nrec.value, nch.value
            lw, rw = (params.k_left + 31) // 32, (params.k_right + 31) // 32

            def grab(p_, nbytes):
                return np.frombuffer(C.string_at(p_, nbytes), dtype=np.uint8).copy() if nbytes and p_.value else np.zeros(0, np.uint8)
            out["clusters"] = grab(recs, k * C.sizeof(Cluster))
            out["chains"] = grab(chains, m * C.sizeof(Chain))
            out["left"] = grab(left, m * lw * 8)
            out["right"] = grab(right, k * rw * 8)
            out["count"], out["n_chains"] = k, m
            for p_ in (recs, chains, left, right):
                if p_.value:
                    L.orc_free(p_)
        return out


def pack_calls(rec: dict, k_left: int, k_right: int) -> bytes:
    """Byte image understood by eb_calls_unpack (csrc/eb_writer.cpp) from an oracle record capture:
    64-byte header, clusters, chains (padded to 8 bytes), left words, right words."""
    hdr = np.array([0x32534C4C41434245, rec["count"], rec["n_chains"], k_left, k_right, 0, 0, 0], dtype=np.uint64).tobytes()
    chains = rec["chains"].tobytes()
    chains += b"\0" * ((-len(chains)) % 8)
    return hdr + rec["clusters"].tobytes() + chains + rec["left"].tobytes() + rec["right"].tobytes()


def apply_right_gate(rec: dict, k_left: int, k_right: int, mode: int) -> dict:
    """The record capture as a LARGE call set leaves the product (include/ebwt2indel.h, EB_RIGHT_SKIPPED): clusters that can
    never print -- one sample: fewer than two full-length left contexts (ebwt2InDel.cpp:1256 with :317); two samples: no
    full context on each side with different letters (:1161-1163) -- carry no right context (right_len 0, words zero,
    reserved = 1).  The oracle itself always extracts it, like the reference; this applies the definition to its capture."""
    out = dict(rec)
    cl = np.frombuffer(rec["clusters"].tobytes(), dtype=[("b", "<u8"), ("e", "<u8"), ("fc", "<u4"), ("ns", "<u2"), ("rf", "<u2"), ("rl", "<u4"), ("res", "<u4")]).copy()
    ch = np.frombuffer(rec["chains"].tobytes(), dtype=[("side", "u1"), ("sym", "u1"), ("r", "<u2"), ("len", "<u4"), ("sup", "<u4")])
    rw = (k_right + 31) // 32
    right = np.frombuffer(rec["right"].tobytes(), dtype=np.uint64).reshape(-1, rw).copy() if len(cl) else np.zeros((0, rw), np.uint64)
    skipped = 0
    for i, c in enumerate(cl):
        k = ch[c["fc"]: c["fc"] + c["ns"]]
        full = k[k["len"] == k_left]
        s0, s1 = set(full["sym"][full["side"] == 0].tolist()), set(full["sym"][full["side"] == 1].tolist())
        walk = len(s0) >= 2 if mode == 1 else bool(s0) and bool(s1) and len(s0 | s1) >= 2
        if not walk:
            cl[i]["rl"], cl[i]["res"] = 0, 1
            right[i] = 0
            skipped += 1
    out["clusters"] = np.frombuffer(cl.tobytes(), dtype=np.uint8).copy()
    out["right"] = np.frombuffer(right.tobytes(), dtype=np.uint8).copy()
    out["skipped"] = skipped
    return out

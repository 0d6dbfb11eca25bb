"""GPU parity past 2^32 BWT positions -- where every BASELINE config from [1] on lives.  The reference is 64-bit with
2^32-symbol superblocks (dna_string.hpp:36-38,140-152: `superblock_ranks[i>>32]` + u32 block counters relative to the
superblock); here that is x.super[] indexing for superblock >= 1, 40-bit traversal records, `begin_hi` of the wire
image, DA rank groups and select-directory offsets past 2^32.

  * a random 5-symbol string of n = 2^32 + 2^22 built on the device: parallel_rank / operator[] / LF / FL at positions
    straddling 2^32 against the DEFINITION (chunked prefix counts computed with torch on the device), which also pins
    the oracle's own superblock arithmetic there;
  * phase 4 on that string under adversarial flag patterns in the window [2^32 - 2^20, 2^32 + 2^20) against the oracle's
    scan of the same window (records, counters, CLUST_SIZES; whole window and cut exactly at 2^32), modes 1, 3 and 2;
  * the whole pipeline (index -> traversal -> scan -> writer) on a 4.35 G-position read collection against the
    UNMODIFIED REFERENCE BINARY's .snp (SHA-256 recorded by scripts/ref_past_2p32.py in the build container).
About 4 minutes on the GPU box (host side: two oracle indexes of 4.3 G symbols)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from ebwt2indel_amd import api, sim
from tests import _oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWO32 = 1 << 32
N = TWO32 + (1 << 22)
W0, W1 = TWO32 - (1 << 20), TWO32 + (1 << 20)         # the window phase 4 is checked on


class _DevBytes:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}


@pytest.fixture(scope="module")
def big():
    """Random string over {A,C,G,T,#} of N symbols in HBM + its index; host copy for the oracle."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(20261004)
    lut = torch.tensor(list(b"ACGT#"), dtype=torch.uint8, device=dev)
    text = torch.empty(N, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for a in range(0, N, step):
        b = min(N, a + step)
        # TERM a little rarer than the letters, as in an eBWT
        code = torch.randint(0, 21, (b - a,), dtype=torch.int64, device=dev, generator=gen) // 5
        text[a:b] = lut[code]
        del code
    text[:16] = ord("#")                                  # row 0 is a TERM row like in any eBWT
    torch.cuda.synchronize()
    bwt = api.DnaBwt(None, device_ptr=text.data_ptr(), n=N)
    return dict(torch=torch, dev=dev, text=text, bwt=bwt)


def _def_rank(big, pos):
    """#A,#C,#G,#T in text[0,p) for every p in pos, by chunked prefix counts on the device (the definition)."""
    torch, text = big["torch"], big["text"]
    pos = np.asarray(pos, dtype=np.uint64)
    order = np.argsort(pos, kind="stable")
    sp = pos[order].astype(np.int64)
    out = np.zeros((len(pos), 4), dtype=np.uint64)
    base = np.zeros(4, dtype=np.int64)
    step = 1 << 27
    for a in range(0, N + 1, step):
        b = min(N, a + step)
        lo, hi = np.searchsorted(sp, a, side="left"), np.searchsorted(sp, b, side="left" if b < N else "right")
        chunk = text[a:b]
        if hi > lo:
            idx = torch.as_tensor(sp[lo:hi] - a, device=big["dev"])
        for c, ch in enumerate(b"ACGT"):
            m = chunk == ch
            if hi > lo:
                cs = torch.cumsum(m, dim=0, dtype=torch.int32)
                cs = torch.cat([torch.zeros(1, dtype=torch.int32, device=big["dev"]), cs])
                out[order[lo:hi], c] = (cs[idx].cpu().numpy().astype(np.int64) + base[c]).astype(np.uint64)
                del cs
            base[c] += int(m.sum())
        if b == N:
            break
    return out


def _positions_around_2p32(rng):
    edge = np.arange(TWO32 - 2200, TWO32 + 2200, dtype=np.uint64)                  # every offset of the blocks around the boundary
    blocks = (TWO32 + 128 * np.arange(-40, 40, dtype=np.int64)).astype(np.uint64)
    ends = np.array([0, 1, 127, 128, N - 129, N - 128, N - 1, N], dtype=np.uint64)
    rnd = rng.integers(0, N + 1, size=600_000, dtype=np.uint64)
    hi = rng.integers(TWO32, N + 1, size=400_000, dtype=np.uint64)                 # superblock 1
    return np.concatenate([edge, blocks, blocks - np.uint64(1), ends, rnd, hi])


def test_index_totals_past_2p32(big):
    g = big["bwt"]
    tot = [int(x) for x in _def_rank(big, [N])[0]]
    n_term = N - sum(tot)
    assert g.size() == N and g.n_blocks == (N + 128) // 128
    assert g.F_arr == [n_term, n_term + tot[0], n_term + tot[0] + tot[1], n_term + tot[0] + tot[1] + tot[2]]
    assert all((1 << 29) < t < TWO32 for t in tot)                 # ~1.02 G of each letter, ~0.2 G TERM


def test_parallel_rank_and_access_straddling_2p32_vs_definition(big, oracle_big):
    g = big["bwt"]
    rng = np.random.Generator(np.random.PCG64(1))
    pos = _positions_around_2p32(rng)
    want = _def_rank(big, pos)
    got = g.parallel_rank(pos)
    assert np.array_equal(got, want)
    assert (want[pos >= np.uint64(TWO32)].sum(axis=1) > 3_000_000_000).all()
    # the checker itself (the oracle's superblock arithmetic) against the definition, then the rest against the checker
    assert np.array_equal(oracle_big.rank4(pos), want)
    acc = pos[pos < np.uint64(N)]
    sym = big["text"][big["torch"].as_tensor(acc.astype(np.int64), device=big["dev"])].cpu().numpy()
    assert np.array_equal(g[acc], sym)
    assert np.array_equal(oracle_big.access(acc), sym)


def test_lf_and_fl_straddling_2p32(big, oracle_big):
    g = big["bwt"]
    rng = np.random.Generator(np.random.PCG64(2))
    F = np.array(g.F_arr, dtype=np.uint64)
    # LF(range) / LF(range,c): ranges that begin below 2^32 and end above it, ranges inside superblock 1, empty ranges
    l = np.concatenate([rng.integers(TWO32 - 5000, TWO32 + 5000, size=40_000, dtype=np.uint64), rng.integers(TWO32, N - 400, size=40_000, dtype=np.uint64)])
    w = rng.integers(0, 400, size=len(l), dtype=np.uint64)
    w[:3000] = 0
    lr = np.stack([l, np.minimum(l + w, np.uint64(N))], axis=1)
    ra, rb = _def_rank(big, lr[:, 0]), _def_rank(big, lr[:, 1])
    want = np.empty((len(lr), 8), dtype=np.uint64)
    want[:, 0::2] = F[None, :] + ra
    want[:, 1::2] = F[None, :] + rb
    got = g.LF(lr)
    assert np.array_equal(got, want)
    assert np.array_equal(oracle_big.lf_range(lr), want)
    letters = rng.integers(0, 4, size=len(lr))
    one = g.LF(lr, np.frombuffer(b"ACGT", dtype=np.uint8)[letters])
    assert np.array_equal(one[:, 0], want[np.arange(len(lr)), 2 * letters]) and np.array_equal(one[:, 1], want[np.arange(len(lr)), 2 * letters + 1])
    # FL = select (dna_string.hpp:182-188): j = FL(i) holds letter c = F(i) and exactly i - F_c occurrences of c precede it.
    # Rows i whose image lies around 2^32 (select directory entries on both sides of the boundary), plus random rows.
    at = _def_rank(big, [TWO32])[0]
    near = np.concatenate([F[c] + at[c] + np.arange(-3000, 3000, dtype=np.int64).astype(np.uint64) for c in range(4)])
    rows = np.concatenate([near, rng.integers(int(F[0]), N, size=200_000, dtype=np.uint64)])
    j = g.FL(rows)
    assert (j[: len(near)] > np.uint64(TWO32 - 20_000)).all() and (j[: len(near)] < np.uint64(TWO32 + 20_000)).all()
    c = np.searchsorted(F, rows, side="right") - 1                      # F(i): letter class of row i
    sym = big["text"][big["torch"].as_tensor(j.astype(np.int64), device=big["dev"])].cpu().numpy()
    assert np.array_equal(sym, np.frombuffer(b"ACGT", dtype=np.uint8)[c])
    rj = _def_rank(big, j)
    assert np.array_equal(rj[np.arange(len(rows)), c], rows - F[c])
    some = rows[:: 97]
    assert np.array_equal(g.FL(some), oracle_big.fl(some))               # and the reference's bisection agrees
    # the same rows through the FL table (38-bit positions: bits 32..37 travel in the chunk's spare bits)
    assert not g.psi()[0]
    g.build_psi()
    assert g.psi()[0] and np.array_equal(g.FL(rows), j)


def test_oracle_directory_select_equals_bisection_around_2p32(oracle, oracle_big):
    """The checker's own select (directory form, what the full-size records run on) against the reference's bisection
    (dna_string.hpp:254-272) for the occurrences on both sides of the superblock boundary, and at random."""
    import ctypes
    cnt = ctypes.c_uint64(0)
    assert oracle.lib.orc_select_check_boundaries(oracle_big.h, 3000, ctypes.byref(cnt)) == 0
    assert cnt.value == 4 * 6000
    assert oracle.lib.orc_select_check(oracle_big.h, 100_000, 23) == 0


def test_lf_node_boundaries_past_2p32(big, oracle_big):
    g = big["bwt"]
    rng = np.random.Generator(np.random.PCG64(3))
    m = 20_000
    b = np.sort(rng.integers(TWO32 - 300, TWO32 + 300, size=(m, 6), dtype=np.uint64), axis=1)      # six boundaries around 2^32
    b[m // 2:] = np.sort(rng.integers(TWO32, N + 1, size=(m - m // 2, 6), dtype=np.uint64), axis=1)
    nodes = np.concatenate([b, rng.integers(0, 100, size=(m, 1), dtype=np.uint64)], axis=1)
    assert np.array_equal(g.LF_node(nodes), oracle_big.lf_node(nodes))


@pytest.fixture(scope="module")
def host_text(big):
    out = np.empty(N, dtype=np.uint8)
    sim._hip_copy_to_host(out, big["text"].data_ptr(), N)
    return out


@pytest.fixture(scope="module")
def oracle_big(oracle, host_text):
    return oracle.index(host_text, threads=min(os.cpu_count() or 1, 16))


def _window_flags(rng):
    """thrK / thrR / minima words for N positions: zero outside [W0, W1), adversarial inside -- runs that end exactly at
    2^32, begin exactly at 2^32, cross it by a few rows / by hundreds, nine clusters per word, word-aligned runs."""
    idx = np.arange(W0, W1, dtype=np.int64)
    rel = idx - W0
    bits = np.zeros(len(idx), dtype=bool)
    q = len(idx) // 8
    bits[0 * q:1 * q] = rng.random(q) < 0.95                                   # dense runs of ~20
    bits[1 * q:2 * q] = (rel[1 * q:2 * q] % 7) != 6                            # nine clusters per word
    bits[2 * q:3 * q] = (rel[2 * q:3 * q] % 151) != 0                          # runs of 150 across words
    bits[3 * q:4 * q] = rng.random(q) < 0.6                                    # short runs
    bits[4 * q:5 * q] = rng.random(q) < 0.9                                    # straddles 2^32 (4q = 2^20): runs of ~10
    bits[5 * q:6 * q] = (rel[5 * q:6 * q] % 64) != 63                          # word-aligned runs
    bits[6 * q:7 * q] = (rel[6 * q:7 * q] % 128) != 0                          # two-word runs
    bits[7 * q:] = rng.random(len(idx) - 7 * q) < 0.9993                       # runs across many words
    at = TWO32 - W0
    bits[at - 700:at + 900] = True                                             # one long run over the boundary ...
    bits[at - 701] = False
    bits[at + 900] = False
    bits[at + 1000:at + 1012] = True                                           # ... a run that begins right after
    bits[at + 1012] = False
    bits[at - 2000:at - 1900] = True                                           # a run that ends exactly at 2^32 - 1900 ...
    bits[at - 1900] = False
    bits[at - 1500:at - 1400] = True
    bits[at - 1400] = False
    words = (N + 63) // 64 + 1

    def pack(b):
        v = np.zeros(words, dtype=np.uint64)
        pk = np.packbits(b, bitorder="little")
        assert W0 % 64 == 0 and len(pk) % 8 == 0
        v[W0 // 64: W0 // 64 + len(pk) // 8] = pk.view(np.uint64)
        return v
    thrK = pack(bits)
    thrR = pack(bits & (rng.random(len(idx)) < 0.5))
    minima = pack(rng.random(len(idx)) < 0.01)
    return thrK, thrR, minima


def _compare_scan(of, g1, gf, p, pp, cuts, g2=None):
    seen = []
    for b, e in zip(cuts[:-1], cuts[1:]):
        want = of.scan(p, b, e, capture=True)
        seen.append(want["count"])
        calls = api.scan_clusters(g1, gf, pp, bwt2=g2, begin=b, end=e)
        cl, chains, left, right = calls.arrays()
        st = calls.stats
        assert dict(runs_closed=st.runs_closed, clusters=st.clusters, bases_in_runs=st.bases_in_runs,
                    survivors=st.survivors, chains=st.chains) == want["stats"]
        assert np.array_equal(np.array(list(st.hist), dtype=np.uint64), want["hist"])
        assert np.array_equal(cl, want["clusters"])
        assert np.array_equal(chains, want["chains"])
        assert np.array_equal(left, want["left"])
        assert np.array_equal(right, want["right"])
        # the compact wire image (begin_lo / begin_hi) of the same call set unpacks to the same records
        if want["count"]:
            job = api.scan_clusters_begin(g1, gf, pp, bwt2=g2, begin=b, end=e, device_only=True, wire=True)
            job.wait()
            ptr, length = job.device_wire()
            wire = np.empty(length, dtype=np.uint8)
            sim._hip_copy_to_host(wire, ptr, length)
            job.finish()
            assert wire[:8].tobytes() == b"EBCALLS4"
            assert bytes(api.Calls.unpack(wire.tobytes()).blob()) == bytes(calls.blob())
            begins = np.frombuffer(cl.tobytes(), dtype=np.uint64).reshape(-1, 4)[:, 0]
            seen.append(-int((begins >= np.uint64(TWO32)).sum()))
    return seen


@pytest.mark.parametrize("mode", [1, 3])
def test_scan_window_across_2p32_vs_oracle(big, oracle, oracle_big, mode):
    rng = np.random.Generator(np.random.PCG64(40 + mode))
    thrK, thrR, minima = _window_flags(rng)
    da = None
    if mode == 3:
        da = rng.integers(0, 1 << 63, size=len(thrK), dtype=np.uint64) ^ (rng.integers(0, 2, size=len(thrK), dtype=np.uint64) << np.uint64(63))
    p, pp = _oracle.Params.from_args([]), api.make_params([])
    of = _oracle.OracleFlags(oracle, oracle_big, None, vectors=(thrK, thrR, minima, da))
    gf = api.Flags.from_host(N, thrK, thrR, minima, da)
    whole = _compare_scan(of, big["bwt"], gf, p, pp, [W0, W1])
    assert whole[0] > 1000 and -whole[1] > 300                       # thousands of records, hundreds with begin >= 2^32
    parts = _compare_scan(of, big["bwt"], gf, p, pp, [W0, TWO32 - 37, TWO32, TWO32 + 64 * 5 + 1, W1])
    assert sum(x for x in parts if x > 0) == whole[0]                # the cuts (one exactly at 2^32) partition the clusters
    # nothing closes outside the window
    out = api.scan_clusters(big["bwt"], gf, pp, begin=0, end=W0)
    assert out.stats.runs_closed == 0 and len(out) == 0


def test_scan_window_two_bwts_merged_past_2p32(big, oracle, host_text):
    """Mode 2: two BWTs of ~2.15 G symbols each whose merge by the document array is the random string; the flags live in
    merged coordinates (past 2^32), the symbols in the BWTs' own: da_rank groups and the merged scan planes past 2^32."""
    torch = big["torch"]
    rng = np.random.Generator(np.random.PCG64(52))
    thrK, thrR, minima = _window_flags(rng)
    da = rng.integers(0, 1 << 63, size=len(thrK), dtype=np.uint64) ^ (rng.integers(0, 2, size=len(thrK), dtype=np.uint64) << np.uint64(63))
    da[N // 64 + 1:] = 0
    if N % 64:
        da[N // 64] &= np.uint64((1 << (N % 64)) - 1)
    da_bits = np.unpackbits(da.view(np.uint8), bitorder="little")[:N].astype(bool)
    sym1, sym2 = host_text[~da_bits].copy(), host_text[da_bits].copy()
    sym1[:4] = ord("#")
    sym2[:4] = ord("#")
    del da_bits
    thr = min(os.cpu_count() or 1, 16)
    o1, o2 = oracle.index(sym1, threads=thr), oracle.index(sym2, threads=thr)
    g1, g2 = api.DnaBwt(sym1), api.DnaBwt(sym2)
    assert g1.n + g2.n == N
    p, pp = _oracle.Params.from_args([]), api.make_params([])
    of = _oracle.OracleFlags(oracle, o1, o2, vectors=(thrK, thrR, minima, da))
    gf = api.Flags.from_host(N, thrK, thrR, minima, da)
    whole = _compare_scan(of, g1, gf, p, pp, [W0, W1], g2=g2)
    assert whole[0] > 1000 and -whole[1] > 300
    parts = _compare_scan(of, g1, gf, p, pp, [W0, TWO32, W1], g2=g2)
    assert sum(x for x in parts if x > 0) == whole[0]
    del torch


# ---------------------------------------------------------------------------------------------------------------------
REF_JSON = os.path.join(ROOT, "profiles", "r02_reference_past_2p32.json")


def test_whole_pipeline_past_2p32_equals_the_reference_binary():
    """4.35 G positions end to end on the GPU against the unmodified reference binary (scripts/ref_past_2p32.py ran it in
    the build container on the same bytes: the input's SHA-256 is compared first)."""
    torch = pytest.importorskip("torch")
    assert os.path.exists(REF_JSON), "profiles/r02_reference_past_2p32.json is missing: run scripts/ref_past_2p32.py in the build container"
    with open(REF_JSON) as f:
        ref = json.load(f)
    i = ref["input"]
    text = sim.replicated_ebwt_gpu(i["n_reads_base"], i["read_len"], i["coverage_base"], i["copies"], i["seed"])
    n = int(text.numel())
    assert n == i["n"] and n > TWO32
    host = np.empty(n, dtype=np.uint8)
    sim._hip_copy_to_host(host, text.data_ptr(), n)
    assert hashlib.sha256(host).hexdigest() == i["sha256"], "the GPU generator and the host builder disagree on the input"
    del host
    p = api.make_params()
    bwt = api.DnaBwt(None, device_ptr=text.data_ptr(), n=n)
    del text
    torch.cuda.empty_cache()
    flags = api.navigate_one_bwt(bwt, p)
    rc = ref["reference_counters"]
    assert (flags.stats.leaves, flags.stats.nodes, flags.stats.minima) == (rc["leaves"], rc["nodes"], rc["minima"])
    assert flags.stats.lcp_values == n
    calls = api.scan_clusters(bwt, flags, p)
    assert calls.stats.clusters == rc["clusters"]
    got, cluster_nr, events = calls.format_snp(p, False)
    assert events == got.count(b">") == ref["snp_records"] and len(got) == ref["snp_bytes"]
    assert hashlib.sha256(got).hexdigest() == ref["snp_sha256"]
    v = calls.view()
    begins = np.frombuffer(C.string_at(v.clusters, v.count * 32), dtype=np.uint64).reshape(-1, 4)[:, 0]
    assert int((begins >= np.uint64(TWO32)).sum()) > 100            # clusters of superblock 1 are in the output
    # and the same call set cut into two shards at 2^32 (the N-GPU path's ownership rule) gives the same text
    text2, cn, ev = b"", 1, 0
    for b, e in ((0, TWO32), (TWO32, n)):
        t, cn, ev = api.scan_clusters(bwt, flags, p, begin=b, end=e).format_snp(p, False, cn, ev)
        text2 += t
    assert text2 == got

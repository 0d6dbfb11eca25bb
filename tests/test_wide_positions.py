"""Host-side checks of everything that carries a BWT position wider than 32 bits (configs[1..4] all live past 2^32;
the reference is 64-bit throughout with 2^32-symbol superblocks, dna_string.hpp:36-38,140-152):
  * the traversal's packed stack records (eb_nav_common.h: 40-bit positions, saturating 16-bit depth) round trip through
    the same __host__ __device__ code the kernels run, with every field beyond 2^32;
  * the compact wire images of a call set (EBCALLS4: begin_lo + begin_hi; EBCALLS3: u64 begin) unpack to the friendly
    records with begins >= 2^32, and the writer numbers and prints them.
No GPU needed."""
import ctypes as C
import struct

import numpy as np
import pytest

from ebwt2indel_amd import _lib, api

M40 = (1 << 40) - 1


def _roundtrip(nf, fields, depth):
    L = api.lib()
    f = np.asarray(fields, dtype=np.uint64)
    out = np.zeros(nf, dtype=np.uint64)
    words = np.zeros(16, dtype=np.uint64)
    d_out, nw = C.c_uint64(0), C.c_uint32(0)
    rc = L.eb_selftest_pack_fields(nf, f.ctypes.data, depth, out.ctypes.data, C.byref(d_out), words.ctypes.data, C.byref(nw))
    assert rc == 0, L.eb_last_error()
    return out, d_out.value, words[: nw.value].copy()


@pytest.mark.parametrize("nf,nwords", [(6, 4), (12, 8), (4, 4)])
def test_pack_fields_roundtrip_beyond_32_bits(nf, nwords):
    rng = np.random.Generator(np.random.PCG64(nf))
    cases = [
        [M40] * nf,                                   # all ones: neighbours must not bleed into each other
        [0] * nf,
        [(1 << 32) + k for k in range(nf)],           # just past the first superblock
        [(1 << 32) - 1 - k for k in range(nf)],
        [(k % 2) * M40 for k in range(nf)],           # alternating all-ones / zero
        [(1 << 39) | (1 << k) for k in range(nf)],
        [7_550_000_000 - 3 * k for k in range(nf)],   # configs[1]
        [151_000_000_000 + 1_000_003 * k for k in range(nf)],  # configs[4]
    ] + [list(rng.integers(0, 1 << 40, size=nf, dtype=np.uint64)) for _ in range(200)]
    for fields in cases:
        for depth in (0, 1, 15, 16, 30, 31, 150, 65534, 65535):
            out, d, words = _roundtrip(nf, fields, depth)
            assert len(words) == nwords
            assert [int(v) for v in out] == [int(v) for v in fields], (fields, depth)
            assert d == depth
    # the depth saturates instead of wrapping (it is only ever compared with K and k_right < 65535)
    for depth in (65536, 1 << 20, (1 << 40) + 5):
        out, d, _ = _roundtrip(nf, [M40] * nf, depth)
        assert d == 65535 and all(int(v) == M40 for v in out)


def test_pack_fields_layout_is_40_bit_dense():
    # field k occupies bits [40k, 40k+40) of the little-endian word array; the depth follows the last field
    for nf in (4, 6, 12):
        for k in range(nf):
            fields = [0] * nf
            fields[k] = (1 << 39) | 1
            _, _, words = _roundtrip(nf, fields, 0)
            big = sum(int(w) << (64 * i) for i, w in enumerate(words))
            assert big == ((1 << 39) | 1) << (40 * k)
        _, _, words = _roundtrip(nf, [0] * nf, 0xABCD)
        big = sum(int(w) << (64 * i) for i, w in enumerate(words))
        assert big == 0xABCD << (40 * nf)


def test_pack_fields_rejects_unknown_record_kind():
    L = api.lib()
    f = np.zeros(8, dtype=np.uint64)
    d = C.c_uint64(0)
    assert L.eb_selftest_pack_fields(5, f.ctypes.data, 0, f.ctypes.data, C.byref(d), None, None) == _lib.EB_ERR_ARG


# ---------------------------------------------------------------------------------------------------------------- wire images
HDR = struct.Struct("<8Q")


def _ctx_words(bases_2bit, n_words):
    w = [0] * n_words
    for e, b in enumerate(bases_2bit):
        w[e >> 5] |= (b & 3) << (2 * (e & 31))
    return w


def _make_clusters():
    """Three two-allele clusters: begins in superblock 0, straddling 2^32, and deep inside superblock 1 / 34."""
    rng = np.random.Generator(np.random.PCG64(5))
    begins = [(1 << 32) - 7, (1 << 32) + 123_456_789, 7_549_999_000, 150_999_999_000]
    out = []
    for i, b in enumerate(begins):
        left0 = [int(v) for v in rng.integers(0, 4, size=31)]
        left1 = list(left0)
        left0[0], left1[0] = 0, 2                       # base 0 of a chain = the frequent symbol itself (A / G): a SNP
        right = [int(v) for v in rng.integers(0, 4, size=30)]
        right[1] = (right[0] + 1) & 3                   # not low-complexity
        out.append(dict(begin=b, len=12 + i, chains=[(0, 0, 31, 6 + i, left0), (0, 2, 31, 5, left1)], right=right, right_len=30))
    return out


def _wire4(clusters, k_left=31, k_right=30):
    lw, rw = (k_left + 31) // 32, (k_right + 31) // 32
    n_chains = sum(len(c["chains"]) for c in clusters)
    img = bytearray(HDR.pack(0x34534C4C41434245, len(clusters), n_chains, k_left, k_right, 0, 0, 0))
    for c in clusters:
        img += struct.pack("<IBBHHH", c["begin"] & 0xFFFFFFFF, c["begin"] >> 32, len(c["chains"]) | (1 << 4), c["len"], c["right_len"], 0)
    for c in clusters:
        for side, letter, ln, sup, _ in c["chains"]:
            img += struct.pack("<HH", side | (letter << 1) | (ln << 3), sup)
    while len(img) % 8:
        img += b"\0"
    for c in clusters:
        for *_, bases in c["chains"]:
            img += struct.pack(f"<{lw}Q", *_ctx_words(bases, lw))
    for c in clusters:
        img += struct.pack(f"<{rw}Q", *_ctx_words(c["right"], rw))
    return bytes(img)


def _wire3(clusters, k_left=31, k_right=30):
    lw, rw = (k_left + 31) // 32, (k_right + 31) // 32
    n_chains = sum(len(c["chains"]) for c in clusters)
    img = bytearray(HDR.pack(0x33534C4C41434245, len(clusters), n_chains, k_left, k_right, 0, 0, 0))
    for c in clusters:
        img += struct.pack("<QIBBH", c["begin"], c["len"], len(c["chains"]), 1, c["right_len"])
    for c in clusters:
        for side, letter, ln, sup, _ in c["chains"]:
            img += struct.pack("<BBHI", side, b"ACGT"[letter], ln, sup)
    for c in clusters:
        for *_, bases in c["chains"]:
            img += struct.pack(f"<{lw}Q", *_ctx_words(bases, lw))
    for c in clusters:
        img += struct.pack(f"<{rw}Q", *_ctx_words(c["right"], rw))
    return bytes(img)


@pytest.mark.parametrize("maker", [_wire4, _wire3], ids=["EBCALLS4", "EBCALLS3"])
def test_wire_images_carry_begins_beyond_32_bits(maker):
    cl = _make_clusters()
    if maker is _wire4:
        cl = [c for c in cl if c["begin"] < (1 << 40)]
    calls = api.Calls.unpack(maker(cl))
    v = calls.view()
    assert v.count == len(cl) and v.n_chains == 2 * len(cl)
    chain = 0
    for i, c in enumerate(cl):
        r = v.clusters[i]
        assert r.begin == c["begin"] and r.end == c["begin"] + c["len"], (hex(r.begin), hex(c["begin"]))
        assert r.first_chain == chain and r.n_slots == 2 and r.right_found == 1 and r.right_len == 30
        for s, (side, letter, ln, sup, bases) in enumerate(c["chains"]):
            k = v.chains[chain + s]
            assert (k.side, k.sym, k.len, k.support) == (side, b"ACGT"[letter], ln, sup)
            assert v.left[chain + s] == _ctx_words(bases, 1)[0]
        assert v.right[i] == _ctx_words(c["right"], 1)[0]
        chain += 2
    # the friendly blob of the same call set round-trips too, and the writer prints every cluster (one SNP each)
    again = api.Calls.unpack(calls.pack())
    p = api.make_params()
    text, cluster_nr, events = again.format_snp(p, False)
    assert cluster_nr == 1 + len(cl) and events == 2 * len(cl)      # cluster_nr starts at 1 (ebwt2InDel.cpp:52)
    assert text.count(b"_type:_SNP_event:A/G") == 2 * len(cl)
    assert text == calls.format_snp(p, False)[0]


def test_every_slot_once_check_by_power_sums():
    """eb_nav_covers_every_slot_once: count, sum and sum of squares (mod 2^64) of the written LCP positions against the
    closed forms for 1 .. n-1 -- also past 2^32 and at configs[4] size, where the products wrap many times."""
    from ebwt2indel_amd import _lib
    L = api.lib()
    M = (1 << 64) - 1
    for n in (2, 3, 1000, (1 << 32) + 12345, 7_550_000_000, 151_000_000_000, (1 << 40) - 1):
        st = _lib.NavStats()
        st.n = n
        st.lcp_values = n
        st.lcp_pos_sum = (n * (n - 1) // 2) & M
        st.lcp_pos_sq_sum = ((n - 1) * n * (2 * n - 1) // 6) & M
        assert L.eb_nav_covers_every_slot_once(st) == 1, n
        # one slot written twice, another never: the count still says n
        a, b = n // 3 + 1, n // 2 + 1
        if a != b and b < n:
            st.lcp_pos_sum = (st.lcp_pos_sum + a - b) & M
            st.lcp_pos_sq_sum = (st.lcp_pos_sq_sum + a * a - b * b) & M
            assert L.eb_nav_covers_every_slot_once(st) == 0, n
        st.lcp_pos_sum = (n * (n - 1) // 2) & M
        st.lcp_pos_sq_sum = ((n - 1) * n * (2 * n - 1) // 6) & M
        st.lcp_values = n - 1
        assert L.eb_nav_covers_every_slot_once(st) == 0, n

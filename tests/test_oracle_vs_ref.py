"""Differential runs: CPU oracle vs the reference binary itself on fresh seeded inputs.
Runs only where oracle/_ref/ebwt2InDel_ref exists (the build container); the GPU box has
no reference, there the committed golden fixtures pin the oracle."""
import os
import subprocess

import numpy as np
import pytest

from ebwt2indel_amd import sim
from tests import _oracle

pytestmark = pytest.mark.skipif(not os.path.exists(_oracle.REF_BIN), reason="reference binary not built here")


def _ref(tmp_path, args):
    out = tmp_path / "ref.snp"
    rc = subprocess.run([_oracle.REF_BIN] + args + ["-o", str(out)], stdout=subprocess.DEVNULL).returncode
    return out.read_bytes(), rc


@pytest.mark.parametrize("seed,extra", [(101, []), (102, ["-m", "4", "-k", "18"]), (103, ["-L", "36", "-R", "28", "-g", "4", "-q", "2"])])
def test_mode1_fresh(oracle, tmp_path, seed, extra):
    reads, _, _ = sim.diploid_sample(20000, 5000, 90, seed=seed, snp_every=350, indel_every=900)
    e = sim.build_ebwt(reads)
    (tmp_path / "x.ebwt").write_bytes(e.tobytes())
    want, rc = _ref(tmp_path, ["-1", str(tmp_path / "x.ebwt")] + extra)
    got, _ = oracle.run(e, params=_oracle.Params.from_args(extra))
    assert rc == 0 and want.count(b">") > 20
    assert got == want


@pytest.mark.parametrize("seed,het,extra", [(201, False, []), (202, True, []), (203, True, ["-m", "2", "-v", "3", "-c", "3"])])
def test_modes23_fresh(oracle, tmp_path, seed, het, extra):
    r0, r1 = sim.two_samples(15000, 3500, 80, seed=seed, het_second=het, snp_every=250, indel_every=700)
    a, b = sim.build_ebwt(r0), sim.build_ebwt(r1)
    m, da = sim.build_merged_ebwt(r0, r1)
    for name, x in (("a", a), ("b", b), ("m", m), ("da", da)):
        (tmp_path / name).write_bytes(x.tobytes())
    p = _oracle.Params.from_args(extra)
    want2, rc2 = _ref(tmp_path, ["-1", str(tmp_path / "a"), "-2", str(tmp_path / "b")] + extra)
    got2, _ = oracle.run(a, bwt2=b, params=p)
    assert rc2 == 0 and got2 == want2 and want2.count(b">") > 10
    want3, rc3 = _ref(tmp_path, ["-1", str(tmp_path / "m"), "-d", str(tmp_path / "da")] + extra)
    got3, _ = oracle.run(m, da=da, params=p)
    assert rc3 == 0 and got3 == want3
    if not het:
        assert want2 == want3      # restricted equivalence of modes 2 and 3


def test_merged_bwt_consistency():
    """The merged eBWT restricted to DA==0 / DA==1 equals the two single-collection eBWTs."""
    r0, r1 = sim.two_samples(3000, 400, 60, seed=5)
    a, b = sim.build_ebwt(r0), sim.build_ebwt(r1)
    m, da = sim.build_merged_ebwt(r0, r1)
    assert np.array_equal(m[da == ord("0")], a)
    assert np.array_equal(m[da == ord("1")], b)


@pytest.mark.parametrize("reads", [[b"A"], [b"A"] * 4, [b"ACGTTGCAACGT"], [b"ACGTACGGTCAATCGGCTAAGCTTAGCCTAGGATCCA"] * 12,
                                   [b"A", b"C", b"G", b"T"] * 5, [b"A" * 60] * 8 + [b"A" * 59 + b"C"] * 8],
                         ids=["one_base", "same_base_x4", "one_read", "identical_reads", "length_one_reads", "homopolymer"])
def test_degenerate_inputs_against_reference(oracle, tmp_path, reads):
    """Tiny / duplicate-only / homopolymer collections.  On cluster-free inputs the reference dies with SIGFPE after the
    (empty) .snp is complete: only the file is compared."""
    e = sim.build_ebwt(reads)
    (tmp_path / "x.ebwt").write_bytes(e.tobytes())
    want, rc = _ref(tmp_path, ["-1", str(tmp_path / "x.ebwt")])
    assert rc in (0, -8)
    got, _ = oracle.run(e)
    assert got == want


def test_short_document_array_against_reference(oracle, tmp_path):
    """A -d file shorter than the BWT: the reference's read() fails and its one-byte buffer keeps the last byte read, so the
    rest of the DA repeats it (ebwt2InDel.cpp:1503-1508).  Cuts inside printed clusters, so that what fills the tail
    decides which side those clusters' last rows count for: ending on '1' and on '0' must differ somewhere."""
    r0, r1 = sim.two_samples(15000, 3500, 80, seed=211, het_second=True, snp_every=250, indel_every=700)
    m, da = sim.build_merged_ebwt(r0, r1)
    (tmp_path / "m").write_bytes(m.tobytes())
    p = _oracle.Params.from_args([])
    full = _oracle.OracleFlags(oracle, oracle.index(m), None, p.K, p.k_right, da).scan(p, 0, len(m), capture=True)
    cl = np.frombuffer(full["clusters"].tobytes(), dtype=np.uint64).reshape(-1, 4)
    cuts = [int(b + (e - b) // 2) for b, e in cl[len(cl) // 3: len(cl) // 3 + 6, :2]] + [len(da) // 3 + 7]
    differ = 0
    for cut in cuts:
        outs = []
        for last in (b"1", b"0"):
            d = bytearray(da.tobytes()[:cut])
            d[-1] = last[0]
            (tmp_path / "da").write_bytes(bytes(d))
            want, rc = _ref(tmp_path, ["-1", str(tmp_path / "m"), "-d", str(tmp_path / "da")])
            got, _ = oracle.run(m, da=bytes(d))
            assert rc == 0 and got == want
            outs.append(want)
        differ += outs[0] != outs[1]
    assert differ > 0

"""N>1 path on CPU: two gloo ranks shard the phase-4 range, each rank's cluster records (captured from the
oracle -- no GPU here, and the product has no CPU compute path) go through the product's pack ->
torch.distributed gather -> unpack -> writer chain; rank 0 must reproduce the golden .snp byte for byte."""
import os
import socket

import pytest

from tests.conftest import CASES, golden_bytes


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out_path, arena=False):
    import torch.distributed as dist

    from ebwt2indel_amd import api
    from ebwt2indel_amd import distributed as ebd
    from tests import _oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = _oracle.Oracle()
        p = _oracle.Params.from_args(case["args"])
        inp = case["inputs"]
        x1 = orc.index(golden_bytes(inp["bwt1"]), p.term)
        x2 = orc.index(golden_bytes(inp["bwt2"]), p.term) if case["mode"] == 2 else None
        da = golden_bytes(inp["da"]) if case["mode"] == 3 else None
        fl = _oracle.OracleFlags(orc, x1, x2, p.K, p.k_right, da)
        n = x1.n + (x2.n if x2 is not None else 0)
        b, e = ebd.shard_range(n, rank, world)
        rec = fl.scan(p, b, e, capture=True)
        calls = api.Calls.unpack(_oracle.pack_calls(rec, p.k_left, p.k_right))     # product container
        if arena:
            # round 5: every rank puts its records into its slot of a host arena all ranks map (on GPUs: downloaded over the
            # rank's own PCIe link), rank 0 reads the slots -- no collective on the data path.  Three steps through a ring of two
            # slots: the third must wait until rank 0 has released the first.
            ar = ebd.NodeArena(cap=len(calls.pack()) + 100, depth=2, pin=False)
            parts = None
            for step in range(3):
                ar.publish_bytes(calls.pack())
                if rank == 0:
                    parts = [bytes(v) for v in ar.collect()]
                    ar.release()
            dist.barrier()
            ar.close()
        else:
            parts = ebd.gather_packed(calls.pack())
        if rank == 0:
            assert len(parts) == world
            text, cluster_nr, events, total = ebd.write_gathered(parts, api.make_params(case["args"]), case["mode"] != 1)
            with open(out_path, "wb") as f:
                f.write(text)
        else:
            assert parts is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["m1_default", "m1_edge_default", "m2_het_default", "m3_hom_default"])
def test_two_rank_gather_reproduces_golden(tmp_path, name):
    import torch.multiprocessing as mp

    case = [c for c in CASES if c["name"] == name][0]
    out = tmp_path / "out.snp"
    mp.spawn(_worker, args=(2, _free_port(), case, str(out)), nprocs=2, join=True)
    assert out.read_bytes() == golden_bytes(case["snp"])


@pytest.mark.parametrize("name", ["m1_default", "m3_het_default"])
def test_eight_rank_gather_reproduces_golden(tmp_path, name):
    """World size 8 (the node north_star names) over gloo: eight shards of the scan range, eight packed call sets gathered to
    rank 0 in rank order, one running cluster_nr across them."""
    import torch.multiprocessing as mp

    case = [c for c in CASES if c["name"] == name][0]
    out = tmp_path / "out.snp"
    mp.spawn(_worker, args=(8, _free_port(), case, str(out)), nprocs=8, join=True)
    assert out.read_bytes() == golden_bytes(case["snp"])


@pytest.mark.parametrize("world", [2, 8])
def test_host_arena_reproduces_golden(tmp_path, world):
    """The same with the records handed over through the node's shared host arena (ebd.NodeArena) instead of a gather."""
    import torch.multiprocessing as mp

    case = [c for c in CASES if c["name"] == "m1_default"][0]
    out = tmp_path / "out.snp"
    mp.spawn(_worker, args=(world, _free_port(), case, str(out), True), nprocs=world, join=True)
    assert out.read_bytes() == golden_bytes(case["snp"])


def test_shard_ranges_tile_the_bwt():
    from ebwt2indel_amd import distributed as ebd
    for n in (0, 1, 63, 64, 65, 1000, 1_010_000, 7_550_000_000):
        for world in (1, 2, 3, 8):
            cuts = [ebd.shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            assert all(b <= e for b, e in cuts)
            assert all(c[0] % 64 == 0 or c[0] == n for c in cuts)

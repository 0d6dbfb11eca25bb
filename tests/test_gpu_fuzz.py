"""A seeded slice of the randomised differential run (tests/fuzz_vs_oracle.py): random small inputs in all three modes, with and
without sequencing errors / reverse complements, under random options -- GPU .snp == CPU oracle's, byte for byte."""
import pytest

from tests import fuzz_vs_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [1000, 2000, 3000])
def test_random_inputs_and_options_equal_the_oracle(oracle, first):
    seen = 0
    for seed in range(first, first + 12):
        r = fuzz_vs_oracle.one_case(oracle, seed)
        assert r["same"], r
        seen += r["records"]
    assert seen > 50


@pytest.mark.parametrize("seed", [501, 502, 503])
def test_random_medium_size_inputs_equal_the_oracle(oracle, seed):
    """2-20 M positions from the GPU generator (several tiles per workgroup, many batches, full traversal stacks), random mode and options."""
    r = fuzz_vs_oracle.medium_case(oracle, seed)
    assert r["same"] and r["records"] > 100, r

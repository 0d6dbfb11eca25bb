"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/ebwt2indel.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ebwt2indel_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "ebwt2indel.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(eb_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    L = api.lib()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ebwt2indel.h but not exported"
    assert sorted(_lib.ABI_SYMBOLS) == names


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.Cluster) == 32
    assert C.sizeof(_lib.Chain) == 12
    assert C.sizeof(_lib.CallsView) == 56
    assert C.sizeof(_lib.Params) == 36
    assert C.sizeof(_lib.IndexInfo) == 64


def test_params_zero_means_default():
    p = api.make_params(["-m", "0", "-k", "0", "-L", "0", "-R", "0", "-g", "0", "-v", "0", "-c", "0"])
    assert (p.k_left, p.k_right, p.K, p.max_gap, p.max_snvs, p.mcov_out, p.complexity) == (31, 30, 16, 10, 2, 3, 20)
    p = api.make_params(["-R", "50"])
    assert p.complexity == 20          # complexity_def is frozen at 30-10 (ebwt2InDel.cpp:64)
    p = api.make_params(["-q", "2", "-t", "36"])
    assert p.max_variants_per_position == 2 and p.term == 36


@pytest.mark.skipif(api.device_count() > 0, reason="checks the no-GPU behaviour")
def test_no_gpu_fails_loudly():
    with pytest.raises(_lib.EbError) as e:
        api.DnaBwt(b"ACGT#ACGT")
    assert e.value.code == _lib.EB_ERR_DEVICE
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(_lib.EbError) as e:
        api.Flags.from_host(64, np.zeros(1, np.uint64), np.zeros(1, np.uint64), np.zeros(1, np.uint64))
    assert e.value.code == _lib.EB_ERR_DEVICE


def test_cli_binary_exists_and_prints_help():
    import subprocess
    assert os.path.exists(_lib.CLI_PATH)
    r = subprocess.run([_lib.CLI_PATH, "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "ebwt2InDel [options]" in r.stdout and "-1 <arg>" in r.stdout

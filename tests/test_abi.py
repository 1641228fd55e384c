"""The C-ABI library loads and exports every symbol include/clane_hip.h declares (no GPU needed,
no compute calls)."""
import re
from pathlib import Path

import pytest

from clane_amd import _hip

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "clane_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(clane_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert _hip.LIB_PATH.exists(), "run __graft_entry__.build() first"
    assert _hip.LIB_PATH.parent == ROOT / "clane_amd"


def test_every_declared_symbol_is_exported_and_bound():
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(_hip.SIGNATURES) == declared
    lib = _hip.load_library()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.clane_abi_version() == _hip.ABI_VERSION
    assert lib.clane_spmm_partials_len(10, 0) == 1          # 32 consecutive rows per workgroup
    assert lib.clane_spmm_partials_len(200_000, 7) == 6250 + 7   # small graphs: 32 rows per workgroup
    assert lib.clane_spmm_partials_len(2_000_000, 7) == 31250 + 7  # ~32k workgroups: 64 rows each
    assert lib.clane_spmm_partials_len(10_000_000, 0) == 39063     # capped at 256 rows per workgroup
    assert lib.clane_reduce_ws_len() >= 2 * 1024 + 2


def test_bound_argument_counts_match_the_header():
    text = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "clane_hip.h").read_text(), flags=re.S)
    seen = 0
    for name, params in re.findall(r"\b(clane_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        params = params.strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert len(_hip.SIGNATURES[name][1]) == n, name
        seen += 1
    assert seen == len(_hip.SIGNATURES)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_hip.ClaneHipError, match="no CPU fallback"):
        _hip.load_library(tmp_path / "libclane_hip.so")


def test_argument_validation_reaches_last_error():
    # invalid arguments are rejected on the host before any launch: safe without a GPU
    lib = _hip.load_library()
    rc = lib.clane_spmm_update_f32(None, None, None, 4, 0, None, 8, None, 8, 0.5, None, 8, 8, 0, 0, None, None, None,
                                   None)
    assert rc == -1 and b"delta_partials" in lib.clane_last_error()
    rc = lib.clane_row_sqnorm_f32(None, 4, 0, 0, None, None)
    assert rc == -1 and b"bad shape" in lib.clane_last_error()
    rc = lib.clane_edge_score_f32(None, None, 4, 0, None, 8, 8, 7, None, None, None, 0, 0, None, 0, None)
    assert rc == -1 and b"unknown mode" in lib.clane_last_error()

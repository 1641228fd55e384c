import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str):
    return np.load(GOLDEN / name, allow_pickle=True)


def write_data_root(root: Path, vertex_ids, edge_src, edge_dst, C=None):
    """Write the reference's on-disk graph format (V, E, optional C.npy)."""
    root.mkdir(parents=True, exist_ok=True)
    (root / "V").write_text("\n".join(str(v) for v in vertex_ids) + "\n")
    (root / "E").write_text("\n".join(f"{s}\t{d}" for s, d in zip(edge_src, edge_dst)) + "\n")
    if C is not None:
        np.save(root / "C.npy", C)
    return root


@pytest.fixture
def karate_root(tmp_path):
    g = load_golden("g2_karate_csr.npz")
    return write_data_root(tmp_path / "karate", g["vertex_ids"], g["edge_src"], g["edge_dst"])


def csr_from_golden_edges(vertex_ids, edge_src, edge_dst):
    from oracle import clane_oracle as O
    first = {}
    for i, v in enumerate(vertex_ids):
        first.setdefault(str(v), i)
    src = np.array([first[str(s)] for s in edge_src], dtype=np.int64)
    dst = np.array([first[str(d)] for d in edge_dst], dtype=np.int64)
    return O.build_csr(len(vertex_ids), src, dst)

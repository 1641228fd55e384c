"""CPU restatement of CLANE's iterative embedding path -- TEST INFRASTRUCTURE ONLY.

This module is the parity oracle for the HIP path in ``clane_amd``.  It is a
from-scratch restatement, in PyTorch-CPU ops, of what the reference computes on
its CPU path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; nothing under ``clane_amd/`` does, and the
product path raises when the HIP library is missing instead of coming here.

Parity status: PINNED.  ``oracle/make_goldens.py`` imports the reference from
``/root/reference`` (in the build container only) and writes the fixtures under
``tests/golden/``; ``tests/test_oracle_goldens.py`` checks every function here
against them.

Reference citations (``/root/reference``):

* file formats / index assignment ......... ``clane/graph.py:43-47,72-89``
* adjacency: src=row, dst=col, coalesced .. ``clane/graph.py:104-116``
* CosineSimilarity (GLOBAL denominators) .. ``clane/similarity.py:26-37``
* build_P: per-source-row softmax ......... ``clane/graph.py:118-128``
* sweep ``z_v = x_v + gamma * P_v Z[nbrs]`` ``clane/embedder.py:84-94``
* propagate / iterate tolerance machine ... ``clane/embedder.py:45-69,77-108``
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch


# --------------------------------------------------------------------------
# graph.py:43-47, 72-89 -- the V / E text files
# --------------------------------------------------------------------------
def read_graph_files(data_root: Path) -> Tuple[List[str], np.ndarray, np.ndarray]:
    """Parse ``V`` (ids, one per line) and ``E`` (``src\\tdst`` per line).

    Returns (vertex_ids, src_idx, dst_idx) with one entry per *line* of ``E``
    (duplicates kept: ``len(g.E)`` counts lines, graph.py:78-89).  The vertex
    index is the position of the FIRST occurrence of the id in ``V``
    (``list.index`` semantics, graph.py:81).
    """
    data_root = Path(data_root)
    with open(data_root / "V", "r") as io:
        vertex_ids = io.read().strip().split("\n")
    with open(data_root / "E", "r") as io:
        lines = io.read().strip().split("\n")
    first = {}
    for i, vid in enumerate(vertex_ids):
        first.setdefault(vid, i)
    src = np.empty(len(lines), dtype=np.int64)
    dst = np.empty(len(lines), dtype=np.int64)
    for k, line in enumerate(lines):
        s, d = line.split("\t")          # ValueError on malformed line (graph.py:80)
        if s not in first or d not in first:
            raise ValueError(f"{s if s not in first else d!r} is not in list")
        src[k], dst[k] = first[s], first[d]
    return vertex_ids, src, dst


# --------------------------------------------------------------------------
# graph.py:104-116 -- adjacency as CSR (row = source, col = destination)
# --------------------------------------------------------------------------
def build_csr(num_vertices: int, src: np.ndarray, dst: np.ndarray):
    """``sparse_coo_tensor(...).coalesce()`` restated: sort by (src,dst), merge duplicates.

    Returns (rowptr int64 [V+1], colidx int32 [E']).  Self-loops are kept.
    """
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    key = np.unique(src * np.int64(num_vertices) + dst)
    rows = key // num_vertices
    cols = (key % num_vertices).astype(np.int32)
    rowptr = np.zeros(num_vertices + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=num_vertices), out=rowptr[1:])
    return rowptr, cols


def get_nbrs(rowptr: np.ndarray, colidx: np.ndarray, idx: int) -> np.ndarray:
    """Sorted out-neighbour indices of vertex ``idx`` (graph.py:112-116)."""
    return colidx[rowptr[idx]:rowptr[idx + 1]].astype(np.int64)


def _row_of_edge(rowptr: np.ndarray) -> torch.Tensor:
    deg = np.diff(rowptr)
    return torch.from_numpy(np.repeat(np.arange(len(deg), dtype=np.int64), deg))


# --------------------------------------------------------------------------
# similarity.py:26-37 -- CosineSimilarity, literally
# --------------------------------------------------------------------------
def cosine_similarity(v1: torch.Tensor, v2: torch.Tensor) -> torch.Tensor:
    """Row-wise dot divided by the product of the two GLOBAL Frobenius norms.

    ``v1.pow(2).sum()`` has no ``dim`` (similarity.py:37): for a batch this is
    not a per-pair cosine; for a single pair it is.
    """
    if v1.dim() == 1:
        v1 = v1.unsqueeze(0)
    if v2.dim() == 1:
        v2 = v2.unsqueeze(0)
    dots = (v1.unsqueeze(1) @ v2.unsqueeze(-1)).squeeze(1).squeeze(1)
    return dots / (v1.pow(2).sum().sqrt() * v2.pow(2).sum().sqrt())


def edge_dots(rowptr, colidx, Z: torch.Tensor, chunk: int = 1 << 20) -> torch.Tensor:
    """dot(Z[src_e], Z[dst_e]) for every CSR edge, chunked so [E,d] never materialises."""
    rows = _row_of_edge(rowptr)
    cols = torch.from_numpy(colidx.astype(np.int64))
    out = torch.empty(cols.numel(), dtype=Z.dtype)
    for a in range(0, cols.numel(), chunk):
        b = min(a + chunk, cols.numel())
        out[a:b] = (Z[rows[a:b]] * Z[cols[a:b]]).sum(1)
    return out


def global_denominator(rowptr, colidx, Z: torch.Tensor) -> float:
    """sqrt(sum_e |z_src|^2) * sqrt(sum_e |z_dst|^2) in float64 (similarity.py:37).

    Equal to sqrt(sum_v outdeg_v |z_v|^2) * sqrt(sum_v indeg_v |z_v|^2).
    """
    sq = Z.double().pow(2).sum(1)
    outdeg = torch.from_numpy(np.diff(rowptr)).double()
    indeg = torch.from_numpy(np.bincount(colidx, minlength=len(rowptr) - 1)).double()
    return math.sqrt(float((outdeg * sq).sum())) * math.sqrt(float((indeg * sq).sum()))


def segment_softmax(rowptr, scores: torch.Tensor) -> torch.Tensor:
    """Softmax of ``scores`` within each CSR row (graph.py:122-123)."""
    V = len(rowptr) - 1
    rows = _row_of_edge(rowptr)
    neg_inf = torch.full((V,), -math.inf, dtype=scores.dtype)
    rmax = neg_inf.scatter_reduce(0, rows, scores, reduce="amax", include_self=True)
    ex = (scores - rmax[rows]).exp()
    rsum = torch.zeros(V, dtype=scores.dtype).scatter_add_(0, rows, ex)
    return ex / rsum[rows]


def build_P_values(rowptr, colidx, Z: torch.Tensor, mode: str = "reference",
                   similarity: Optional[Callable] = None) -> torch.Tensor:
    """P values in CSR order = row-softmax of per-edge similarity (graph.py:118-128).

    mode "reference": the global-denominator scores the reference really produces.
    mode "per_edge" : true per-edge cosine (what its docstring describes).
    ``similarity``: if given, the literal path -- gather both endpoints, call it.
    """
    if similarity is not None:
        rows = _row_of_edge(rowptr)
        cols = torch.from_numpy(colidx.astype(np.int64))
        scores = similarity(Z[rows], Z[cols])
    else:
        dots = edge_dots(rowptr, colidx, Z)
        if mode == "reference":
            scores = dots / torch.tensor(global_denominator(rowptr, colidx, Z), dtype=Z.dtype)
        elif mode == "per_edge":
            nrm = Z.pow(2).sum(1).sqrt()
            rows = _row_of_edge(rowptr)
            cols = torch.from_numpy(colidx.astype(np.int64))
            scores = dots / (nrm[rows] * nrm[cols])
        else:
            raise ValueError(mode)
    return segment_softmax(rowptr, scores)


# --------------------------------------------------------------------------
# embedder.py:84-94 -- one Jacobi sweep
# --------------------------------------------------------------------------
def as_sparse(rowptr, colidx, P: torch.Tensor) -> torch.Tensor:
    V = len(rowptr) - 1
    idx = torch.stack([_row_of_edge(rowptr), torch.from_numpy(colidx.astype(np.int64))])
    return torch.sparse_coo_tensor(idx, P, size=(V, V), is_coalesced=True)


def sweep(rowptr, colidx, P, X: torch.Tensor, Z: torch.Tensor, gamma: float,
          P_sparse: Optional[torch.Tensor] = None):
    """Z_new[v] = X[v] + gamma * sum_e P_e Z[col_e]; rows without out-edges keep Z[v].

    Returns (Z_new, sum|Z_new - Z|)  (embedder.py:88-94).
    """
    if P_sparse is None:
        P_sparse = as_sparse(rowptr, colidx, P)
    Z_new = X + gamma * torch.sparse.mm(P_sparse, Z)
    sink = torch.from_numpy(np.diff(rowptr) == 0)
    if bool(sink.any()):
        Z_new[sink] = Z[sink]
    return Z_new, (Z_new - Z).abs().sum()


# --------------------------------------------------------------------------
# embedder.py:45-108 -- tolerance machine, propagate, iterate
# --------------------------------------------------------------------------
class Tolerence:
    """embedder.py:45-54."""

    def __init__(self, initial_value: int):
        self.initial_value = initial_value
        self.value = initial_value

    def reset(self):
        self.value = self.initial_value

    def endure(self):
        self.value -= 1


class OracleEmbedder:
    """State + control flow of ``Embedder`` (embedder.py:12-108) over CSR arrays."""

    def __init__(self, rowptr, colidx, X: torch.Tensor, gamma: float = 0.76,
                 tolerence: int = 10, mode: str = "reference", save_history: bool = False,
                 max_sweeps: Optional[int] = None, plain_c: bool = False):
        """``plain_c``: the same control flow with build_P / sweep done by the plain-C restatement
        (oracle/clane_oracle.c: fp32, reference cosine mode only) -- the independent second oracle, and much the faster
        one at d = 1433."""
        if plain_c and (mode != "reference" or X.dtype != torch.float32):
            raise ValueError("the plain-C oracle restates the fp32 reference-mode path only")
        self.plain_c = plain_c
        self.rowptr, self.colidx = rowptr, colidx
        self.X = X
        self.Z = X.clone()
        self.gamma = gamma
        self.mode = mode
        self.tolerences = {"global": Tolerence(tolerence), "propagation": Tolerence(tolerence)}
        self.minimum_amount_updated_Z = math.inf
        self.save_history = save_history
        self.history = {"Z": []}
        self.sweep_counts: List[int] = []
        self.outer_deltas: List[float] = []
        self.sweep_deltas: List[List[float]] = []
        self.max_sweeps = max_sweeps

    def propagate(self):
        if self.plain_c:
            from . import clane_oracle_c as OC
            P, _ = OC.build_P(self.rowptr, self.colidx, self.Z)
            c_sweep = lambda: OC.sweep(self.rowptr, self.colidx, P, self.X, self.Z, self.gamma)  # noqa: E731
        else:
            P = build_P_values(self.rowptr, self.colidx, self.Z, self.mode)
            P_sparse = as_sparse(self.rowptr, self.colidx, P)
        minimum = math.inf
        tol = self.tolerences["propagation"]
        tol.reset()
        hist, deltas = [], []
        while True:
            Z_new, amount = c_sweep() if self.plain_c else sweep(self.rowptr, self.colidx, P, self.X, self.Z, self.gamma,
                                                                   P_sparse)
            self.Z = Z_new
            deltas.append(float(amount))
            if self.save_history:
                hist.append(Z_new.clone())
            if minimum > amount:
                tol.reset()
                minimum = amount
            else:
                tol.endure()
            if tol.value == 0 or (self.max_sweeps and len(deltas) >= self.max_sweeps):
                if self.save_history:
                    self.history["Z"].append(hist)
                self.sweep_counts.append(len(deltas))
                self.sweep_deltas.append(deltas)
                return P

    def iterate(self):
        tol = self.tolerences["global"]
        while True:
            prev = self.Z.clone()
            self.propagate()
            amount = (self.Z - prev).abs().sum()
            self.outer_deltas.append(float(amount))
            if self.minimum_amount_updated_Z > amount:
                tol.reset()
                self.minimum_amount_updated_Z = amount
            else:
                tol.endure()
            if tol.value == 0:
                return self.Z


def fixed_point(rowptr, colidx, P: torch.Tensor, X: torch.Tensor, gamma: float) -> torch.Tensor:
    """Dense solve of Z* = (I - gamma P)^-1 X in float64 -- size-independent property check."""
    V = len(rowptr) - 1
    Pd = as_sparse(rowptr, colidx, P.double()).to_dense()
    return torch.linalg.solve(torch.eye(V, dtype=torch.float64) - gamma * Pd, X.double())


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))

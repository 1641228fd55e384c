"""Synthetic inputs for benchmarks and scale tests: R-MAT and uniform
random directed graphs as CSR, Gaussian content embeddings.  Host utilities, not on the hot path;
torch is used as a fast RNG / sorter (on the GPU when one is present)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .partition import HostCSR


def _csr_from_keys(num_vertices: int, keys: torch.Tensor) -> HostCSR:
    keys = keys.cpu().numpy()
    rows = keys // num_vertices
    rowptr = np.zeros(num_vertices + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=num_vertices), out=rowptr[1:])
    return HostCSR(num_vertices, rowptr, (keys % num_vertices).astype(np.int32))


def rmat_csr(num_vertices: int, num_edges: int, seed: int = 1, abcd=(0.57, 0.19, 0.19, 0.05),
             device: Optional[str] = None) -> HostCSR:
    """R-MAT on 2^ceil(log2 V) ids folded ``mod V``; exactly ``num_edges`` UNIQUE directed edges
    (self-loops allowed), sorted by (src, dst)."""
    if num_edges > num_vertices * num_vertices:
        raise ValueError("more unique edges requested than the graph can hold")
    dev = torch.device(device) if device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    gen = torch.Generator(device=dev).manual_seed(seed)
    levels = max(1, int(np.ceil(np.log2(num_vertices))))
    a, b, c, _ = abcd
    have = torch.empty(0, dtype=torch.int64, device=dev)
    while have.numel() < num_edges:
        n = int((num_edges - have.numel()) * 1.25) + 1024
        src = torch.zeros(n, dtype=torch.int64, device=dev)
        dst = torch.zeros(n, dtype=torch.int64, device=dev)
        for _ in range(levels):
            r = torch.rand(n, generator=gen, device=dev)
            src = (src << 1) | (r >= a + b).long()
            dst = (dst << 1) | (((r >= a) & (r < a + b)) | (r >= a + b + c)).long()
        have = torch.unique(torch.cat([have, (src % num_vertices) * num_vertices + (dst % num_vertices)]))
    if have.numel() > num_edges:
        keep = torch.randperm(have.numel(), generator=gen, device=dev)[:num_edges]
        have = torch.sort(have[keep]).values
    return _csr_from_keys(num_vertices, have)


def uniform_csr(num_vertices: int, num_edges: int, seed: int = 0) -> HostCSR:
    """``num_edges`` distinct directed edges drawn uniformly (Cora-shaped synthetic, config 1)."""
    rng = np.random.default_rng(seed)
    keys = np.sort(rng.choice(num_vertices * num_vertices, size=num_edges, replace=False))
    return _csr_from_keys(num_vertices, torch.from_numpy(keys))


def gaussian_X(num_vertices: int, d: int, seed: int = 2, dtype=torch.float32) -> torch.Tensor:
    return torch.normal(0, 1, [num_vertices, d], generator=torch.Generator().manual_seed(seed)).to(dtype)


def bow_X(num_vertices: int, d: int, ones_per_row: float = 18.0, seed: int = 0) -> torch.Tensor:
    """Binary bag-of-words rows (Cora-like), at least one 1 per row."""
    rng = np.random.default_rng(seed)
    X = (rng.random((num_vertices, d)) < ones_per_row / d).astype(np.float32)
    X[X.sum(1) == 0, 0] = 1.0
    return torch.from_numpy(X)

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


def powerlaw_csr(num_vertices: int, num_edges: int, alpha: float = 2.1, max_degree: int = 100_000, seed: int = 5,
                 device: Optional[str] = None) -> HostCSR:
    """Zipf(alpha) out-degrees clipped to [1, max_degree] and scaled to sum ~ num_edges; destinations drawn
    proportionally to an independent Zipf in-weight.  Duplicate edges are merged and
    more are drawn until at least ``num_edges`` distinct ones exist; a random subset of exactly ``num_edges`` is kept
    (rounds 1-3 stopped at 97 %: config 4 ran with 198M edges instead of BASELINE's 200M)."""
    dev = torch.device(device) if device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    gen = torch.Generator(device=dev).manual_seed(seed)
    u = torch.rand(num_vertices, generator=gen, device=dev).clamp_min(1e-12)
    shape = u.pow(-1.0 / (alpha - 1.0)).clamp(1, max_degree)
    w = torch.rand(num_vertices, generator=gen, device=dev).clamp_min(1e-12).pow(-1.0 / (alpha - 1.0)).clamp(1, max_degree)
    cdf = torch.cumsum(w.double(), 0)
    cdf /= cdf[-1].clone()
    target, keys = float(num_edges), None
    for _ in range(8):      # hub destinations collide inside a row: draw more until num_edges survive the merge
        deg = (shape * (target / float(shape.sum()))).round().clamp(0, max_degree).long()
        src_all = torch.repeat_interleave(torch.arange(num_vertices, device=dev), deg)
        parts, step = [], 1 << 26
        for a in range(0, src_all.numel(), step):
            src = src_all[a:a + step]
            dst = torch.searchsorted(cdf, torch.rand(src.numel(), generator=gen, device=dev, dtype=torch.float64))
            parts.append(src * num_vertices + dst.clamp_max(num_vertices - 1))
        del src_all
        keys = torch.unique(torch.cat(parts))
        del parts
        if keys.numel() >= num_edges:
            break
        target *= 1.01 * num_edges / keys.numel()
    if keys.numel() < num_edges:        # the docstring's promise: never hand back a smaller graph silently
        raise RuntimeError(f"powerlaw_csr: only {keys.numel()} distinct edges after 8 rounds of drawing, {num_edges} wanted "
                           f"(|V|={num_vertices}, max_degree={max_degree}: too dense for this generator)")
    if keys.numel() > num_edges:
        keep = torch.randperm(keys.numel(), generator=gen, device=dev)[:num_edges]
        keys = torch.sort(keys[keep]).values
    return _csr_from_keys(num_vertices, keys)


def _unique_keys_csr(num_vertices: int, rows: torch.Tensor, cols: torch.Tensor) -> HostCSR:
    """CSR of the distinct (row, col) pairs, sorted."""
    return _csr_from_keys(num_vertices, torch.unique(rows.long() * num_vertices + cols.long()))


def regular_csr(num_vertices: int, lo: int, hi: int, seed: int = 11, device: Optional[str] = None) -> HostCSR:
    """Near-regular graph: every row draws lo..hi (uniform) destinations uniformly at random (the few collisions
    inside a row are merged) -- no hubs, no skew in either direction: the opposite of R-MAT."""
    dev = torch.device(device) if device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    gen = torch.Generator(device=dev).manual_seed(seed)
    deg = torch.randint(lo, hi + 1, (num_vertices,), generator=gen, device=dev)
    rows = torch.repeat_interleave(torch.arange(num_vertices, device=dev), deg)
    cols = torch.randint(0, num_vertices, (rows.numel(),), generator=gen, device=dev)
    return _unique_keys_csr(num_vertices, rows, cols)


def uniform_random_csr(num_vertices: int, num_edges: int, seed: int = 12, device: Optional[str] = None) -> HostCSR:
    """``num_edges`` (src, dst) pairs drawn uniformly, duplicates merged (Erdos-Renyi like: Poisson degrees, a quarter
    of a percent of the pairs collide at 2M / 40M)."""
    dev = torch.device(device) if device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    gen = torch.Generator(device=dev).manual_seed(seed)
    rows = torch.randint(0, num_vertices, (num_edges,), generator=gen, device=dev)
    cols = torch.randint(0, num_vertices, (num_edges,), generator=gen, device=dev)
    return _unique_keys_csr(num_vertices, rows, cols)


def star_csr(num_vertices: int, num_stars: int, star_degree: int, background_max_degree: int = 3, seed: int = 13,
             device: Optional[str] = None) -> HostCSR:
    """``num_stars`` rows of ``star_degree`` distinct destinations each (star_degree = num_vertices: a row that reads
    every vertex) over a sparse background of 0..background_max_degree random edges per row: almost all of the
    edges sit in a handful of mega-hub rows."""
    dev = torch.device(device) if device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    gen = torch.Generator(device=dev).manual_seed(seed)
    deg = torch.randint(0, background_max_degree + 1, (num_vertices,), generator=gen, device=dev)
    rows = torch.repeat_interleave(torch.arange(num_vertices, device=dev), deg)
    cols = torch.randint(0, num_vertices, (rows.numel(),), generator=gen, device=dev)
    stars = torch.randperm(num_vertices, generator=gen, device=dev)[:num_stars]
    keys = [rows.long() * num_vertices + cols.long()]
    for s_ in stars.tolist():
        dst = torch.randperm(num_vertices, generator=gen, device=dev)[:star_degree]
        keys.append(s_ * num_vertices + dst.long())
    return _csr_from_keys(num_vertices, torch.unique(torch.cat(keys)))


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

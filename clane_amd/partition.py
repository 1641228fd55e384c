"""Row partition of the graph over the GPUs of one box, and the HBM layout that goes with it.

The reference runs in one process and has no partitioning (the path
shards by rows with one exchange per sweep).  Layout chosen for RCCL over xGMI:

* every rank owns ``n_local = chunks * rows_per_chunk`` rows and keeps the FULL ``Z``
  (two ping-pong buffers) in its own HBM, so the gather in a sweep never leaves the GPU;
* rows are stored *chunk-major*:  position ``g = c*(W*Vc) + r*Vc + i``  holds row ``i`` of
  chunk ``c`` of rank ``r``.  The rows that chunk ``c`` of all ranks produce are therefore one
  contiguous span ``[c*W*Vc, (c+1)*W*Vc)`` and rank ``r``'s piece sits at offset ``r*Vc`` in
  it -- exactly the in-place form of an all-gather (send = recv + rank*count).  Splitting a
  sweep into ``chunks`` launches lets the all-gather of chunk ``c`` run over xGMI while the
  kernel of chunk ``c+1`` is still reading HBM;
* vertices are assigned to positions through an optional random permutation, which
  balances edges per rank in expectation and spreads hub rows (R-MAT / power-law inputs).

Everything here is host-side index arithmetic (numpy); it runs once per graph.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class HostCSR:
    """Adjacency in CSR on the host: row = source, col = destination, columns sorted and
    unique within a row (what ``Graph.A`` -- reference graph.py:104-110 -- coalesces to)."""
    num_vertices: int
    rowptr: np.ndarray   # int64 [V+1]
    colidx: np.ndarray   # int32 [E]

    @property
    def num_edges(self) -> int:
        return int(self.colidx.shape[0])

    def outdeg(self) -> np.ndarray:
        return np.diff(self.rowptr)

    def indeg(self) -> np.ndarray:
        return np.bincount(self.colidx, minlength=self.num_vertices).astype(np.int32)


@dataclass
class RowPartition:
    num_vertices: int
    world_size: int
    rank: int
    chunks: int
    rows_per_chunk: int
    perm: Optional[np.ndarray] = None   # vertex -> natural slot (None = identity)

    @classmethod
    def create(cls, num_vertices: int, world_size: int = 1, rank: int = 0, chunks: int = 1,
               shuffle: Optional[bool] = None, seed: int = 0) -> "RowPartition":
        if not (0 <= rank < world_size) or chunks < 1 or num_vertices < 1:
            raise ValueError(f"bad partition: V={num_vertices} world={world_size} rank={rank} chunks={chunks}")
        vc = -(-num_vertices // (world_size * chunks))
        if shuffle is None:
            shuffle = world_size > 1
        perm = np.random.default_rng(seed).permutation(num_vertices).astype(np.int64) if shuffle else None
        return cls(num_vertices, world_size, rank, chunks, vc, perm)

    # ---- sizes ------------------------------------------------------------------------
    @property
    def n_local(self) -> int:
        return self.chunks * self.rows_per_chunk

    @property
    def padded_vertices(self) -> int:
        return self.world_size * self.n_local

    # ---- index maps -------------------------------------------------------------------
    def slot_to_position(self, slot: np.ndarray) -> np.ndarray:
        """natural slot p = r*n_local + c*Vc + i  ->  chunk-major position g."""
        vc, w = self.rows_per_chunk, self.world_size
        r, l = np.divmod(slot, self.n_local)
        c, i = np.divmod(l, vc)
        return c * (w * vc) + r * vc + i

    def position_of_vertex(self) -> np.ndarray:
        """int64 [V]: row of the full Z buffer that holds vertex v."""
        slot = self.perm if self.perm is not None else np.arange(self.num_vertices, dtype=np.int64)
        return self.slot_to_position(slot)

    def local_positions(self, rank: Optional[int] = None) -> np.ndarray:
        """int64 [n_local]: positions of this rank's rows, in local order (chunk by chunk)."""
        r = self.rank if rank is None else rank
        return self.slot_to_position(r * self.n_local + np.arange(self.n_local, dtype=np.int64))

    def chunk_row0(self, c: int) -> int:
        """position of the first row of this rank's chunk c (= `row0` of the kernel call)."""
        return c * self.world_size * self.rows_per_chunk + self.rank * self.rows_per_chunk

    def chunk_span(self, c: int):
        """[begin, end) positions written by the all-gather of chunk c."""
        n = self.world_size * self.rows_per_chunk
        return c * n, (c + 1) * n


@dataclass
class LocalCSR:
    """This rank's rows, columns relabelled to positions, sorted within each row."""
    rowptr: np.ndarray          # int64 [n_local+1]
    colidx: np.ndarray          # int32 [E_local]  (positions in the full Z buffer)
    indeg: np.ndarray           # int32 [n_local]  global in-degree of the vertex in each local row
    vertex: np.ndarray          # int64 [n_local]  vertex id of each local row, -1 for padding rows
    edge_origin: np.ndarray     # int64 [E_local]  index of each local edge in the global CSR order


def localize(csr: HostCSR, part: RowPartition) -> LocalCSR:
    """Slice + relabel the global CSR for one rank."""
    V = csr.num_vertices
    if V != part.num_vertices:
        raise ValueError("partition built for a different vertex count")
    pos = part.position_of_vertex()
    vertex_at = np.full(part.padded_vertices, -1, dtype=np.int64)
    vertex_at[pos] = np.arange(V, dtype=np.int64)
    verts = vertex_at[part.local_positions()]
    valid = verts >= 0
    safe = np.where(valid, verts, 0)
    deg = np.where(valid, csr.outdeg()[safe], 0).astype(np.int64)
    rowptr = np.zeros(part.n_local + 1, dtype=np.int64)
    np.cumsum(deg, out=rowptr[1:])
    # original edge id of every local edge: start-of-row + offset within the row
    row_of = np.repeat(np.arange(part.n_local, dtype=np.int64), deg)
    origin = csr.rowptr[safe][row_of] + (np.arange(rowptr[-1], dtype=np.int64) - rowptr[:-1][row_of])
    cols = pos[csr.colidx[origin]]
    identity = part.perm is None and part.world_size == 1 and part.chunks == 1
    if not identity:
        order = np.lexsort((cols, row_of))      # stable: by row, then by new column
        cols, origin = cols[order], origin[order]
    indeg = np.where(valid, csr.indeg()[safe], 0).astype(np.int32)
    return LocalCSR(rowptr, cols.astype(np.int32), indeg, verts, origin)

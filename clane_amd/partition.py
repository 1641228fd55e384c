"""Row partition of the graph over the GPUs of one box, and the HBM layout that goes with it.

The reference runs in one process and has no partitioning (the path
shards by rows with one exchange per sweep).  Layout chosen for RCCL over xGMI:

* every rank keeps a FULL-size ``Z`` (two ping-pong buffers) in its own HBM, so the gathers of a
  sweep never leave the GPU, and owns ``n_local`` rows of it;
* only rows that can both CHANGE and BE READ have to travel: a row without out-edges never
  changes (embedder.py:88-89) and a row without in-edges is never gathered by anyone.  Rows
  with ``outdeg > 0 and indeg > 0`` are *live*, the rest *quiet*.  A rank's rows are ordered
  ``[live chunk 0 | ... | live chunk C-1 | quiet]``; only live chunks are exchanged during
  sweeps (on R-MAT 2M/40M that is ~45 % of the rows), quiet rows are synchronised once when
  ``Z`` is read out;
* live rows are stored *chunk-major*:  position ``g = c*(W*Lc) + r*Lc + i``  holds live row ``i``
  of chunk ``c`` of rank ``r``.  The rows that chunk ``c`` of all ranks produce are one contiguous
  span and rank ``r``'s piece sits at offset ``r*Lc`` in it -- exactly the in-place form of an
  all-gather (send = recv + rank*count).  A sweep is one launch per chunk, so the all-gather
  of chunk ``c`` runs over xGMI while chunk ``c+1`` (and finally the quiet block) computes;
* quiet rows follow, rank-major: position ``live_total + r*Q + i``;
* vertices are dealt to ranks through a seeded random permutation (per class), which balances
  edges per rank in expectation and spreads hub rows (R-MAT / power-law inputs).

With one rank there is nothing to exchange: all rows form ``chunks`` plain blocks.
Everything here is host-side index arithmetic (numpy); it runs once per graph.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from .xcd import XCD_CLASSES, row_pieces, xcd_subclass


@dataclass
class HostCSR:
    """Adjacency in CSR on the host: row = source, col = destination, columns sorted and
    unique within a row (what ``Graph.A`` -- reference graph.py:104-110 -- coalesces to)."""
    num_vertices: int
    rowptr: np.ndarray   # int64 [V+1]
    colidx: np.ndarray   # int32 [E]
    _indeg: Optional[np.ndarray] = field(default=None, repr=False, compare=False)

    @property
    def num_edges(self) -> int:
        return int(self.colidx.shape[0])

    def validate(self) -> None:
        """ValueError unless this is a CSR the kernels can gather through: rowptr int64 [V + 1] from 0 to E,
        non-decreasing; colidx int32 [E] within [0, V).  (Sortedness / uniqueness within a row is the loader's business:
        ``graph.csr_from_edges``.)  On the host, before anything is uploaded: an index out of range on the device ends
        the process at best."""
        V, rp, ci = self.num_vertices, self.rowptr, self.colidx
        if rp.dtype != np.int64 or ci.dtype != np.int32 or rp.shape != (V + 1,) or ci.ndim != 1:
            raise ValueError("HostCSR: rowptr must be int64 [V + 1] and colidx int32 [E]")
        if rp[0] != 0 or rp[-1] != ci.size or (V and bool((rp[1:] < rp[:-1]).any())):
            raise ValueError("HostCSR: rowptr must rise from 0 to the number of edges")
        if ci.size and (int(ci.min()) < 0 or int(ci.max()) >= V):
            raise ValueError(f"HostCSR: colidx holds entries outside [0, {V})")

    def outdeg(self) -> np.ndarray:
        return np.diff(self.rowptr)

    def indeg(self, device=None) -> np.ndarray:
        """In-degree of every vertex (int32 [V]), counted once and kept; the CSR is not meant to be edited afterwards.
        Counted on the host unless the caller names a GPU (``SweepEngine`` passes its own: 40M edges take 0.1 s on the
        host): a host utility never opens a HIP context on a device nobody asked for, and a card without room for the
        scratch falls back to the host."""
        if self._indeg is None:
            import torch
            on_card = (device is not None and torch.device(device).type == "cuda" and self.colidx.size >= (1 << 22))
            if on_card:
                try:
                    total = torch.zeros(self.num_vertices, dtype=torch.int64, device=device)
                    for a in range(0, self.colidx.size, 1 << 28):            # pieces: bounded scratch on the card
                        total += torch.bincount(torch.from_numpy(self.colidx[a:a + (1 << 28)]).to(device),
                                                minlength=self.num_vertices)
                    self._indeg = total.to(torch.int32).cpu().numpy()
                except torch.cuda.OutOfMemoryError:
                    on_card = False
            if not on_card:
                self._indeg = np.bincount(self.colidx, minlength=self.num_vertices).astype(np.int32)
        return self._indeg

    def live_mask(self) -> np.ndarray:
        """Rows that change during sweeps AND are read by some row: the only ones worth exchanging."""
        return (self.outdeg() > 0) & (self.indeg() > 0)


@dataclass
class Block:
    """One kernel launch of a sweep: ``nrows`` consecutive local rows starting at ``local_start``,
    stored at positions ``row0 ...``; ``span`` = positions the all-gather after it fills (None: no exchange)."""
    local_start: int
    nrows: int
    row0: int
    span: Optional[Tuple[int, int]]
    exchange: Optional[object] = None      # halo.HaloExchange in halo mode


@dataclass
class RowPartition:
    num_vertices: int
    world_size: int
    rank: int
    chunks: int
    live_per_chunk: int                 # Lc: live rows per (rank, chunk), padded
    quiet_per_rank: int                 # Q : quiet rows per rank, padded
    live_per_rank: int                  # live vertices dealt to each rank before chunk padding (ceil(nlive / W))
    vertex_slot: np.ndarray = field(repr=False, default=None)   # int64 [V]: slot within its class
    vertex_live: np.ndarray = field(repr=False, default=None)   # bool  [V]

    @classmethod
    def create(cls, num_vertices: int, world_size: int = 1, rank: int = 0, chunks: int = 1,
               live_mask: Optional[np.ndarray] = None, shuffle: Optional[bool] = None, seed: int = 0,
               priority: Optional[np.ndarray] = None) -> "RowPartition":
        """``priority`` (one value per vertex, used when not shuffling): vertices are laid out in DESCENDING
        priority.  With priority = in-degree the rows every sweep gathers most often sit next to each other at
        the front of Z (the top 250k rows of R-MAT 2M/40M take 91 % of the gathers and fill exactly the
        256 MiB Infinity Cache): measured +29 % on the pure gather, same multiset of reads."""
        if not (0 <= rank < world_size) or chunks < 1 or num_vertices < 1:
            raise ValueError(f"bad partition: V={num_vertices} world={world_size} rank={rank} chunks={chunks}")
        if shuffle is None:
            shuffle = world_size > 1
        if world_size == 1 or live_mask is None:
            live = np.ones(num_vertices, dtype=bool)          # nothing to exchange / nothing known: all rows chunked
        else:
            live = np.asarray(live_mask, dtype=bool)
            if live.shape != (num_vertices,):
                raise ValueError("live_mask must have one entry per vertex")
        rng = np.random.default_rng(seed)
        slot = np.empty(num_vertices, dtype=np.int64)
        for mask in (live, ~live):
            ids = np.nonzero(mask)[0]
            if shuffle:
                order = rng.permutation(ids.size)
            elif priority is not None:
                order = np.argsort(-np.asarray(priority)[ids], kind="stable")
            else:
                order = np.arange(ids.size)
            slot[ids[order]] = np.arange(ids.size, dtype=np.int64)
        n_live, n_quiet = int(live.sum()), int((~live).sum())
        live_per_rank = max(1, -(-n_live // world_size))
        lc = -(-live_per_rank // chunks)
        q = -(-n_quiet // world_size)
        return cls(num_vertices, world_size, rank, chunks, lc, q, live_per_rank, slot, live)

    # ---- sizes ------------------------------------------------------------------------
    @property
    def live_total(self) -> int:
        return self.world_size * self.chunks * self.live_per_chunk

    @property
    def n_local(self) -> int:
        return self.chunks * self.live_per_chunk + self.quiet_per_rank

    @property
    def padded_vertices(self) -> int:
        return self.live_total + self.world_size * self.quiet_per_rank

    # ---- index maps -------------------------------------------------------------------
    def _live_position(self, r, l):
        c, i = np.divmod(l, self.live_per_chunk)
        return c * (self.world_size * self.live_per_chunk) + r * self.live_per_chunk + i

    def position_of_vertex(self) -> np.ndarray:
        """int64 [V]: row of the full Z buffer that holds vertex v."""
        s = self.vertex_slot
        r_live, l_live = np.divmod(s, self.live_per_rank)
        pos_live = self._live_position(r_live, l_live)
        pos_quiet = self.live_total + s          # slot t -> rank t // Q, offset t % Q  ==  live_total + t
        return np.where(self.vertex_live, pos_live, pos_quiet).astype(np.int64)

    def local_positions(self, rank: Optional[int] = None) -> np.ndarray:
        """int64 [n_local]: positions of this rank's rows in local order (live chunks, then quiet)."""
        r = self.rank if rank is None else rank
        live = self._live_position(r, np.arange(self.chunks * self.live_per_chunk, dtype=np.int64))
        quiet = self.live_total + r * self.quiet_per_rank + np.arange(self.quiet_per_rank, dtype=np.int64)
        return np.concatenate([live, quiet])

    def blocks(self, spans_for_one_rank: bool = False) -> List[Block]:
        """Launch plan of one sweep for this rank.  ``spans_for_one_rank``: keep the (then trivial) all-gather span
        of every chunk in a one-rank partition too (TorchComm(force_collectives=True))."""
        lc, w = self.live_per_chunk, self.world_size
        out = []
        for c in range(self.chunks):
            b = c * w * lc
            span = (b, b + w * lc) if (w > 1 or spans_for_one_rank) else None
            out.append(Block(c * lc, lc, b + self.rank * lc, span))
        if self.quiet_per_rank:
            out.append(Block(self.chunks * lc, self.quiet_per_rank,
                             self.live_total + self.rank * self.quiet_per_rank, None))
        return out

    def quiet_span(self) -> Optional[Tuple[int, int, int]]:
        """(begin, end, rows per rank) of the quiet region -- all-gathered only when Z is read out."""
        if self.world_size == 1 or self.quiet_per_rank == 0:
            return None
        return self.live_total, self.padded_vertices, self.quiet_per_rank


@dataclass
class LocalCSR:
    """This rank's rows, columns relabelled to positions, sorted within each row."""
    rowptr: np.ndarray          # int64 [n_local+1]
    colidx: np.ndarray          # int32 [E_local]  (positions in the full Z buffer)
    indeg: np.ndarray           # int32 [n_local]  global in-degree of the vertex in each local row
    vertex: np.ndarray          # int64 [n_local]  vertex id of each local row, -1 for padding rows
    edge_origin: np.ndarray     # int64 [E_local]  index of each local edge in the global CSR order


LOCALIZE_PIECE_EDGES = 1 << 28


def edge_order(row_of, cols, sub, table_rows: int, n_sub: int = XCD_CLASSES, two_pass: Optional[bool] = None):
    """Permutation (torch) that sorts edges by (row, sub-class, column) -- `sub` None: by (row, column); `sub` in
    [0, n_sub).  (row, column) pairs are unique, so any sort gives THE order.  One sort of a fused int64 key while it
    fits; two stable sorts beyond that (rows x n_sub x table rows >= 2^62)."""
    import torch
    n_rows = int(row_of.max()) + 1 if row_of.numel() else 1
    width = table_rows * (n_sub if sub is not None else 1)
    if two_pass is None:
        two_pass = n_rows * width >= 2 ** 62
    if not two_pass:
        key = row_of * width + cols
        if sub is not None:
            key = key + sub * table_rows
        return torch.argsort(key)
    first = torch.argsort(cols, stable=True)
    major = row_of if sub is None else row_of * n_sub + sub
    return first[torch.argsort(major[first], stable=True)]


def localize(csr: HostCSR, part: RowPartition, device=None, class_threshold: int = 0, phase_threshold: int = 0,
             phases: int = 1) -> LocalCSR:
    """Slice + relabel the global CSR for one rank.  The one heavy step -- re-sorting every row's edges by their
    new column -- is a single sort of unique (row, column) keys; with ``device`` = a GPU it runs there (40M edges:
    seconds on the host, milliseconds on the card).
    ``class_threshold`` > 0: rows with more edges than that are sorted by (xcd_subclass(column), column) instead, so
    that the edges to one XCD class (of one phase, for rows above ``phase_threshold``) are contiguous (engine:
    class-affine rows, csrc/spmm_update.h; clane_amd/xcd.py)."""
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
    indeg = np.where(valid, csr.indeg()[safe], 0).astype(np.int32)
    by_class = deg > class_threshold if class_threshold > 0 else np.zeros_like(deg, dtype=bool)
    identity = (part.world_size == 1 and part.chunks == 1 and not by_class.any()
                and np.array_equal(pos, np.arange(V)))
    if identity:
        return LocalCSR(rowptr, csr.colidx.astype(np.int32), indeg, verts, np.arange(csr.num_edges, dtype=np.int64))
    import torch
    dev = torch.device(device) if device is not None and torch.device(device).type == "cuda" else torch.device("cpu")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    pos_t, start_t = t(pos), t(csr.rowptr[safe])
    colidx_t = t(csr.colidx)
    E_loc = int(rowptr[-1])
    cols_out = np.empty(E_loc, dtype=np.int32)
    origin_out = np.empty(E_loc, dtype=np.int64)
    # rows are independent and their edges contiguous: work through the rows in pieces of at most
    # LOCALIZE_PIECE_EDGES edges (one sort each; torch sorts at most 2^31 - 1 elements, and the pieces bound the
    # scratch memory: ~10 int64 vectors of the piece's length)
    for a, b in row_pieces(rowptr, LOCALIZE_PIECE_EDGES):
        e0, e1 = int(rowptr[a]), int(rowptr[b])
        if e1 == e0:
            continue
        deg_t = t(deg[a:b])
        row_of = torch.repeat_interleave(torch.arange(b - a, device=dev), deg_t)
        # original edge id of every local edge: start-of-row + offset within the row
        origin = start_t[a:b][row_of] + (torch.arange(e1 - e0, device=dev) - (t(rowptr[a:b]) - e0)[row_of])
        cols = pos_t[colidx_t[origin].long()]
        sub = (xcd_subclass(cols, deg_t[row_of], class_threshold, phase_threshold, phases)
               if by_class[a:b].any() else None)
        order = edge_order(row_of, cols, sub, part.padded_vertices, XCD_CLASSES * max(1, phases))
        cols_out[e0:e1] = cols[order].to(torch.int32).cpu().numpy()
        origin_out[e0:e1] = origin[order].cpu().numpy()
    return LocalCSR(rowptr, cols_out, indeg, verts, origin_out)


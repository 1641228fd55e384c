"""SweepEngine -- device-resident state and the launch sequence of the embedding loop.

Replaces, for one rank, what the reference spreads over ``Graph.build_P`` (graph.py:118-128),
the body of ``Embedder.propagate`` (embedder.py:84-94) and the per-vertex tensors behind
``Graph.Z`` (graph.py:130-138).  All arithmetic is done by the HIP kernels behind
``clane_amd._hip.HipKernels`` (C ABI in include/clane_hip.h); torch is used for device memory,
the D2H copy of one scalar per sweep and ``torch.distributed`` (RCCL).

HBM layout per rank (T = float32 / float64 / bfloat16, ld = d rounded up to a 16-byte pack,
pad columns zero; on N > 1 GPUs with the column split d is the width of the rank's column slice and
every rank holds all rows; with a row split n_loc < V_pad):

    Zbuf[3]   [V_pad, ld] T    embedding matrix: a sweep reads one, writes another; the third holds the embeddings of
                               the start of the outer round (embedder.py:58) while the other two ping-pong
    X_loc     [n_loc, ld] T    content embeddings of the owned rows
    rowptr    [n_loc+1] i64, colidx [E_loc] i32 (positions), P [E_loc] acc, indeg [n_loc] i32
    partials  fixed-order L1-delta partial sums (double)

A sweep is one launch group per *block* of owned rows (one block, unless rows are divided over GPUs:
then each live chunk is followed by the exchange of its rows -- async on RCCL's stream, overlapping
the next block's kernels).  The scalar delta is all-reduced.  On one GPU no collective is issued; with
the column split (the default division) that scalar is the only per-sweep collective.
"""
from __future__ import annotations

import contextlib
import threading
from typing import List, Optional

import numpy as np
import torch

from . import _hip, xcd
from .comm import TorchComm
from .diagnostics import DiagnosticsMixin
from .halo import build_halo_layout
from .partition import Block, HostCSR, LocalCSR, RowPartition, localize
from .plan import (HEAVY_ROW_EDGES, HUB_FACTOR, INFINITY_CACHE_BYTES, L2_BYTES_ALL_XCDS,            # noqa: F401 (re-exported)
                   LONG_THRESHOLD_BY_ROWS_PER_WAVE, MIN_HOT_READ_SHARE, MIN_SEGMENT_EDGES, MIN_SLICE_ROW_BYTES,
                   SOFTMAX_EDGES_PER_WORKGROUP, SPLIT_EDGES, TARGET_SEGMENTS, UNSKEWED_LONG_THRESHOLD, _round_up,
                   column_slice, hot_read_share, lanes_per_row, pick_exchange)
from .staging import StagedZ, StagingMixin, place_piece                                                # noqa: F401 (re-exported)
from .xcd import (CLASS_CHUNK, CLASS_THRESHOLD_BY_ROWS_PER_WAVE, PHASE_THRESHOLD, PHASES_BY_ROWS_PER_WAVE,
                  class_items)

class SweepEngine(StagingMixin, DiagnosticsMixin):
    STAGE_SLOTS = 3          # copies of Z that may be in flight to the host at once (stage_Z)
    # Z tables.  Two do for the sweeps (read old / write new).  A third replaces the reference's
    # `prev_Z = graph.Z.clone()` (embedder.py:58): snapshot() PINS the current table instead of copying 2 GB, and the
    # sweeps of the round ping-pong between the other two.  It is allocated by the first snapshot(): a propagate-only
    # run never pays for it (2 GB at config 3, 16 GiB at the 16M-vertex capacity run) -- except in halo_p2p mode, where
    # the tables are shared allocations mapped by every peer and all three are made at start-up.
    N_TABLES = 3
    # time_kernels: at most this many sweeps get HIP events.  Hundreds of live timing events slow every launch down
    # (config 2, 200 timed sweeps: 0.475 ms per step with 1 000 events alive, 0.248 ms with 160)
    MAX_TIMED_SWEEPS = 32

    EXCHANGES = ("auto", "columns", "halo", "halo_p2p", "allgather", "allgather_all")

    def __init__(self, csr: HostCSR, X: torch.Tensor, device, kernels=None, *, cosine_mode: str = "reference",
                 process_group=None, chunks: Optional[int] = None, long_threshold: Optional[int] = None,
                 hub_threshold: Optional[int] = None, shuffle: Optional[bool] = None, seed: int = 0,
                 exchange: str = "auto", comm=None, hot_rows_first: bool = True, split_hubs: bool = True,
                 overlap_chunks: bool = True, fused_pack: bool = True, class_threshold: Optional[int] = None,
                 class_chunk: int = CLASS_CHUNK, class_k1: bool = True, class_phases: Optional[int] = None,
                 phase_threshold: int = PHASE_THRESHOLD, delta_stream: bool = False, table_skew=None,
                 table_alloc: str = "torch", column_tiles: Optional[int] = None):
        """``exchange`` (N > 1 only) -- how the sweep is divided over the GPUs:
        "auto" (default) -- "columns" while a rank's slice of a row is >= 64 bytes, else "halo" (pick_exchange).
        "columns" -- every GPU holds the whole graph and d/N COLUMNS of X and Z.  ``Z[:, c] = X[:, c] + gamma P Z[:, c]``
        is independent per column, so a sweep needs no exchange at all (only the delta scalar is all-reduced);
        build_P all-reduces the partial dot products (E values, once per outer iteration).
        "halo" -- rows are divided; a compact per-rank table, rows sent only to the ranks that read them (halo.py);
        "halo_p2p" -- the same tables, mapped into every process of the box (hipIpc): the kernel that finishes a row
        stores it into the tables of the ranks that read it (peer-to-peer over xGMI), there is no exchange step;
        "allgather" -- rows divided, full-size Z on every rank, in-place all-gather of the live rows
        (partition.py); "allgather_all" -- the same without the live/quiet split.
        The constructor is a sequence of steps, each a method: the division over the ranks, the class-pass thresholds,
        the row layout, the row bins, the structure upload, the launch lists, the exchange lists, the tables, scratch."""
        if X.dim() != 2 or X.shape[0] != csr.num_vertices:
            raise ValueError(f"X must be [V, d] with V={csr.num_vertices}, got {tuple(X.shape)}")
        if cosine_mode not in ("reference", "per_edge"):
            raise ValueError(f"cosine_mode must be 'reference' or 'per_edge', got {cosine_mode!r}")
        csr.validate()                      # before anything is uploaded or indexed on the device
        csr.indeg(device)                   # counted once, on THIS engine's card (partition / halo layouts reuse it)
        self.k = kernels if kernels is not None else _hip.kernels()
        self.device = torch.device(device)
        self.dtype = X.dtype
        self.acc_dtype = _hip.acc_dtype(X.dtype)
        self.cosine_mode = cosine_mode
        self.table_skew = table_skew
        if table_alloc not in ("torch", "contiguous"):
            raise ValueError("table_alloc must be 'torch' or 'contiguous'")
        self.table_alloc, self._own_tables, self.table_alloc_note = table_alloc, [], None
        self._column_tiles_asked = column_tiles
        X = self._choose_division(csr, X, process_group, comm, exchange)
        self._choose_class_pass(csr, class_threshold, class_chunk, class_k1, class_phases, phase_threshold)
        self._build_layout(csr, chunks, shuffle, seed, hot_rows_first)
        self._choose_row_bins(long_threshold, hub_threshold)
        deg = self._upload_structure()
        self._bin_rows(deg, split_hubs)
        self._setup_exchange(deg, fused_pack)
        self._alloc_tables(X, deg)
        self._alloc_scratch(delta_stream, overlap_chunks)

    # ---- constructor steps ----------------------------------------------------------------------------------
    def _choose_division(self, csr: HostCSR, X: torch.Tensor, process_group, comm, exchange: str) -> torch.Tensor:
        """Who the ranks are and how the sweep is divided over them; returns this rank's columns of X.  Either the
        COLUMNS are divided ("columns": C = N column groups, every rank holds all rows) or the ROWS are (C = 1)."""
        self.pg = process_group
        self.comm = comm if comm is not None else (TorchComm(process_group) if process_group is not None else None)
        self.world = self.comm.world if self.comm is not None else 1
        rank = self.comm.rank if self.comm is not None else 0
        if exchange not in self.EXCHANGES:
            raise ValueError("exchange must be 'auto', 'columns', 'halo', 'halo_p2p', 'allgather' or "
                             f"'allgather_all', got {exchange!r}")
        # a one-rank group whose comm insists on its collectives keeps the division it is given (RCCL rehearsal on a
        # one-GPU box, comm.TorchComm(force_collectives=True)); "auto" is then the plain one-GPU plan
        self._forced = self.world == 1 and self.comm is not None and bool(self.comm.force) and exchange in (
            "columns", "allgather", "allgather_all")
        divided = self.world > 1 or self._forced
        if exchange == "auto":
            exchange = pick_exchange(int(X.shape[1]), X.dtype, self.world)
        self._exchange_asked = exchange
        self.exchange = exchange if divided else "none"
        self.columns = divided and exchange == "columns"
        self.C = self.world if self.columns else 1      # column groups
        self.c = rank if self.columns else 0
        self.V, self.d_full = csr.num_vertices, int(X.shape[1])
        self.E_total = csr.num_edges
        self.col0, self.col1 = column_slice(self.d_full, X.dtype, self.C, self.c) if self.columns else (0, self.d_full)
        if self.columns:
            X = X[:, self.col0:self.col1]
            self.ld_max = _round_up(column_slice(self.d_full, X.dtype, self.C, 0)[1], _hip.VEC_ELEMS[X.dtype])
        self.d = self.col1 - self.col0                 # columns this rank computes (0: an idle rank, d < C packs)
        self.ld = _round_up(self.d, _hip.VEC_ELEMS[X.dtype])
        # rows are divided over `row_world` ranks; with the column split a rank owns every row
        self.row_world, self.row_rank = (1, 0) if self.columns else (self.world, rank)
        self.halo = self.row_world > 1 and exchange in ("halo", "halo_p2p")
        self.p2p = self.halo and exchange == "halo_p2p"       # finished rows are stored straight into the readers' tables
        if self.p2p and self.world > 8:
            raise ValueError("halo_p2p addresses at most 8 GPUs (one box)")
        # Everything that decides the LAYOUT (which rows are class rows, hence the order of every row's edges) follows
        # from `d_plan`: the rank's own width -- except with the column split, where it is the widest slice (rank 0's)
        # on EVERY rank, one without columns included: build_P all-reduces the partial dot products element by
        # element, so all ranks must hold their edges in one order, even when their slices straddle a lane-layout
        # boundary (33 packs over two ranks: 32 and 16 lanes per row, class thresholds 64 and 256).
        self.d_plan = self.d
        if self.columns:
            c0, c1 = column_slice(self.d_full, X.dtype, self.C, 0)
            self.d_plan = c1 - c0
        # COLUMN TILES (one GPU): a sweep as T passes over T column ranges of the same tables -- the update is independent
        # per column (embedder.py:92: P mixes rows, never columns), and with rows of T-th the width the L2s and the
        # Infinity Cache hold T times as many of the hot rows per pass; colidx / P are read T times.  See `_pick_tiles`.
        T = self._pick_tiles(csr, X.dtype) if not divided else 1
        cuts = [column_slice(self.d, X.dtype, T, t) for t in range(T)] if self.d > 0 else [(0, 0)]
        self.tiles = [c for c in cuts if c[1] > c[0]] or [(0, self.d)]
        if len(self.tiles) > 1:
            self.d_plan = max(c1 - c0 for c0, c1 in self.tiles)
        # a hint for the kernels (CLANE_SPMM_TABLE_BEYOND_CACHE): with the table far beyond the Infinity Cache the
        # 512-byte-row instances keep fewer row loads in flight per wave and run more waves (csrc/clane_abi.hip)
        self.beyond_cache = csr.num_vertices * self.ld * torch.empty(0, dtype=X.dtype).element_size() > 2 * INFINITY_CACHE_BYTES
        self.ld_plan = _round_up(self.d_plan, _hip.VEC_ELEMS[X.dtype])
        self.rows_per_wave = 64 // lanes_per_row(self.d_plan, X.dtype) if self.d_plan > 0 else 1
        return X

    def _pick_tiles(self, csr: HostCSR, dtype: torch.dtype) -> int:
        """Column tiles of a one-GPU sweep (``column_tiles``; None = the rule below).  Measured on R-MAT 2M / 40M
        (profiles/r05_column_tiles_ab.md), sweep ms untiled -> with tiles of 128 fp32 columns (512 bytes): d=256 3.92 ->
        3.79 (2 tiles; 4 tiles of 64 columns: 4.08), d=384 6.40 -> 5.81 (3), d=512 8.70 -> 7.58 (4; 8 tiles of 64: 8.27),
        d=1024 19.5 -> 15.4 (8); the 16M-vertex run 38.6 -> 38.1.  build_P, which keeps the full width on the tiles'
        layout, pays 4-6 %; a propagate runs >= 11 sweeps per build_P.  Losses: uniform-random pairs 7.45 -> 7.63 (nothing
        to keep in a cache: colidx / P read twice for nothing), rows below 1 KiB (cache-resident config 2 0.187 -> 0.211,
        config 4's bf16 rows 7.04 -> 8.08), bf16 rows of 1 KiB (4.51 -> 4.72), tiles whose edges are not 128-byte
        aligned (d=256 in 3 tiles: 5.11).  Hence: fp32, d a multiple of 128 and at least 256, tiles of 128 columns (at
        most 8), skewed reads at the tile's width, a table well beyond the Infinity Cache."""
        asked = self._column_tiles_asked
        if asked is not None:
            if asked < 1:
                raise ValueError("column_tiles must be >= 1")
            return int(asked)
        if dtype != torch.float32 or self.d < 256 or self.d % 128:
            return 1
        if csr.num_vertices * self.ld * 4 <= 2 * INFINITY_CACHE_BYTES:
            return 1
        return min(8, self.d // 128) if hot_read_share(csr, 512) >= MIN_HOT_READ_SHARE else 1

    def _choose_class_pass(self, csr: HostCSR, class_threshold, class_chunk, class_k1, class_phases,
                           phase_threshold) -> None:
        """Which rows take the XCD-affine pass (xcd.py), in how many phases."""
        row_bytes = self.ld_plan * torch.empty(0, dtype=self.dtype).element_size()
        # how much of the gather traffic XCD affinity could serve from the L2s at all (same number on every rank)
        self.hot_read_share = hot_read_share(csr, row_bytes) if self.d_plan > 0 else 1.0
        self.class_affinity = True
        if class_threshold is None:         # XCD-affine long rows, whatever the division
            class_threshold = CLASS_THRESHOLD_BY_ROWS_PER_WAVE[self.rows_per_wave]
            if self.rows_per_wave == 2 and csr.num_vertices * row_bytes <= INFINITY_CACHE_BYTES:
                # a table that sits in the Infinity Cache: misses of an L2 are cheap, so the class pass pays from
                # twice the degree on (512-byte rows, 200k x 4M: 0.207 ms per sweep at 64, 0.192 at 128, 0.196 at 256;
                # 1-KiB rows keep 64, narrower rows are at 256 anyway: profiles/r04_class_threshold_cache_resident.md)
                class_threshold = max(class_threshold, 128)
            if self.hot_read_share < MIN_HOT_READ_SHARE:    # evenly spread reads: only rows that need splitting anyway
                class_threshold = max(class_threshold, HEAVY_ROW_EDGES)
                self.class_affinity = False
        self.class_threshold = int(class_threshold)
        self.class_chunk = int(class_chunk)
        self.class_k1 = bool(class_k1) and self.class_threshold > 0    # build_P scores the class rows XCD-affine too
        # heavy class rows are also phased in time (xcd.py): phases 1 = off
        if class_phases is None:
            class_phases = (xcd.PHASES_UNDER_COLUMN_TILES if len(self.tiles) > 1 else 0) or \
                PHASES_BY_ROWS_PER_WAVE[self.rows_per_wave]
        self.class_phases = int(class_phases)
        if self.class_threshold == 0 or self.class_phases < 1:
            self.class_phases = 1
        if self.class_phases & (self.class_phases - 1) or self.class_phases > 8:
            raise ValueError("class_phases must be 1, 2, 4 or 8")
        self.phase_threshold = max(int(phase_threshold), self.class_threshold) if self.class_phases > 1 else 0
        if self.class_threshold and not (64 <= self.class_chunk <= 4096 and self.class_chunk % 64 == 0):
            raise ValueError("class_chunk must be a multiple of 64 in [64, 4096]")
        # a row one of whose (phase, class) segments alone is more than one XCD's L2 holds: its chunks are scheduled by
        # column, next to the other such rows' (xcd.class_items)
        self.mega_segment_edges = int(L2_BYTES_ALL_XCDS // 8 // max(row_bytes, 1)) if self.d_plan > 0 else 0

    def _build_layout(self, csr: HostCSR, chunks, shuffle, seed: int, hot_rows_first: bool) -> None:
        """Which rows this rank owns, where every row sits in the tables, the rank's CSR relabelled to table rows."""
        if chunks is None:
            chunks = 1 if self.row_world == 1 else 4
        self.hot_rows_first = bool(hot_rows_first)
        if self.halo:
            self.part = build_halo_layout(csr, self.row_world, self.row_rank, chunks, shuffle=shuffle is not False, seed=seed,
                                          hot_rows_first=hot_rows_first, class_threshold=self.class_threshold,
                                          phase_threshold=self.phase_threshold, phases=self.class_phases)
            self.blocks: List[Block] = self.part.blocks
            self.local: LocalCSR = self.part.local
        else:
            live = csr.live_mask() if (self.row_world > 1 and self._exchange_asked == "allgather") else None
            # all rows on this GPU: lay Z out by descending in-degree, so the rows gathered most often are contiguous
            hot = csr.indeg() if (self.row_world == 1 and hot_rows_first and not shuffle) else None
            self.part = RowPartition.create(self.V, self.row_world, self.row_rank, chunks, live_mask=live, shuffle=shuffle,
                                            seed=seed, priority=hot)
            self.blocks = self.part.blocks(spans_for_one_rank=self._forced and not self.columns)
            self.local = localize(csr, self.part, self.device, class_threshold=self.class_threshold,
                                  phase_threshold=self.phase_threshold, phases=self.class_phases)
        if self.part.padded_vertices >= 2 ** 31:
            raise ValueError(f"{self.part.padded_vertices} table rows: column indices are 32-bit (ABI v1); divide the "
                             "rows over more GPUs (exchange='halo') or wait for 64-bit indices")

    def _choose_row_bins(self, long_threshold, hub_threshold) -> None:
        """The degrees at which K1 / K2 / K3 hand a row from the one-(sub-)wave kernels to the workgroup-per-row ones."""
        if long_threshold is None:
            long_threshold = LONG_THRESHOLD_BY_ROWS_PER_WAVE[self.rows_per_wave]
            if self.rows_per_wave == 1 and self.class_threshold and self.class_affinity:
                # with the class pass taking the rows above class_threshold, the 33..64-edge rows are better off with
                # one wave each than with a 16-wave workgroup of which 15 waves leave at once (4.34 -> 4.30 ms)
                long_threshold = max(long_threshold, self.class_threshold)
            elif self.rows_per_wave == 1 and not self.class_affinity:
                # evenly spread reads: no hubs for a lone wave to crawl along, and plenty of similar rows in flight --
                # one wave per row up to UNSKEWED_LONG_THRESHOLD edges (near-regular 2M / 128M, rows of 48..80 edges:
                # 23.4 ms with T = 32, 22.2 with 128 or 256; uniform and star-heavy graphs: no difference,
                # profiles/r04_threshold_robustness.md)
                long_threshold = max(long_threshold, UNSKEWED_LONG_THRESHOLD)
            if self.rows_per_wave > 1:
                # A T-edge row walked by one sub-wave takes T/8 gather groups in sequence -- the tail of its launch.
                # That is nothing next to a 40M-edge pass and a third of a 4M-edge one (R-MAT 200k/4M/d=128:
                # 3 545 sweeps/s at T=128, 2 013 at T=1024), so T also scales with the edges of the pass.
                by_size = 1 << max(7, int(self.local.colidx.shape[0] // 32768).bit_length() - 1)
                long_threshold = min(long_threshold, by_size)
        # K1 / K2 (build_P) cut rows at the same degree as before; K3's one-(sub-)wave pass also stops below the class rows
        self.score_threshold = int(long_threshold)
        # K1 with the class pass: its one-(sub-)wave kernel stops below the class rows as well
        self.k1_threshold = (min(self.score_threshold, self.class_threshold) if self.score_threshold > 0
                             else self.class_threshold) if self.class_k1 else self.score_threshold
        if self.class_threshold:            # the class rows are nobody else's
            long_threshold = min(long_threshold, self.class_threshold) if long_threshold > 0 else self.class_threshold
        self.long_threshold = int(long_threshold)
        self.hub_threshold = int(hub_threshold) if hub_threshold is not None else HUB_FACTOR * self.long_threshold

    def _upload_structure(self) -> np.ndarray:
        """rowptr / colidx / in-degrees on the card, checked there once; P allocated.  Returns the local out-degrees."""
        dev = self.device
        self.rowptr = torch.from_numpy(self.local.rowptr).to(dev)
        self.colidx = torch.from_numpy(self.local.colidx).to(dev)
        if self.colidx.numel() == 0:       # a graph without edges: the ABI still wants a real pointer
            self.colidx = torch.zeros(1, dtype=torch.int32, device=dev)
        self.indeg = torch.from_numpy(self.local.indeg).to(dev)
        self.E_loc = int(self.local.colidx.shape[0])
        # what every gather kernel takes on trust, checked once on the device (a bad index faults the GPU)
        self.k.check_csr(self.rowptr, self.colidx, self.part.n_local, self.E_loc, self.part.padded_vertices)
        self.P = torch.zeros(max(self.E_loc, 1), dtype=self.acc_dtype, device=dev)
        self.P_valid = False
        deg = np.diff(self.local.rowptr)
        self.max_degree = int(deg.max()) if deg.size else 0
        # workgroups that share the final softmax rescale of one class row in build_P: one, until a row is long enough for
        # that single workgroup to be the tail of build_P (a 2M-edge row: 2.3 ms of 6.6; config 3's 70k-edge hub: one)
        self.softmax_row_parts = int(min(64, max(1, -(-self.max_degree // SOFTMAX_EDGES_PER_WORKGROUP))))
        return deg

    def _bin_rows(self, deg: np.ndarray, split_hubs: bool) -> None:
        """Per launch block: the row lists of each kernel (relative to the block's first row: the kernels get rowptr / X /
        Z_new offset to it), the class items, the slots of the fixed-order delta partials, the slabs."""
        dev = self.device
        self.long_rows: List[Optional[torch.Tensor]] = []     # every row above score_threshold (K1 / K2 slice these)
        self.mid_rows: List[Optional[torch.Tensor]] = []      # long_threshold < deg <= hub_threshold: 4 waves/row
        self.hub_rows: List[Optional[torch.Tensor]] = []      # hub_threshold < deg <= SPLIT_EDGES: 16 waves/row
        self.split_rows: List[Optional[tuple]] = []           # deg > SPLIT_EDGES: (rows, seg_ptr, seg_row) on device
        self.class_rows: List[Optional[tuple]] = []           # deg > class_threshold: (rows, slot_ptr, e0, len, slot, row)
        self.class_slots: List[int] = []                      # slot_ptr[-1] of each block's class rows, known on the host
        self.k1_long_rows: List[Optional[torch.Tensor]] = []  # K1's workgroup-per-row list: above k1_threshold, not class
        self.split_edges = SPLIT_EDGES if split_hubs else 0
        hub_edges = int(deg[deg > SPLIT_EDGES].sum()) // max(1, len(self.blocks))
        self.segment_edges = int(min(SPLIT_EDGES, max(MIN_SEGMENT_EDGES, hub_edges // TARGET_SEGMENTS // 1024 * 1024)))
        max_segments = max_slots = 0
        self.partial_off = [0]
        to_dev = lambda a: torch.from_numpy(a.astype(np.int32)).to(dev) if a.size else None  # noqa: E731
        for b in self.blocks:
            db = deg[b.local_start:b.local_start + b.nrows]
            is_long = db > self.long_threshold if self.long_threshold > 0 else np.zeros_like(db, dtype=bool)
            is_class = db > self.class_threshold if self.class_threshold > 0 else np.zeros_like(db, dtype=bool)
            is_long = is_long | is_class
            is_split = (is_long & (db > self.split_edges) if self.split_edges > 0 else np.zeros_like(is_long)) & ~is_class
            is_hub = is_long & (db > self.hub_threshold) & ~is_split & ~is_class
            self.long_rows.append(to_dev(np.nonzero(db > self.score_threshold)[0] if self.score_threshold > 0
                                         else np.empty(0, dtype=np.int64)))
            self.mid_rows.append(to_dev(np.nonzero(is_long & ~is_hub & ~is_split & ~is_class)[0]))
            rows_c = np.nonzero(is_class)[0]
            if rows_c.size:
                rows_abs = rows_c + b.local_start
                phased = dict(phase_threshold=self.phase_threshold, phases=self.class_phases,
                              mega_segment_edges=self.mega_segment_edges,
                              mega_min_edges=self.part.padded_vertices // 4)
                # items per workgroup follow the item count (few chunks in a launch -- a chunk of a rank's rows --
                # get smaller workgroups): class_items picks it after its one counting pass over the edges
                items = class_items(self.local.rowptr, self.local.colidx, rows_abs, self.class_chunk, None,
                                    row_ids=rows_c, colidx_dev=self.colidx, **phased)
                ipb = items["items_per_block"]
                self.class_rows.append((to_dev(rows_c), torch.from_numpy(items["slot_ptr"]).to(dev),
                                        torch.from_numpy(items["e0"]).to(dev), torch.from_numpy(items["len"]).to(dev),
                                        torch.from_numpy(items["slot"]).to(dev), torch.from_numpy(items["row"]).to(dev),
                                        ipb))
                self.class_slots.append(int(items["slot_ptr"][-1]))
                max_slots = max(max_slots, self.class_slots[-1])
            else:
                self.class_rows.append(None)
                self.class_slots.append(0)
            k1_long = (db > self.k1_threshold) & ~(is_class if self.class_k1 else np.zeros_like(is_class)) \
                if self.k1_threshold > 0 else np.zeros_like(is_class)
            self.k1_long_rows.append(to_dev(np.nonzero(k1_long)[0]))
            self.hub_rows.append(to_dev(np.nonzero(is_hub)[0]))
            rows_s = np.nonzero(is_split)[0]
            if rows_s.size:
                nseg = -(-db[rows_s] // self.segment_edges)
                seg_ptr = np.zeros(rows_s.size + 1, dtype=np.int64)
                np.cumsum(nseg, out=seg_ptr[1:])
                seg_row = np.repeat(np.arange(rows_s.size, dtype=np.int32), nseg)
                self.split_rows.append((to_dev(rows_s), torch.from_numpy(seg_ptr).to(dev),
                                        torch.from_numpy(seg_row).to(dev)))
                max_segments = max(max_segments, int(seg_ptr[-1]))
            else:
                self.split_rows.append(None)
            self.partial_off.append(self.partial_off[-1] + self.k.spmm_partials_len(b.nrows, int(is_long.sum())))
        # one set of fixed-order delta partials per column tile, summed together by the sweep's final reduction
        self.partials = torch.zeros(self.partial_off[-1] * len(self.tiles), dtype=torch.float64, device=dev)
        # segment sums of the split hub rows: one slab per launch stream (blocks on a stream run in order)
        slab_len = max(1, self.k.spmm_split_slab_len(max_segments, max(self.d_plan, 1)))
        if max_slots:
            slab_len = max(slab_len, self.k.spmm_class_slab_len(max_slots, max(self.d_plan, 1)), 2 * max_slots)   # K1: 2 stats per slot
        # one slab per launch stream: blocks alternate between two side streams when there are several
        self.slabs = [torch.zeros(slab_len, dtype=self.acc_dtype, device=dev)
                      for _ in range(2 if len(self.blocks) > 1 else 1)]

    def _setup_exchange(self, deg: np.ndarray, fused_pack: bool) -> None:
        """Halo exchange: send lists and send buffers (one per own chunk).  The kernel that finishes a row also stores
        it to its slots of the send buffer (`mirrors`: row -> slots, the inverse of send_rows), so no separate packing
        pass runs between the kernels and the exchange."""
        dev = self.device
        self.send_rows: List[Optional[torch.Tensor]] = []
        self.send_buf: List[Optional[torch.Tensor]] = []
        self.mirrors: List[Optional[object]] = []
        self.fused_pack = bool(fused_pack)
        for b in self.blocks:
            ex = b.exchange
            has = ex is not None and ex.send_rows.size > 0 and not self.p2p
            self.send_rows.append(torch.from_numpy(ex.send_rows).to(dev) if has else None)
            self.send_buf.append(torch.zeros(ex.send_rows.size, self.ld, dtype=self.dtype, device=dev) if has
                                 else (torch.zeros(0, self.ld, dtype=self.dtype, device=dev) if ex is not None else None))
            mirror = None
            if has and self.fused_pack:
                rel = ex.send_rows.astype(np.int64) - b.local_start
                if rel.min() < 0 or rel.max() >= b.nrows or (deg[ex.send_rows] == 0).any():
                    raise AssertionError("halo send list holds a row outside its chunk or a row that never changes")
                row_ptr = np.zeros(b.nrows + 1, dtype=np.int64)
                np.cumsum(np.bincount(rel, minlength=b.nrows), out=row_ptr[1:])
                slot = np.argsort(rel, kind="stable").astype(np.int32)
                mirror = self._make_mirror(torch.from_numpy(row_ptr).to(dev), torch.from_numpy(slot).to(dev),
                                           self.send_buf[-1])
            self.mirrors.append(mirror)

    def _new_table(self, tag: str, rows: int) -> torch.Tensor:
        """One of the big [rows, ld] tables (Zbuf0 / Zbuf1 / X), zeroed.  ``table_skew`` = {tag: bytes} (experiments:
        tools/placement_probe.py) carves it out of a private allocation at that offset from a 2-MiB boundary."""
        skew = (self.table_skew or {}).get(tag)
        if self.table_alloc == "contiguous":
            try:        # physically contiguous backing: its own allocation, kept alive by this engine
                buf = self.k.contiguous_matrix((rows, self.ld), self.dtype, self.device)
                self._own_tables.append(buf)
                return buf.tensor
            except _hip.ClaneHipError as exc:               # no contiguous range free: an ordinary allocation will do
                self.table_alloc_note = f"{tag}: {exc}"
        if skew is None or self.device.type != "cuda":
            return torch.zeros(rows, self.ld, dtype=self.dtype, device=self.device)
        es = torch.empty(0, dtype=self.dtype).element_size()
        nbytes = rows * self.ld * es
        raw = torch.zeros(nbytes + (4 << 20) + int(skew), dtype=torch.uint8, device=self.device)
        start = (-raw.data_ptr()) % (2 << 20) + int(skew)
        return raw[start:start + nbytes].view(self.dtype).view(rows, self.ld)

    def _alloc_tables(self, X: torch.Tensor, deg: np.ndarray) -> None:
        """The embedding tables (two ping-pong Z buffers, the owned rows of X) and the maps between vertex order and
        engine order, which is applied ON THE DEVICE (an upload / download plus one indexed copy): a 2 GB matrix permuted
        with host fancy-indexing costs more than a whole iterate() at config 3."""
        dev = self.device
        if self.halo:
            self.table_vertex = torch.from_numpy(self.part.table_vertex).to(dev)   # int64 [table_rows]
            self.slot = torch.from_numpy(self.part.vertex_slot).to(dev)            # int64 [V]
        else:
            self.pos = torch.from_numpy(self.part.position_of_vertex()).to(dev)    # int64 [V]
        if self.p2p:        # own allocations that the other processes map; views[q][p] = rank q's table p
            self._shared = [self.k.shareable_matrix((self.part.padded_vertices, self.ld), self.dtype, dev)
                            for _ in range(self.N_TABLES)]
            self.Zbuf = [b.tensor for b in self._shared]
            self.peer_tables, self._mapped = self.comm.share_matrices(self.k, self._shared)
            self._build_p2p_mirrors(deg)
        else:           # the third table comes with the first snapshot() (_third_table)
            self.Zbuf = [self._new_table(f"Z{i}", self.part.padded_vertices) for i in range(2)]
        # cur: the table holding the current embeddings; hold: the one snapshot() pinned (never a sweep's destination);
        # _prev_cur: what cur was before the latest sweep_launch (discard_launch); _tick: which of the two delta slots
        # the next launch uses
        self.cur, self.hold, self._prev_cur, self._tick = 0, None, 0, 0
        Xd = X.to(dev)                                                             # this rank's columns, [V, d]
        verts = torch.from_numpy(self.local.vertex).to(dev)
        ok = verts >= 0
        self.X_loc = self._new_table("X", self.part.n_local)
        self.X_loc[ok, :self.d] = Xd[verts[ok]]
        self.quiet_stale = False          # other ranks' quiet rows are only refreshed when Z is read out
        self._load_Z(Xd)

    def _alloc_scratch(self, delta_stream: bool, overlap_chunks: bool) -> None:
        """Scalars, reduction workspaces, streams, timing state."""
        dev = self.device
        self.ws = torch.zeros(self.k.reduce_ws_len(), dtype=torch.float64, device=dev)
        self.sums2 = torch.zeros(2, dtype=torch.float64, device=dev)
        self.delta = torch.zeros(1, dtype=torch.float64, device=dev)
        # per ping-pong parity: the sweep's delta on the device, its copy in pinned host memory, the copy's event
        self.delta_pp = torch.zeros(2, dtype=torch.float64, device=dev)
        self._delta_host = torch.zeros(2, dtype=torch.float64)
        self._delta_ev = None
        if self.device.type == "cuda":
            self._delta_host = self._delta_host.pin_memory()
            self._delta_ev = [torch.cuda.Event() for _ in range(2)]
        # delta_stream (several ranks, opt-in): the delta's all-reduce and host copy run on a stream of their own, so a
        # sweep launched ahead does not queue behind RCCL's latency.  With ONE rank (no xGMI hop) the extra event costs
        # what it saves (profiles/r02_delta_stream_ab.md); to be decided on a real multi-GPU box.
        self._delta_stream = None
        self._delta_busy = [False, False]
        if self.device.type == "cuda" and (self.world > 1 or self._forced) and delta_stream:
            self._delta_stream = torch.cuda.Stream(self.device)
        self.block_out = torch.zeros(len(self.blocks), dtype=torch.float64, device=dev)
        # |z_v|^2 of the owned rows, one copy per Z table, and whether it matches the table (`sq_ok`): written by K0 in
        # build_P or by the outer-delta pass (l1_between), which reads every row of the new Z anyway -- free, and what
        # Embedder.iterate() lives on.  (The K3 kernels could leave them behind too: built in round 4, +1.1-1.4 % per
        # sweep to save 0.34 ms per build_P, a net loss -- removed; profiles/r04_fused_norms_ab.jsonl.)
        self.sq_pp = [torch.zeros(self.part.n_local, dtype=self.acc_dtype, device=dev) for _ in self.Zbuf]
        self.sq_ok = [False] * len(self.Zbuf)
        self.sq_full: Optional[torch.Tensor] = None
        self.sweeps_done = 0
        # optional per-kernel timing with HIP events on the launch stream (bench.py)
        self._plans = {}                 # (src, dst, tick, gamma, stream) -> launch list (see _build_plan)
        # Alternate chunks go to two side streams, so the tail of one chunk's kernels overlaps the head of the
        # next chunk's (a chunk at 8 GPUs is only ~0.25 ms of kernels: ramp-up and tail are a third of it).
        self.side_streams = None
        if overlap_chunks and len(self.blocks) > 1 and self.device.type == "cuda":
            self.side_streams = [torch.cuda.Stream(self.device) for _ in range(2)]
        self.time_kernels = False
        # bench.py, N > 1: HIP events on the sweep's stream around what it spends waiting for the other ranks -- the
        # row exchange still outstanding after its own kernels (what the overlap did not hide) and the scalar all-reduce
        self.time_collectives = False
        self.collective_events = []      # [(after kernels, after exchange waits, after all-reduce)]
        self._stage_cv, self._stage_free = threading.Condition(), None      # stage_Z: slots made on first use
        self._own_vertex = None          # stage_Z(pieces=True) on a row split: vertex id of every own row
        self.kernel_events = []          # [(block, start, after_hub, after_mid, after_main)]
    def use_delta_stream(self, on: bool) -> None:
        """Switch the delta's all-reduce + host copy onto a stream of their own (or back onto the sweep's) between
        sweeps: the constructor's ``delta_stream`` made switchable, so that one run on real GPUs can time both
        (bench.py's `comm.delta_stream_ab`).  Results are bit-identical either way."""
        if self.device.type != "cuda" or not (self.world > 1 or self._forced):
            return
        torch.cuda.synchronize(self.device)         # nothing of the other mode is in flight
        self._delta_busy = [False, False]
        self._delta_stream = torch.cuda.Stream(self.device) if on else None

    def _build_p2p_mirrors(self, deg) -> None:
        """halo_p2p: for each destination table and own chunk, where every finished row has to go -- (rank q,
        row of q's table) for each rank q that reads it.  q's table rows come from q's own layout (the start of
        the chunk's halo slice and the rows the ranks before me put there), gathered once."""
        me, W, dev = self.comm.rank, self.world, self.device
        mine = [(b.exchange.recv_start, list(b.exchange.out_splits)) for b in self.blocks]
        layouts = self.comm.all_gather_object(mine)
        self.mirrors_p2p = [[None] * len(self.blocks) for _ in range(self.N_TABLES)]
        for i, b in enumerate(self.blocks):
            ex = b.exchange
            if ex.send_rows.size == 0:
                continue
            place = np.empty(ex.send_rows.size, dtype=np.int64)
            off = 0
            for q in range(W):
                n = ex.in_splits[q]
                if n == 0:
                    continue
                start_q, from_ranks = layouts[q][i]
                if from_ranks[me] != n:
                    raise AssertionError(f"rank {q} expects {from_ranks[me]} rows of chunk {i} from rank {me}, not {n}")
                base = start_q + sum(from_ranks[:me])
                place[off:off + n] = (q << _hip.MIRROR_ROW_BITS) | (base + np.arange(n, dtype=np.int64))
                off += n
            rel = ex.send_rows.astype(np.int64) - b.local_start
            if rel.min() < 0 or rel.max() >= b.nrows or (deg[ex.send_rows] == 0).any():
                raise AssertionError("halo send list holds a row outside its chunk or a row that never changes")
            row_ptr = np.zeros(b.nrows + 1, dtype=np.int64)
            np.cumsum(np.bincount(rel, minlength=b.nrows), out=row_ptr[1:])
            slot = place[np.argsort(rel, kind="stable")].astype(np.int32)
            rp_d, slot_d = torch.from_numpy(row_ptr).to(dev), torch.from_numpy(slot).to(dev)
            for table in range(self.N_TABLES):
                self.mirrors_p2p[table][i] = self._make_mirror(rp_d, slot_d,
                                                               [self.peer_tables[q][table] for q in range(W)])

    def _barrier(self) -> None:
        """Every rank has got here and its queued device work is done (an all-reduce, then the host waits)."""
        flag = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._all_reduce(flag)
        flag.item()

    def _make_mirror(self, row_ptr, slot, buf):
        return self.k.make_mirror(row_ptr, slot, buf)

    # ---- views of one block -------------------------------------------------------------
    def _rows(self, b: Block):
        return slice(b.local_start, b.local_start + b.nrows)

    def _zrows(self, Z: torch.Tensor, b: Block) -> torch.Tensor:
        return Z[b.row0:b.row0 + b.nrows]

    # ---- Z in / out -------------------------------------------------------------------
    @property
    def Zcur(self) -> torch.Tensor:
        return self.Zbuf[self.cur]

    def set_Z(self, Z: torch.Tensor) -> None:
        """Load a [V, d] matrix (vertex order) into the full-Z buffers."""
        if tuple(Z.shape) != (self.V, self.d_full):
            raise ValueError(f"set_Z: expected {(self.V, self.d_full)}, got {tuple(Z.shape)}")
        self._load_Z(Z.detach()[:, self.col0:self.col1].to(self.device, self.dtype))

    def _load_Z(self, Zd: torch.Tensor) -> None:
        """`Zd`: this rank's columns of the matrix, on the device, vertex order."""
        first = self.Zbuf[0]
        first.zero_()
        if self.halo:       # own rows, every remote row this rank reads, and the constant (sink) halo rows
            held = self.table_vertex >= 0
            first[held, :self.d] = Zd[self.table_vertex[held]]
        else:
            first[self.pos, :self.d] = Zd
        # EVERY table: rows without out-edges never change (embedder.py:88-89), so the sweep
        # kernel leaves them alone (CLANE_SPMM_SINKS_UNTOUCHED) and relies on all copies agreeing.
        for other in self.Zbuf[1:]:
            other.copy_(first)
        self.cur, self.hold, self._prev_cur = 0, None, 0
        self.sq_ok = [False] * len(self.Zbuf)
        self.P_valid = False
        self.quiet_stale = False
        if self.p2p:        # nobody may store into a table that its owner is still loading
            self._barrier()

    def _own_columns_in_vertex_order(self) -> torch.Tensor:
        """[V, ld] on the device: every row (vertex order) of the columns this rank holds (collective when rows are
        divided: the ranks of a column group put their rows together)."""
        if self.halo:       # own rows of every rank of the column group, in slot order (slot = owner * n_local + local row)
            n = self.part.n_local
            everyone = torch.empty(self.row_world * n, self.ld, dtype=self.dtype, device=self.device)
            self.comm.all_gather_into(everyone, self.Zcur[:n].contiguous())
            return everyone[self.slot]
        self._sync_quiet_rows()
        return self.Zcur[self.pos]

    def get_Z(self) -> torch.Tensor:
        """Current embeddings as a fresh CPU tensor [V, d] in vertex order (collective when N > 1)."""
        Zv = self._own_columns_in_vertex_order()
        if not self.columns:
            return Zv[:, :self.d].cpu()
        # column slices of every column group, padded to the widest one
        mine = torch.zeros(self.V, self.ld_max, dtype=self.dtype, device=self.device)
        mine[:, :self.ld] = Zv
        del Zv
        everyone = torch.empty(self.C * self.V, self.ld_max, dtype=self.dtype, device=self.device)
        self.comm.all_gather_into(everyone, mine)
        everyone = everyone.view(self.C, self.V, self.ld_max)
        out = torch.empty(self.V, self.d_full, dtype=self.dtype, device=self.device)
        for c in range(self.C):
            c0, c1 = column_slice(self.d_full, self.dtype, self.C, c)
            if c1 > c0:
                out[:, c0:c1] = everyone[c][:, :c1 - c0]
        return out.cpu()

    def _sync_quiet_rows(self) -> None:
        """Quiet rows (no out-edges, or never read) are not exchanged during sweeps; bring the other
        ranks' copies up to date before the matrix leaves the engine."""
        span = None if self.halo else self.part.quiet_span()
        if span is None or not self.quiet_stale:
            return
        begin, end, q = span
        mine = self.Zcur[begin + self.part.rank * q: begin + (self.part.rank + 1) * q]
        self.comm.all_gather_into(self.Zcur[begin:end], mine)
        self.quiet_stale = False

    # ---- build_P (graph.py:118-128) -----------------------------------------------------
    def build_P(self) -> None:
        k, part = self.k, self.part
        Z = self.Zcur
        mode = _hip.SCORE_MODES[self.cosine_mode]
        busy = self.d > 0                   # a column-split rank without columns only joins the collectives
        sq = None
        # the own rows' norms -- unless a row division in per_edge mode is about to norm the WHOLE table (own rows and
        # the copies of the rows it reads) in one pass below
        whole_table = self.cosine_mode == "per_edge" and self.row_world > 1
        if not self.sq_ok[self.cur] and not whole_table:    # nobody has left this table's norms behind: K0
            if busy:
                for b in self.blocks:
                    k.row_sqnorm(self._zrows(Z, b), self.d, self.sq_pp[self.cur][self._rows(b)])
            self.sq_ok[self.cur] = True
        sq_own = self.sq_pp[self.cur]       # K0's bits, whoever wrote them (tests: test_row_norms_are_bitwise_k0)
        if self.cosine_mode == "reference":
            k.degree_weighted_sums(sq_own, self.rowptr, self.indeg, part.n_local, self.ws, self.sums2)
            self._all_reduce(self.sums2)    # partial over the owned rows, or over the owned columns: a sum either way
        else:
            if self.sq_full is None:
                self.sq_full = torch.zeros(part.padded_vertices, dtype=self.acc_dtype, device=self.device)
            if self.row_world == 1:         # every table row is an own row, in table order (a column rank: of its columns)
                self.sq_full.copy_(sq_own)
            elif busy:
                k.row_sqnorm(Z, self.d, self.sq_full)   # row split: every rank holds valid copies of all rows it reads
            if self.columns:
                self._all_reduce(self.sq_full)                    # partial norms over the ranks' column slices
            sq = self.sq_full
        if self.E_loc > 0 and not self.columns:
            for i, b in enumerate(self.blocks):
                rp = self.rowptr[b.local_start:]
                # K1 soft-maxes every row it scores: one wave in registers / online, a listed row by its workgroup
                k.edge_score(rp, self.colidx, b.nrows, b.row0, Z, self.d, mode, self.sums2, sq, self.P,
                             self.k1_threshold, self.k1_long_rows[i], fuse_softmax=True)
                if self.class_k1 and self.class_rows[i] is not None:
                    rows_c, slot_ptr, it_e0, it_len, it_slot, it_row, ipb = self.class_rows[i]
                    k.edge_score_class(rp, self.colidx, it_e0, it_len, it_slot, it_row, ipb, rows_c,
                                       slot_ptr, b.row0, Z, self.d, mode, self.sums2, sq, self.P,
                                       self.slabs[i % len(self.slabs)], fuse_softmax=True,
                                       n_slots=self.class_slots[i], row_parts=self.softmax_row_parts)
        elif self.E_loc > 0:
            # column split: dot products of the owned columns, summed over the ranks, then denominators + softmax
            if busy:
                for i, b in enumerate(self.blocks):
                    k.edge_score(self.rowptr[b.local_start:], self.colidx, b.nrows, b.row0, Z, self.d,
                                 _hip.SCORE_RAW_DOT, None, None, self.P, self.k1_threshold, self.k1_long_rows[i])
                    if self.class_k1 and self.class_rows[i] is not None:
                        rows_c, slot_ptr, it_e0, it_len, it_slot, it_row, ipb = self.class_rows[i]
                        k.edge_score_class(self.rowptr[b.local_start:], self.colidx, it_e0, it_len, it_slot, it_row,
                                           ipb, rows_c, slot_ptr, b.row0, Z, self.d,
                                           _hip.SCORE_RAW_DOT, None, None, self.P)
            else:
                self.P.zero_()
            self._all_reduce(self.P)                              # every rank holds the same rows, other columns
            for i, b in enumerate(self.blocks):
                rp = self.rowptr[b.local_start:]
                k.edge_score_finalize(rp, self.colidx, b.nrows, b.row0, mode, self.sums2, sq, self.P)
                k.segment_softmax(rp, b.nrows, self.P, 0, self.score_threshold if self.long_rows[i] is not None else 0,
                                  self.long_rows[i])
        self.P_valid = True

    def set_cosine_mode(self, mode: str) -> None:
        """Switch between the reference's scores (global Frobenius denominators, similarity.py:35-37) and true per-edge
        cosine for the NEXT build_P; the row layout does not depend on it."""
        if mode not in ("reference", "per_edge"):
            raise ValueError(f"cosine_mode must be 'reference' or 'per_edge', got {mode!r}")
        if mode != self.cosine_mode:
            self.cosine_mode, self.P_valid = mode, False

    def P_global(self) -> torch.Tensor:
        """P values of the rows this rank owns as a CPU tensor in the GLOBAL (row, col)-sorted edge order,
        zeros elsewhere (the engine stores them in its own row / column order)."""
        out = torch.zeros(int(self.local.edge_origin.max()) + 1 if self.E_loc else 0, dtype=self.acc_dtype)
        out[torch.from_numpy(self.local.edge_origin)] = self.P[:self.E_loc].to("cpu")
        return out

    # ---- one sweep (embedder.py:84-94) --------------------------------------------------
    def _bind(self, method: str, *args, **kwargs):
        return self.k.bind(method, *args, **kwargs)              # HipKernels: the pre-marshalled ABI call

    def _build_plan(self, cur: int, dst: int, tick: int, gamma: float):
        """Launch lists of one sweep reading Zbuf[cur] and writing Zbuf[dst]: per block, bound kernel calls, event
        marks and the exchange; then the final reduction (into delta slot `tick`).  Everything that can be computed once (views, pointers, offsets,
        the stream each call goes to) is, so the per-sweep host cost is a few microseconds per launch -- it
        matters at 8 GPUs, where a sweep is ~1 ms of GPU time."""
        k = self.k
        Zold, Znew = self.Zbuf[cur], self.Zbuf[dst]
        per_block = []
        for i, b in enumerate(self.blocks):
            steps = []
            ctx = torch.cuda.stream(self.side_streams[i % 2]) if self.side_streams else contextlib.nullcontext()
            with ctx:            # bound calls capture the current stream
                rp, Xb, Zn = self.rowptr[b.local_start:], self.X_loc[self._rows(b)], self._zrows(Znew, b)
                po = self.partial_off[i]
                mir = self.mirrors_p2p[dst][i] if self.p2p else self.mirrors[i]
                po_mid = po + k.spmm_partials_len(b.nrows, 0)
                po_hub = po_mid + (0 if self.mid_rows[i] is None else self.mid_rows[i].numel())
                po_split = po_hub + (0 if self.hub_rows[i] is None else self.hub_rows[i].numel())
                po_class = po_split + (0 if self.split_rows[i] is None else self.split_rows[i][0].numel())
                # biggest rows first: split hubs, 16-wave rows, (4-wave rows), then the one-(sub-)wave-per-row pass
                steps.append(("event", i, 0))
                if self.d == 0:              # column-split rank without columns: nothing to launch, delta stays 0
                    per_block.append(steps + [("event", i, e) for e in (4, 1, 2, 3)])
                    continue
                # one launch per kernel and COLUMN TILE (one tile unless column_tiles > 1): views of the same tables
                T = len(self.tiles)
                cut = (lambda M, t: M) if T == 1 else (lambda M, t: M[:, self.tiles[t][0]:self.tiles[t][1]])
                width = (lambda t: self.d) if T == 1 else (lambda t: self.tiles[t][1] - self.tiles[t][0])
                part = lambda off, t: self.partials[t * self.partial_off[-1] + off:]      # noqa: E731
                if mir is not None and T > 1:
                    raise AssertionError("column tiles and mirrored launches do not combine")
                for t in range(T):
                    if self.class_rows[i] is not None:
                        rows_c, slot_ptr, it_e0, it_len, it_slot, _, ipb = self.class_rows[i]
                        steps.append(("call", self._bind("spmm_update_class", self.colidx, self.P, it_e0, it_len, it_slot,
                                                         ipb, rows_c, slot_ptr, b.row0, cut(Zold, t), cut(Xb, t), gamma,
                                                         cut(Zn, t), width(t), self.slabs[i % len(self.slabs)],
                                                         part(po_class, t), mirror=mir, beyond_cache=self.beyond_cache)))
                    if self.split_rows[i] is not None:
                        rows_s, seg_ptr, seg_row = self.split_rows[i]
                        steps.append(("call", self._bind("spmm_update_split", rp, self.colidx, self.P, rows_s, seg_ptr,
                                                         seg_row, self.segment_edges, b.row0, cut(Zold, t), cut(Xb, t), gamma,
                                                         cut(Zn, t), width(t), self.slabs[i % len(self.slabs)],
                                                         part(po_split, t), mirror=mir)))
                steps.append(("event", i, 4))
                for t in range(T):
                    if self.hub_rows[i] is not None:
                        steps.append(("call", self._bind("spmm_update_long", rp, self.colidx, self.P, self.hub_rows[i], 16,
                                                         b.row0, cut(Zold, t), cut(Xb, t), gamma, cut(Zn, t), width(t),
                                                         part(po_hub, t), mirror=mir)))
                steps.append(("event", i, 1))
                for t in range(T):
                    if self.mid_rows[i] is not None:
                        steps.append(("call", self._bind("spmm_update_long", rp, self.colidx, self.P, self.mid_rows[i], 4,
                                                         b.row0, cut(Zold, t), cut(Xb, t), gamma, cut(Zn, t), width(t),
                                                         part(po_mid, t), mirror=mir)))
                steps.append(("event", i, 2))
                for t in range(T):
                    steps.append(("call", self._bind("spmm_update", rp, self.colidx, self.P, b.nrows, b.row0, cut(Zold, t),
                                                     cut(Xb, t), gamma, cut(Zn, t), width(t), self.long_threshold,
                                                     part(po, t), sinks_untouched=True, mirror=mir,
                                                     beyond_cache=self.beyond_cache)))
                steps.append(("event", i, 3))
                if b.span is not None:
                    steps.append(("allgather", Znew[b.span[0]:b.span[1]], Zn))
                elif b.exchange is not None and not self.p2p:  # halo: the rows others read are packed; swap, no unpack
                    ex = b.exchange
                    if self.send_rows[i] is not None and mir is None:
                        steps.append(("call", self._bind("gather_rows", Znew, self.send_rows[i], self.d,
                                                         self.send_buf[i])))
                    steps.append(("alltoall", Znew[ex.recv_start:ex.recv_start + ex.recv_rows], self.send_buf[i],
                                  ex.out_splits, ex.in_splits))
            per_block.append(steps)
        final = self._bind("reduce_partials", self.partials, self.partials.numel(), self.ws,
                           self.delta_pp[tick:tick + 1])
        return per_block, final

    def sweep(self, gamma: float) -> float:
        """Z <- X + gamma * P Z on the owned rows, exchange, return sum|Z_new - Z_old| (global)."""
        return self.sweep_wait(self.sweep_launch(gamma))

    def sweep_launch(self, gamma: float) -> int:
        """Enqueue one sweep (kernels, exchange, delta reduction, all-reduce, copy of the delta to pinned host
        memory) and return a ticket for ``sweep_wait``.  Nothing blocks the host, so the NEXT sweep can be
        launched before this one's delta is read: it reads the table this one writes, and if the host
        then decides to stop, ``discard_launch()`` drops it -- the table it read still holds this sweep's Z.
        The destination is the table that is neither the current one nor the one ``snapshot()`` pinned."""
        if not self.P_valid:
            raise RuntimeError("sweep() before build_P()")
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
        src = self.cur
        dst = next(i for i in range(len(self.Zbuf)) if i != src and i != self.hold)
        parity = self._tick                 # delta slot of this launch (two launches may be in flight)
        self._tick ^= 1
        if self._delta_stream is not None and self._delta_busy[parity]:
            torch.cuda.current_stream(self.device).wait_event(self._delta_ev[parity])   # its last all-reduce has read delta_pp[parity]
            self._delta_busy[parity] = False
        key = (src, dst, parity, float(gamma), stream)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self._build_plan(src, dst, parity, float(gamma))
        events = None
        if self.time_kernels and len(self.kernel_events) < self.MAX_TIMED_SWEEPS * len(self.blocks):
            events = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in self.blocks]
            self.kernel_events.extend((i,) + tuple(ev) for i, ev in enumerate(events))
        per_block, final = plan
        works = []
        side = self.side_streams
        if side:
            main = torch.cuda.current_stream(self.device)
            start = torch.cuda.Event()
            start.record(main)
            for st in side:
                st.wait_event(start)
        for i, steps in enumerate(per_block):
            ctx = torch.cuda.stream(side[i % 2]) if side else contextlib.nullcontext()
            with ctx:
                for step in steps:
                    kind = step[0]
                    if kind == "call":
                        step[1]()
                    elif kind == "event":
                        if events is not None:
                            events[step[1]][step[2]].record()
                    elif kind == "allgather":
                        works.append(self.comm.all_gather_into(step[1], step[2], async_op=True))
                    else:
                        works.append(self.comm.all_to_all_rows(step[1], step[2], step[3], step[4], async_op=True))
        if side:
            for st in side:
                done = torch.cuda.Event()
                done.record(st)
                main.wait_event(done)
        final()
        mine = self.delta_pp[parity:parity + 1]
        cev = None
        if self.time_collectives and self._delta_stream is None and len(self.collective_events) < self.MAX_TIMED_SWEEPS:
            cev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            self.collective_events.append(cev)
            cev[0].record()
        for w in works:
            w.wait()
        if cev is not None:
            cev[1].record()
        if self._delta_stream is not None:
            # the scalar all-reduce and the copy of the delta leave the sweep's stream: the next sweep (launched ahead
            # by the lagged check) does not queue behind RCCL's latency.  delta_pp[parity] is next written two sweeps
            # on; sweep_launch makes that sweep wait for this event first.
            reduced = torch.cuda.Event()
            reduced.record()
            with torch.cuda.stream(self._delta_stream):
                self._delta_stream.wait_event(reduced)
                self._all_reduce(mine)
                self._delta_host[parity:parity + 1].copy_(mine, non_blocking=True)
                self._delta_ev[parity].record()
            self._delta_busy[parity] = True
        else:
            self._all_reduce(mine)
            if cev is not None:
                cev[2].record()
            self._delta_host[parity:parity + 1].copy_(mine, non_blocking=True)
            if self._delta_ev is not None:
                self._delta_ev[parity].record()
        self._prev_cur, self.cur = src, dst
        self.sq_ok[dst] = False
        self.sweeps_done += 1
        self.quiet_stale = True
        return parity

    def sweep_wait(self, ticket: int) -> float:
        """The delta of the sweep `ticket` came from (blocks until its copy has landed)."""
        if self._delta_ev is not None:
            self._delta_ev[ticket].synchronize()
        return float(self._delta_host[ticket])

    def discard_launch(self) -> None:
        """Forget the most recent ``sweep_launch`` (a sweep launched ahead of a stop decision): the current
        embeddings are again those of the sweep before it.  What it wrote sits in a table that a later sweep
        overwrites; rows without out-edges were not touched by it either.  (One launch can be taken back, not two.)"""
        self.cur = self._prev_cur
        self.sweeps_done -= 1

    # ---- outer-loop delta (embedder.py:58-60) -------------------------------------------
    def snapshot(self) -> None:
        """Remember the current embeddings (the reference's ``prev_Z = graph.Z.clone()``, embedder.py:58) WITHOUT a
        copy: the current table is pinned -- no sweep writes to it until the next snapshot() or set_Z()."""
        if len(self.Zbuf) < self.N_TABLES:
            self._third_table()
        self.hold = self.cur

    def _third_table(self) -> None:
        """The table the sweeps need once one is pinned, made on the first snapshot(): a copy of the current one, so the
        rows the kernels never write (no out-edges; other ranks' constant halo rows) agree in all three."""
        table = self._new_table(f"Z{len(self.Zbuf)}", self.part.padded_vertices)
        table.copy_(self.Zcur)
        self.Zbuf.append(table)
        self.sq_pp.append(torch.zeros_like(self.sq_pp[0]))
        self.sq_ok.append(False)

    def l1_between(self, i: int, j: int) -> float:
        """sum |Zbuf[i] - Zbuf[j]| over the owned rows, all ranks (embedder.py:60's reduction).  The pass reads every
        row of table i anyway, so it also leaves that table's row norms behind (sq_pp[i]: bit for bit K0's) -- the next
        build_P of the outer loop starts without a pass over Z."""
        for n, b in enumerate(self.blocks):
            if self.d > 0:
                self.k.l1_distance(self._zrows(self.Zbuf[i], b), self._zrows(self.Zbuf[j], b), self.d, self.ws,
                                   self.block_out[n:n + 1], sq_a=self.sq_pp[i][self._rows(b)])
        self.sq_ok[i] = True
        self.k.reduce_partials(self.block_out, len(self.blocks), self.ws, self.delta)
        self._all_reduce(self.delta)
        return float(self.delta.item())

    def distance_from_snapshot(self) -> float:
        if self.hold is None:
            raise RuntimeError("distance_from_snapshot() before snapshot()")
        return self.l1_between(self.cur, self.hold)

    # ---- collectives ------------------------------------------------------------------
    def _all_reduce(self, t: torch.Tensor) -> None:
        """Sum over all ranks."""
        if self.world > 1 or self._forced:
            self.comm.all_reduce_sum(t)

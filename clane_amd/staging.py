"""Embeddings on their way to the host while the sweeps go on (``--save_history`` at scale; the
reference stacks and copies Z synchronously after every sweep, embedder.py:96-97 / __main__.py:73-86)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


class StagedZ:
    """The embeddings of one moment on their way to the host (``SweepEngine.stage_Z``): the sweeps go on while
    the copy drains over PCIe; ``result()`` waits for it and returns a fresh ``[V, d]`` CPU tensor in vertex order.
    On several GPUs (``stage_Z(pieces=True)``) every rank stages only what it holds -- its columns of all rows, or its
    own rows -- and ``piece()`` returns that part with where it belongs (``place_piece`` puts parts together)."""

    def __init__(self, engine=None, slot=None, ready: Optional[torch.Tensor] = None, where: Optional[dict] = None):
        self._engine, self._slot, self._ready, self._where = engine, slot, ready, where

    def _resolve(self) -> torch.Tensor:
        if self._ready is None:
            eng, slot = self._engine, self._slot
            slot["done"].synchronize()
            self._ready = slot["host"][:, :eng.d].clone()
            eng._release_stage_slot(slot)
            self._engine = self._slot = None
        return self._ready

    def result(self) -> torch.Tensor:
        if self._where is not None:
            raise RuntimeError("this copy holds one rank's part of the matrix: use piece() / place_piece()")
        return self._resolve()

    def piece(self) -> dict:
        """{'kind': 'columns', 'c0', 'c1', 'Z': [V, c1 - c0]} or {'kind': 'rows', 'vertex': int64 [n], 'Z': [n, d]}."""
        if self._where is None:
            raise RuntimeError("this copy holds the whole matrix: use result()")
        return dict(self._where, Z=self._resolve())


def place_piece(out: torch.Tensor, piece: dict) -> None:
    """Write one rank's part (``StagedZ.piece()``) into the full ``[V, d]`` matrix ``out``."""
    c0, c1 = piece.get("c0", 0), piece.get("c1", out.shape[1])
    if "vertex" not in piece:                           # every row, some columns
        out[:, c0:c1] = piece["Z"]
    else:
        real = piece["vertex"] >= 0                     # padding rows of an equal-size row division hold no vertex
        out[piece["vertex"][real], c0:c1] = piece["Z"][real]


class StagingMixin:
    """``SweepEngine.stage_Z``: kept apart from the launch logic (engine.py)."""

    def stage_Z(self, pieces: bool = False) -> StagedZ:
        """Start copying the current embeddings to the host WITHOUT stalling the sweeps (``--save_history`` at
        scale): a device-to-device copy (into vertex order) on the sweep stream (the ping-pong buffer is overwritten two
        sweeps later, long before 2 GB have crossed PCIe), then an asynchronous D2H into pinned memory on a copy
        stream.  At most STAGE_SLOTS copies are in flight; with none free this call waits for ``result()`` of an
        earlier one (possibly on another thread).
        Several GPUs: ``pieces=True`` stages only what THIS rank holds -- its column slice of every row (column
        split) or its own rows (row splits) -- the same way and with no collective: N PCIe links drain in parallel and
        whoever wants the whole matrix puts the ranks' pieces together on the host (``StagedZ.piece``,
        ``place_piece``; ``Embedder`` does, through files).  Without ``pieces`` a multi-GPU run gathers synchronously
        (``get_Z``: collective), as a host-memory engine does."""
        if self.world > 1 and not pieces:
            return StagedZ(ready=self.get_Z())
        by_rows = self.row_world > 1            # this rank's own rows; else all rows (of its columns)
        cols = {"c0": self.col0, "c1": self.col1} if self.columns else {}
        if self.device.type != "cuda":          # host-memory engine (the CPU suite's test double): nothing to overlap
            if self.world == 1:
                return StagedZ(ready=self.get_Z())
            if by_rows:
                own = torch.cat([self._zrows(self.Zcur, b)[:, :self.d] for b in self.blocks]).clone()
                return StagedZ(ready=own, where=dict(cols, kind="rows",
                                                     vertex=torch.from_numpy(self.local.vertex.astype(np.int64))))
            return StagedZ(ready=self.Zcur[self.pos, :self.d].clone(), where=dict(cols, kind="columns"))
        n_rows = self.part.n_local if by_rows else self.V
        with self._stage_cv:
            if self._stage_free is None:
                self._stage_free, self._stage_made = [], 0
                self._copy_stream = torch.cuda.Stream(self.device)
            while not self._stage_free and self._stage_made >= self.STAGE_SLOTS:
                self._stage_cv.wait()
            if self._stage_free:
                slot = self._stage_free.pop()
            else:
                self._stage_made += 1
                slot = {"dev": torch.empty(n_rows, self.ld, dtype=self.dtype, device=self.device),
                        "host": torch.empty(n_rows, self.ld, dtype=self.dtype, pin_memory=True),
                        "done": torch.cuda.Event()}
        main = torch.cuda.current_stream(self.device)
        where = None
        if by_rows:                                             # own rows, block by block (local row order)
            for b in self.blocks:
                slot["dev"][self._rows(b)].copy_(self._zrows(self.Zcur, b))
            if self._own_vertex is None:
                self._own_vertex = torch.from_numpy(self.local.vertex.astype(np.int64))
            where = dict(cols, kind="rows", vertex=self._own_vertex)
        else:
            torch.index_select(self.Zcur, 0, self.pos, out=slot["dev"])     # vertex order, on the sweep stream
            if self.world > 1:
                where = dict(cols, kind="columns")
        copied = torch.cuda.Event()
        copied.record(main)
        self._copy_stream.wait_event(copied)
        with torch.cuda.stream(self._copy_stream):
            slot["host"].copy_(slot["dev"], non_blocking=True)
            slot["done"].record(self._copy_stream)
        return StagedZ(self, slot, where=where)

    def _release_stage_slot(self, slot) -> None:
        with self._stage_cv:
            self._stage_free.append(slot)
            self._stage_cv.notify()

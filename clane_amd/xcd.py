"""XCD affinity -- host side of the class-affine row kernels (csrc/spmm_update.h, csrc/edge_score.h).

MI355X has 8 XCDs with a private 4 MiB L2 each and deals the workgroups of a launch to them round-robin (workgroup w
runs on XCD (w + c) % 8 with c fixed within a launch; tools/xcc_map.hip, clane_xcc_ids).  When any workgroup may gather any row of Z, all eight L2s cache the same few
thousand hottest rows.  The class-affine kernels keep every gathered row on ONE XCD instead: each table row gets a
class 0..7 (`xcd_class`), the edges of a long row are sorted by (class of the column, column) -- `partition.localize`,
`halo.build_halo_layout` -- and cut into work items of one class (`class_items`), laid out so that the items of
class b run on the workgroups 8 j + b.  No counterpart in the reference (its loop is embedder.py:84-92 for every row
alike); measurements in profiles/r02_gather_rows_ceiling.md and r02_class_threshold_sweep.md.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

XCD_CLASSES = 8          # MI355X: 8 XCDs, one private L2 each


def xcd_class(position):
    """XCD class (0..7) of the table row at `position` (numpy array, torch tensor or int): an xor-fold of the
    position's 3-bit groups.  Two things matter.  (1) NOT position % 8: that pins three low address bits of every
    row an XCD gathers, and only part of its L2's channels / sets get used -- measured on the pure gather
    (profiles/r02_gather_rows_ceiling.md): 128-byte rows 13.4 TB/s with % 8, 19.4 TB/s with this; 256-byte rows
    17.1 -> 19.5; 1-KiB rows the same.  (2) Heat must be dealt evenly whatever the layout: 8 consecutive rows go to 8
    different classes, so the hottest rows of every sorted run of a table (hot-rows-first Z, but also each
    (chunk, source rank) list of a halo table) are spread over all XCDs.  ((position / 8) % 8 is as fast on the
    hot-first layout and 40 % slower on a halo table, whose runs each start with their 8 hottest rows.)"""
    p = position
    return (p ^ (p >> 3) ^ (p >> 6) ^ (p >> 9)) & (XCD_CLASSES - 1)


def xcd_subclass(position, row_degree, class_threshold: int, phase_threshold: int = 0, phases: int = 1):
    """Sort key of an edge inside its row, ahead of the column: 0 for rows of at most `class_threshold` edges (plain
    column order); the XCD class of the column for class rows; `phase * 8 + class` for HEAVY rows (more than
    `phase_threshold` edges, `phases` > 1), where phase = (column / 64) % phases.  Works on numpy arrays and torch
    tensors alike (`position` and `row_degree` of equal shape)."""
    cls = xcd_class(position)
    if phases > 1 and phase_threshold > 0:
        cls = cls + XCD_CLASSES * (((position >> 6) % phases) * (row_degree > phase_threshold))
    return cls * (row_degree > class_threshold)


# Rows above CLASS_THRESHOLD edges are gathered XCD-affine (csrc/spmm_update.h, spmm_class_chunk_kernel): their edges
# are sorted by (class of the column, column) -- xcd_class above --, cut into chunks of at most CLASS_CHUNK edges
# of one class, and the chunks of class b run on the workgroups 8 j + b = XCD b, so each XCD's 4 MiB L2 caches its own
# eighth of the hot rows instead of all eight caching the same ones.  Costs one partial sum (d accumulators, written +
# read once) per chunk, which is why short rows stay with the row kernels.  Threshold by rows per wave-instruction,
# 0 = off; measured (profiles/r02_class_threshold_sweep.md), sweep ms without -> with: config 3 (1-KiB rows)
# 5.36 -> 4.36 at 64; its column slices: 512-B rows 2.55 -> 1.99 at 64, 256-B rows 1.26 -> 1.04 at 128..256, 128-B rows
# 0.716 -> 0.632 at 256; config 4's shape (bf16, 256-B rows) 8.24 -> 7.44 at 256; config 2 (Z fits the Infinity Cache)
# 0.257 -> 0.249.  build_P scores these rows over the same chunks (class_k1: 5.7 -> 4.7 ms at config 3).
CLASS_THRESHOLD_BY_ROWS_PER_WAVE = {1: 64, 2: 64, 4: 256, 8: 256}
CLASS_CHUNK = 256
# HEAVY class rows are additionally PHASED in time: the rows of a class are split into sub-classes ((column / 64) %
# phases: every sub-class gets the same mix of hot and cold rows) and all chunks of sub-class 0 are scheduled before
# those of sub-class 1, ... -- at any moment an XCD's hot working set is 1/phases of its class.  At 1-KiB rows the 4096
# hottest rows of a class ARE the 4 MiB L2; halving / quartering that working set lifts the pure gather from 14.3 to
# 15.5 / 16.7 TB/s (profiles/r02_gather_rows_ceiling.md).  Every phase multiplies a row's pieces and partial sums,
# hence heavy rows only (above PHASE_THRESHOLD edges).  Measured (profiles/r02_class_threshold_sweep.md): config 3
# 4.31 -> 4.01 ms with 4 phases (build_P 4.72 -> 4.44); its 512-byte column slice 1.98 -> 1.91 with 2; narrower rows
# lose (their hot rows already fit).
PHASES_BY_ROWS_PER_WAVE = {1: 4, 2: 2, 4: 1, 8: 1}
# A one-GPU sweep cut into column tiles (engine._pick_tiles) plans for the TILE's row width, but build_P's K1 still reads
# whole rows over the same edge order.  0 = by the tile's width (the table above); 4 is the measured better choice at
# config 3 (two tiles of 512 bytes under 1-KiB rows: build_P 3.81 -> 3.65-3.67 ms, sweep 3.66-3.67 -> 3.65-3.66,
# profiles/r05_tiles_phases_ab.jsonl) -- switch it together with a re-take of the PMC passes: class_phases is part of
# kernel_config(), profiles/traffic.json's entry was measured with two.
PHASES_UNDER_COLUMN_TILES = 0
PHASE_THRESHOLD = 512
CLASS_ITEMS_PIECE_EDGES = 1 << 28
# Chunks per workgroup, at most.  Round 2 found 16 / 32 / 64 alike on config 3 and took 32; measured again in round 3
# across shapes (profiles/r03_items_per_block.md): 16 is within 1 % of the best everywhere and ahead of 32 where a
# launch holds fewer chunks -- config 2 (108 k chunks) 0.141 -> 0.107 ms for the class pass, config 3 2.355 -> 2.29 ms.
CLASS_ITEMS_PER_BLOCK = 16



def row_pieces(rowptr: np.ndarray, max_edges: int):
    """Consecutive row ranges [a, b) covering all rows, each with at most ``max_edges`` edges (a single row with more
    is a piece of its own)."""
    n = rowptr.size - 1
    a = 0
    while a < n:
        b = int(np.searchsorted(rowptr, rowptr[a] + max_edges, side="right")) - 1
        b = min(max(b, a + 1), n)
        yield a, b
        a = b


def class_items(rowptr: np.ndarray, colidx: np.ndarray, rows: np.ndarray, chunk: int, items_per_block: Optional[int],
                row_ids: Optional[np.ndarray] = None, colidx_dev: Optional[torch.Tensor] = None,
                phase_threshold: int = 0, phases: int = 1, mega_segment_edges: int = 0, mega_min_edges: int = 0) -> dict:
    """Work items of the XCD-affine pass over `rows` (absolute local row ids whose edges are sorted by
    (xcd_class(column), column)): every class segment of a row is cut into chunks of at most `chunk` edges.
    Slots -- where the partial sums go -- are numbered row by row, class by class, chunk by chunk, so a row's slots
    are contiguous (`slot_ptr`) and summed in that order.  Items are laid out for the kernel: blocks of
    `items_per_block` items of ONE class, block j of class b at block index 8 j + b, padded with empty items
    (len 0, slot -1).  With `phases` > 1 the rows above `phase_threshold` edges are cut by (phase, class) -- their
    edges sorted by xcd_subclass -- and their blocks come first, phase by phase (each phase a whole number of
    8-block rounds, so block index % 8 stays the class), then the blocks of the other rows.  Inside a class the items go row
    by row, except those of mega rows (`mega_segment_edges`), which come first ordered by their first column.  Returns int64 e0, int32 len, int32 slot, int32 row (flat, whole blocks; row = `row_ids` of the
    item's row, default `rows` itself), int64 slot_ptr [rows + 1] and the items_per_block used (None on entry: chosen
    by `items_per_block_for` from the item count, once the O(E) counting pass has it -- the layout itself is cheap)."""
    n = rows.size
    NS = XCD_CLASSES * max(1, phases)                       # sub-classes per row (phase-major)
    sizes = (rowptr[rows + 1] - rowptr[rows]).astype(np.int64)
    start = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    if colidx_dev is not None and colidx_dev.is_cuda:        # the O(E) part on the card (40M edges: 0.3 s on the host)
        dev = colidx_dev.device
        seg_len = np.empty(n * NS, dtype=np.int64)
        unsorted = False
        bounds = np.concatenate([[0], np.cumsum(sizes)])
        for a, b in row_pieces(bounds, CLASS_ITEMS_PIECE_EDGES):    # pieces bound the scratch memory on the card
            if bounds[b] == bounds[a]:
                seg_len[a * NS:b * NS] = 0
                continue
            sizes_t = torch.from_numpy(sizes[a:b]).to(dev)
            rid = torch.repeat_interleave(torch.arange(b - a, device=dev), sizes_t)
            idx = (torch.from_numpy(rowptr[rows[a:b]] - (start[a:b] - bounds[a])).to(dev)[rid]
                   + torch.arange(int(bounds[b] - bounds[a]), device=dev))
            sub = xcd_subclass(colidx_dev[idx].long(), sizes_t[rid], 0, phase_threshold, phases)
            key = rid * NS + sub
            unsorted = unsorted or (bool((key[1:] < key[:-1]).any()) if key.numel() > 1 else False)
            seg_len[a * NS:b * NS] = torch.bincount(key, minlength=(b - a) * NS).cpu().numpy()
    else:
        idx = np.repeat(rowptr[rows] - start, sizes) + np.arange(int(sizes.sum()), dtype=np.int64)
        rid = np.repeat(np.arange(n, dtype=np.int64), sizes)
        sub = xcd_subclass(colidx[idx].astype(np.int64), sizes[rid], 0, phase_threshold, phases)
        unsorted = idx.size > 1 and bool((np.diff(rid * NS + sub) < 0).any())
        seg_len = np.bincount(rid * NS + sub, minlength=n * NS)
    if unsorted:
        raise AssertionError("class rows must have their edges sorted by (xcd_subclass(column), column)")
    seg_e0 = np.repeat(rowptr[rows], NS) + (np.cumsum(seg_len) - seg_len - np.repeat(start, NS))
    nchunk = -(-seg_len // chunk)
    tot = int(nchunk.sum())
    if items_per_block is None:
        items_per_block = items_per_block_for(tot)
    seg_of = np.repeat(np.arange(n * NS), nchunk)
    within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(nchunk) - nchunk, nchunk)
    e0 = seg_e0[seg_of] + within * chunk
    ln = np.minimum(chunk, seg_len[seg_of] - within * chunk)
    slot_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(nchunk.reshape(n, NS).sum(1), out=slot_ptr[1:])
    item_cls = (seg_of % NS) % XCD_CLASSES
    heavy = sizes > phase_threshold if (phases > 1 and phase_threshold > 0) else np.zeros(n, dtype=bool)
    # rows whose average (phase, class) segment exceeds `mega_segment_edges` edges (0 = none): see the item order below
    # and that read at least `mega_min_edges` table rows (the engine passes a quarter of the table: two such rows
    # share much of what they gather; 256 hubs that each read 3.6 % of a 33M-row table do not, and lost 4 % by column)
    mega = ((sizes > mega_segment_edges * XCD_CLASSES * np.where(heavy, max(1, phases), 1)) & (sizes >= mega_min_edges)) \
        if mega_segment_edges > 0 else np.zeros(n, dtype=bool)
    # launch groups, in order: the heavy rows' items phase by phase, then everybody else's
    item_group = np.where(heavy[seg_of // NS], (seg_of % NS) // XCD_CLASSES, max(1, phases))
    ids = (rows if row_ids is None else row_ids).astype(np.int32)
    item_row = ids[seg_of // NS]
    pieces = []
    for g in range(max(1, phases) + 1):
        in_group = item_group == g
        if not in_group.any():
            continue
        per_class = [np.nonzero(in_group & (item_cls == c))[0] for c in range(XCD_CLASSES)]
        if mega.any():
            # MEGA rows -- one (phase, class) segment of the row alone is more than an L2 holds -- come first in their
            # class, their items in the order of their first column instead of row by row: chunks of DIFFERENT mega rows
            # that gather the same stretch of the table then run next to each other and share it through the L2 (ten rows
            # that read all 2M vertices: class pass 3.03 -> 1.68 ms).  Row by row, such a row has left the L2 long
            # before the next one comes by; for rows whose segments fit (config 3's 70k-edge hub: 2.2 MB) row order is
            # the better one (2.35 vs 2.41 ms).  Slots, and with them the order of every sum, stay row-major.
            ordered = []
            for pc in per_class:
                is_mega = mega[seg_of[pc] // NS]
                first = pc[is_mega]
                ordered.append(np.concatenate([first[np.argsort(colidx[e0[first]], kind="stable")], pc[~is_mega]]))
            per_class = ordered
        nblk = max(-(-len(pc) // items_per_block) for pc in per_class)
        flat = XCD_CLASSES * nblk * items_per_block
        g_e0 = np.zeros(flat, dtype=np.int64)
        g_len = np.zeros(flat, dtype=np.int32)
        g_slot = np.full(flat, -1, dtype=np.int32)
        g_row = np.zeros(flat, dtype=np.int32)
        for c, pc in enumerate(per_class):
            t = np.arange(len(pc))
            where = (t // items_per_block) * (XCD_CLASSES * items_per_block) + c * items_per_block + t % items_per_block
            g_e0[where], g_len[where], g_slot[where], g_row[where] = e0[pc], ln[pc], pc, item_row[pc]
        pieces.append((g_e0, g_len, g_slot, g_row))
    if not pieces:                                          # no edges at all: one round of empty blocks
        flat = XCD_CLASSES * items_per_block
        pieces.append((np.zeros(flat, dtype=np.int64), np.zeros(flat, dtype=np.int32),
                       np.full(flat, -1, dtype=np.int32), np.zeros(flat, dtype=np.int32)))
    out_e0, out_len, out_slot, out_row = (np.concatenate([p[i] for p in pieces]) for i in range(4))
    return {"e0": out_e0, "len": out_len, "slot": out_slot, "row": out_row, "slot_ptr": slot_ptr,
            "items_per_block": int(items_per_block)}


def items_per_block_for(n_items: int) -> int:
    """Chunks per workgroup of the class kernels: CLASS_ITEMS_PER_BLOCK when there are plenty, fewer (down to one per
    wave) when a launch holds few chunks -- a launch wants >= ~4096 workgroups to fill 256 CUs (a rank's quarter of
    the halo split at 8 GPUs holds 21k chunks: 650 workgroups of 32 ran at a third of the rate of 2 600 of 8)."""
    return int(min(CLASS_ITEMS_PER_BLOCK, max(4, 4 * -(-n_items // (4 * 4096)))))

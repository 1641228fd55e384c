"""The launch plan's constants and rules: at which out-degree a row changes kernel, when the XCD-affine pass pays, how
the sweep is divided over several GPUs.  Every number is a measurement on MI355X (profiles/); nothing here has a
counterpart in the reference, whose loop treats every row alike (embedder.py:84-92)."""
from __future__ import annotations

import numpy as np
import torch

from . import _hip
from .partition import HostCSR

# Rows are binned by out-degree once per graph (profiles/r01_threshold_sweep.md, r02_class_threshold_sweep.md):
#   deg <= T                 one (sub-)wave per row, rows claimed dynamically inside a workgroup
#                            (spmm_update_kernel when a row fills a wave, spmm_update_subrow_kernel otherwise)
#   deg >  class threshold   XCD-affine chunks + fixed-order combine (spmm_class_chunk_kernel; below)
#   in between               16-wave workgroup per row, 64-aligned slices, idle waves exit at once (spmm_long_kernel);
#                            rows above SPLIT_EDGES cut into segments -- only without the class pass
# The thresholds below are those of the row kernels on their own (class pass off):
# Measured on MI355X: a single wave walking a 65..1024-edge row of 1-KiB rows streams at a fraction of
# what the multi-wave kernel reaches, so T is small when a row fills a wave (d=256 fp32: T=32); with
# narrow rows (d=128 bf16: 4 rows per wave-instruction) the sub-wave kernel is the efficient one and a
# workgroup per 100-edge row is not, so T grows with the rows a wave covers per instruction.  Since every
# sub-wave claims its own rows the optimum is ~1024 (R-MAT 2M/40M: 2 rows/wave 2.82 ms at T=64 -> 2.58 at 1024;
# 4 rows/wave 1.40 at 384 -> 1.31 at 1024; the 10M-vertex power-law graph prefers 384..1024 and loses 5-19 % at
# 2048: a 2000-edge row walked by 8 lanes is the tail of its launch).
# A 4-wave bin (T < deg <= hub_threshold) exists in the ABI; it did not pay.
LONG_THRESHOLD_BY_ROWS_PER_WAVE = {1: 32, 2: 1024, 4: 1024, 8: 512}
HUB_FACTOR = 1
# Rows above SPLIT_EDGES edges are cut into segments, one 16-wave workgroup each: a 70k-edge hub done by ONE
# workgroup is a ~0.25 ms tail on every launch.  Segments are 4096 edges (256 per wave) when there are plenty of
# hub edges, down to 1024 when a rank holds few (8 GPUs: ~30 hub rows per rank would give < 100 workgroups).
SPLIT_EDGES = 4096
MIN_SEGMENT_EDGES = 1024
TARGET_SEGMENTS = 512                   # two workgroups per CU
# The XCD-affine pass buys L2 hits; it has something to buy only when the gathers are SKEWED -- when the rows the
# eight 4-MiB L2s can hold between them take a real share of all edge reads (config 3: 61 %, config 4's shape: ~65 %).
# On a graph whose destinations are spread evenly (a near-regular or uniform random graph: that share is the rows'
# share of the table, 1.6 % at 2M x 1 KiB) cutting a 70-edge row into 8 class pieces is pure overhead: measured 25.7 ms
# against 20.7 on a near-regular 2M / 128M graph (profiles/r03_threshold_robustness.md).  Below MIN_HOT_READ_SHARE only
# rows that need their work spread anyway (above HEAVY_ROW_EDGES: the class pass doubles as the hub splitter, a
# 2M-edge row scored by ONE workgroup took build_P from 6.6 to 29 ms) take the pass.
SOFTMAX_EDGES_PER_WORKGROUP = 1 << 17     # build_P: edges of one class row that one workgroup of the rescale pass takes
L2_BYTES_ALL_XCDS = 8 * 4 * 1024 * 1024
INFINITY_CACHE_BYTES = 256 * 1024 * 1024
MIN_HOT_READ_SHARE = 0.2
HEAVY_ROW_EDGES = 4096
UNSKEWED_LONG_THRESHOLD = 128


def hot_read_share(csr: HostCSR, row_bytes: int) -> float:
    """Share of all edge reads that go to the rows the eight L2s can hold between them (the most-read rows first) --
    what XCD affinity can turn into L2 hits at best.  The same number on every rank (global in-degrees)."""
    V, E = csr.num_vertices, csr.num_edges
    k = L2_BYTES_ALL_XCDS // max(int(row_bytes), 1)
    if E == 0 or k >= V:
        return 1.0
    indeg = csr.indeg()
    return float(np.partition(indeg, V - k)[V - k:].sum(dtype=np.int64)) / E


def lanes_per_row(d: int, dtype: torch.dtype) -> int:
    """Lanes that cover one row with 16-byte packs (mirrors pick_layout in csrc/clane_abi.hip)."""
    packs = -(-d // _hip.VEC_ELEMS[dtype])
    return 8 if packs <= 8 else 16 if packs <= 16 else 32 if packs <= 32 else 64


def _round_up(a: int, b: int) -> int:
    return -(-a // b) * b


MIN_SLICE_ROW_BYTES = 64


def pick_exchange(d: int, dtype: torch.dtype, world: int) -> str:
    """The division ``exchange="auto"`` takes.  Column split while a rank's slice of a row is at least
    MIN_SLICE_ROW_BYTES (HBM is fetched in 64/128-byte lines: below that every gather drags in bytes of columns the rank
    does not own -- measured on the 10M-vertex bf16 graph); else divide the rows and exchange halo rows.
    (A 2-D division -- R row groups x C column groups -- was built and measured in round 4 and removed in round 5: on a
    fully connected fabric it loses to the halo division, because a column group's row exchange runs over R - 1 of a
    GPU's 7 links instead of all of them; numbers in profiles/HISTORY.md, profiles/r04_rank_compute_grid_powerlaw10m.jsonl.)"""
    row_bytes = d * torch.empty(0, dtype=dtype).element_size()
    return "columns" if row_bytes // max(world, 1) >= MIN_SLICE_ROW_BYTES else "halo"


def column_slice(d: int, dtype: torch.dtype, world: int, rank: int):
    """Columns [c0, c1) of the embedding matrix held by `rank` in a column-split run: contiguous, in whole
    16-byte packs, as even as the pack count allows (a rank may hold none when d is tiny)."""
    vec = _hip.VEC_ELEMS[dtype]
    packs = -(-d // vec)
    base, rem = divmod(packs, world)
    p0 = rank * base + min(rank, rem)
    p1 = p0 + base + (1 if rank < rem else 0)
    return min(d, p0 * vec), min(d, p1 * vec)

"""Workloads, tolerances, constants and the clock of one bench.py command."""
from __future__ import annotations

import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
BENCH_SCRIPT = ROOT / "bench.py"

WORKLOADS = {
    # name: (generator, V, E, d, dtype, graph seed, X seed)            -- SURVEY.md section 8d
    "rmat2m": ("rmat", 2_000_000, 40_000_000, 256, "f32", 3, 4),      # BASELINE config 3 (headline metric)
    "rmat200k": ("rmat", 200_000, 4_000_000, 128, "f32", 1, 2),       # BASELINE config 2
    "powerlaw10m": ("powerlaw", 10_000_000, 200_000_000, 128, "bf16", 5, 6),   # BASELINE config 4 (shape)
    # The roofline's ANCHOR: uniform-random (src, dst) pairs, no hubs, no skew -- nothing for the L2s or the Infinity Cache
    # to reuse (the 2 GB table is 8x the cache), so the PMC traffic equals the algorithmic bytes and `frac` is a true HBM
    # fraction with no cache caveat.  Same |V|, |E|, d as the headline.
    "uniform2m": ("uniform", 2_000_000, 40_000_000, 256, "f32", 12, 4),
    "rmat200k256": ("rmat", 200_000, 4_000_000, 256, "f32", 1, 2),    # config 2's graph at 1-KiB rows: a cache-resident table
    # config 3's graph at other row widths: where do column tiles pay? (profiles/r05_column_tiles_ab.md)
    "rmat2m512": ("rmat", 2_000_000, 40_000_000, 512, "f32", 3, 4),
    "rmat2m384": ("rmat", 2_000_000, 40_000_000, 384, "f32", 3, 4),
    "rmat2m1024": ("rmat", 2_000_000, 40_000_000, 1024, "f32", 3, 4),
    "rmat2m512bf16": ("rmat", 2_000_000, 40_000_000, 512, "bf16", 3, 4),
    "tiny": ("rmat", 20_000, 200_000, 64, "f32", 7, 8),
    "tiny12": ("rmat", 20_000, 200_000, 12, "f32", 7, 8),             # 3 packs a row: more ranks than packs leaves idle column ranks
    # 8x config 3: a 16 GiB embedding matrix (byte offsets beyond 32 bits, ~85 GB of HBM in use) -- capacity check
    "rmat16m": ("rmat", 16_000_000, 320_000_000, 256, "f32", 9, 10),
}
GENERATOR_NAMES = {"rmat": "R-MAT", "powerlaw": "power-law", "uniform": "uniform-random pairs (duplicates merged)"}
DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "f64": torch.float64}
PARITY_TOL = {"f32": 1e-4, "f64": 1e-10, "bf16": 8e-3}       # bf16: 2^-8 rounding of every stored value
PARITY_P_TOL = {"f32": 2e-6, "f64": 1e-12, "bf16": 1e-4}     # P itself (fp32 arithmetic on the stored values)
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_NOTE = ("traffic = rocprofv3 FETCH_SIZE x calibrated factor + WRITE_SIZE per launch, taken in separate --pmc "
                "passes of this same command (profiles/) and matched to the live run by kernel configuration: bytes "
                "and time come from different runs (kernels[..].pmc_run_over_live_time says how the kernel's duration "
                "in the PMC run compares with this run's); these counters sit on the L2's fabric side, so Infinity-Cache "
                "hits are counted as traffic: frac = min(algorithmic, traffic) bytes / kernel time / 8 TB/s, capped "
                "at 1, is an UPPER bound of the HBM share; achieved_algorithmic is the no-reuse gather model "
                "(SURVEY 8d), which also counts L2 hits")
# north_star_literal block: past this many seconds the main record is printed without it (a healthy block takes ~10 s at
# config 3; the driver's own limit for the whole command is 600 s, and the main record must come out well inside it)
LITERAL_DEADLINE_S = float(os.environ.get("CLANE_BENCH_LITERAL_DEADLINE_S", "150"))
DRIVER_LIMIT_S = 600.0
T_START = time.perf_counter()
TIMELINE = {}           # seconds spent per phase of this command (N > 1: reported as time_plan.spent_s)


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)

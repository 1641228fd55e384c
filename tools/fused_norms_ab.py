#!/usr/bin/env python3
"""A/B on ONE box, interleaved (so both variants see the same physical pages and clocks): the K3 kernels with the row
norms fused into their epilogue (round 4) against the same kernels without (sq_out = NULL) + the separate K0 pass in
build_P.  Sweep ms and build_P ms per variant, median of 3 rounds of 3 blocks.
Usage: tools/fused_norms_ab.py [--out gpurun_out/r04/fused_norms_ab.jsonl]"""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip, synth
from clane_amd.engine import SweepEngine

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--steps", type=int, default=15)
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
CASES = [
    ("R-MAT 2M/40M d=256 fp32 (config 3)", lambda: synth.rmat_csr(2_000_000, 40_000_000, seed=3, device=str(dev)), 256, torch.float32),
    ("uniform 2M/40M d=256 fp32", lambda: synth.uniform_random_csr(2_000_000, 40_000_000, device=str(dev)), 256, torch.float32),
    ("uniform 8M/400M d=32 fp32 (128-byte rows, the shape of the 2^31-edge test)",
     lambda: synth.uniform_random_csr(8_000_000, 400_000_000, device=str(dev)), 32, torch.float32),
    ("power-law 10M/200M d=128 bf16 (config 4's shape)", lambda: synth.powerlaw_csr(10_000_000, 200_000_000, seed=5, device=str(dev)), 128,
     torch.bfloat16),
]


def timed(eng):
    eng.build_P()
    for _ in range(3):
        eng.sweep(0.76)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.build_P()
    torch.cuda.synchronize()
    bp = (time.perf_counter() - t0) * 1e3
    blocks = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.sweep(0.76)
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t0) / args.steps * 1e3)
    return float(np.median(blocks)), bp


out = open(args.out, "a") if args.out else None
for name, make, d, dtype in CASES:
    csr = make()
    X = synth.gaussian_X(csr.num_vertices, d, seed=5).to(dtype)
    runs = {True: [], False: []}
    for rnd in range(3):
        for fused in (True, False):
            eng = SweepEngine(csr, X, dev, fused_norms=fused)
            runs[fused].append(timed(eng))
            del eng                     # back to torch's cache: the next engine reuses the pages
    rec = {"case": name, "edges": int(csr.num_edges)}
    for fused in (True, False):
        ms = [a for a, _ in runs[fused]]
        rec["fused" if fused else "separate K0"] = {"sweep_ms": round(float(np.median(ms)), 3), "sweep_ms_min": round(min(ms), 3),
                                                    "sweep_ms_max": round(max(ms), 3),
                                                    "build_P_ms": round(float(np.median([b for _, b in runs[fused]])), 3)}
    print(json.dumps(rec), flush=True)
    if out:
        out.write(json.dumps(rec) + "\n")
        out.flush()
    del csr, X
    torch.cuda.empty_cache()

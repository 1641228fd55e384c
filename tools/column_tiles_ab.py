#!/usr/bin/env python3
"""One-GPU sweeps as T column tiles (SweepEngine(column_tiles=T)) against the plain sweep, interleaved on ONE box.
Usage: tools/column_tiles_ab.py --workload rmat2m --tiles 1 2 4 [--out gpurun_out/r05/column_tiles_ab.jsonl]"""
import argparse
import json
import statistics
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from clane_amd import _hip, synth  # noqa: E402
from clane_amd.engine import SweepEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rmat2m")
ap.add_argument("--tiles", type=int, nargs="+", default=[1, 2])
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--sweeps", type=int, default=40)
ap.add_argument("--out", default=None)
ap.add_argument("--class-threshold", type=int, default=None)
ap.add_argument("--class-phases", type=int, default=None)
ap.add_argument("--long-threshold", type=int, default=None)
ap.add_argument("--class-chunk", type=int, default=256)
args = ap.parse_args()
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[args.workload]
dev = _hip.require_gpu("cuda:0")
csr = {"rmat": synth.rmat_csr, "powerlaw": synth.powerlaw_csr, "uniform": synth.uniform_random_csr}[gen](V, E, seed=gseed, device=str(dev))
X = synth.gaussian_X(V, d, seed=xseed).to(bench.DTYPES[dname])
rec = {"workload": args.workload, "device": torch.cuda.get_device_name(dev), "tiles": {}}
ref = None
for T in args.tiles:            # one engine at a time (config 4 x 3 engines would not leave room), rounds inside
    eng = SweepEngine(csr, X, dev, column_tiles=T, class_threshold=args.class_threshold, class_phases=args.class_phases,
                      long_threshold=args.long_threshold, class_chunk=args.class_chunk)
    t0 = time.perf_counter()
    eng.build_P()
    torch.cuda.synchronize()
    for _ in range(3):
        eng.sweep(0.76)
    times, bp = [], []
    for _ in range(args.rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.sweeps):
            eng.sweep(0.76)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / args.sweeps * 1e3)
        eng.snapshot()                     # (in-loop build_P: the outer-delta pass leaves the norms behind)
        eng.sweep(0.76)
        eng.distance_from_snapshot()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.build_P()
        torch.cuda.synchronize()
        bp.append((time.perf_counter() - t0) * 1e3)
    eng.time_kernels, eng.kernel_events = True, []
    for _ in range(8):
        eng.sweep(0.76)
    torch.cuda.synchronize()
    kt = eng.kernel_times_ms()
    eng.time_kernels = False
    Z = eng.get_Z().float()
    if ref is None:
        ref = Z
    rec["tiles"][T] = {"ms_per_sweep": statistics.median(times), "all": [round(x, 4) for x in times], "kernel_ms": kt,
                       "build_P_ms": min(bp), "tiles": eng.tiles, "class_rows": sum(0 if c is None else c[0].numel() for c in eng.class_rows),
                       "thresholds": [eng.long_threshold, eng.class_threshold, eng.class_phases],
                       "rel_l2_vs_first": float((Z - ref).norm() / ref.norm())}
    print(T, rec["tiles"][T], flush=True)
    del eng
    torch.cuda.empty_cache()
print(json.dumps(rec))
if args.out:
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    with open(args.out, "a") as f:
        f.write(json.dumps(rec) + "\n")

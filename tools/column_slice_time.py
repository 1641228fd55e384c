#!/usr/bin/env python3
"""Sweep time of ONE rank of a column-split W-GPU run, measured on one GPU: every rank holds the whole graph and
d/W columns of X and Z, so its sweep is exactly a 1-GPU sweep over [V, d/W].
Usage: tools/column_slice_time.py [--workload rmat2m] [--world 1 2 4 8] [--long-threshold T]"""
import argparse, json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip, synth
from clane_amd.engine import SweepEngine
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rmat2m")
ap.add_argument("--world", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--long-threshold", type=int, default=None)
ap.add_argument("--chunks", type=int, default=1)
ap.add_argument("--class-threshold", type=int, nargs="+", default=[None],
                help="rows above this many edges take the XCD-affine pass (0 = off; default: the engine's choice)")
ap.add_argument("--class-chunk", type=int, nargs="+", default=[256])
ap.add_argument("--class-phases", type=int, default=None)
ap.add_argument("--phase-threshold", type=int, default=512)
ap.add_argument("--calibrate", action="store_true",
                help="also launch l1_distance over two [V, d/N] matrices (known bytes) -- PMC calibration")
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[args.workload]
csr = synth.rmat_csr(V, E, seed=gseed) if gen == "rmat" else synth.powerlaw_csr(V, E, seed=gseed)
X = synth.gaussian_X(V, d, seed=xseed).to(bench.DTYPES[dname])
for W, ct, cc in [(W, ct, cc) for W in args.world for ct in args.class_threshold for cc in args.class_chunk]:
    if ct == 0 and cc != args.class_chunk[0]:
        continue
    dl = d // W
    eng = SweepEngine(csr, X[:, :dl].contiguous(), dev, chunks=args.chunks, long_threshold=args.long_threshold,
                      class_threshold=ct, class_chunk=cc, class_phases=args.class_phases,
                      phase_threshold=args.phase_threshold)
    eng.build_P()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.build_P()
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    eng.P.copy_(torch.rand(eng.P.numel(), device=dev) / 20)       # any frozen weights: traffic is what is timed
    eng.P_valid = True
    if args.calibrate:
        eng.l1_between(0, 1)
    for _ in range(5):
        eng.sweep(0.76)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.sweep(0.76)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    eng.time_kernels = True
    for _ in range(5):
        eng.sweep(0.76)
    torch.cuda.synchronize()
    kt = eng.kernel_times_ms()
    kb = eng.kernel_bytes()
    nbytes = sum(kb.values())
    print(json.dumps({"world": W, "d_local": dl, "ms_per_sweep": round(ms, 3), "long_threshold": eng.long_threshold,
                      "class_threshold": eng.class_threshold, "class_chunk": eng.class_chunk,
                      "class_phases": eng.class_phases, "phase_threshold": eng.phase_threshold,
                      "build_P_ms": round(build_ms, 3),
                      "algorithmic_GB": round(nbytes / 1e9, 2), "TBps": round(nbytes / ms / 1e9, 2),
                      "kernels_ms": {k: round(v, 3) for k, v in kt.items()},
                      "kernels_GB": {k: round(v / 1e9, 2) for k, v in kb.items()}}), flush=True)
    del eng
    torch.cuda.empty_cache()

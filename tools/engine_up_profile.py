#!/usr/bin/env python3
"""Where does SweepEngine's start-up go?  cProfile of the constructor on a bench workload (second construction, so
code objects and allocator are warm).  Usage: tools/engine_up_profile.py [--workload rmat2m]"""
import argparse, cProfile, pstats, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip, synth
from clane_amd.engine import SweepEngine
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rmat2m")
ap.add_argument("--lines", type=int, default=45)
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[args.workload]
csr = synth.rmat_csr(V, E, seed=gseed, device=str(dev)) if gen == "rmat" else synth.powerlaw_csr(V, E, seed=gseed, device=str(dev))
X = synth.gaussian_X(V, d, seed=xseed).to(bench.DTYPES[dname])
for rep in range(2):
    t0 = time.perf_counter()
    eng = SweepEngine(csr, X, dev)
    torch.cuda.synchronize()
    print(f"construction {rep}: {time.perf_counter() - t0:.3f} s", flush=True)
    del eng
pr = cProfile.Profile()
pr.enable()
eng = SweepEngine(csr, X, dev)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(args.lines)

#!/usr/bin/env python3
"""A graph dominated by a few rows that read every vertex (`synth.star_csr`): build_P + sweeps, for a kernel trace
(`rocprofv3 --kernel-trace --stats -- python3 tools/star_probe.py`): how long the many-slot combine of a mega-hub row is.
Usage: tools/star_probe.py [--vertices 2000000] [--stars 10] [--d 256] [--steps 20]"""
import argparse, json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip, synth
from clane_amd.engine import SweepEngine

ap = argparse.ArgumentParser()
ap.add_argument("--vertices", type=int, default=2_000_000)
ap.add_argument("--stars", type=int, default=10)
ap.add_argument("--d", type=int, default=256)
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
csr = synth.star_csr(args.vertices, args.stars, args.vertices, device=str(dev))
X = synth.gaussian_X(args.vertices, args.d, seed=5)
eng = SweepEngine(csr, X, dev)
eng.build_P()
for _ in range(3):
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
print(json.dumps({"edges": csr.num_edges, "class_rows": int(eng.class_rows[0][0].numel()), "slots": eng.class_slots[0],
                  "ms_per_sweep": round(ms, 3), "kernels_ms": eng.kernel_times_ms(), "build": eng.k.build_info()}))

#!/usr/bin/env python3
"""Class rows finished inside the chunk launch (class_fused=True: the last-arriving chunk adds the row's slots) against
the two-launch form (chunks, then one workgroup per row over the slab), interleaved on ONE box, same tables.
Usage: tools/class_fused_ab.py [--workload rmat2m] [--out gpurun_out/r05/class_fused_ab.jsonl]"""
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
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--sweeps", type=int, default=50)
ap.add_argument("--out", default=None)
args = ap.parse_args()
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[args.workload]
dev = _hip.require_gpu("cuda:0")
csr = {"rmat": synth.rmat_csr, "powerlaw": synth.powerlaw_csr, "uniform": synth.uniform_random_csr}[gen](V, E, seed=gseed, device=str(dev))
X = synth.gaussian_X(V, d, seed=xseed).to(bench.DTYPES[dname])
engines = {name: SweepEngine(csr, X, dev, class_fused=fused) for name, fused in (("fused", True), ("two_launches", False))}
for eng in engines.values():
    eng.build_P()
    for _ in range(5):
        eng.sweep(0.76)
same = torch.equal(engines["fused"].Zcur, engines["two_launches"].Zcur)
times = {name: [] for name in engines}
for _ in range(args.rounds):
    for name, eng in engines.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.sweeps):
            eng.sweep(0.76)
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / args.sweeps * 1e3)
ktimes = {}
for name, eng in engines.items():
    eng.time_kernels, eng.kernel_events = True, []
    for _ in range(16):
        eng.sweep(0.76)
    torch.cuda.synchronize()
    ktimes[name] = eng.kernel_times_ms()
    eng.time_kernels = False
rec = {"workload": args.workload, "bit_identical_after_5_sweeps": same,
       "still_bit_identical": torch.equal(engines["fused"].Zcur, engines["two_launches"].Zcur),
       "ms_per_sweep": {n: {"median": statistics.median(t), "all": [round(x, 4) for x in t]} for n, t in times.items()},
       "kernel_ms": ktimes, "build": engines["fused"].k.build_info(), "device": torch.cuda.get_device_name(dev)}
print(json.dumps(rec))
if args.out:
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    with open(args.out, "a") as f:
        f.write(json.dumps(rec) + "\n")

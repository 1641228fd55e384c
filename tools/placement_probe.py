#!/usr/bin/env python3
"""Why does the SAME configuration time in two states (7.96 / 8.58 ms on the uniform-random 2M / 40M / d=256 graph, 3.045 /
3.173 on the star-heavy one: profiles/r03_threshold_robustness.md)?  ONE run that logs, for a series of engines built on
the same graph, where the allocator put the big tables (Zbuf0 / Zbuf1 / X / P / colidx: address, address mod 2 MiB ...
1 GiB, pairwise distances), the clocks sysfs reports, and every timed block -- first with whatever placement the caching
allocator gives after differently sized junk allocations, then with the tables carved at CONTROLLED offsets from a 2-MiB
boundary (SweepEngine(table_skew=...)).  The placement that correlates with the slow state is then visible in one table.
Usage: tools/placement_probe.py [--graph uniform|star] [--out gpurun_out/placement_probe.jsonl]"""
import argparse, glob, json, sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip, synth
from clane_amd.engine import SweepEngine

ap = argparse.ArgumentParser()
ap.add_argument("--graph", default="uniform", choices=["uniform", "star", "rmat"])
ap.add_argument("--steps", type=int, default=15)
ap.add_argument("--blocks", type=int, default=5)
ap.add_argument("--out", default=None)
ap.add_argument("--contiguous", action="store_true",
                help="also: the three big tables in physically contiguous allocations (SweepEngine(table_alloc='contiguous'))")
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
V = 2_000_000
csr = {"uniform": lambda: synth.uniform_random_csr(V, 40_000_000, device=str(dev)),
       "star": lambda: synth.star_csr(V, 10, V, device=str(dev)),
       "rmat": lambda: synth.rmat_csr(V, 40_000_000, seed=3, device=str(dev))}[args.graph]()
X = synth.gaussian_X(V, 256, seed=5)
out = open(args.out, "a") if args.out else None


def clocks():
    got = {}
    for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
        for f in glob.glob(f"/sys/class/drm/card*/device/{name}"):
            try:
                cur = [l.strip() for l in open(f).read().splitlines() if l.strip().endswith("*")]
                got.setdefault(name, []).append(cur[0] if cur else "?")
            except OSError:
                pass
    return got


def tables(eng):
    t = {"Z0": eng.Zbuf[0], "Z1": eng.Zbuf[1], "X": eng.X_loc, "P": eng.P, "colidx": eng.colidx, "rowptr": eng.rowptr}
    return {k_: v.data_ptr() for k_, v in t.items()}


def timed(eng):
    eng.build_P()
    for _ in range(3):
        eng.sweep(0.76)
    blocks = []
    for _ in range(args.blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.sweep(0.76)
        torch.cuda.synchronize()
        blocks.append(round((time.perf_counter() - t0) / args.steps * 1e3, 3))
    return blocks


def run(label, junk_mib=0, **kw):
    torch.cuda.empty_cache()
    junk = torch.empty(junk_mib << 20, dtype=torch.uint8, device=dev) if junk_mib else None
    eng = SweepEngine(csr, X, dev, **kw)
    ptr = tables(eng)
    blocks = timed(eng)
    rec = {"graph": args.graph, "label": label, "table_alloc": eng.table_alloc, "table_alloc_note": eng.table_alloc_note, "junk_mib": junk_mib, "skew": kw.get("table_skew"),
           "ms_blocks": blocks, "ms_median": float(np.median(blocks)),
           "ptr": {k_: hex(v) for k_, v in ptr.items()},
           "mod_2MiB": {k_: v % (2 << 20) for k_, v in ptr.items()},
           "mod_1GiB_MiB": {k_: (v % (1 << 30)) >> 20 for k_, v in ptr.items()},
           "Z1_minus_Z0_MiB": (ptr["Z1"] - ptr["Z0"]) / 2 ** 20, "X_minus_Z0_MiB": (ptr["X"] - ptr["Z0"]) / 2 ** 20,
           "clocks": clocks(), "reserved_GiB": round(torch.cuda.memory_reserved(dev) / 2 ** 30, 2)}
    print(json.dumps(rec), flush=True)
    if out:
        out.write(json.dumps(rec) + "\n")
        out.flush()
    del eng, junk
    return rec


# 1. whatever the caching allocator gives, perturbed by junk allocated first (kept alive while the engine is built)
for i, junk in enumerate((0, 0, 3, 0, 70, 0, 513, 1031, 0)):
    run(f"allocator #{i}", junk_mib=junk)
# 2. the three streamed tables at controlled offsets from a 2-MiB boundary
K, M = 1 << 10, 1 << 20
for name, skew in (("all aligned", {"Z0": 0, "Z1": 0, "X": 0}),
                   ("Z1 +4K, X +8K", {"Z0": 0, "Z1": 4 * K, "X": 8 * K}),
                   ("Z1 +64K, X +128K", {"Z0": 0, "Z1": 64 * K, "X": 128 * K}),
                   ("Z1 +256K, X +512K", {"Z0": 0, "Z1": 256 * K, "X": 512 * K}),
                   ("Z1 +683K, X +1365K", {"Z0": 0, "Z1": 683 * K, "X": 1365 * K}),
                   ("Z1 +1M, X +1M", {"Z0": 0, "Z1": M, "X": M}),
                   ("all aligned (again)", {"Z0": 0, "Z1": 0, "X": 0})):
    run(name, table_skew=skew)
run("allocator (last)")
if args.contiguous:
    for i in range(6):
        run(f"contiguous #{i}", table_alloc="contiguous")
        run(f"torch again #{i}")

#!/usr/bin/env python3
"""Experiment (r02): does the ORDER in which the short rows (<= T edges) are processed matter to the one-wave pass?
The engine walks them in table order (hottest-first by in-degree).  Candidates: grouped by their hottest neighbour
(rows that share it run close in time, so it stays in L2), or shuffled.  Timed with the production spmm_update_kernel
on a virtual CSR holding the short rows in the given order (outputs to dummy rows).
Usage: tools/short_row_order_experiment.py [--workload rmat2m]"""
import argparse, json, sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip, synth
from clane_amd.engine import SweepEngine
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rmat2m")
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
k = _hip.kernels()
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[args.workload]
csr = synth.rmat_csr(V, E, seed=gseed) if gen == "rmat" else synth.powerlaw_csr(V, E, seed=gseed)
X = synth.gaussian_X(V, d, seed=xseed).to(bench.DTYPES[dname])
eng = SweepEngine(csr, X, dev)
rowptr, colidx = eng.local.rowptr, eng.local.colidx.astype(np.int64)
deg = np.diff(rowptr)
T = eng.long_threshold
rows = np.nonzero((deg > 0) & (deg <= T))[0]
first_col = colidx[rowptr[rows]]                 # columns are sorted within these rows: the hottest neighbour
orders = {"table order (as shipped)": rows,
          "grouped by hottest neighbour": rows[np.lexsort((rows, first_col))],
          "shuffled": rows[np.random.default_rng(0).permutation(rows.size)]}
P_all = torch.rand(colidx.size, device=dev) / 20
for name, order in orders.items():
    dg = deg[order]
    rp = np.zeros(order.size + 1, dtype=np.int64)
    np.cumsum(dg, out=rp[1:])
    take = np.repeat(rowptr[order] - rp[:-1], dg) + np.arange(int(dg.sum()))
    rp_d = torch.from_numpy(rp).to(dev)
    ci_d = torch.from_numpy(colidx[take].astype(np.int32)).to(dev)
    P_d = P_all[torch.from_numpy(take).to(dev)]
    n = order.size
    Xv = torch.zeros(n, eng.ld, dtype=eng.dtype, device=dev)
    Zn = torch.zeros(n, eng.ld, dtype=eng.dtype, device=dev)
    part = torch.zeros(k.spmm_partials_len(n, 0), dtype=torch.float64, device=dev)
    launch = lambda: k.spmm_update(rp_d, ci_d, P_d, n, 0, eng.Zbuf[0], Xv, 0.76, Zn, eng.d, 0, part)  # noqa: E731
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.steps):
        launch()
    ev[1].record()
    torch.cuda.synchronize()
    print(json.dumps({"order": name, "rows": int(n), "edges": int(dg.sum()), "ms": round(ev[0].elapsed_time(ev[1]) / args.steps, 3)}),
          flush=True)

#!/usr/bin/env python3
"""Experiment: do the long rows of K3 run faster when every XCD only gathers ITS share of the hot rows?

Z is laid out hot rows first, so the H*8 most-gathered rows are positions [0, 8H).  Each long row's edge list
is cut into 8 hot buckets (position % 8 of the rows below 8H: equal heat per bucket) plus an eighth of its cold tail, giving 8 "virtual
rows" per long row, listed so that virtual row (r, b) is workgroup 8*i + b: blocks are dealt round-robin over
the XCDs, so bucket b's 1-KiB rows (H KiB in all) are only ever gathered through one XCD's 4 MiB L2.
Timed with the production spmm_long kernel on the virtual CSR (the partial sums land in dummy rows: this measures
the gather, not the combine).  Usage: tools/xcd_bucket_experiment.py [--hot 3072 3584 4096] [--min-degree 256]"""
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
ap.add_argument("--hot", type=int, nargs="+", default=[3584])
ap.add_argument("--min-degree", type=int, nargs="+", default=[32, 256])
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
k = _hip.kernels()
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[args.workload]
csr = synth.rmat_csr(V, E, seed=gseed)
X = synth.gaussian_X(V, d, seed=xseed)
eng = SweepEngine(csr, X, dev)                 # hot-rows-first layout, position-relabelled CSR
rowptr, colidx = eng.local.rowptr, eng.local.colidx.astype(np.int64)
deg = np.diff(rowptr)
P = torch.rand(colidx.size, device=dev) / 20
Zold = eng.Zbuf[0]


def timed(rp, ci, Pv, rows, label, extra):
    n = rows.size
    rp_d, ci_d = torch.from_numpy(rp).to(dev), torch.from_numpy(ci.astype(np.int32)).to(dev)
    rows_d = torch.from_numpy(rows.astype(np.int32)).to(dev)
    nv = rp.size - 1
    Xv = torch.zeros(max(nv, 1), d, device=dev)
    Zn = torch.zeros(max(nv, 1), d, device=dev)
    partials = torch.zeros(n, dtype=torch.float64, device=dev)
    for _ in range(3):
        k.spmm_update_long(rp_d, ci_d, Pv, rows_d, 16, 0, Zold, Xv, 0.76, Zn, d, partials)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.steps):
        k.spmm_update_long(rp_d, ci_d, Pv, rows_d, 16, 0, Zold, Xv, 0.76, Zn, d, partials)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / args.steps
    edges = int((rp[rows + 1] - rp[rows]).sum())
    print(json.dumps({"variant": label, "workgroups": int(n), "edges": edges, "ms": round(ms, 3),
                      "gather_TBps": round(edges * d * 4 / ms / 1e9, 2), **extra}), flush=True)
    del Xv, Zn
    torch.cuda.empty_cache()


for dmin in args.min_degree:
    long_rows = np.nonzero((deg > dmin) & (deg <= 4096))[0]
    timed(rowptr, colidx, P, long_rows, "baseline: one workgroup per long row", {"min_degree": dmin})
    for H in args.hot:
        # per long row: offsets of the 8 hot buckets and of the cold tail
        a, b = rowptr[long_rows], rowptr[long_rows + 1]
        n = long_rows.size
        sizes = b - a
        # vectorised: edges of the long rows, with their row index
        idx = np.repeat(a - np.concatenate([[0], np.cumsum(sizes)[:-1]]), sizes) + np.arange(sizes.sum())
        cols = colidx[idx]
        rid = np.repeat(np.arange(n, dtype=np.int64), sizes)
        bucket = np.where(cols < 8 * H, cols % 8, 8)             # hot rows dealt round-robin to 8 buckets; 8 = cold
        # cold edges: dealt to the 8 virtual rows in equal contiguous shares
        cold = bucket == 8
        cold_rank = np.zeros(idx.size, dtype=np.int64)
        cold_cnt = np.bincount(rid[cold], minlength=n)
        first_cold = np.concatenate([[0], np.cumsum(sizes)[:-1]]) + (sizes - cold_cnt)
        cold_rank[cold] = np.arange(idx.size)[cold] - first_cold[rid[cold]]
        share = np.maximum(1, -(-cold_cnt // 8))
        vb = np.where(cold, np.minimum(cold_rank // share[rid], 7), bucket)
        order = np.lexsort((np.arange(idx.size), vb, rid))       # by row, virtual bucket, original order
        v_cols, v_src = cols[order], idx[order]
        counts = np.bincount(rid * 8 + vb, minlength=n * 8)
        v_rowptr = np.zeros(n * 8 + 1, dtype=np.int64)
        np.cumsum(counts, out=v_rowptr[1:])
        Pv = P[torch.from_numpy(v_src).to(dev)]
        hot_edges = int((~cold).sum())
        extra = {"min_degree": dmin, "H": H, "hot_MB_per_xcd": round(H * d * 4 / 2**20, 2),
                 "hot_edge_share": round(hot_edges / idx.size, 3)}
        rows_v = np.arange(n * 8)
        timed(v_rowptr, v_cols, Pv, rows_v, "bucketed, bucket b -> workgroup 8i+b (one XCD)", extra)
        perm = np.random.default_rng(0).permutation(n * 8)
        timed(v_rowptr, v_cols, Pv, perm, "bucketed, virtual rows in random order (no affinity)", extra)

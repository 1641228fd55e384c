#!/usr/bin/env python3
"""Experiment (r02): how fast do the long rows of K3 gather when EVERY gathered row is read through one XCD only?

tools/gather_rows_ceiling.hip says: with the read skew of config 3 and the hottest-first layout, pure 1-KiB-row
gathers run at 8.9 TB/s when any workgroup may read any row, and at 14.3 TB/s when workgroup i (XCD i % 8) only
reads rows r with r % 8 == i % 8 -- each XCD's 4 MiB L2 then caches its own eighth of the hot rows.

Here the same idea on the real CSR with a production kernel: every row above `--min-degree` edges has its edge
list sorted by (column % 8, column) and cut into chunks of at most `--chunk` edges of ONE class; a chunk is a
"virtual row" done by one wave of spmm_update_kernel (one wave per row, partial sums to dummy rows: this times the
gather, not the combine).  Virtual rows are laid out in blocks of `rows_per_block` of one class, block j of class b
at workgroup 8 j + b.  Compared with the same virtual rows in shuffled block order (no affinity) and with the
production row-split kernels on the original rows.   Usage: tools/xcd_class_experiment.py [--min-degree 32 128]
"""
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
ap.add_argument("--min-degree", type=int, nargs="+", default=[32, 128])
ap.add_argument("--chunk", type=int, nargs="+", default=[64, 128, 256])
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--block-class", action="store_true", help="class = (row / 8) % 8 instead of row % 8")
ap.add_argument("--phases", type=int, default=1,
                help="P > 1: every class is split into P sub-classes (an xor-fold of higher position bits) and the chunk "
                     "blocks are laid out phase-major: all chunks of sub-class 0 before those of sub-class 1, ... "
                     "(XCD-affine in space AND phased in time, profiles/r02_gather_rows_ceiling.md)")
ap.add_argument("--phase-bits", default="granules", choices=["granules", "bands"])
ap.add_argument("--cols", type=int, default=None, help="use only the first COLS columns (a column slice)")
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
k = _hip.kernels()
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[args.workload]
if args.cols:
    d = args.cols
csr = synth.rmat_csr(V, E, seed=gseed) if gen == "rmat" else synth.powerlaw_csr(V, E, seed=gseed)
X = synth.gaussian_X(V, d, seed=xseed).to(bench.DTYPES[dname])
eng = SweepEngine(csr, X, dev, class_threshold=0)   # hot-rows-first layout, position-relabelled CSR, columns sorted
rowptr, colidx = eng.local.rowptr, eng.local.colidx.astype(np.int64)
deg = np.diff(rowptr)
P = torch.rand(colidx.size, device=dev) / 20
Zold = eng.Zbuf[0]
es = Zold.element_size()


def rows_per_block(nrows):                      # mirrors csrc/clane_abi.hip
    r = -(-(-(-nrows // 32768)) // 4) * 4
    return min(max(r, 32), 256)


def run(label, launch, edges, extra):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.steps):
        launch()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / args.steps
    print(json.dumps({"variant": label, "edges": edges, "ms": round(ms, 3),
                      "gather_TBps": round(edges * d * es / ms / 1e9, 2), **extra}), flush=True)


for dmin in args.min_degree:
    long_rows = np.nonzero(deg > dmin)[0]
    n = long_rows.size
    a, sizes = rowptr[long_rows], deg[long_rows]
    n_edges = int(sizes.sum())
    # production kernels on these rows: 16-wave row kernel up to 4096 edges, segment split above
    small = long_rows[sizes <= 4096]
    rows_d = torch.from_numpy(small.astype(np.int32)).to(dev)
    rp_d, ci_d = eng.rowptr, eng.colidx
    Zn = torch.zeros_like(Zold)
    part = torch.zeros(n + 8, dtype=torch.float64, device=dev)
    run("production: one 16-wave workgroup per row (rows <= 4096 edges only)",
        lambda: k.spmm_update_long(rp_d, ci_d, P, rows_d, 16, 0, Zold, eng.X_loc, 0.76, Zn, eng.d, part),
        int(deg[small].sum()), {"min_degree": dmin, "rows": int(small.size)})
    del Zn
    # edges of the long rows with their row index, sorted by (row, class, column)
    start = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    idx = np.repeat(a - start, sizes) + np.arange(n_edges)
    cols = colidx[idx]
    rid = np.repeat(np.arange(n, dtype=np.int64), sizes)
    from clane_amd.xcd import xcd_class
    cls = xcd_class(cols) if not args.block_class else (cols >> 3) & 7
    NP = args.phases
    if NP > 1 and args.phase_bits == "bands":       # 4096-row bands of the table, dealt to the phases
        phase = ((cols >> 12) ^ (cols >> 14) ^ (cols >> 16)) % NP
    elif NP > 1:                                     # 64-row granules: every phase gets the same mix of hot and cold rows
        phase = (cols >> 6) % NP
    else:
        phase = np.zeros_like(cols)
    sub = phase * 8 + cls                                              # 8 * NP sub-classes; XCD = sub % 8
    order = np.lexsort((cols, sub, rid))
    cols, idx, rid, cls, sub = cols[order], idx[order], rid[order], cls[order], sub[order]
    Pv = P[torch.from_numpy(idx).to(dev)]
    NS = 8 * NP
    seg_key = rid * NS + sub
    seg_len = np.bincount(seg_key, minlength=n * NS)
    seg_start = np.concatenate([[0], np.cumsum(seg_len)[:-1]])
    for C in args.chunk:
        nchunk = -(-seg_len // C)                                   # chunks per (row, class) segment
        tot = int(nchunk.sum())
        seg_of = np.repeat(np.arange(n * NS), nchunk)
        within = np.arange(tot) - np.repeat(np.concatenate([[0], np.cumsum(nchunk)[:-1]]), nchunk)
        c_e0 = seg_start[seg_of] + within * C
        c_e1 = np.minimum(c_e0 + C, seg_start[seg_of] + seg_len[seg_of])
        c_sub = seg_of % NS
        # blocks of rpb chunks of one sub-class; phase-major: for every phase, block j of class b -> workgroup 8 j + b
        per_sub = [np.nonzero(c_sub == q)[0] for q in range(NS)]
        rpb = rows_per_block(tot)
        for _ in range(4):                                          # fixed point: padding changes the row count
            Jp = [max(-(-len(per_sub[ph * 8 + b]) // rpb) for b in range(8)) for ph in range(NP)]
            rpb_new = rows_per_block(8 * sum(Jp) * rpb)
            if rpb_new == rpb:
                break
            rpb = rpb_new
        nv = 8 * sum(Jp) * rpb
        v_e0 = np.zeros(nv, dtype=np.int64)
        v_e1 = np.zeros(nv, dtype=np.int64)
        base = 0
        for ph in range(NP):
            for b in range(8):
                pc = per_sub[ph * 8 + b]
                slots = base + (np.arange(len(pc)) // rpb) * 8 * rpb + b * rpb + np.arange(len(pc)) % rpb
                v_e0[slots], v_e1[slots] = c_e0[pc], c_e1[pc]
            base += 8 * Jp[ph] * rpb
        assert rows_per_block(nv) == rpb, (nv, rpb)

        def virtual_csr(e0, e1):
            ln = e1 - e0
            rp = np.zeros(e0.size + 1, dtype=np.int64)
            np.cumsum(ln, out=rp[1:])
            take = np.repeat(e0 - rp[:-1], ln) + np.arange(int(ln.sum()))
            return (torch.from_numpy(rp).to(dev), torch.from_numpy(cols[take].astype(np.int32)).to(dev),
                    Pv[torch.from_numpy(take).to(dev)])
        Xv = torch.zeros(nv, eng.ld, dtype=Zold.dtype, device=dev)
        Znv = torch.zeros(nv, eng.ld, dtype=Zold.dtype, device=dev)
        pv = torch.zeros(k.spmm_partials_len(nv, 0), dtype=torch.float64, device=dev)
        extra = {"min_degree": dmin, "chunk": C, "phases": NP, "virtual_rows": tot, "padded": nv, "rows_per_block": rpb,
                 "slab_GB_write_plus_read": round(2 * tot * d * 4 / 1e9, 2)}
        rp_v, ci_v, P_v = virtual_csr(v_e0, v_e1)
        # Z_old rows are read at [row0 + r]: virtual rows have no Z_old row of their own -> row0 = 0 reads rows 0..nv
        # of the real table (nv < V here), harmless for a timing run
        assert nv <= Zold.shape[0]
        run("class-affine: chunk blocks of class b on workgroups 8j+b",
            lambda: k.spmm_update(rp_v, ci_v, P_v, nv, 0, Zold, Xv, 0.76, Znv, eng.d, 0, pv, sinks_untouched=True),
            n_edges, extra)
        perm_blocks = np.random.default_rng(0).permutation(nv // rpb)
        perm = (perm_blocks[:, None] * rpb + np.arange(rpb)[None, :]).reshape(-1)
        rp_s, ci_s, P_s = virtual_csr(v_e0[perm], v_e1[perm])
        run("same chunks, blocks in shuffled order (no affinity)",
            lambda: k.spmm_update(rp_s, ci_s, P_s, nv, 0, Zold, Xv, 0.76, Znv, eng.d, 0, pv, sinks_untouched=True),
            n_edges, extra)
        del Xv, Znv, rp_v, ci_v, P_v, rp_s, ci_s, P_s
        torch.cuda.empty_cache()

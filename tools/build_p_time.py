#!/usr/bin/env python3
"""build_P kernel by kernel (K0, K1 one-(sub-)wave pass, K1 long rows, K1 class pass + its softmax) with HIP events,
on one GPU, for the whole matrix or for one rank's column slice of a W-GPU run.
Usage: tools/build_p_time.py [--workload powerlaw10m] [--world 1 8] [--reps 5] [--no-softmax]
Library variants: CLANE_HIP_LIB=build/variants/libclane_hip_<name>.so tools/build_p_time.py ..."""
import argparse, json, sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip, synth
from clane_amd.engine import SweepEngine
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="powerlaw10m")
ap.add_argument("--world", type=int, nargs="+", default=[1])
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--no-softmax", action="store_true", help="K1 without its fused softmax (what the scores alone cost)")
ap.add_argument("--class-threshold", type=int, default=None)
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[args.workload]
csr = synth.rmat_csr(V, E, seed=gseed) if gen == "rmat" else synth.powerlaw_csr(V, E, seed=gseed)
X = synth.gaussian_X(V, d, seed=xseed).to(bench.DTYPES[dname])
deg = np.diff(csr.rowptr)
for W in args.world:
    dl = d // W
    eng = SweepEngine(csr, X[:, :dl].contiguous(), dev, class_threshold=args.class_threshold)
    k, Z, b = eng.k, eng.Zcur, eng.blocks[0]
    mode = _hip.SCORE_REFERENCE
    fuse = not args.no_softmax
    s = Z.element_size()
    ldeg = np.diff(eng.local.rowptr)
    in_main = ldeg <= eng.k1_threshold if eng.k1_threshold > 0 else np.ones_like(ldeg, dtype=bool)
    rows_main = int((in_main & (ldeg > 0)).sum())
    e_main = int(ldeg[in_main].sum())
    main_bytes = rows_main * (dl * s + 8) + e_main * (dl * s + 8)     # source row + rowptr; per edge: row, colidx, score
    steps = {
        "K0 row_sqnorm + degree sums": lambda: (k.row_sqnorm(Z[b.row0:b.row0 + b.nrows], eng.d, eng.sq_pp[eng.cur]),
                                                k.degree_weighted_sums(eng.sq_pp[eng.cur], eng.rowptr, eng.indeg, eng.part.n_local,
                                                                       eng.ws, eng.sums2)),
        "K1 one (sub-)wave per row": lambda: k.edge_score(eng.rowptr, eng.colidx, b.nrows, b.row0, Z, eng.d, mode,
                                                          eng.sums2, None, eng.P, eng.k1_threshold, None, fuse_softmax=fuse),
    }
    if eng.k1_long_rows[0] is not None:
        steps["K1 + long rows"] = lambda: k.edge_score(eng.rowptr, eng.colidx, b.nrows, b.row0, Z, eng.d, mode, eng.sums2,
                                                       None, eng.P, eng.k1_threshold, eng.k1_long_rows[0], fuse_softmax=fuse)
    if eng.class_k1 and eng.class_rows[0] is not None:
        rows_c, slot_ptr, it_e0, it_len, it_slot, it_row, ipb = eng.class_rows[0]
        steps["K1 class pass (+ softmax)"] = lambda: k.edge_score_class(
            eng.rowptr, eng.colidx, it_e0, it_len, it_slot, it_row, ipb, rows_c, slot_ptr, b.row0, Z, eng.d, mode,
            eng.sums2, None, eng.P, eng.slabs[0], fuse_softmax=fuse, n_slots=eng.class_slots[0])
    steps["build_P (engine)"] = eng.build_P
    out = {"workload": args.workload, "world": W, "d_local": dl, "k1_threshold": eng.k1_threshold,
           "class_threshold": eng.class_threshold, "fused_softmax": fuse, "build": k.build_info(),
           "K1_main_rows": rows_main, "K1_main_edges": e_main, "K1_main_algorithmic_GB": round(main_bytes / 1e9, 2)}
    for name, fn in steps.items():
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            z.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(z))
        out[name + " ms"] = round(min(ts), 3)
    out["K1_main_TBps_algorithmic"] = round(main_bytes / out["K1 one (sub-)wave per row ms"] / 1e9, 2)
    print(json.dumps(out), flush=True)
    del eng
    torch.cuda.empty_cache()

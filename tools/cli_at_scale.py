#!/usr/bin/env python3
"""End-to-end run of the reference's workflow at scale: write a synthetic data_root (V, E as text, C.npy), then do
what `python -m clane_amd --data_root ... --gpu` does, timing each phase -- loader, engine start-up,
Embedder.iterate(), read-back, np.save.   Usage: tools/cli_at_scale.py [--workload rmat200k|rmat2m] [--cache]"""
import argparse, json, sys, tempfile, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip, synth
from clane_amd.embedder import Embedder
from clane_amd.graph import Graph
from clane_amd.similarity import CosineSimilarity
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rmat200k")
ap.add_argument("--cache", action="store_true", help="Graph(cache=True): second load reuses the parsed edge list")
ap.add_argument("--tolerence", type=int, default=10)
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[args.workload]
out = {"workload": args.workload, "V": V, "E": E, "d": d}
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    root = Path(tmp) / "data_root"
    root.mkdir()
    t0 = time.perf_counter()
    csr = synth.rmat_csr(V, E, seed=gseed, device=str(dev)) if gen == "rmat" else synth.powerlaw_csr(V, E, seed=gseed, device=str(dev))
    X = synth.gaussian_X(V, d, seed=xseed)
    import pandas as pd
    ids = np.char.add("v", np.arange(V).astype(str))                 # string ids, as in the reference's files
    rng = np.random.default_rng(0)
    perm = rng.permutation(csr.num_edges)                            # edge lines in arbitrary order
    src = np.repeat(np.arange(V, dtype=np.int64), np.diff(csr.rowptr))[perm]
    dst = csr.colidx.astype(np.int64)[perm]
    (root / "V").write_text("\n".join(ids.tolist()))
    pd.DataFrame({"s": ids[src], "d": ids[dst]}).to_csv(root / "E", sep="\t", header=False, index=False)
    np.save(root / "C.npy", X.numpy())
    out["write_dataset_s"] = round(time.perf_counter() - t0, 2)
    out["E_file_MB"] = round((root / "E").stat().st_size / 1e6)

    t0 = time.perf_counter()
    g = Graph(root, embedding_dim=d, cache=args.cache)
    out["load_graph_s"] = round(time.perf_counter() - t0, 2)
    if args.cache:
        t0 = time.perf_counter()
        g = Graph(root, embedding_dim=d, cache=True)
        out["load_graph_cached_s"] = round(time.perf_counter() - t0, 2)
    assert len(g) == V and g.csr.num_edges == csr.num_edges
    t0 = time.perf_counter()
    eng = g.engine(dev)
    torch.cuda.synchronize()
    out["engine_up_s"] = round(time.perf_counter() - t0, 2)
    emb = Embedder(g, CosineSimilarity(), dev, gamma=0.76, tolerence=args.tolerence, verbose=False, max_sweeps=2000)
    t0 = time.perf_counter()
    emb.iterate()
    torch.cuda.synchronize()
    out["iterate_s"] = round(time.perf_counter() - t0, 2)
    out["outer_rounds"], out["sweeps"], out["sweeps_launched"] = len(emb.sweep_counts), sum(emb.sweep_counts), emb.sweeps_launched
    t0 = time.perf_counter()
    Z = g.Z
    out["read_back_s"] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    np.save(Path(tmp) / "Z.npy", Z.numpy())
    out["save_s"] = round(time.perf_counter() - t0, 2)
print(json.dumps(out))

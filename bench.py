#!/usr/bin/env python3
"""Headline benchmark: embedding-update sweeps/sec + achieved HBM GB/s of the K3 SpMM kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload rmat2m|rmat200k|tiny]

One "step" = one Jacobi sweep  Z <- X + gamma * P Z  over the whole graph, P frozen: the K3
kernels, the deterministic L1-delta reduction, the host read-back of that scalar (the
reference decides after every sweep, embedder.py:94-105) and, for N > 1, the all-reduce of that
scalar.  Inputs are resident in HBM before the timed region.
N > 1 is launched by torchrun (one rank per GPU, RCCL).  The graph is fixed and divided, so scaling
is STRONG: by default every GPU sweeps d/N columns of all rows (no exchange per sweep);
--exchange halo|allgather divides the rows instead.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

WORKLOADS = {
    # name: (generator, V, E, d, dtype, graph seed, X seed)            -- SURVEY.md section 8d
    "rmat2m": ("rmat", 2_000_000, 40_000_000, 256, "f32", 3, 4),      # BASELINE config 3 (headline metric)
    "rmat200k": ("rmat", 200_000, 4_000_000, 128, "f32", 1, 2),       # BASELINE config 2
    "powerlaw10m": ("powerlaw", 10_000_000, 200_000_000, 128, "bf16", 5, 6),   # BASELINE config 4 (shape)
    "tiny": ("rmat", 20_000, 200_000, 64, "f32", 7, 8),
    # 8x config 3: a 16 GiB embedding matrix (byte offsets beyond 32 bits, ~85 GB of HBM in use) -- capacity check
    "rmat16m": ("rmat", 16_000_000, 320_000_000, 256, "f32", 9, 10),
}
DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "f64": torch.float64}
PARITY_TOL = {"f32": 1e-4, "f64": 1e-10, "bf16": 8e-3}       # bf16: 2^-8 rounding of every stored value
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(csr, X, P_host, gamma, Z1_gpu, budget_s=20.0):
    """The oracle's sweep timed on this box's host cores: the plain-C restatement (oracle/clane_oracle.c,
    OpenMP over rows, same CSR / fp32) -- kind "port".  Also the parity check of the first GPU sweep."""
    from oracle import clane_oracle as O
    from oracle import clane_oracle_c as OC
    if X.dtype != torch.float32:        # the oracle computes in fp32 on the (bf16-)rounded inputs
        X = X.float()
    P_host = P_host.float()
    threads = OC.threads()
    out = torch.empty_like(X)
    t0 = time.perf_counter()
    Z, _ = OC.sweep(csr.rowptr, csr.colidx, P_host, X, X, gamma, out=out)      # warm-up, also the parity sweep
    first = time.perf_counter() - t0
    parity = O.rel_l2(Z1_gpu.float(), Z)
    n = int(max(1, min(20, budget_s // max(first, 1e-3))))
    Za, Zb = Z.clone(), out
    t0 = time.perf_counter()
    for _ in range(n):
        Zb, _ = OC.sweep(csr.rowptr, csr.colidx, P_host, X, Za, gamma, out=Zb)
        Za, Zb = Zb, Za
    per = (time.perf_counter() - t0) / n
    return {"value": 1.0 / per, "unit": "sweeps/s", "cores": threads, "kind": "port",
            "sample": f"{n} full sweeps of the same graph by oracle/clane_oracle.c (plain C, OpenMP over rows, "
                      f"{threads} threads), P taken from the GPU build_P"}, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="rmat2m", choices=sorted(WORKLOADS))
    ap.add_argument("--gamma", type=float, default=0.76)
    ap.add_argument("--chunks", type=int, default=None)
    ap.add_argument("--long-threshold", type=int, default=None)
    ap.add_argument("--hub-threshold", type=int, default=None)
    ap.add_argument("--no-split-hubs", action="store_true", help="hub rows by one workgroup each (no segment split)")
    ap.add_argument("--natural-order", action="store_true", help="keep vertex order (default: hot rows first)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "columns", "halo", "halo_p2p", "allgather", "allgather_all"],
                    help="N > 1: auto = columns while a rank's row slice is >= 64 bytes, else halo; columns = every GPU holds d/N columns of every row, no exchange per sweep; the others "
                         "divide the rows and say how updated rows travel (clane_amd/halo.py, partition.py)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 flow)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (RCCL refuses that: use --backend gloo)")
    ap.add_argument("--pipelined", action="store_true",
                    help="opt-in (SURVEY H5): launch sweep t+1 before the host has read sweep t's delta "
                         "(SweepEngine.sweep_launch / sweep_wait); every delta is still read, one sweep later")
    ap.add_argument("--iterate", action="store_true",
                    help="after the timed sweeps also run the WHOLE algorithm from Z = X -- Embedder.iterate() to "
                         "tolerance (build_P + propagate per outer round) -- and report rounds, sweeps, wall time")
    ap.add_argument("--tolerence", type=int, default=10, help="(reference spelling) for --iterate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch l1_distance over two [V,d] matrices (known bytes) -- PMC calibration")
    args = ap.parse_args()

    import torch.distributed as dist
    from clane_amd import _hip, synth
    from clane_amd.engine import SweepEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    # one visible device per rank (a launcher that masks HIP_VISIBLE_DEVICES per process): it is cuda:0 there
    masked = torch.cuda.device_count() == 1 and world > 1
    dev = _hip.require_gpu("cuda:0" if (args.share_gpu or masked) else f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    pg = None
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        pg = dist.group.WORLD

    gen, V, E, d, dname, gseed, xseed = WORKLOADS[args.workload]
    if os.environ.get("CLANE_BENCH_PERTURB_RANK") == str(rank) and world > 1:
        gseed += 1000           # test hook: this rank draws a different graph, the agreement check must repair it
    t0 = time.perf_counter()
    if gen == "rmat":
        csr = synth.rmat_csr(V, E, seed=gseed, device=str(dev))
    else:
        csr = synth.powerlaw_csr(V, E, seed=gseed, device=str(dev))
    E = csr.num_edges
    X = synth.gaussian_X(V, d, seed=xseed).to(DTYPES[dname])
    if world > 1:       # every rank generated the graph on its own GPU from the same seed: make sure they agree
        mine = (E, int(csr.colidx.astype(np.int64).sum()), int(csr.rowptr[::997].sum()), float(X[::9973].double().sum()))
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        if any(e != everyone[0] for e in everyone):
            # should not happen (counter-based RNG, same seed, same GPU model); if it does, rank 0's input wins
            log(f"ranks disagree on the synthetic input ({everyone}): broadcasting rank 0's graph and X")
            from clane_amd.partition import HostCSR
            n_edges = torch.tensor([csr.num_edges], dtype=torch.int64, device=dev)
            dist.broadcast(n_edges, 0)
            rp = torch.from_numpy(csr.rowptr).to(dev)
            ci = torch.from_numpy(csr.colidx).to(dev) if rank == 0 else torch.empty(int(n_edges), dtype=torch.int32,
                                                                                    device=dev)
            Xd = X.to(dev)
            for t in (rp, ci, Xd):
                dist.broadcast(t, 0)
            csr, X = HostCSR(V, rp.cpu().numpy(), ci.cpu().numpy()), Xd.cpu()
            E = csr.num_edges
    log(f"{args.workload}: |V|={V} |E|={csr.num_edges} d={d} max outdeg={int(np.diff(csr.rowptr).max())} "
        f"generated in {time.perf_counter() - t0:.1f}s")

    t0 = time.perf_counter()
    eng = SweepEngine(csr, X, dev, process_group=pg, chunks=args.chunks, long_threshold=args.long_threshold,
                      hub_threshold=args.hub_threshold, exchange=args.exchange,
                      hot_rows_first=not args.natural_order, split_hubs=not args.no_split_hubs)
    torch.cuda.synchronize()
    log(f"engine up in {time.perf_counter() - t0:.1f}s; rank rows={eng.part.n_local} edges={eng.E_loc} "
        f"rows/kernel: mid(4 waves)={sum(0 if l is None else l.numel() for l in eng.mid_rows)} "
        f"hub(16 waves)={sum(0 if l is None else l.numel() for l in eng.hub_rows)} "
        f"split={sum(0 if l is None else l[0].numel() for l in eng.split_rows)} "
        f"thresholds {eng.long_threshold}/{eng.hub_threshold}")

    # build_P once (timed separately, not part of a step), P frozen afterwards
    eng.build_P()                      # first call loads the code objects; time the second
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.build_P()
    torch.cuda.synchronize()
    build_p_ms = (time.perf_counter() - t0) * 1e3

    if args.calibrate:      # known-size streaming read in this library's own 16 B/lane access pattern
        eng.snapshot()
        eng.distance_from_snapshot()
    Z1 = None
    if world == 1 and not args.no_cpu_baseline:     # the sweep the oracle is checked against (Z = X before it)
        eng.sweep(args.gamma)
        Z1 = eng.get_Z()
    for _ in range(args.warmup):
        eng.sweep(args.gamma)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eng.time_kernels = True
    eng.kernel_events = []
    barrier()
    t0 = time.perf_counter()
    if args.pipelined:
        ticket = eng.sweep_launch(args.gamma)
        for _ in range(args.steps - 1):
            following = eng.sweep_launch(args.gamma)
            delta = eng.sweep_wait(ticket)
            ticket = following
        delta = eng.sweep_wait(ticket)
    else:
        for _ in range(args.steps):
            delta = eng.sweep(args.gamma)
    barrier()
    elapsed = time.perf_counter() - t0
    eng.time_kernels = False
    ktimes = eng.kernel_times_ms()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3

    # Roofline of the DOMINANT K3 kernel (largest share of the sweep), from HIP events recorded on the
    # launch stream inside the timed region.  One launch of each kernel per chunk, so per-launch
    # bytes = that kernel's algorithmic bytes per sweep / chunks (SURVEY.md section 8d gather model).
    chunks = len(eng.blocks)          # launches of each kernel per sweep
    kbytes = eng.kernel_bytes()
    names = {"main": "spmm_update_kernel", "mid": "spmm_long_kernel<4 waves>", "hub": "spmm_long_kernel<16 waves>",
             "split": "spmm_split_segment_kernel+combine"}
    per_kernel = {}
    for key, ms in ktimes.items():
        if kbytes[key] > 0 and ms > 0:       # ms = per sweep, summed over the blocks
            gbps = kbytes[key] / (ms * 1e-3) / 1e9
            per_kernel[names[key]] = {"avg_launch_ms": ms / chunks, "algorithmic_bytes_per_launch": kbytes[key] / chunks,
                                      "GBps": gbps, "frac": gbps / HBM_PEAK_GBPS}
    dom = max(per_kernel, key=lambda n: per_kernel[n]["avg_launch_ms"])
    pass_ms = sum(ktimes.values())
    pass_bytes = sum(kbytes.values())
    pass_gbps = pass_bytes / (pass_ms * 1e-3) / 1e9
    traffic = None
    tfile = ROOT / "profiles" / "traffic.json"
    if tfile.exists():
        table = json.loads(tfile.read_text())
        entry = table.get(f"{args.workload}_n{world}")
        if entry is None and world > 1 and eng.columns:     # measured on one GPU over the same column slice
            entry = table.get(f"{args.workload}_column_slice_of_{world}")
        traffic = (entry or {}).get(dom, {}).get("bytes_per_launch")

    if world == 1:
        parallelism = f"1 GPU, {chunks} launch block(s)/sweep"
    elif eng.columns:
        parallelism = (f"column split x{world}: every GPU holds the whole graph and columns [{eng.col0}:{eng.col1}) "
                       f"(rank 0) of X and Z; no exchange per sweep, one scalar all-reduce (RCCL); build_P all-reduces "
                       f"the {E} partial dot products")
    else:
        how = ("stored by the producing kernels straight into the readers' tables (hipIpc peer mappings)" if eng.p2p
               else f"exchange={eng.exchange} over RCCL per chunk")
        parallelism = (f"row split x{world}, {chunks} launch block(s)/sweep, {how} "
                       f"({eng.exchange_bytes_per_sweep() / 1e6:.0f} MB received/rank/sweep) + scalar all-reduce")
    result = {
        "metric": "embedding-update iters/sec (Jacobi sweeps of Z <- X + gamma*P*Z, P frozen)",
        "value": args.steps / elapsed, "unit": "sweeps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": dname, "data": "synthetic",
        "config": {"workload": f"{'R-MAT' if gen == 'rmat' else 'power-law'} |V|={V} |E|={E} d={d} {dname}, "
                               f"gamma={args.gamma}, CosineSimilarity "
                               f"(reference mode), seeds {gseed}/{xseed}",
                   "parallelism": parallelism,
                   "host_sync": ("pipelined: the delta of sweep t is read while sweep t+1 runs" if args.pipelined
                                 else "after every sweep (reference semantics)")},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": per_kernel[dom]["GBps"], "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": per_kernel[dom]["frac"], "traffic": traffic,
                     "algorithmic_bytes_per_launch": per_kernel[dom]["algorithmic_bytes_per_launch"],
                     "avg_launch_ms": per_kernel[dom]["avg_launch_ms"], "kernels": per_kernel,
                     "k3_pass": {"bytes": pass_bytes, "ms": pass_ms, "GBps": pass_gbps,
                                 "frac": pass_gbps / HBM_PEAK_GBPS}},
        "build_P_ms": build_p_ms, "last_delta": delta,
    }
    if world == 1 and not args.no_cpu_baseline:
        base, parity = cpu_baseline(csr, X, eng.P_global(), args.gamma, Z1)
        result["cpu_baseline"] = base
        result["parity_rel_l2_vs_oracle_after_1_sweep"] = parity
        if not parity < PARITY_TOL[dname]:
            raise SystemExit(f"parity check failed: rel-L2 {parity}")
    if args.iterate:
        from clane_amd.embedder import Embedder
        from clane_amd.graph import Graph
        from clane_amd.similarity import CosineSimilarity
        g = Graph.from_csr(csr, X)
        eng.set_Z(X)
        g._attach_engine(eng)
        emb = Embedder(g, CosineSimilarity(), dev, gamma=args.gamma, tolerence=args.tolerence, verbose=False,
                       max_sweeps=2000)
        barrier()
        t0 = time.perf_counter()
        emb.iterate()
        barrier()
        wall = time.perf_counter() - t0
        result["iterate"] = {"wall_s": wall, "outer_rounds": len(emb.sweep_counts), "sweeps": sum(emb.sweep_counts),
                             "sweeps_launched": emb.sweeps_launched,
                             "sweeps_per_round": emb.sweep_counts, "tolerence": args.tolerence,
                             "last_outer_delta": emb.outer_deltas[-1],
                             "note": "Embedder.iterate() from Z = X: build_P + propagate per round, reference "
                                     "stopping rule (embedder.py:56-108); sweeps whose delta is provably 0 (after an "
                                     "exactly-zero delta with P frozen) are counted, not launched; not part of the "
                                     "headline value"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()              # leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

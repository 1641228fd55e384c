#!/usr/bin/env python3
"""Headline benchmark: embedding-update sweeps/sec + achieved HBM GB/s of the K3 SpMM kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload rmat2m|rmat200k|powerlaw10m|tiny]

One "step" = one Jacobi sweep  Z <- X + gamma * P Z  over the whole graph, P frozen: the K3
kernels, the deterministic L1-delta reduction, the host read-back of that scalar (the
reference decides after every sweep, embedder.py:94-105) and, for N > 1, the all-reduce of that
scalar.  Inputs are resident in HBM before the timed region.

N > 1: one process per GPU over RCCL.  `python bench.py --gpus N` starts the N ranks itself (fresh child
processes through torch.distributed.run, before this process has touched a GPU) and relays rank 0's JSON
line; started under torchrun (WORLD_SIZE set) it is one of the ranks.  The graph is fixed and divided, so
scaling is STRONG: by default every GPU sweeps d/N columns of all rows (no exchange per sweep);
`--exchange allgather_all` is north_star's literal plan (rows divided, one in-place RCCL all-gather of the
updated rows per sweep), `halo` / `halo_p2p` / `allgather` are the leaner row splits.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
TRAFFIC_NOTE = ("traffic = rocprofv3 FETCH_SIZE x calibrated factor + WRITE_SIZE per launch (separate --pmc passes, "
                "profiles/); these counters sit on the L2's fabric side, so Infinity-Cache hits are counted as "
                "traffic: frac = min(algorithmic, traffic) bytes / kernel time / 8 TB/s is an UPPER bound of the "
                "HBM share, never quoted above the roof; achieved_algorithmic is the no-reuse gather model "
                "(SURVEY 8d), which also counts L2 hits")


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) and hand back
    their exit code.  Nothing in THIS process has initialised the GPU (importing torch does not), and nothing is
    exec'd: the ranks are children, rank 0 writes the JSON line to the inherited stdout."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    log("starting ranks: " + " ".join(cmd))
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode


def oracle_first_sweep(csr, X, P_host, gamma):
    """The C oracle's first sweep from Z = X (oracle/clane_oracle.c); P from the GPU when given, else the oracle's
    own build_P (graph.py:118-128).  Returns (Z1, seconds of the sweep, threads, the P used, X as fp32)."""
    from oracle import clane_oracle_c as OC
    Xf = X.float() if X.dtype != torch.float32 else X    # the oracle computes in fp32 on the (bf16-)rounded inputs
    if P_host is None:
        P_host, _ = OC.build_P(csr.rowptr, csr.colidx, Xf)
    out = torch.empty_like(Xf)
    t0 = time.perf_counter()
    Z, _ = OC.sweep(csr.rowptr, csr.colidx, P_host.float(), Xf, Xf, gamma, out=out)
    return Z, time.perf_counter() - t0, OC.threads(), P_host.float(), Xf


def cpu_baseline(csr, X, P_host, gamma, Z1_gpu, budget_s=15.0):
    """The oracle's sweep timed on this box's host cores: the plain-C restatement (oracle/clane_oracle.c,
    OpenMP over rows, same CSR / fp32) -- kind "port".  Also the parity check of the first GPU sweep."""
    from oracle import clane_oracle as O
    from oracle import clane_oracle_c as OC
    Z, first, threads, P_host, Xf = oracle_first_sweep(csr, X, P_host, gamma)     # warm-up, also the parity sweep
    parity = O.rel_l2(Z1_gpu.float(), Z)
    n = int(max(1, min(20, budget_s // max(first, 1e-3))))
    Za, Zb = Z.clone(), torch.empty_like(Z)
    t0 = time.perf_counter()
    for _ in range(n):
        Zb, _ = OC.sweep(csr.rowptr, csr.colidx, P_host, Xf, Za, gamma, out=Zb)
        Za, Zb = Zb, Za
    per = (time.perf_counter() - t0) / n
    return {"value": 1.0 / per, "unit": "sweeps/s", "cores": threads, "kind": "port",
            "sample": f"{n} full sweeps of the same graph by oracle/clane_oracle.c (plain C, OpenMP over rows, "
                      f"{threads} threads), P taken from the GPU build_P"}, parity, P_host, Xf


def cpu_baseline_torch(csr, Xf, P_host, gamma, budget_s=8.0):
    """The PyTorch-CPU restatement SURVEY 8d names --  Z = X + gamma * torch.sparse.mm(P, Z)  plus the L1 delta,
    as in oracle/clane_oracle.py:sweep -- on all host threads and on ONE thread.  A full sweep takes seconds to
    minutes that way, so each figure is timed on a bounded SAMPLE of the workload, a seeded random 1/m of the rows
    (same degree mix; every m-th row would not do: R-MAT's hubs sit on the ids with trailing zero bits), and scaled
    by the share of the edges the sample holds."""
    deg = np.diff(csr.rowptr)
    E, V = int(csr.rowptr[-1]), csr.num_vertices
    Z = Xf
    all_threads = torch.get_num_threads()

    def timed(stride, threads, reps):
        rows = np.arange(V, dtype=np.int64) if stride == 1 else \
            np.sort(np.random.default_rng(stride).choice(V, size=max(1, V // stride), replace=False))
        take = np.repeat(csr.rowptr[rows], deg[rows]) + (np.arange(int(deg[rows].sum())) -
                                                         np.repeat(np.cumsum(deg[rows]) - deg[rows], deg[rows]))
        idx = torch.stack([torch.from_numpy(np.repeat(np.arange(rows.size), deg[rows])),
                           torch.from_numpy(csr.colidx[take].astype(np.int64))])
        Ps = torch.sparse_coo_tensor(idx, P_host[torch.from_numpy(take)], size=(rows.size, V), is_coalesced=True)
        rows_t = torch.from_numpy(rows)
        Xs, Zs = Xf[rows_t], Z[rows_t]
        sink = torch.from_numpy(deg[rows] == 0)
        torch.set_num_threads(threads)
        try:
            best = float("inf")
            for _ in range(reps + 1):                          # first pass = warm-up
                t0 = time.perf_counter()
                Zn = Xs + gamma * torch.sparse.mm(Ps, Z)
                Zn[sink] = Zs[sink]
                (Zn - Zs).abs().sum()
                best = min(best, time.perf_counter() - t0)
        finally:
            torch.set_num_threads(all_threads)
        share = take.size / max(E, 1)
        return best / share, share, rows.size

    # size the samples from one small probe so that each figure costs a few seconds at most
    probe, share, _ = timed(max(1, V // 20_000), all_threads, 1)
    stride_all = max(1, int(np.ceil(probe * 3 / budget_s)))     # warm-up + 2 timed passes within the budget
    per_all, share_all, n_all = timed(stride_all, all_threads, 2)
    probe1, _, _ = timed(max(1, V // 5_000), 1, 1)
    stride_1 = max(1, int(np.ceil(probe1 * 2 / budget_s)))
    per_1, share_1, n_1 = timed(stride_1, 1, 1)
    return {"value": 1.0 / per_all, "unit": "sweeps/s", "cores": all_threads, "kind": "port",
            "sample": f"X + gamma*torch.sparse.mm(P, Z) + L1 delta (oracle/clane_oracle.py:sweep) on a random "
                      f"1/{stride_all} of the rows ({n_all} rows, {share_all:.1%} of the edges), best of 2, scaled to a whole "
                      f"sweep; torch {torch.__version__}, {all_threads} threads",
            "one_thread": {"value": 1.0 / per_1, "unit": "sweeps/s", "cores": 1,
                           "sample": f"the same on a random 1/{stride_1} of the rows ({n_1} rows, {share_1:.2%} of the edges), "
                                     f"torch.set_num_threads(1)"}}


def traffic_entry(workload: str, world: int, eng, dom: str, slice_of=None):
    """PMC traffic of kernel `dom` from profiles/traffic.json -- only if it was measured with THIS kernel
    configuration (thresholds, launch blocks, compile-time tuning, ...); else (None, why)."""
    tfile = ROOT / "profiles" / "traffic.json"
    if not tfile.exists():
        return None, "no profiles/traffic.json"
    table = json.loads(tfile.read_text())
    key = f"{workload}_column_slice_of_{slice_of}" if slice_of else f"{workload}_n{world}"
    entry = table.get(key)
    if entry is None and world > 1 and eng.columns:          # measured on one GPU over the same column slice
        key = f"{workload}_column_slice_of_{world}"
        entry = table.get(key)
    if entry is None:
        return None, f"no PMC measurement for {key}"
    live = eng.kernel_config()
    then = entry.get("kernel_config")
    if then is None:
        return None, f"{key}: measured before kernel configurations were recorded -- treated as stale"
    diff = sorted(k for k in set(live) | set(then) if live.get(k) != then.get(k) and k != "exchange")
    if diff:
        return None, f"{key}: stale, measured with another kernel configuration (differs in {', '.join(diff)})"
    got = entry.get(dom, {}).get("bytes_per_launch")
    return got, (entry.get("source") if got is not None else f"{key}: kernel {dom} not in the measurement")


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
    ap.add_argument("--class-threshold", type=int, default=None,
                    help="rows above this many edges take the XCD-affine pass (0 = off; default: by row width)")
    ap.add_argument("--class-chunk", type=int, default=256)
    ap.add_argument("--no-split-hubs", action="store_true", help="hub rows by one workgroup each (no segment split)")
    ap.add_argument("--natural-order", action="store_true", help="keep vertex order (default: hot rows first)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "columns", "halo", "halo_p2p", "allgather", "allgather_all"],
                    help="N > 1: auto = columns while a rank's row slice is >= 64 bytes, else halo; columns = every GPU "
                         "holds d/N columns of every row, no exchange per sweep; allgather_all = north_star's literal "
                         "plan: rows divided, one in-place RCCL all-gather of the updated rows per sweep; allgather = "
                         "the same for the live rows only; halo / halo_p2p = rows sent only to the ranks that read "
                         "them (clane_amd/halo.py, partition.py; DESIGN.md section 6)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 flow)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (RCCL refuses that: use --backend gloo)")
    ap.add_argument("--host-sync", default="auto", choices=["auto", "every-sweep", "pipelined"],
                    help="every-sweep: the host reads a sweep's delta before launching the next (the reference's literal "
                         "order); pipelined (SURVEY H5): sweep t+1 is launched before the delta of sweep t is read -- every "
                         "delta is still read, one sweep later (SweepEngine.sweep_launch / sweep_wait; Embedder's "
                         "lagged_check, bit-identical results); auto (default) = what Embedder does by default: pipelined "
                         "when a sweep is estimated below 1 ms (2 ms at N > 1, where the scalar all-reduce adds to the round trip; "
                         "SweepEngine.estimated_sweep_seconds), else every-sweep")
    ap.add_argument("--pipelined", action="store_true", help="same as --host-sync pipelined")
    ap.add_argument("--iterate", action="store_true",
                    help="after the timed sweeps also run the WHOLE algorithm from Z = X -- Embedder.iterate() to "
                         "tolerance (build_P + propagate per outer round) -- and report rounds, sweeps, wall time")
    ap.add_argument("--tolerence", type=int, default=10, help="(reference spelling) for --iterate")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baselines (the parity check stays)")
    ap.add_argument("--no-parity", action="store_true", help="skip the first-sweep check against the C oracle too")
    ap.add_argument("--column-slice-of", type=int, default=None, metavar="N",
                    help="one-GPU rehearsal of ONE rank of the N-GPU column split: sweep only the first d/N columns "
                         "(what every rank of `--gpus N` does); for profiling that rank's kernels, not a headline number")
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch l1_distance over two [V,d] matrices (known bytes) -- PMC calibration")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:        # no launcher: be the launcher
        raise SystemExit(launch_ranks(args.gpus))
    if os.environ.get("CLANE_BENCH_WATCHDOG_S"):                # debugging aid: every rank dumps its stack every S seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["CLANE_BENCH_WATCHDOG_S"]), repeat=True, file=sys.stderr)

    import torch.distributed as dist
    from clane_amd import _hip, synth
    from clane_amd.engine import SweepEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    if world > 1:       # torchrun starts every rank with OMP_NUM_THREADS=1: give each rank its share of the host cores
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))
    n_dev = torch.cuda.device_count()
    if world > 1 and not args.share_gpu and n_dev not in (1, world) and n_dev < world:
        raise SystemExit(f"--gpus {world} but this box shows {n_dev} GPU(s); a rehearsal on fewer GPUs needs "
                         f"--backend gloo --share-gpu")
    # one visible device per rank (a launcher that masks HIP_VISIBLE_DEVICES per process): it is cuda:0 there
    masked = n_dev == 1 and world > 1
    dev = _hip.require_gpu("cuda:0" if (args.share_gpu or masked) else f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    pg = None
    if world > 1:
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)
            except dist.DistBackendError:
                if masked:      # RCCL refuses two ranks on one device: most likely a one-GPU box, not a masking launcher
                    print(f"[bench] rank {rank}: RCCL could not start with {world} ranks and ONE visible GPU; to rehearse "
                          f"the N > 1 flow on a one-GPU box use --backend gloo --share-gpu", file=sys.stderr)
                raise
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
    if args.column_slice_of:
        if world != 1:
            raise SystemExit("--column-slice-of is a one-GPU rehearsal")
        from clane_amd.engine import column_slice
        c0, c1 = column_slice(d, X.dtype, args.column_slice_of, 0)
        X = X[:, c0:c1].contiguous()
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
                      hot_rows_first=not args.natural_order, split_hubs=not args.no_split_hubs,
                      class_threshold=args.class_threshold, class_chunk=args.class_chunk)
    torch.cuda.synchronize()
    log(f"engine up in {time.perf_counter() - t0:.1f}s; rank rows={eng.part.n_local} edges={eng.E_loc} "
        f"rows/kernel: mid(4 waves)={sum(0 if l is None else l.numel() for l in eng.mid_rows)} "
        f"hub(16 waves)={sum(0 if l is None else l.numel() for l in eng.hub_rows)} "
        f"split={sum(0 if l is None else l[0].numel() for l in eng.split_rows)} "
        f"class={sum(0 if l is None else l[0].numel() for l in eng.class_rows)} "
        f"thresholds {eng.long_threshold}/{eng.hub_threshold}")

    # build_P once (timed separately, not part of a step), P frozen afterwards
    eng.build_P()                      # first call loads the code objects; time the second
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.build_P()
    torch.cuda.synchronize()
    build_p_ms = (time.perf_counter() - t0) * 1e3

    calibration_bytes = None
    if args.calibrate:      # known-size streaming read in this library's own 16 B/lane access pattern
        eng.snapshot()
        eng.distance_from_snapshot()
        calibration_bytes = 2 * eng.part.n_local * eng.ld * eng.Zcur.element_size()    # l1_distance reads two matrices
    Z1 = None
    if not args.no_parity:              # the sweep the oracle is checked against (Z = X before it); collective
        eng.sweep(args.gamma)
        Z1 = eng.get_Z()
        if rank != 0:
            Z1 = None
    for _ in range(args.warmup):
        eng.sweep(args.gamma)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pipelined:
        args.host_sync = "pipelined"
    from clane_amd.embedder import Embedder
    pipelined = args.host_sync == "pipelined" or (
        args.host_sync == "auto" and eng.estimated_sweep_seconds() < (Embedder.LAGGED_BELOW_ESTIMATE_S if world > 1
                                                                       else Embedder.LAGGED_BELOW_S))
    eng.time_kernels = True
    eng.kernel_events = []
    barrier()
    t0 = time.perf_counter()
    if pipelined:
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
    # the fraction is quoted from the SMALLER of (algorithmic, measured) bytes, so that cache hits
    # cannot inflate it.
    chunks = len(eng.blocks)          # launches of each kernel per sweep
    kbytes = eng.kernel_bytes()
    per_kernel = {}
    KERNEL_NAMES = eng.kernel_names()
    for key, ms in ktimes.items():
        if kbytes[key] > 0 and ms > 0:       # ms = per sweep, summed over the blocks
            gbps = kbytes[key] / (ms * 1e-3) / 1e9
            per_kernel[KERNEL_NAMES[key]] = {"avg_launch_ms": ms / chunks,
                                             "algorithmic_bytes_per_launch": kbytes[key] / chunks,
                                             "achieved_algorithmic": gbps}
    dom = max(per_kernel, key=lambda n: per_kernel[n]["avg_launch_ms"])
    pass_ms = sum(ktimes.values())
    pass_bytes = sum(kbytes.values())
    pass_traffic = 0.0
    for name, pk in per_kernel.items():
        tr, why = traffic_entry(args.workload, world, eng, name, args.column_slice_of)
        alg = pk["algorithmic_bytes_per_launch"]
        pk["traffic"] = tr
        counted = min(alg, tr) if tr is not None else alg
        rate = counted / (pk["avg_launch_ms"] * 1e-3) / 1e9
        pk["achieved"] = min(rate, HBM_PEAK_GBPS) if tr is None else rate
        pk["frac"] = pk["achieved"] / HBM_PEAK_GBPS
        pk["traffic_over_algorithmic"] = None if tr is None else tr / alg
        if tr is None:
            pk["traffic_missing"] = why
        pass_traffic = None if (tr is None or pass_traffic is None) else pass_traffic + tr * chunks
    pass_counted = min(pass_bytes, pass_traffic) if pass_traffic is not None else pass_bytes
    pass_rate = pass_counted / (pass_ms * 1e-3) / 1e9
    if pass_traffic is None:
        pass_rate = min(pass_rate, HBM_PEAK_GBPS)
    pd = per_kernel[dom]
    note = TRAFFIC_NOTE if pd["traffic"] is not None else (
        f"no valid PMC traffic for this configuration ({pd['traffic_missing']}): frac is the algorithmic rate, capped "
        f"at the roof -- rates above 8 TB/s mean rows served from L2 / the Infinity Cache, not HBM")

    if args.column_slice_of:
        parallelism = (f"REHEARSAL on 1 GPU of one rank of the column split x{args.column_slice_of}: columns "
                       f"[0:{X.shape[1]}) of X and Z, whole graph; not a headline number")
    elif world == 1:
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
                   "host_sync": (f"pipelined ({args.host_sync}): the delta of sweep t is read while sweep t+1 runs"
                                 if pipelined else f"after every sweep ({args.host_sync}; the reference's order)")},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": pd["achieved"], "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": pd["frac"], "traffic": pd["traffic"],
                     "achieved_algorithmic": pd["achieved_algorithmic"],
                     "traffic_over_algorithmic": pd["traffic_over_algorithmic"],
                     "algorithmic_bytes_per_launch": pd["algorithmic_bytes_per_launch"],
                     "avg_launch_ms": pd["avg_launch_ms"], "note": note, "kernels": per_kernel,
                     "k3_pass": {"algorithmic_bytes": pass_bytes, "traffic": pass_traffic, "ms": pass_ms,
                                 "achieved": pass_rate, "frac": pass_rate / HBM_PEAK_GBPS,
                                 "achieved_algorithmic": pass_bytes / (pass_ms * 1e-3) / 1e9},
                     "kernel_config": eng.kernel_config()},
        "build_P_ms": build_p_ms, "last_delta": delta,
    }
    if calibration_bytes is not None:
        result["calibration"] = {"kernel": "l1_distance_kernel", "bytes_read": calibration_bytes}
    if Z1 is not None:              # rank 0: the first sweep against the C oracle -- at any N
        from oracle import clane_oracle as O
        if world == 1 and not args.no_cpu_baseline:
            base, parity, P_host, Xf = cpu_baseline(csr, X, eng.P_global(), args.gamma, Z1)
            result["cpu_baseline"] = base
            result["cpu_baseline_torch"] = cpu_baseline_torch(csr, Xf, P_host, args.gamma)
        else:                       # P from the oracle's own build_P when the ranks hold only their rows of it
            if world > 1:           # torchrun gives every rank OMP_NUM_THREADS=1; the others are idle at the barrier now
                from oracle import clane_oracle_c as OC
                OC.set_threads(os.cpu_count() or 1)
            P_host = eng.P_global() if (world == 1 or eng.columns) else None
            Zo, _, _, _, _ = oracle_first_sweep(csr, X, P_host, args.gamma)
            parity = O.rel_l2(Z1.float(), Zo)
        result["parity_rel_l2_vs_oracle_after_1_sweep"] = parity
        if not parity < PARITY_TOL[dname]:
            print(json.dumps(result), flush=True)
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

#!/usr/bin/env python3
"""Headline benchmark: embedding-update sweeps/sec + achieved HBM GB/s of the K3 SpMM kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks B] [--workload rmat2m|uniform2m|rmat200k|powerlaw10m|tiny]

One "step" = one Jacobi sweep  Z <- X + gamma * P Z  over the whole graph, P frozen: the K3
kernels, the deterministic L1-delta reduction, the host read-back of that scalar (the
reference decides after every sweep, embedder.py:94-105) and, for N > 1, the all-reduce of that
scalar.  Inputs are resident in HBM before the timed region.

Timing protocol: W warm-up sweeps, then B blocks (default 5) of EXACTLY K sweeps, each block bracketed
by a barrier + torch.cuda.synchronize() on both sides (host clock, MAX over ranks) and by a HIP event pair on the
sweep's stream; `value` / `ms_per_step` are the MEDIAN block, `ms_per_step_min` / `_max` the spread.

N > 1: one process per GPU over RCCL.  `python bench.py --gpus N` starts the N ranks itself (fresh child
processes through torch.distributed.run, before this process has touched a GPU) and relays rank 0's JSON
line; started under torchrun (WORLD_SIZE set) it is one of the ranks.  The graph is fixed and divided, so
scaling is STRONG: by default every GPU sweeps d/N columns of all rows (no exchange per sweep);
`--exchange allgather_all` is north_star's literal plan (rows divided, one in-place RCCL all-gather of the
updated rows per sweep), `halo` / `halo_p2p` / `allgather` are the leaner row splits.  At N > 1 the record also
carries a `comm` block (what RCCL saw, time inside the collectives, the delta-stream variant timed beside the default)
and -- `--also-exchange`, default allgather_all,allgather -- the same measurement of north_star's literal division
in `north_star_literal` and of the live-rows all-gather in `other_divisions`, so ONE record answers both "what
scales" and "what north_star asked for".  `--rehearse-rccl` runs that whole N > 1 flow through the real RCCL
library with the one rank a one-GPU box can give it.  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import socket
import statistics
import subprocess
import sys
import threading
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
    # The roofline's ANCHOR: uniform-random (src, dst) pairs, no hubs, no skew -- nothing for the L2s or the Infinity Cache
    # to reuse (the 2 GB table is 8x the cache), so the PMC traffic equals the algorithmic bytes and `frac` is a true HBM
    # fraction with no cache caveat.  Same |V|, |E|, d as the headline.
    "uniform2m": ("uniform", 2_000_000, 40_000_000, 256, "f32", 12, 4),
    "rmat200k256": ("rmat", 200_000, 4_000_000, 256, "f32", 1, 2),    # config 2's graph at 1-KiB rows: a cache-resident table
    # config 3's graph at other row widths: where do column tiles pay? (profiles/r05_column_tiles_ab.md)
    "rmat2m512": ("rmat", 2_000_000, 40_000_000, 512, "f32", 3, 4),
    "rmat2m384": ("rmat", 2_000_000, 40_000_000, 384, "f32", 3, 4),
    "rmat2m1024": ("rmat", 2_000_000, 40_000_000, 1024, "f32", 3, 4),
    "rmat2m512bf16": ("rmat", 2_000_000, 40_000_000, 512, "bf16", 3, 4),
    "tiny": ("rmat", 20_000, 200_000, 64, "f32", 7, 8),
    "tiny12": ("rmat", 20_000, 200_000, 12, "f32", 7, 8),             # 3 packs a row: more ranks than packs leaves idle column ranks
    # 8x config 3: a 16 GiB embedding matrix (byte offsets beyond 32 bits, ~85 GB of HBM in use) -- capacity check
    "rmat16m": ("rmat", 16_000_000, 320_000_000, 256, "f32", 9, 10),
}
GENERATOR_NAMES = {"rmat": "R-MAT", "powerlaw": "power-law", "uniform": "uniform-random pairs (duplicates merged)"}
DTYPES = {"f32": torch.float32, "bf16": torch.bfloat16, "f64": torch.float64}
PARITY_TOL = {"f32": 1e-4, "f64": 1e-10, "bf16": 8e-3}       # bf16: 2^-8 rounding of every stored value
PARITY_P_TOL = {"f32": 2e-6, "f64": 1e-12, "bf16": 1e-4}     # P itself (fp32 arithmetic on the stored values)
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TRAFFIC_NOTE = ("traffic = rocprofv3 FETCH_SIZE x calibrated factor + WRITE_SIZE per launch, taken in separate --pmc "
                "passes of this same command (profiles/) and matched to the live run by kernel configuration: bytes "
                "and time come from different runs (kernels[..].pmc_run_over_live_time says how the kernel's duration "
                "in the PMC run compares with this run's); these counters sit on the L2's fabric side, so Infinity-Cache "
                "hits are counted as traffic: frac = min(algorithmic, traffic) bytes / kernel time / 8 TB/s, capped "
                "at 1, is an UPPER bound of the HBM share; achieved_algorithmic is the no-reuse gather model "
                "(SURVEY 8d), which also counts L2 hits")
# north_star_literal block: past this many seconds the main record is printed without it (a healthy block takes ~10 s at
# config 3; the driver's own limit for the whole command is 600 s, and the main record must come out well inside it)
LITERAL_DEADLINE_S = float(os.environ.get("CLANE_BENCH_LITERAL_DEADLINE_S", "150"))


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
    then, live = dict({"column_tiles": 1}, **then), dict({"column_tiles": 1}, **live)   # before tiles existed: one tile
    diff = sorted(k for k in set(live) | set(then) if live.get(k) != then.get(k) and k != "exchange")
    if diff:
        return None, f"{key}: stale, measured with another kernel configuration (differs in {', '.join(diff)})"
    got = entry.get(dom, {}).get("bytes_per_launch")
    if got is None:
        return None, f"{key}: kernel {dom} not in the measurement"
    return got, {"source": entry.get("source"), "avg_us_under_pmc": entry[dom].get("avg_us_under_pmc")}


def make_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=5,
                    help="timed blocks of --steps sweeps each (SURVEY 8d: median of 5); the JSON line reports the median "
                         "block and the spread")
    ap.add_argument("--workload", default="rmat2m", choices=sorted(WORKLOADS))
    ap.add_argument("--gamma", type=float, default=0.76)
    ap.add_argument("--chunks", type=int, default=None)
    ap.add_argument("--long-threshold", type=int, default=None)
    ap.add_argument("--hub-threshold", type=int, default=None)
    ap.add_argument("--class-threshold", type=int, default=None,
                    help="rows above this many edges take the XCD-affine pass (0 = off; default: by row width)")
    ap.add_argument("--class-chunk", type=int, default=256)
    ap.add_argument("--column-tiles", type=int, default=None,
                    help="one GPU: a sweep as T passes over T column ranges of the tables (default: the engine's rule)")
    ap.add_argument("--no-split-hubs", action="store_true", help="hub rows by one workgroup each (no segment split)")
    ap.add_argument("--natural-order", action="store_true", help="keep vertex order (default: hot rows first)")
    ap.add_argument("--exchange", default="auto",
                    choices=["auto", "columns", "halo", "halo_p2p", "allgather", "allgather_all"],
                    help="N > 1: auto = columns while a rank's row slice is >= 64 bytes, else halo; columns = every GPU "
                         "holds d/N columns of every row, no exchange per sweep; allgather_all = north_star's literal "
                         "plan: rows divided, one in-place RCCL all-gather of the updated rows per sweep; allgather = "
                         "the same for the live rows only; halo / halo_p2p = rows sent only to the ranks that read "
                         "them (clane_amd/halo.py, partition.py; DESIGN.md section 6)")
    ap.add_argument("--also-exchange", default="allgather_all,allgather",
                    help="N > 1: after the main division's timed blocks, rebuild the engine with each of these divisions "
                         "(comma-separated) and report its sweeps/s, parity and collective time in the same record: "
                         "allgather_all = north_star's row partition + one all-gather of the owned rows per sweep -> "
                         "`north_star_literal`; the others (allgather = the same for the rows that change and are read; "
                         "halo, halo_p2p) -> `other_divisions`.  The main division is skipped; none = off.  The default "
                         "adds the two in-place all-gather forms; `halo` (all_to_all_single with uneven splits) is left to "
                         "an explicit request: nothing measured after the main division may put its record at risk.")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 flow)")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="one GPU, ONE rank, but a real RCCL process group: everything an N > 1 run does -- group "
                         "start-up, the agreement check, the column split's collectives, the comm block, the "
                         "north_star_literal block with its in-place all-gathers -- goes through the real library (each "
                         "collective is the identity on one rank).  Shows no scaling; catches API, dtype and stream "
                         "mistakes that the gloo rehearsals cannot.  Not a headline number.")
    ap.add_argument("--no-fabric-probe", action="store_true",
                    help="N > 1: skip tools/fabric_probe.py (child processes that time the all-gather / direct exchange of "
                         "a rank's slice of Z and log RCCL's algorithm / protocol / channels before the timed run)")
    ap.add_argument("--no-delta-stream-ab", action="store_true",
                    help="N > 1: skip the extra timed blocks with the delta's all-reduce on its own stream (comm.delta_stream_ab)")
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
    ap.add_argument("--legs", default="auto",
                    help="further workloads measured after the main one, in the same process and JSON line (one GPU only): "
                         "auto (default) = uniform2m,rmat200k,powerlaw10m when the main workload is the headline (rmat2m): the "
                         "roofline's no-reuse anchor -> roofline.anchor, BASELINE configs 2 and 4 (the latter incl. "
                         "Embedder.iterate() to tolerence) -> other_workloads; a comma-separated list; none = off")
    ap.add_argument("--budget-s", type=float, default=270.0,
                    help="wall-time plan of the whole command: a leg is only started while the time used so far plus its "
                         "estimate fits (the driver allows 600 s; the main record never waits for a leg)")
    ap.add_argument("--no-parity", action="store_true", help="skip the first-sweep check against the C oracle too")
    ap.add_argument("--column-slice-of", type=int, default=None, metavar="N",
                    help="one-GPU rehearsal of ONE rank of the N-GPU column split: sweep only the first d/N columns "
                         "(what every rank of `--gpus N` does); for profiling that rank's kernels, not a headline number")
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch l1_distance over two [V,d] matrices (known bytes) -- PMC calibration")
    return ap


class Ranks:
    """The process group as bench.py uses it (a no-op on one GPU)."""

    def __init__(self, world, rank, dev, pg, rehearsal=False, fabric_probe=None):
        self.world, self.rank, self.dev, self.pg = world, rank, dev, pg
        self.fabric_probe = fabric_probe          # rank 0: what tools/fabric_probe.py measured before the GPUs were touched
        # `grouped`: there is a process group and every collective is really issued -- N > 1, or the one-rank RCCL
        # rehearsal (--rehearse-rccl), where each is the identity but goes through the real library
        self.rehearsal = bool(rehearsal)
        self.grouped = world > 1 or self.rehearsal

    def barrier(self):
        if self.grouped:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, values):
        t = torch.tensor(list(values), dtype=torch.float64, device=self.dev)
        if self.grouped:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    def gather_objects(self, obj):
        if not self.grouped:
            return [obj]
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def agree_to_fail(self, failed: bool) -> bool:
        """True on EVERY rank when any rank says so (one scalar all-reduce): a failed check on rank 0 must not leave
        the others inside a collective until the launcher tears them down."""
        return self.max_over_ranks([1.0 if failed else 0.0])[0] > 0


def generate_input(args, ranks: Ranks):
    """The synthetic graph + content embeddings, the same on every rank."""
    import torch.distributed as dist
    from clane_amd import synth
    gen, V, E, d, dname, gseed, xseed = WORKLOADS[args.workload]
    world, rank, dev = ranks.world, ranks.rank, ranks.dev
    if os.environ.get("CLANE_BENCH_PERTURB_RANK") == str(rank) and world > 1:
        gseed += 1000           # test hook: this rank draws a different graph, the agreement check must repair it
    make = {"rmat": lambda: synth.rmat_csr(V, E, seed=gseed, device=str(dev)),
            "powerlaw": lambda: synth.powerlaw_csr(V, E, seed=gseed, device=str(dev)),
            "uniform": lambda: synth.uniform_random_csr(V, E, seed=gseed, device=str(dev))}[gen]
    if world > 1 and args.share_gpu:
        # Rehearsal with every rank on ONE card: the generator's rocPRIM sort / unique kernels (decoupled look-back:
        # workgroups spin on their predecessors) crawl when several processes run them on a time-sliced GPU -- four
        # ranks sat in torch.unique for minutes at config 3 (round 2, gpurun_out/final/bench_n4.err).  One rank at
        # a time, the others wait at a barrier on the host.
        csr = None
        for turn in range(world):
            if turn == rank:
                csr = make()
                torch.cuda.synchronize()
            dist.barrier()
    else:
        csr = make()
    X = synth.gaussian_X(V, d, seed=xseed).to(DTYPES[dname])
    if args.column_slice_of:
        if world != 1:
            raise SystemExit("--column-slice-of is a one-GPU rehearsal")
        from clane_amd.engine import column_slice
        c0, c1 = column_slice(d, X.dtype, args.column_slice_of, 0)
        X = X[:, c0:c1].contiguous()
    if ranks.grouped:   # every rank generated the graph on its own GPU from the same seed: make sure they agree
        mine = (csr.num_edges, int(csr.colidx.astype(np.int64).sum()), int(csr.rowptr[::997].sum()),
                float(X[::9973].double().sum()))
        everyone = ranks.gather_objects(mine)
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
    return csr, X


def run_sweeps(eng, args, pipelined: bool) -> float:
    """Exactly args.steps sweeps the way the host loop runs them; returns the last delta."""
    delta = float("nan")
    if pipelined:
        ticket = eng.sweep_launch(args.gamma)
        for _ in range(args.steps - 1):
            following = eng.sweep_launch(args.gamma)
            delta = eng.sweep_wait(ticket)
            ticket = following
        return eng.sweep_wait(ticket)
    for _ in range(args.steps):
        delta = eng.sweep(args.gamma)
    return delta


def timed_blocks(eng, args, ranks: Ranks, pipelined: bool, n_blocks: int):
    """`n_blocks` blocks of exactly args.steps sweeps, each bracketed by barrier + torch.cuda.synchronize() on both
    sides (host clock, MAX over ranks) and by a HIP event pair on the sweep's stream.  Returns (wall seconds per block,
    this rank's own seconds per block, HIP-event ms per block, last delta)."""
    wall, local_wall, hip_ms = [], [], []
    delta = float("nan")
    for _ in range(n_blocks):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ranks.barrier()
        t0 = time.perf_counter()
        ev0.record()
        delta = run_sweeps(eng, args, pipelined)
        ev1.record()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        ranks.barrier()
        elapsed = time.perf_counter() - t0
        wall.append(ranks.max_over_ranks([elapsed])[0])
        local_wall.append(mine)
        hip_ms.append(ev0.elapsed_time(ev1))
    return wall, local_wall, hip_ms, delta


def delta_stream_ab(eng, args, ranks: Ranks, pipelined: bool, off_wall) -> dict:
    """The same protocol once more with the delta's all-reduce on a stream of its own (SweepEngine(delta_stream=True)):
    off by default because one rank cannot show whether it pays (profiles/r02_delta_stream_ab.md) -- a run between
    real GPUs can, and this is where it says so.  `value` stays the default's."""
    try:
        eng.use_delta_stream(True)
        for _ in range(3):
            eng.sweep(args.gamma)
        on, _, _, _ = timed_blocks(eng, args, ranks, pipelined, min(3, max(1, args.blocks)))
        return {"off_ms_per_step": statistics.median(off_wall) / args.steps * 1e3,
                "on_ms_per_step": statistics.median(on) / args.steps * 1e3,
                "note": "on = the all-reduce of the delta and its copy to the host on a stream of their own "
                        "(SweepEngine(delta_stream=True)); the record's value is the default (off)"}
    except Exception as exc:            # noqa: BLE001 -- an extra; the measurement above stands
        return {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        eng.use_delta_stream(False)


def measure_division(args, ranks: Ranks, csr, X, exchange: str, time_kernels: bool):
    """Engine for `exchange`, build_P (timed on its second call), the parity sweep, warm-up, and the timed blocks.
    Returns a dict of everything measured (every rank gets the same numbers where they are reduced)."""
    from clane_amd.embedder import Embedder
    from clane_amd.engine import SweepEngine
    world, rank, dev = ranks.world, ranks.rank, ranks.dev
    t0 = time.perf_counter()
    comm = None
    if ranks.rehearsal:     # one rank, real RCCL: the engine keeps the division it is given and issues every collective
        from clane_amd.comm import TorchComm
        comm = TorchComm(ranks.pg, force_collectives=True)
        exchange = "columns" if exchange == "auto" else exchange
    eng = SweepEngine(csr, X, dev, process_group=ranks.pg, comm=comm, chunks=args.chunks,
                      long_threshold=args.long_threshold, hub_threshold=args.hub_threshold, exchange=exchange,
                      hot_rows_first=not args.natural_order, split_hubs=not args.no_split_hubs,
                      class_threshold=args.class_threshold, class_chunk=args.class_chunk, column_tiles=args.column_tiles)
    torch.cuda.synchronize()
    log(f"engine up in {time.perf_counter() - t0:.1f}s ({eng.exchange}); rank rows={eng.part.n_local} edges={eng.E_loc} "
        f"rows/kernel: mid(4 waves)={sum(0 if l is None else l.numel() for l in eng.mid_rows)} "
        f"hub(16 waves)={sum(0 if l is None else l.numel() for l in eng.hub_rows)} "
        f"split={sum(0 if l is None else l[0].numel() for l in eng.split_rows)} "
        f"class={sum(0 if l is None else l[0].numel() for l in eng.class_rows)} "
        f"thresholds {eng.long_threshold}/{eng.hub_threshold}")

    # build_P (timed separately, not part of a step), P frozen afterwards.  Twice: "cold" = the row norms recomputed by
    # K0 (a pass over Z: the first build_P of a run, or after set_Z), and as it runs in every later outer round of
    # Embedder.iterate(), where the outer-delta pass (embedder.py:60) has left the norms behind.
    eng.build_P()                      # first call loads the code objects
    times = {}
    for name, cold in (("cold", True), ("in_loop", False)):
        if cold:
            eng.sq_ok[eng.cur] = False
        ranks.barrier()
        t0 = time.perf_counter()
        eng.build_P()
        torch.cuda.synchronize()
        times[name] = ranks.max_over_ranks([(time.perf_counter() - t0) * 1e3])[0]

    out = {"eng": eng, "build_P_ms": times["in_loop"], "build_P_cold_ms": times["cold"], "calibration_bytes": None, "Z1": None}
    if args.calibrate:      # known-size streaming read in this library's own 16 B/lane access pattern
        eng.l1_between(0, 1)
        out["calibration_bytes"] = 2 * eng.part.n_local * eng.ld * eng.Zcur.element_size()    # l1_distance reads two matrices
    if not args.no_parity:              # the sweep the oracle is checked against (Z = X before it); collective
        eng.sweep(args.gamma)
        Z1 = eng.get_Z()
        out["Z1"] = Z1 if rank == 0 else None
    for _ in range(args.warmup):
        eng.sweep(args.gamma)

    host_sync = "pipelined" if args.pipelined else args.host_sync
    pipelined = host_sync == "pipelined" or (
        host_sync == "auto" and eng.estimated_sweep_seconds() < (Embedder.LAGGED_BELOW_ESTIMATE_S if ranks.grouped
                                                                  else Embedder.LAGGED_BELOW_S))
    out["pipelined"], out["host_sync"] = pipelined, host_sync
    eng.time_kernels = time_kernels
    eng.kernel_events = []
    eng.time_collectives = ranks.grouped
    eng.collective_events = []
    wall, local_wall, hip_ms, delta = timed_blocks(eng, args, ranks, pipelined, max(1, args.blocks))
    eng.time_kernels = eng.time_collectives = False
    ab = delta_stream_ab(eng, args, ranks, pipelined, wall) if (ranks.grouped and not args.no_delta_stream_ab) else None
    med = statistics.median(wall)
    out.update({
        "delta": delta, "elapsed": med, "value": args.steps / med, "ms_per_step": med / args.steps * 1e3,
        "ms_per_step_min": min(wall) / args.steps * 1e3, "ms_per_step_max": max(wall) / args.steps * 1e3,
        "block_ms_per_step": [w / args.steps * 1e3 for w in wall],
        "ms_per_step_hip_events": ranks.max_over_ranks([statistics.median(hip_ms) / args.steps])[0],
        "ktimes": eng.kernel_times_ms() if time_kernels else {}, "ctimes": eng.collective_times_ms(),
        # this rank's own clock, before the closing barrier: the spread over the ranks says who waits for whom
        "rank_ms_per_step": ranks.gather_objects(statistics.median(local_wall) / args.steps * 1e3),
        "delta_stream_ab": ab,
    })
    return out


def comm_block(args, ranks: Ranks, m) -> dict:
    """What a reader of an N > 1 record must be able to check: which backend, how many ranks and which devices the
    process group really had, what a rank receives per sweep and how long the sweep's stream spends in collectives."""
    import torch.distributed as dist
    eng, dev = m["eng"], ranks.dev
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": ranks.rank, "device": str(dev), "name": props.name,
            "pci_bus_id": f"{getattr(props, 'pci_domain_id', 0):04x}:{getattr(props, 'pci_bus_id', 0):02x}:"
                          f"{getattr(props, 'pci_device_id', 0):02x}",
            "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid(),
            "visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", "all"))}
    devices = ranks.gather_objects(mine)
    recv = ranks.gather_objects(int(eng.exchange_bytes_per_sweep()))
    per_rank = m["rank_ms_per_step"]
    c = m["ctimes"]
    collective = ranks.max_over_ranks([c.get("exchange_exposed", 0.0) + c.get("allreduce", 0.0),
                                       c.get("exchange_exposed", 0.0), c.get("allreduce", 0.0)])
    return {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(),
            "distinct_devices": len({(d_["uuid"], d_["pci_bus_id"]) for d_ in devices}), "devices": devices,
            "shared_gpu_rehearsal": bool(args.share_gpu), "exchange": eng.exchange,
            "exchange_bytes_per_sweep": max(recv), "exchange_bytes_per_sweep_by_rank": recv,
            "collective_ms_per_sweep": collective[0],
            "collective_detail": {"exchange_exposed_ms": collective[1], "allreduce_ms": collective[2],
                                  "sweeps_timed": c.get("sweeps_timed", 0),
                                  "how": "HIP events on the sweep's stream, max over ranks: the wait for the row "
                                         "exchange still outstanding once the rank's own kernels are done (what the "
                                         "per-chunk overlap did not hide) + the all-reduce of the delta scalar"},
            "ms_per_step_by_rank": per_rank, "ms_per_step_rank_min": min(per_rank), "ms_per_step_rank_max": max(per_rank),
            "collectives_issued": dict(eng.comm.calls) if hasattr(eng.comm, "calls") else None,
            "delta_stream_ab": m.get("delta_stream_ab"),
            # what RCCL chose and what a link delivers at the literal plan's message size (ring vs direct),
            # measured by child processes before the timed run (tools/fabric_probe.py); None off rank 0 / --no-fabric-probe
            "fabric_probe": ranks.fabric_probe}


def describe_parallelism(args, world, eng, X, E) -> str:
    chunks = len(eng.blocks)
    if args.column_slice_of:
        return (f"REHEARSAL on 1 GPU of one rank of the column split x{args.column_slice_of}: columns "
                f"[0:{X.shape[1]}) of X and Z, whole graph; not a headline number")
    if world == 1 and eng.exchange == "none":
        tiles = (f", {len(eng.tiles)} column tiles per sweep ({', '.join(f'[{a}:{b})' for a, b in eng.tiles)}: the update "
                 f"is independent per column, embedder.py:92)") if len(eng.tiles) > 1 else ""
        return f"1 GPU, {chunks} launch block(s)/sweep{tiles}"
    if eng.columns:
        return (f"column split x{world}: every GPU holds the whole graph and columns [{eng.col0}:{eng.col1}) "
                f"(rank 0) of X and Z; no exchange per sweep, one scalar all-reduce (RCCL); build_P all-reduces "
                f"the {E} partial dot products")
    how = ("stored by the producing kernels straight into the readers' tables (hipIpc peer mappings)" if eng.p2p
           else f"exchange={eng.exchange} over RCCL per chunk")
    return (f"row split x{world}, {chunks} launch block(s)/sweep, {how} "
            f"({eng.exchange_bytes_per_sweep() / 1e6:.0f} MB received/rank/sweep) + scalar all-reduce")


def roofline_block(args, world, m) -> dict:
    """Roofline of the DOMINANT K3 kernel (largest share of the sweep), from HIP events recorded on the launch stream
    inside the timed region.  One launch of each kernel per chunk, so per-launch bytes = that kernel's algorithmic
    bytes per sweep / chunks (SURVEY.md section 8d gather model).  The fraction is quoted from the SMALLER of
    (algorithmic, measured) bytes and never above the roof, so that cache hits cannot inflate it."""
    eng, ktimes = m["eng"], m["ktimes"]
    chunks = eng.launches_per_sweep()     # launches of each kernel per sweep (launch blocks x column tiles)
    kbytes = eng.kernel_bytes()
    per_kernel = {}
    names = eng.kernel_names()
    for key, ms in ktimes.items():
        if kbytes[key] > 0 and ms > 0:       # ms = per sweep, summed over the blocks
            per_kernel[names[key]] = {"avg_launch_ms": ms / chunks,
                                      "algorithmic_bytes_per_launch": kbytes[key] / chunks,
                                      "achieved_algorithmic": kbytes[key] / (ms * 1e-3) / 1e9}
    if not per_kernel:                       # a rank without columns (d < N packs) launches nothing
        return {"bound": "hbm", "kernel": None, "achieved": 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": 0.0,
                "traffic": None, "note": "rank 0 holds no columns of this matrix: no kernel to time",
                "kernel_config": eng.kernel_config()}
    dom = max(per_kernel, key=lambda n: per_kernel[n]["avg_launch_ms"])
    pass_ms = sum(ktimes.values())
    pass_bytes = sum(kbytes.values())
    pass_traffic = 0.0
    for name, pk in per_kernel.items():
        tr, why = traffic_entry(args.workload, world, eng, name, args.column_slice_of)
        alg = pk["algorithmic_bytes_per_launch"]
        pk["traffic"] = tr
        counted = min(alg, tr) if tr is not None else alg
        pk["achieved"] = min(counted / (pk["avg_launch_ms"] * 1e-3) / 1e9, HBM_PEAK_GBPS)
        pk["frac"] = pk["achieved"] / HBM_PEAK_GBPS
        pk["traffic_over_algorithmic"] = None if tr is None else tr / alg
        if tr is None:
            pk["traffic_missing"] = why
        elif why.get("avg_us_under_pmc"):
            # bytes and time come from different runs of the same configuration: how far apart were those runs' kernels?
            pk["pmc_run_avg_launch_ms"] = why["avg_us_under_pmc"] / 1e3
            pk["pmc_run_over_live_time"] = why["avg_us_under_pmc"] / 1e3 / pk["avg_launch_ms"]
        pass_traffic = None if (tr is None or pass_traffic is None) else pass_traffic + tr * chunks
    pass_counted = min(pass_bytes, pass_traffic) if pass_traffic is not None else pass_bytes
    pass_rate = min(pass_counted / (pass_ms * 1e-3) / 1e9, HBM_PEAK_GBPS)
    pd = per_kernel[dom]
    note = TRAFFIC_NOTE if pd["traffic"] is not None else (
        f"no valid PMC traffic for this configuration ({pd['traffic_missing']}): frac is the algorithmic rate, capped "
        f"at the roof -- rates above 8 TB/s mean rows served from L2 / the Infinity Cache, not HBM")
    return {"bound": "hbm", "kernel": dom, "achieved": pd["achieved"], "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": pd["frac"], "traffic": pd["traffic"],
            "traffic_source": None if pd["traffic"] is None else
            "separate --pmc runs of this command (profiles/traffic.json), not the timed run",
            "achieved_algorithmic": pd["achieved_algorithmic"],
            "traffic_over_algorithmic": pd["traffic_over_algorithmic"],
            "pmc_run_over_live_time": pd.get("pmc_run_over_live_time"),
            "algorithmic_bytes_per_launch": pd["algorithmic_bytes_per_launch"],
            "avg_launch_ms": pd["avg_launch_ms"], "note": note, "kernels": per_kernel,
            "k3_pass": {"algorithmic_bytes": pass_bytes, "traffic": pass_traffic, "ms": pass_ms,
                        "achieved": pass_rate, "frac": pass_rate / HBM_PEAK_GBPS,
                        "achieved_algorithmic": pass_bytes / (pass_ms * 1e-3) / 1e9},
            "kernel_config": eng.kernel_config()}


def start_ranks(args) -> Ranks:
    """This process as one rank: device, process group (RCCL, or gloo for rehearsals), host threads."""
    import torch.distributed as dist
    from clane_amd import _hip
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    if world > 1:       # torchrun starts every rank with OMP_NUM_THREADS=1: give each rank its share of the host cores
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))
    probe = None
    if (world > 1 or args.rehearse_rccl) and not args.no_fabric_probe:
        # BEFORE this process touches its GPU: child processes measure what RCCL and the links do with the literal
        # plan's message (this rank's slice of Z) and log RCCL's choices; the timed run below never logs
        import importlib.util
        spec = importlib.util.spec_from_file_location("clane_fabric_probe", ROOT / "tools" / "fabric_probe.py")
        fp = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(fp)
        _, V_, _, d_, dname_, _, _ = WORKLOADS[args.workload]
        es = torch.empty(0, dtype=DTYPES[dname_]).element_size()
        t0 = time.perf_counter()
        probe = fp.run(world, rank, local_rank, slice_bytes=max(16, V_ * d_ * es // max(world, 1)),
                       backend="nccl" if args.rehearse_rccl else args.backend, share_gpu=args.share_gpu)
        TIMELINE["fabric_probe_s"] = time.perf_counter() - t0
    n_dev = torch.cuda.device_count()
    if world > 1 and not args.share_gpu and n_dev not in (1, world) and n_dev < world:
        raise SystemExit(f"--gpus {world} but this box shows {n_dev} GPU(s); a rehearsal on fewer GPUs needs "
                         f"--backend gloo --share-gpu")
    # one visible device per rank (a launcher that masks HIP_VISIBLE_DEVICES per process): it is cuda:0 there
    masked = n_dev == 1 and world > 1
    dev = _hip.require_gpu("cuda:0" if (args.share_gpu or masked) else f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    pg = None
    if args.rehearse_rccl:
        if world != 1:
            raise SystemExit("--rehearse-rccl is the ONE-rank rehearsal of the N > 1 flow")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(minutes=30))
        pg = dist.group.WORLD
    if world > 1:
        # rank 0 spends seconds in the CPU oracle while the others wait inside a collective: well within this
        patience = datetime.timedelta(minutes=30)
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev, timeout=patience)
            except dist.DistBackendError:
                if masked:      # RCCL refuses two ranks on one device: most likely a one-GPU box, not a masking launcher
                    print(f"[bench] rank {rank}: RCCL could not start with {world} ranks and ONE visible GPU; to rehearse "
                          f"the N > 1 flow on a one-GPU box use --backend gloo --share-gpu", file=sys.stderr)
                raise
        else:
            dist.init_process_group("gloo", timeout=patience)
        pg = dist.group.WORLD
    return Ranks(world, rank, dev, pg, rehearsal=args.rehearse_rccl, fabric_probe=probe)


def main_record(args, ranks: Ranks, m, X, E) -> dict:
    """The JSON line of the main division (the driver's contract + roofline; parity and baselines are added later)."""
    gen, V, _, d, dname, gseed, xseed = WORKLOADS[args.workload]
    eng, world = m["eng"], ranks.world
    result = {
        "metric": "embedding-update iters/sec (Jacobi sweeps of Z <- X + gamma*P*Z, P frozen)",
        "value": m["value"], "unit": "sweeps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": dname, "data": "synthetic",
        "blocks": max(1, args.blocks), "ms_per_step_min": m["ms_per_step_min"], "ms_per_step_max": m["ms_per_step_max"],
        "block_ms_per_step": m["block_ms_per_step"], "ms_per_step_hip_events": m["ms_per_step_hip_events"],
        "timing": f"{args.warmup} warm-up sweeps, then {max(1, args.blocks)} blocks of exactly {args.steps} sweeps, each "
                  f"bracketed by barrier + torch.cuda.synchronize() (host clock, max over ranks) and by a HIP event pair on "
                  f"the sweep's stream (ms_per_step_hip_events); value / ms_per_step = the median block",
        "config": {"workload": f"{GENERATOR_NAMES[gen]} |V|={V} |E|={E} d={d} {dname}, "
                               f"gamma={args.gamma}, CosineSimilarity "
                               f"(reference mode), seeds {gseed}/{xseed}",
                   "parallelism": describe_parallelism(args, world, eng, X, E),
                   "host_sync": (f"pipelined ({m['host_sync']}): the delta of sweep t is read while sweep t+1 runs"
                                 if m["pipelined"] else f"after every sweep ({m['host_sync']}; the reference's order)")},
        "roofline": roofline_block(args, world, m),
        "build_P_ms": m["build_P_ms"], "build_P_cold_ms": m["build_P_cold_ms"],
        "build_P_note": "build_P_ms: as in every outer round of Embedder.iterate() after the first -- the row norms "
                        "(similarity.py:37) are left behind by the pass that measures the round's outer delta "
                        "(embedder.py:60), which reads every row of the new Z anyway; build_P_cold_ms: the norms "
                        "recomputed by row_sqnorm first (the first round, or after set_Z)",
        "last_delta": m["delta"],
    }
    if m["calibration_bytes"] is not None:
        result["calibration"] = {"kernel": "l1_distance_kernel", "bytes_read": m["calibration_bytes"]}
    if ranks.grouped:
        result["comm"] = comm_block(args, ranks, m)
    if ranks.rehearsal:
        result["rehearsal"] = ("ONE rank in a real RCCL group with every collective issued (--rehearse-rccl): the N > 1 "
                               "flow through the real library; not a headline number")
    return result


DEGENERATE_SOFTMAX_NOTE = (
    "reference-mode scores are dot / (||Z[src_all]||_F * ||Z[dst_all]||_F) (similarity.py:35-37: GLOBAL denominators): at "
    "|E| >= 4M they are O(1e-9), exp() of them is exactly 1 in fp32 and P = 1/deg on both sides, so parity_P_rel_l2_vs_oracle "
    "checks the softmax plumbing and the edge order, not K1's dot products.  Those are checked by "
    "parity_P_per_edge_rel_l2_vs_oracle here (the same kernels with per-edge cosine scores, all edges, against the C "
    "oracle) and in the GPU suite by test_k1_scores_at_scale / test_config3_per_edge_P_sampled_rows (raw dots, per-edge "
    "cosine, million-edge rows)")


def check_parity(args, ranks: Ranks, eng, m, csr, X, result, baselines: bool = True):
    """The first GPU sweep (and P itself) against the C oracle, which builds its OWN P; at N = 1 also the CPU
    baselines (the oracle timed on this box's host cores).  Returns (the oracle's first sweep on rank 0, failed) -- the
    verdict is all-reduced, so every rank leaves together."""
    dname = WORKLOADS[args.workload][4]
    Z1_oracle, failed = None, False
    P_gpu = eng.P_global() if eng.row_world == 1 else None      # row splits: each rank holds its rows of P
    if ranks.rank == 0:
        from oracle import baseline as B
        from oracle import clane_oracle as O
        from oracle import clane_oracle_c as OC
        if ranks.world > 1:     # torchrun gives every rank OMP_NUM_THREADS=1; the others are idle in the collective below
            OC.set_threads(os.cpu_count() or 1)
        Z1_oracle, first, _, P_oracle, Xf = B.oracle_first_sweep(csr, X, None, args.gamma)
        parity = O.rel_l2(m["Z1"].float(), Z1_oracle)
        result["parity_rel_l2_vs_oracle_after_1_sweep"] = parity
        result["parity_note"] = ("first sweep from Z = X on the GPU(s), P from the GPU build_P, against "
                                 "oracle/clane_oracle.c running its own build_P and sweep (fp32, on the bf16-rounded "
                                 "inputs where the storage is bf16); " + DEGENERATE_SOFTMAX_NOTE)
        failed = not parity < PARITY_TOL[dname]
        if P_gpu is not None:
            parity_p = O.rel_l2(P_gpu.float(), P_oracle)
            result["parity_P_rel_l2_vs_oracle"] = parity_p
            failed = failed or not parity_p < PARITY_P_TOL[dname]
        if ranks.world == 1 and not ranks.rehearsal and not args.no_cpu_baseline and not failed and baselines:
            result["cpu_baseline"] = B.cpu_baseline(csr, Xf, P_oracle, args.gamma, Z1_oracle, first)
            result["cpu_baseline_torch"] = B.cpu_baseline_torch(csr, Xf, P_oracle, args.gamma)
    return Z1_oracle, ranks.agree_to_fail(failed)


def per_edge_parity(args, ranks: Ranks, eng, csr, X, result) -> bool:
    """A K1 check that is NOT degenerate at full size (one GPU): the same engine scores every edge with the per-edge
    cosine (cosine_mode "per_edge": dot / (|z_src| |z_dst|), the cosine similarity.py's docstring describes) from Z = X
    and all of P is compared with the C oracle's per-edge build_P.  Leaves the engine with a reference-mode P of Z = X."""
    from oracle import clane_oracle as O
    from oracle import clane_oracle_c as OC
    dname = WORKLOADS[args.workload][4]
    eng.set_Z(X)
    eng.set_cosine_mode("per_edge")
    eng.build_P()
    P_pe = eng.P_global().float()
    eng.set_cosine_mode("reference")
    eng.build_P()
    P_or, _ = OC.build_P(csr.rowptr, csr.colidx, X.float(), mode="per_edge")
    err = O.rel_l2(P_pe, P_or)
    deg = np.diff(csr.rowptr)
    hub = int(np.argmax(deg))
    ph = P_or[int(csr.rowptr[hub]):int(csr.rowptr[hub + 1])]
    result["parity_P_per_edge_rel_l2_vs_oracle"] = err
    result["parity_P_per_edge_note"] = (f"all {csr.num_edges} values of P with per-edge cosine scores (not 1/deg: the "
                                        f"heaviest row's P spans a factor {float(ph.max() / ph.min()):.2f}) against "
                                        f"oracle/clane_oracle.c's per-edge build_P")
    return not err < PARITY_P_TOL[dname] * 5         # scores O(0.1): exp and the softmax sums see real arguments


def run_iterate(args, ranks: Ranks, eng, csr, X) -> dict:
    """The WHOLE algorithm from Z = X on the engine just measured: Embedder.iterate() to tolerance."""
    from clane_amd.embedder import Embedder
    from clane_amd.graph import Graph
    from clane_amd.similarity import CosineSimilarity
    g = Graph.from_csr(csr, X)
    eng.set_Z(X)
    g._attach_engine(eng)
    emb = Embedder(g, CosineSimilarity(), ranks.dev, gamma=args.gamma, tolerence=args.tolerence, verbose=False,
                   max_sweeps=2000)
    ranks.barrier()
    t0 = time.perf_counter()
    emb.iterate()
    ranks.barrier()
    wall = time.perf_counter() - t0
    return {"wall_s": wall, "outer_rounds": len(emb.sweep_counts), "sweeps": sum(emb.sweep_counts),
            "sweeps_launched": emb.sweeps_launched,
            "sweeps_per_round": emb.sweep_counts, "tolerence": args.tolerence,
            "last_outer_delta": emb.outer_deltas[-1],
            "note": "Embedder.iterate() from Z = X: build_P + propagate per round, reference "
                    "stopping rule (embedder.py:56-108); sweeps whose delta is provably 0 (after an "
                    "exactly-zero delta with P frozen) are counted, not launched; not part of the "
                    "headline value"}


LEG_SETTINGS = {    # workload: (steps, warmup, blocks, iterate) of the short legs after the headline measurement
    "uniform2m": (20, 5, 3, False), "rmat200k": (200, 20, 3, False), "powerlaw10m": (20, 5, 3, True),
}
LEG_WHAT = {
    "uniform2m": "the roofline's ANCHOR: uniform-random pairs at the headline's |V|, |E|, d -- no hubs, nothing for the L2s "
                 "or the Infinity Cache to reuse (PMC traffic == algorithmic bytes), so its frac is a true HBM fraction",
    "rmat200k": "BASELINE config 2 (R-MAT 200k / 4M / d=128 fp32; the table sits in the Infinity Cache)",
    "powerlaw10m": "BASELINE config 4 (power-law 10M / 200M / d=128, bf16 storage, fp32 accumulate) on ONE GPU, incl. "
                   "Embedder.iterate() from Z = X to `tolerence` convergence -- that run IS configs[4]",
}


def workload_leg(args, ranks: Ranks, name: str) -> dict:
    """One more workload measured in the same process with the same protocol (fewer blocks), parity-checked against the
    C oracle, as a compact block of the main record."""
    import copy
    steps, warmup, blocks, iterate = LEG_SETTINGS[name]
    la = copy.copy(args)
    la.workload, la.steps, la.warmup, la.blocks, la.calibrate, la.column_slice_of = name, steps, warmup, blocks, False, None
    t0 = time.perf_counter()
    csr, X = generate_input(la, ranks)
    m = measure_division(la, ranks, csr, X, "auto", time_kernels=True)
    eng = m["eng"]
    rec = main_record(la, ranks, m, X, csr.num_edges)
    _, failed = check_parity(la, ranks, eng, m, csr, X, rec, baselines=False)
    roof = rec["roofline"]
    out = {"what": LEG_WHAT[name], "workload": rec["config"]["workload"], "value": rec["value"], "unit": rec["unit"],
           "ms_per_step": rec["ms_per_step"], "ms_per_step_min": rec["ms_per_step_min"],
           "ms_per_step_max": rec["ms_per_step_max"], "steps": steps, "warmup": warmup, "blocks": blocks,
           "dtype": rec["dtype"], "host_sync": rec["config"]["host_sync"],
           "build_P_ms": rec["build_P_ms"], "build_P_cold_ms": rec["build_P_cold_ms"],
           "parity_rel_l2_vs_oracle_after_1_sweep": rec.get("parity_rel_l2_vs_oracle_after_1_sweep"),
           "parity_P_rel_l2_vs_oracle": rec.get("parity_P_rel_l2_vs_oracle"),
           "parity_tolerance": {"Z1": PARITY_TOL[rec["dtype"]], "P": PARITY_P_TOL[rec["dtype"]]},
           "roofline": {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                 "traffic_over_algorithmic", "achieved_algorithmic",
                                                 "algorithmic_bytes_per_launch", "avg_launch_ms", "pmc_run_over_live_time",
                                                 "k3_pass")}}
    if roof.get("traffic") is None:
        out["roofline"]["traffic_missing"] = roof.get("kernels", {}).get(roof.get("kernel"), {}).get("traffic_missing")
    if failed:
        out["error"] = "parity check failed"
    elif iterate:
        out["iterate"] = run_iterate(la, ranks, eng, csr, X)
    out["leg_wall_s"] = time.perf_counter() - t0
    return out


DIVISION_NOTES = {
    "allgather_all": "north_star's division: node rows of Z partitioned across the GPUs, every GPU holds the full Z, ONE "
                     "in-place RCCL all-gather of the owned rows per sweep (per launch chunk, overlapped with the next "
                     "chunk's kernels)",
    "allgather": "the same row partition and in-place all-gather, of the rows that CAN change and ARE read only "
                 "(outdeg > 0 and indeg > 0): rows without out-edges are never updated (embedder.py:88-89), rows nobody "
                 "reads are synchronised once at the end",
    "halo": "rows partitioned, each updated row sent only to the ranks that read it: one all_to_all_single per launch "
            "chunk into a compact per-rank table",
}


def division_block(args, ranks: Ranks, csr, X, E, exchange: str, main_division: str, main_value: float, Z1_oracle):
    """One division measured AFTER the main one, as a block of the same record.  Returns (block, parity failed)."""
    dname = WORKLOADS[args.workload][4]
    block = {"exchange": exchange}
    failed = False
    try:
        torch.cuda.empty_cache()
        m2 = measure_division(args, ranks, csr, X, exchange, time_kernels=False)
        e2 = m2.pop("eng")
        block.update({
            "what": DIVISION_NOTES.get(exchange, f"the same graph divided with exchange={exchange}"),
            "value": m2["value"], "unit": "sweeps/s", "ms_per_step": m2["ms_per_step"],
            "ms_per_step_min": m2["ms_per_step_min"], "ms_per_step_max": m2["ms_per_step_max"],
            "ms_per_step_hip_events": m2["ms_per_step_hip_events"], "steps": args.steps,
            "blocks": max(1, args.blocks), "build_P_ms": m2["build_P_ms"], "build_P_cold_ms": m2["build_P_cold_ms"],
            "last_delta": m2["delta"],
            "parallelism": describe_parallelism(args, ranks.world, e2, X, E),
            "host_sync": "pipelined" if m2["pipelined"] else "after every sweep",
            "vs_main_division": m2["value"] / main_value, "main_division": main_division,
            "comm": comm_block(args, ranks, dict(m2, eng=e2))})
        bad = False
        if ranks.rank == 0 and Z1_oracle is not None and m2["Z1"] is not None:
            from oracle import clane_oracle as O
            block["parity_rel_l2_vs_oracle_after_1_sweep"] = O.rel_l2(m2["Z1"].float(), Z1_oracle)
            bad = not block["parity_rel_l2_vs_oracle_after_1_sweep"] < PARITY_TOL[dname]
        del e2, m2
        if ranks.agree_to_fail(bad):
            block["error"] = "parity check failed"
            failed = True
    except Exception as exc:        # noqa: BLE001 -- reported in the record; the main measurement stands
        block["error"] = f"{type(exc).__name__}: {exc}"
    return block, failed


T_START = time.perf_counter()
TIMELINE = {}           # seconds spent per phase of this command (N > 1: reported as time_plan.spent_s)
DRIVER_LIMIT_S = 600.0


def main():
    args = make_parser().parse_args()
    unknown = [x for x in args.also_exchange.split(",") if x not in ("", "none", "columns", "halo", "halo_p2p",
                                                                     "allgather", "allgather_all")]
    if unknown:
        raise SystemExit(f"--also-exchange: unknown division(s) {unknown}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:        # no launcher: be the launcher
        raise SystemExit(launch_ranks(args.gpus))
    if os.environ.get("CLANE_BENCH_WATCHDOG_S"):                # debugging aid: every rank dumps its stack every S seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["CLANE_BENCH_WATCHDOG_S"]), repeat=True, file=sys.stderr)

    import torch.distributed as dist
    n_also = 0 if args.also_exchange == "none" else len([x for x in args.also_exchange.split(",") if x])
    plan = {"driver_limit_s": DRIVER_LIMIT_S, "fabric_probe_s_at_most": 120.0 if args.gpus > 1 and not args.no_fabric_probe else 0.0,
            "generation_and_main_division_s_estimate": 60.0,
            "each_division_after_the_main_one_s_at_most": LITERAL_DEADLINE_S, "divisions_after_the_main_one": n_also if args.gpus > 1 else 0}
    plan["worst_case_s"] = (plan["fabric_probe_s_at_most"] + plan["generation_and_main_division_s_estimate"]
                            + plan["divisions_after_the_main_one"] * LITERAL_DEADLINE_S)
    if args.gpus > 1:
        log(f"time plan: {json.dumps(plan)} -- the main division's record goes to stderr as soon as it is measured and is "
            f"the ONE stdout line whatever happens to the divisions after it")
    ranks = start_ranks(args)
    rank = ranks.rank
    V, d = WORKLOADS[args.workload][1], WORKLOADS[args.workload][3]
    t0 = time.perf_counter()
    csr, X = generate_input(args, ranks)
    E = csr.num_edges
    TIMELINE["generation_s"] = time.perf_counter() - t0
    log(f"{args.workload}: |V|={V} |E|={E} d={d} max outdeg={int(np.diff(csr.rowptr).max())} "
        f"generated in {time.perf_counter() - t0:.1f}s")

    t0 = time.perf_counter()
    m = measure_division(args, ranks, csr, X, args.exchange, time_kernels=True)
    TIMELINE["main_division_s"] = time.perf_counter() - t0
    eng = m["eng"]
    result = main_record(args, ranks, m, X, E)
    if ranks.grouped:
        result["time_plan"] = dict(plan, spent_s=TIMELINE)

    Z1_oracle, failed = None, False
    if not args.no_parity:
        Z1_oracle, failed = check_parity(args, ranks, eng, m, csr, X, result)
        if failed:                  # every rank leaves, non-zero, together
            if rank == 0:
                print(json.dumps(result), flush=True)
            if ranks.grouped:
                dist.destroy_process_group()
            raise SystemExit(f"parity check failed: {result.get('parity_rel_l2_vs_oracle_after_1_sweep')} "
                             f"(P: {result.get('parity_P_rel_l2_vs_oracle')})")
    if not args.no_parity and not ranks.grouped and not args.column_slice_of and WORKLOADS[args.workload][4] != "f64":
        if per_edge_parity(args, ranks, eng, csr, X, result):
            print(json.dumps(result), flush=True)
            raise SystemExit(f"per-edge parity check of P failed: {result['parity_P_per_edge_rel_l2_vs_oracle']}")
    if args.iterate:
        result["iterate"] = run_iterate(args, ranks, eng, csr, X)

    # ---- further workloads in the same record (one GPU): the roofline's anchor, BASELINE configs 2 and 4 --------
    legs = [] if (ranks.grouped or args.column_slice_of or args.legs == "none") else \
        (list(LEG_SETTINGS) if args.workload == "rmat2m" else []) if args.legs == "auto" else \
        [x for x in args.legs.split(",") if x]
    if legs:
        log("main workload measured; the record so far (the ONE stdout line follows after the legs): " + json.dumps(result))
        del m["eng"]
        eng = None
        estimate = {"uniform2m": 25.0, "rmat200k": 10.0, "powerlaw10m": 70.0}
        result["legs_plan"] = {"budget_s": args.budget_s, "estimate_s": {n: estimate.get(n, 60.0) for n in legs},
                               "used_before_legs_s": time.perf_counter() - T_START}
        for name in legs:
            used = time.perf_counter() - T_START
            if name not in LEG_SETTINGS:
                block = {"error": f"no leg settings for workload {name}"}
            elif used + estimate.get(name, 60.0) > args.budget_s:
                block = {"skipped": f"{used:.0f} s used, estimate {estimate.get(name, 60.0):.0f} s: over the {args.budget_s:.0f} s plan"}
            else:
                try:
                    torch.cuda.empty_cache()
                    block = workload_leg(args, ranks, name)
                    log(f"leg {name}: {block['value']:.1f} sweeps/s, frac {block['roofline']['frac']:.3f}, "
                        f"{block['leg_wall_s']:.1f} s")
                except Exception as exc:    # noqa: BLE001 -- reported in the record; the main measurement stands
                    block = {"error": f"{type(exc).__name__}: {exc}"}
            if name == "uniform2m":
                result["roofline"]["anchor"] = block
            else:
                result.setdefault("other_workloads", {})[name] = block
        result["legs_plan"]["used_s"] = time.perf_counter() - T_START
        blocks_ = [result["roofline"].get("anchor", {})] + list(result.get("other_workloads", {}).values())
        if any(b_.get("error") == "parity check failed" for b_ in blocks_):     # loud: a wrong result is not a measurement
            print(json.dumps(result), flush=True)
            raise SystemExit("a workload measured after the main one failed its parity check (see its block in the record)")

    # ---- north_star's literal division beside the default one, in the same record -----------------------------
    printed = threading.Event()
    lock = threading.Lock()
    state = {"failed": False}

    def emit(extra=None):
        """Print the record once (rank 0).  `result` is only ever written under `lock`, so the copy that is dumped --
        by the main thread here, or by the deadline thread -- is never a half-written one."""
        with lock:
            if not printed.is_set():
                printed.set()
                if rank == 0:
                    print(json.dumps(dict(result, **(extra or {}))), flush=True)

    also = [] if (args.also_exchange == "none" or not ranks.grouped or args.column_slice_of) else \
        [x for x in args.also_exchange.split(",") if x and x != eng.exchange
         and not (ranks.rehearsal and x not in ("columns", "allgather", "allgather_all"))]   # what ONE rank can be made to issue
    if also:
        # Whatever happens in here, the main record above must come out: past the deadline (per division) every rank
        # prints / leaves on its own (a rank stuck in a collective cannot be talked to).
        def give_up():
            # the main record still comes out, but the process leaves NON-ZERO: a hang is a failure the launcher and
            # the driver must see (1 if a parity check had already failed, else 3); never re-exec, never retry
            extra = {"also_exchange_error": (f"a division measured after the main one gave no result within "
                                             f"{LITERAL_DEADLINE_S:.0f} s: printed as far as it got")}
            if "north_star_literal" not in result:
                extra["north_star_literal"] = {"exchange": "allgather_all"}
            emit(extra)
            os._exit(1 if state["failed"] else 3)
        # a safety copy for the logs: should a later division take the process down (a fault, an abort inside the
        # library), the measurement of the main division has been seen
        log("main division measured; the record so far (the ONE stdout line follows after the other divisions): "
            + json.dumps(result))
        main_division, main_value = eng.exchange, m["value"]
        del m["eng"]
        eng = None                          # the first engine's tables go back to the allocator before the next is built
        timer = None
        for exchange in also:
            if timer is not None:
                timer.cancel()
            timer = threading.Timer(LITERAL_DEADLINE_S, give_up)
            timer.daemon = True
            timer.start()
            t_div = time.perf_counter()
            block, bad = division_block(args, ranks, csr, X, E, exchange, main_division, main_value, Z1_oracle)
            TIMELINE[f"division_{exchange}_s"] = time.perf_counter() - t_div
            state["failed"] = state["failed"] or bad
            with lock:
                if exchange == "allgather_all":
                    result["north_star_literal"] = block
                else:
                    result.setdefault("other_divisions", {})[exchange] = block
        if timer is not None:               # every block is stored: the deadline of the last division is over
            timer.cancel()
    TIMELINE["total_s"] = time.perf_counter() - T_START
    emit()
    if ranks.grouped:
        # leave together.  A rank that fell out of a block above on its own (an exception the others did not have)
        # must not wait for ever for ranks stuck in a collective: the record is out, so past this deadline just leave,
        # non-zero.
        last = threading.Timer(LITERAL_DEADLINE_S, lambda: os._exit(1 if state["failed"] else 3))
        last.daemon = True
        last.start()
        dist.barrier()
        last.cancel()
        dist.destroy_process_group()
    if state["failed"]:
        raise SystemExit("a division measured after the main one failed its parity check")


if __name__ == "__main__":
    main()

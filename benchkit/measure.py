"""The timed protocol: warm-up, blocks of exactly K sweeps between barriers, HIP events per kernel."""
from __future__ import annotations

import statistics
import time

import torch

from .common import log
from .ranks import Ranks


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

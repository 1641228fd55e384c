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
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from benchkit import common  # noqa: E402
from benchkit.common import DRIVER_LIMIT_S, DTYPES, LITERAL_DEADLINE_S, TIMELINE, T_START, WORKLOADS, log  # noqa: E402,F401
from benchkit.legs import LEG_SETTINGS, division_block, workload_leg  # noqa: E402
from benchkit.measure import measure_division, run_iterate  # noqa: E402
from benchkit.parity import check_parity, per_edge_parity  # noqa: E402
from benchkit.ranks import launch_ranks, start_ranks, generate_input  # noqa: E402
from benchkit.record import main_record, traffic_entry  # noqa: E402,F401


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
            "generation_and_main_division_s_estimate": 60.0, "divisions_after_the_main_one": n_also if args.gpus > 1 else 0}
    # every division after the main one gets the same share of what the driver's limit leaves (at most LITERAL_DEADLINE_S)
    division_deadline = min(LITERAL_DEADLINE_S, (DRIVER_LIMIT_S - 30.0 - plan["fabric_probe_s_at_most"]
                                                 - plan["generation_and_main_division_s_estimate"]) / max(1, n_also))
    plan["each_division_after_the_main_one_s_at_most"] = division_deadline
    plan["worst_case_s"] = (plan["fabric_probe_s_at_most"] + plan["generation_and_main_division_s_estimate"]
                            + plan["divisions_after_the_main_one"] * division_deadline)
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
                                             f"{division_deadline:.0f} s: printed as far as it got")}
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
            timer = threading.Timer(division_deadline, give_up)
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
        last = threading.Timer(division_deadline, lambda: os._exit(1 if state["failed"] else 3))
        last.daemon = True
        last.start()
        dist.barrier()
        last.cancel()
        dist.destroy_process_group()
    if state["failed"]:
        raise SystemExit("a division measured after the main one failed its parity check")


if __name__ == "__main__":
    main()

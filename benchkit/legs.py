"""More blocks of the same record: further workloads on one GPU (the roofline's anchor, BASELINE configs 2 and 4),
further divisions at N > 1 (north_star's literal plan beside the main one)."""
from __future__ import annotations

import time

import torch

from .common import PARITY_P_TOL, PARITY_TOL, WORKLOADS
from .measure import measure_division, run_iterate
from .parity import check_parity, per_edge_parity
from .ranks import Ranks, generate_input
from .record import comm_block, describe_parallelism, main_record


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
    if not failed and rec["dtype"] != "f64":        # all of P once more with per-edge cosine scores (not 1/deg)
        failed = per_edge_parity(la, ranks, eng, csr, X, rec)
    roof = rec["roofline"]
    out = {"what": LEG_WHAT[name], "workload": rec["config"]["workload"], "value": rec["value"], "unit": rec["unit"],
           "ms_per_step": rec["ms_per_step"], "ms_per_step_min": rec["ms_per_step_min"],
           "ms_per_step_max": rec["ms_per_step_max"], "steps": steps, "warmup": warmup, "blocks": blocks,
           "dtype": rec["dtype"], "host_sync": rec["config"]["host_sync"],
           "build_P_ms": rec["build_P_ms"], "build_P_cold_ms": rec["build_P_cold_ms"],
           "parity_rel_l2_vs_oracle_after_1_sweep": rec.get("parity_rel_l2_vs_oracle_after_1_sweep"),
           "parity_P_rel_l2_vs_oracle": rec.get("parity_P_rel_l2_vs_oracle"),
           "parity_P_per_edge_rel_l2_vs_oracle": rec.get("parity_P_per_edge_rel_l2_vs_oracle"),
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

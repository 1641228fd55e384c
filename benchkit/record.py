"""The JSON record: the driver's contract, the roofline of the dominant kernel, what an N > 1 run really was."""
from __future__ import annotations

import json
import os

import torch

from . import common
from .common import GENERATOR_NAMES, HBM_PEAK_GBPS, TRAFFIC_NOTE, WORKLOADS
from .ranks import Ranks


def traffic_entry(workload: str, world: int, eng, dom: str, slice_of=None):
    """PMC traffic of kernel `dom` from profiles/traffic.json -- only if it was measured with THIS kernel
    configuration (thresholds, launch blocks, compile-time tuning, ...); else (None, why)."""
    tfile = common.ROOT / "profiles" / "traffic.json"
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
    older = {"column_tiles": 1, "fewer_loads_in_flight": False}     # entries from before these keys existed
    then, live = dict(older, **then), dict(older, **live)
    diff = sorted(k for k in set(live) | set(then) if live.get(k) != then.get(k) and k != "exchange")
    if diff:
        return None, f"{key}: stale, measured with another kernel configuration (differs in {', '.join(diff)})"
    got = entry.get(dom, {}).get("bytes_per_launch")
    if got is None:
        return None, f"{key}: kernel {dom} not in the measurement"
    return got, {"source": entry.get("source"), "avg_us_under_pmc": entry[dom].get("avg_us_under_pmc")}


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

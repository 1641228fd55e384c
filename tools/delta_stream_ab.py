#!/usr/bin/env python3
"""What does the per-sweep scalar all-reduce cost a short sweep, and does moving it off the sweep's stream pay?
One rank's column slice of config 3 (d = 32 / 64: the N = 8 / 4 shares) with a ONE-rank RCCL group whose collectives are
really issued (TorchComm(force_collectives=True)): 300 sweeps launched one ahead of the host check, with the all-reduce
on the sweep's stream (round-2 form) and on a stream of its own.  One rank has no xGMI hop: real groups pay more per
all-reduce than shown here.   Usage: tools/delta_stream_ab.py"""
import json, socket, sys, time
from pathlib import Path
import torch
import torch.distributed as dist
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip, synth
from clane_amd.comm import TorchComm
from clane_amd.engine import SweepEngine
import bench

dev = _hip.require_gpu("cuda:0")
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS["rmat2m"]
csr = synth.rmat_csr(V, E, seed=gseed, device=str(dev))
Xfull = synth.gaussian_X(V, d, seed=xseed)
for cols in (32, 64):
    X = Xfull[:, :cols].contiguous()
    for variant in ("no collective", "all-reduce on the sweep's stream", "all-reduce on its own stream"):
        comm = None if variant == "no collective" else TorchComm(dist.group.WORLD, force_collectives=True)
        eng = SweepEngine(csr, X, dev, comm=comm, exchange="columns" if comm else "auto",
                          delta_stream=variant.endswith("own stream"))
        eng.build_P()
        deltas = []
        for rep in range(3):
            eng.set_Z(X)
            eng.build_P()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ticket = eng.sweep_launch(0.76)
            for _ in range(299):
                nxt = eng.sweep_launch(0.76)
                deltas.append(eng.sweep_wait(ticket))
                ticket = nxt
            deltas.append(eng.sweep_wait(ticket))
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 300 * 1e3
            if rep == 0:
                first = list(deltas)
            deltas = []
            best = ms if rep == 0 else min(best, ms)
        print(json.dumps({"d": cols, "variant": variant, "ms_per_sweep": round(best, 4), "delta_10": first[10],
                          "collective_calls": dict(comm.calls) if comm else None}), flush=True)
        del eng
dist.destroy_process_group()

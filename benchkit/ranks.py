"""One process per GPU: the launcher for `python bench.py --gpus N`, the process group, the synthetic input."""
from __future__ import annotations

import datetime
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

from .common import BENCH_SCRIPT, DTYPES, ROOT, TIMELINE, WORKLOADS, log


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) and hand back
    their exit code.  Nothing in THIS process has initialised the GPU (importing torch does not), and nothing is
    exec'd: the ranks are children, rank 0 writes the JSON line to the inherited stdout."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(BENCH_SCRIPT)] + sys.argv[1:]
    log("starting ranks: " + " ".join(cmd))
    return subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode


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

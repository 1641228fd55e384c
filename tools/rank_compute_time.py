#!/usr/bin/env python3
"""One rank's share of a W-GPU sweep, timed alone on one GPU: the engine is built for (world W, rank 0) with a
comm whose collectives do nothing, so only the kernels (and the send-buffer packing) run.  Halo rows go stale,
which does not change the memory traffic.  Usage: tools/rank_compute_time.py [--workload rmat2m] [--world 8]"""
import argparse, json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip, synth
from clane_amd.engine import SweepEngine
import bench


class NullComm:
    force = False                   # clane_amd.comm's interface: not a one-rank group insisting on its collectives
    def __init__(self, world, rank=0):
        self.world, self.rank = world, rank
    def all_reduce_sum(self, t): pass
    def all_gather_into(self, out, inp, async_op=False): return bench_done
    def all_to_all_rows(self, out, inp, o, i, async_op=False): return bench_done
    def split(self, groups):            # 2-D division: the sub-group this rank is in, as silent as the whole
        mine = next(g for g in groups if self.rank in g)
        return NullComm(len(mine), list(mine).index(self.rank))
    def all_gather_object(self, obj):
        return self.layouts if self.layouts is not None else [obj] * self.world
    layouts = None
    def share_matrices(self, kernels, mine):
        """halo_p2p timed on one GPU: the other ranks' tables are stand-ins in LOCAL memory of the sizes those
        ranks would have, so the kernels issue exactly the stores they would send over xGMI."""
        views = []
        for q in range(self.world):
            if q == self.rank:
                views.append([b.tensor for b in mine])
            else:
                views.append([torch.zeros(self.table_rows[q], mine[0].shape[1], dtype=mine[0].dtype,
                                          device=mine[0].tensor.device) for _ in range(len(mine))])
        return views, []


class _Done:
    def wait(self): return True


bench_done = _Done()
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rmat2m")
ap.add_argument("--world", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--exchange", default="halo")
ap.add_argument("--chunks", type=int, default=4)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--natural-order", action="store_true")
ap.add_argument("--no-split-hubs", action="store_true")
ap.add_argument("--no-overlap", action="store_true")
ap.add_argument("--no-fused-pack", action="store_true")
ap.add_argument("--class-threshold", type=int, default=None, help="0 = no XCD-affine class pass")
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[args.workload]
csr = synth.rmat_csr(V, E, seed=gseed) if gen == "rmat" else synth.powerlaw_csr(V, E, seed=gseed)
X = synth.gaussian_X(V, d, seed=xseed).to(bench.DTYPES[dname])
for W in args.world:
    comm = NullComm(W)
    if args.exchange == "halo_p2p":       # the layouts the other ranks would publish
        from clane_amd.halo import build_halo_layout
        lays = [build_halo_layout(csr, W, q, args.chunks, hot_rows_first=not args.natural_order) for q in range(W)]
        comm.layouts = [[(b.exchange.recv_start, list(b.exchange.out_splits)) for b in lay.blocks] for lay in lays]
        comm.table_rows = [lay.table_rows for lay in lays]
        del lays
    eng = SweepEngine(csr, X, dev, comm=comm, chunks=args.chunks, exchange=args.exchange,
                      hot_rows_first=not args.natural_order, split_hubs=not args.no_split_hubs, overlap_chunks=not args.no_overlap,
                      fused_pack=not args.no_fused_pack, class_threshold=args.class_threshold)
    eng.build_P()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.build_P()
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(5):
        eng.sweep(0.76)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.sweep(0.76)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    eng.time_kernels = True
    for _ in range(5):
        eng.sweep(0.76)
    torch.cuda.synchronize()
    kt = {k: round(v, 3) for k, v in eng.kernel_times_ms().items()}
    eng.time_kernels = False
    n_class = sum(0 if c is None else c[0].numel() for c in eng.class_rows)
    n_items = sum(0 if c is None else int((c[3] > 0).sum()) for c in eng.class_rows)
    print(json.dumps({"world": W, "kernels_ms_summed_over_chunks": kt, "class_rows": n_class, "class_items": n_items,
                      "thresholds": [eng.long_threshold, eng.class_threshold],
                      "items_per_workgroup": [c[6] for c in eng.class_rows if c is not None][:1], "exchange": args.exchange, "rank0_compute_ms_per_sweep": round(ms, 3),
                      "rank0_build_P_ms_without_collectives": round(build_ms, 3),
                      "recv_MB_per_sweep": round(eng.exchange_bytes_per_sweep() / 1e6), "table_rows": eng.part.padded_vertices, "d_local": eng.d,
                      "n_local": eng.part.n_local, "E_loc": eng.E_loc, "hot_rows_first": not args.natural_order,
                      "fused_pack": eng.fused_pack, "segment_edges": eng.segment_edges}), flush=True)
    del eng
    torch.cuda.empty_cache()

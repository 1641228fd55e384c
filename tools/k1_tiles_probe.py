#!/usr/bin/env python3
"""Would K1 (edge scores) gain from the column tiles the sweeps use?  Measured with what exists: per tile the RAW_DOT
form of clane_edge_score_* / clane_edge_score_class_* over a strided view (what a column-split rank runs), against the
one fused pass build_P runs today; plus what finishing the tiled form costs with today's entry points (add, finalize,
segmented softmax).  The dots of the tiles summed are compared with the one-pass dots.
Usage: tools/k1_tiles_probe.py [--workload rmat2m] [--tiles 2] [--rounds 5]"""
import argparse
import json
import statistics
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from clane_amd import _hip, synth  # noqa: E402
from clane_amd.engine import SweepEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="rmat2m")
ap.add_argument("--tiles", type=int, default=2)
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[args.workload]
dev = _hip.require_gpu("cuda:0")
csr = {"rmat": synth.rmat_csr, "powerlaw": synth.powerlaw_csr, "uniform": synth.uniform_random_csr}[gen](V, E, seed=gseed, device=str(dev))
X = synth.gaussian_X(V, d, seed=xseed).to(bench.DTYPES[dname])
eng = SweepEngine(csr, X, dev)
k = eng.k
eng.build_P()
for _ in range(3):
    eng.sweep(0.76)
eng.build_P()                       # norms / sums2 of the current table are in place after this
torch.cuda.synchronize()
Z, mode = eng.Zcur, _hip.SCORE_MODES[eng.cosine_mode]
w = d // args.tiles
cuts = [(t * w, (t + 1) * w) for t in range(args.tiles)]
tmp = torch.empty_like(eng.P)
raw_one = torch.empty_like(eng.P)


def k1(view, width, mode_, out, fused):
    for i, b in enumerate(eng.blocks):
        rp = eng.rowptr[b.local_start:]
        k.edge_score(rp, eng.colidx, b.nrows, b.row0, view, width, mode_, eng.sums2 if fused else None, None, out,
                     eng.k1_threshold, eng.k1_long_rows[i], fuse_softmax=fused)
        if eng.class_k1 and eng.class_rows[i] is not None:
            rows_c, slot_ptr, it_e0, it_len, it_slot, it_row, ipb = eng.class_rows[i]
            if fused:
                k.edge_score_class(rp, eng.colidx, it_e0, it_len, it_slot, it_row, ipb, rows_c, slot_ptr, b.row0, view, width,
                                   mode_, eng.sums2, None, out, eng.slabs[i % len(eng.slabs)], fuse_softmax=True,
                                   n_slots=eng.class_slots[i], row_parts=eng.softmax_row_parts)
            else:
                k.edge_score_class(rp, eng.colidx, it_e0, it_len, it_slot, it_row, ipb, rows_c, slot_ptr, b.row0, view, width,
                                   mode_, None, None, out)


def finish(out):
    for i, b in enumerate(eng.blocks):
        rp = eng.rowptr[b.local_start:]
        k.edge_score_finalize(rp, eng.colidx, b.nrows, b.row0, mode, eng.sums2, None, out)
        k.segment_softmax(rp, b.nrows, out, 0, eng.score_threshold if eng.long_rows[i] is not None else 0, eng.long_rows[i])


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


def tiled_raw():
    for t, (c0, c1) in enumerate(cuts):
        k1(Z[:, c0:c1], c1 - c0, _hip.SCORE_RAW_DOT, eng.P if t == 0 else tmp, False)
        if t:
            eng.P.add_(tmp)


res = {"fused_one_pass": [], "raw_one_pass": [], "raw_tiles": [], "raw_tiles_each": [], "finish": []}
assert eng.cosine_mode == "reference"
for _ in range(args.rounds):
    res["fused_one_pass"].append(timed(lambda: k1(Z, d, mode, eng.P, True)))
    res["raw_one_pass"].append(timed(lambda: k1(Z, d, _hip.SCORE_RAW_DOT, raw_one, False)))
    each = [timed(lambda c0=c0, c1=c1: k1(Z[:, c0:c1], c1 - c0, _hip.SCORE_RAW_DOT, tmp, False)) for c0, c1 in cuts]
    res["raw_tiles_each"].append(each)
    res["raw_tiles"].append(timed(tiled_raw))
    rel = float((eng.P - raw_one).norm() / raw_one.norm())
    res["finish"].append(timed(lambda: finish(eng.P)))
P_tiled = eng.P.clone()
k1(Z, d, mode, eng.P, True)
torch.cuda.synchronize()
out = {"workload": args.workload, "device": torch.cuda.get_device_name(dev), "tiles": cuts,
       "ms": {n: statistics.median(v) for n, v in res.items() if n != "raw_tiles_each"},
       "raw_tiles_each_ms": [statistics.median(x[t] for x in res["raw_tiles_each"]) for t in range(len(cuts))],
       "all": {n: [([round(y, 4) for y in x] if isinstance(x, list) else round(x, 4)) for x in v] for n, v in res.items()},
       "tiled_dots_vs_one_pass_rel_l2": rel,
       "P_tiled_vs_P_rel_l2": float((P_tiled - eng.P).norm() / eng.P.norm()),
       "note": "raw_tiles includes the add of the second tile's dots (torch); a fused form would add in the kernel and "
               "finish in the last tile's pass, i.e. cost about raw_tiles - add + (fused_one_pass - raw_one_pass)"}
print(json.dumps(out))

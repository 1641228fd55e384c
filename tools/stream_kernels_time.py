#!/usr/bin/env python3
"""Achieved GB/s of the two pure streaming kernels -- row_sqnorm (K0) and l1_distance (outer delta + norms) -- at the
BASELINE shapes, HIP events over 20 launches.  Usage: tools/stream_kernels_time.py [--out file.jsonl]"""
import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from clane_amd import _hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
k = _hip.kernels()
rows = []
for V, d, dtype in ((2_000_000, 256, torch.float32), (10_000_000, 128, torch.bfloat16), (200_000, 128, torch.float32),
                    (2_000_000, 1433 // 4 * 4, torch.float32)):
    if V * d * 4 > 8e9:
        V = V // 4
    A = torch.randn(V, d, device=dev).to(dtype)
    B = torch.randn(V, d, device=dev).to(dtype)
    acc = _hip.acc_dtype(dtype)
    sq = torch.zeros(V, dtype=acc, device=dev)
    ws = torch.zeros(k.reduce_ws_len(), dtype=torch.float64, device=dev)
    out = torch.zeros(1, dtype=torch.float64, device=dev)
    es = A.element_size()
    for name, fn, nbytes in (("row_sqnorm", lambda: k.row_sqnorm(A, d, sq), V * d * es + V * 4),
                             ("l1_distance+sq", lambda: k.l1_distance(A, B, d, ws, out, sq_a=sq), 2 * V * d * es + V * 4),
                             ("l1_distance", lambda: k.l1_distance(A, B, d, ws, out), 2 * V * d * es)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        rows.append({"kernel": name, "V": V, "d": d, "dtype": str(dtype).replace("torch.", ""), "ms": round(ms, 4),
                     "GBps": round(nbytes / ms / 1e6, 1)})
        print(rows[-1], flush=True)
    del A, B
rec = {"build": k.build_info(), "rows": rows}
if args.out:
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    with open(args.out, "a") as f:
        f.write(json.dumps(rec) + "\n")

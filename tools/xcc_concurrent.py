#!/usr/bin/env python3
"""Does the workgroup -> XCD dealing stay round-robin WITHIN a launch when two launches run at the same time on two
streams?  (The class-affine kernels get their speed from it; row-split engines overlap chunks on two streams.)
Prints, per launch, the share of workgroups on XCD (w + c) % 8 for the best c.  Usage: tools/xcc_concurrent.py"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip

dev = _hip.require_gpu("cuda:0")
k = _hip.kernels()
a, b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def share(ids):
    n = ids.numel()
    w = torch.arange(n, device=ids.device, dtype=torch.int32)
    return max(float(((w + c) % 8 == ids).float().mean()) for c in range(8))


for n, threads in ((200_000, 256), (1_000_000, 64), (2_000_000, 1024)):
    k.xcc_ids(1024, threads, dev)
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):
        with torch.cuda.stream(a):
            x = k.xcc_ids(n, threads, dev)
        with torch.cuda.stream(b):
            y = k.xcc_ids(n, threads, dev)
        outs.append((x, y))
    torch.cuda.synchronize()
    alone = k.xcc_ids(n, threads, dev)
    torch.cuda.synchronize()
    print(f"{n} workgroups x {threads} threads: alone {share(alone):.4f}; two streams at once "
          + " ".join(f"{share(x):.4f}/{share(y):.4f}" for x, y in outs), flush=True)

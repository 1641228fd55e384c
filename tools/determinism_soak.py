#!/usr/bin/env python3
"""Race detector of last resort (there is no GPU sanitizer on this pool): the same sweeps, several times over, at the
bench workloads' full sizes, must give the same bits -- every delta and the final embeddings.  A missing fence or a
mis-ordered LDS ticket in the dynamic row / chunk dispatch shows up here as a flipped low bit sooner or later.
Usage: tools/determinism_soak.py [--workload rmat2m rmat200k powerlaw10m] [--sweeps 60] [--repeats 4]"""
import argparse, hashlib, json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clane_amd import _hip, synth
from clane_amd.engine import SweepEngine
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", nargs="+", default=["rmat200k", "rmat2m", "powerlaw10m"])
ap.add_argument("--sweeps", type=int, default=60)
ap.add_argument("--repeats", type=int, default=4)
ap.add_argument("--cols", type=int, default=None, help="only the first COLS columns of X (a column slice: narrower-row kernels)")
args = ap.parse_args()
dev = _hip.require_gpu("cuda:0")
ok = True
for w in args.workload:
    gen, V, E, d, dname, gseed, xseed = bench.WORKLOADS[w]
    csr = synth.rmat_csr(V, E, seed=gseed, device=str(dev)) if gen == "rmat" else synth.powerlaw_csr(V, E, seed=gseed, device=str(dev))
    X = synth.gaussian_X(V, d, seed=xseed).to(bench.DTYPES[dname])
    if args.cols:
        X = X[:, :args.cols].contiguous()
    eng = SweepEngine(csr, X, dev)
    runs = []
    t0 = time.perf_counter()
    for rep in range(args.repeats):
        eng.set_Z(X)
        eng.build_P()
        p_hash = hashlib.sha256(eng.P.cpu().numpy().tobytes()).hexdigest()[:16]
        deltas = [eng.sweep(0.76) for _ in range(args.sweeps)]
        z = eng.Zcur
        z_hash = hashlib.sha256(z.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]
        runs.append((p_hash, [float.hex(x) for x in deltas], z_hash))
    same = all(r == runs[0] for r in runs)
    ok = ok and same
    print(json.dumps({"workload": w, "dtype": dname, "d": int(X.shape[1]), "sweeps": args.sweeps, "repeats": args.repeats, "bit_identical": same,
                      "P_sha": runs[0][0], "Z_sha": runs[0][2], "last_delta": float.fromhex(runs[0][1][-1]),
                      "seconds": round(time.perf_counter() - t0, 1)}), flush=True)
    del eng
    torch.cuda.empty_cache()
sys.exit(0 if ok else 1)

#!/usr/bin/env python3
"""Run bench.py once per library variant (build/variants/libclane_hip_*.so) on the GPU box and
print one table.  Each run is its own process; order is interleaved over `--rounds`."""
import argparse, glob, json, os, subprocess, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--workload", default="rmat2m")
ap.add_argument("--only", default="")
args = ap.parse_args()
libs = sorted(glob.glob(str(ROOT / "build" / "variants" / "libclane_hip_*.so")))
if args.only:
    libs = [l for l in libs if any(o in l for o in args.only.split(","))]
res = {}
for rnd in range(args.rounds):
    for lib in libs:
        name = Path(lib).stem.replace("libclane_hip_", "")
        env = dict(os.environ, CLANE_HIP_LIB=lib)
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", "--legs", "none", "--steps", str(args.steps),
                              "--warmup", "5", "--workload", args.workload], env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            r = j["roofline"]
            main = r["kernels"].get("spmm_update_kernel", {}).get("avg_launch_ms", 0.0)
            res.setdefault(name, []).append((j["ms_per_step"], main, r["k3_pass"]["ms"], j["build_P_ms"]))
            print(f"round {rnd} {name:24s} step {j['ms_per_step']:.3f} ms  main {main:.3f}  "
                  f"pass {r['k3_pass']['ms']:.3f}  build_P {j['build_P_ms']:.1f}", flush=True)
        except Exception as e:
            print(f"round {rnd} {name}: FAILED {e}\n{out.stderr[-800:]}", flush=True)
print("\n| variant | step ms (min) | main kernel ms (min) | K3 pass ms (min) | build_P ms (min) |\n|---|---|---|---|---|")
for name, v in sorted(res.items(), key=lambda kv: min(x[1] for x in kv[1])):
    print(f"| {name} | {min(x[0] for x in v):.3f} | {min(x[1] for x in v):.3f} | {min(x[2] for x in v):.3f} | "
          f"{min(x[3] for x in v):.1f} |")

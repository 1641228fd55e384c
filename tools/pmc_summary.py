#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into profiles/: per-kernel time (kernel-trace stats) and HBM-side
traffic from the PMC passes (FETCH_SIZE / WRITE_SIZE collected in SEPARATE runs, KiB units).

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of a
16-B-per-lane coalesced read.  The factor is calibrated, not assumed: bench.py --calibrate
launches l1_distance_kernel over two [V, ld] matrices whose byte count is known.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE \
        --known-bytes 4096000000 --tag r02 --workload rmat2m_n1 --bench-json gpurun_out/pmc_fetch.json \
        [--stats gpurun_out/prof1]
"""
import argparse
import collections
import csv
import glob
import json
from pathlib import Path


def _newest(pattern):
    import os
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)   # gpurun merges runs into one directory


def per_kernel(dirname):
    f = _newest(f"{dirname}/**/*counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "clane::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return {k: (sum(a for a, _ in v) / len(v), sum(t for _, t in v) / len(v) / 1e3, len(v)) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--known-bytes", type=float, default=None,
                    help="bytes read by l1_distance_kernel (default: calibration.bytes_read of --bench-json)")
    ap.add_argument("--out", default=None, help="directory to write into (default: profiles/ of this checkout)")
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--workload", default="rmat2m_n1")
    ap.add_argument("--stats", default=None)
    ap.add_argument("--bench-json", default=None,
                    help="the JSON line bench.py printed in the PMC run: its roofline.kernel_config is stored with the "
                         "entry, and bench.py only quotes the traffic for a run with the same configuration")
    args = ap.parse_args()
    repo_profiles = Path(__file__).resolve().parent.parent / "profiles"
    out = Path(args.out) if args.out else repo_profiles
    out.mkdir(parents=True, exist_ok=True)
    bench_line = None
    if args.bench_json:
        bench_line = json.loads([ln for ln in Path(args.bench_json).read_text().splitlines() if ln.startswith("{")][-1])
        if args.known_bytes is None:
            args.known_bytes = float(bench_line["calibration"]["bytes_read"])
    if args.known_bytes is None:
        raise SystemExit("--known-bytes or a --bench-json of a `bench.py --calibrate` run is needed")
    fetch, write = per_kernel(args.fetch_dir), per_kernel(args.write_dir)
    cal = [k for k in fetch if "l1_distance_kernel" in k][0]
    factor = args.known_bytes / (fetch[cal][0] * 1024)
    lines = [f"# HBM-side traffic per launch, {args.workload} ({args.tag})", "",
             f"FETCH_SIZE calibration on `{cal}`: known {args.known_bytes:.4g} B / counted "
             f"{fetch[cal][0] * 1024:.4g} B = **x{factor:.4f}** (guide: exactly 2 for 16 B/lane reads). "
             "WRITE_SIZE is taken as exact.", "",
             "| kernel | launches | avg us (under PMC) | fetch GB (corrected) | write GB | total GB | GB/s |",
             "|---|---|---|---|---|---|---|"]
    summary = {}
    for k in fetch:
        fb = fetch[k][0] * 1024 * factor
        wb = write.get(k, (0, 0, 0))[0] * 1024
        us = fetch[k][1]
        lines.append(f"| `{k}` | {fetch[k][2]} | {us:.1f} | {fb / 1e9:.3f} | {wb / 1e9:.3f} | {(fb + wb) / 1e9:.3f} | "
                     f"{(fb + wb) / us / 1e3:.0f} |")
        summary[k] = {"fetch_bytes": fb, "write_bytes": wb, "avg_us_under_pmc": us}
    (out / f"{args.tag}_pmc_traffic_{args.workload}.md").write_text("\n".join(lines) + "\n")
    def bench_name(k):       # names used by bench.py's roofline.kernels
        if "spmm_update_subrow_kernel" in k:
            return "spmm_update_subrow_kernel"
        if "spmm_update_kernel" in k:
            return "spmm_update_kernel"
        if "spmm_split_segment_kernel" in k:
            return "spmm_split_segment_kernel+combine"
        if "spmm_class_chunk_kernel" in k:
            return "spmm_class_chunk_kernel+combine"
        if "spmm_long_kernel" in k:
            return f"spmm_long_kernel<{k.rstrip('>').split(',')[-1].strip()} waves>"
        return None

    tfile = out / "traffic.json"
    seed = tfile if tfile.exists() else repo_profiles / "traffic.json"
    data = json.loads(seed.read_text()) if seed.exists() else {}
    entry = {"fetch_correction": factor, "source": f"profiles/{args.tag}_pmc_traffic_{args.workload}.md"}
    if bench_line is not None:
        entry["kernel_config"] = bench_line["roofline"]["kernel_config"]
    for k, v in summary.items():
        if bench_name(k):
            entry[bench_name(k)] = {"bytes_per_launch": v["fetch_bytes"] + v["write_bytes"], "rocprof_name": k,
                                    "avg_us_under_pmc": round(v["avg_us_under_pmc"], 1)}
    for k, v in summary.items():        # the combine launch belongs to the same bench entry as its chunk / segment pass
        # (spmm_class_combine_kernel also closes the split-segment pass when the class pass is off)
        for part, whole in (("spmm_class_combine_kernel", "spmm_class_chunk_kernel+combine"),
                            ("spmm_class_combine_kernel", "spmm_split_segment_kernel+combine")):
            if part in k and whole in entry and not (whole.startswith("spmm_split") and
                                                     "spmm_class_chunk_kernel+combine" in entry):
                entry[whole]["bytes_per_launch"] += v["fetch_bytes"] + v["write_bytes"]
                entry[whole]["avg_us_under_pmc"] = round(entry[whole]["avg_us_under_pmc"] + v["avg_us_under_pmc"], 1)
    data[args.workload] = entry
    tfile.write_text(json.dumps(data, indent=1) + "\n")
    if args.stats:
        f = _newest(f"{args.stats}/**/*kernel_stats.csv")
        rows = list(csv.DictReader(open(f)))
        sl = [f"# rocprofv3 --kernel-trace --stats, {args.workload} ({args.tag})", "",
              "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
        for r in rows[:12]:
            sl.append(f"| `{r['Name'].split('(')[0][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                      f"{float(r['TotalDurationNs']) / 1e6:.2f} | {r['Percentage']} |")
        (out / f"{args.tag}_kernel_stats_{args.workload}.md").write_text("\n".join(sl) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

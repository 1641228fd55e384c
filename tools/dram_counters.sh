#!/bin/bash
# Does gfx950 let the L2's fabric-side read requests be split into "went to DRAM" and "served elsewhere"?  rocprofv3 -L
# lists TCC_EA0_RDREQ_DRAM(_sum) / TCC_EA0_RDREQ_DRAM_32B beside TCC_EA0_RDREQ(_sum) (what FETCH_SIZE derives from).  One
# --pmc pass per counter (own runs, no tracing options), per-kernel averages into gpurun_out/profiles/<tag>_dram_counters_<workload>.md.
#   tools/dram_counters.sh <workload> <tag> [extra bench.py arguments]
set -e
W=${1:-rmat2m}; TAG=${2:-r04}; shift 2 || true
R="$(cd "$(dirname "$0")/.." && pwd)"
RAW="/tmp/clane_dram_${TAG}_${W}"; rm -rf "$RAW"; mkdir -p "$RAW" "$R/gpurun_out/profiles"
cd /tmp && export TMPDIR=/tmp
for C in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum; do
  if rocprofv3 --pmc $C --output-format csv -d "$RAW/$C" -- python3 "$R/bench.py" --workload $W --steps 6 --warmup 2 \
      --blocks 1 --no-cpu-baseline --legs none --no-parity "$@" > "$RAW/$C.json" 2> "$RAW/$C.err"; then echo "[dram] $C done"; else echo "[dram] $C FAILED: $(tail -2 "$RAW/$C.err")"; fi
done
python3 - "$RAW" "$W" "$TAG" "$R/gpurun_out/profiles" <<'PY'
import collections, csv, glob, os, sys
raw, w, tag, out = sys.argv[1:5]
table = collections.defaultdict(dict)
for d in sorted(glob.glob(f"{raw}/TCC_*")):
    if not os.path.isdir(d):
        continue
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        if "clane::spmm" in r["Kernel_Name"] or "clane::edge_score" in r["Kernel_Name"] or "l1_distance" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        table[k][c] = sum(v) / len(v)
cols = sorted({c for v in table.values() for c in v})
lines = [f"# Fabric-side L2 request counters per launch, {w} ({tag}; one --pmc pass per counter, `tools/dram_counters.sh`)", "",
         "| kernel | " + " | ".join(cols) + " |", "|---|" + "---|" * len(cols)]
for k, v in sorted(table.items()):
    lines.append(f"| `{k}` | " + " | ".join(f"{v.get(c, float('nan')):.4g}" for c in cols) + " |")
open(f"{out}/{tag}_dram_counters_{w}.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY

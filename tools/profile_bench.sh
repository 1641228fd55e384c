#!/bin/bash
# The three rocprofv3 passes behind bench.py's roofline line, on the GPU box (run from the repo root):
#   1. --kernel-trace --stats           per-kernel average durations (must agree with bench.py's HIP events)
#   2. --pmc FETCH_SIZE  (own pass)     bytes the L2 requested from the fabric, calibrated on l1_distance_kernel
#   3. --pmc WRITE_SIZE  (own pass)
# and their summary (tools/pmc_summary.py) into gpurun_out/profiles/: copy what should be judged into profiles/.
# The raw traces stay in /tmp on the box (hundreds of MB); only the summaries and the bench lines come back.
#   tools/profile_bench.sh <workload> <tag> [extra bench.py arguments]
set -e
W=${1:-rmat2m}; TAG=${2:-r04}; shift 2 || true
KEY=${CLANE_PROFILE_KEY:-${W}_n1}        # name of the traffic.json entry (e.g. rmat2m_column_slice_of_8 with --column-slice-of 8)
R="$(cd "$(dirname "$0")/.." && pwd)"
RAW="/tmp/clane_prof_${TAG}_${W}"; rm -rf "$RAW"; mkdir -p "$RAW" "$R/gpurun_out/profiles"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/stats" -- python3 "$R/bench.py" --workload $W --steps 20 \
    --warmup 5 --blocks 2 --no-cpu-baseline --legs none "$@" > "$RAW/bench_stats.json" 2> "$RAW/bench_stats.err" || { tail -20 "$RAW/bench_stats.err"; exit 1; }
echo "[profile] stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$RAW/fetch" -- python3 "$R/bench.py" --workload $W --steps 10 \
    --warmup 2 --blocks 2 --no-cpu-baseline --legs none --calibrate "$@" > "$RAW/bench_fetch.json" 2> "$RAW/bench_fetch.err" || { tail -20 "$RAW/bench_fetch.err"; exit 1; }
echo "[profile] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$RAW/write" -- python3 "$R/bench.py" --workload $W --steps 10 \
    --warmup 2 --blocks 2 --no-cpu-baseline --legs none --calibrate "$@" > "$RAW/bench_write.json" 2> "$RAW/bench_write.err" || { tail -20 "$RAW/bench_write.err"; exit 1; }
echo "[profile] WRITE_SIZE pass done"
python3 "$R/tools/pmc_summary.py" "$RAW/fetch" "$RAW/write" --tag $TAG --workload ${KEY} \
    --bench-json "$RAW/bench_fetch.json" --stats "$RAW/stats" --out "$R/gpurun_out/profiles"
cp "$RAW/bench_stats.json" "$R/gpurun_out/profiles/${TAG}_bench_under_rocprof_${KEY}.json"
cp "$RAW/bench_fetch.json" "$R/gpurun_out/profiles/${TAG}_bench_under_pmc_fetch_${KEY}.json"

#!/bin/bash
# One rank's column slice of config 3 at N = 1 / 2 / 4 / 8 on one GPU (tools/column_slice_time.py) -> gpurun_out/profiles/
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"; TAG=${1:-r02}
mkdir -p "$R/gpurun_out/profiles"
python3 "$R/tools/column_slice_time.py" --world 1 2 4 8 2>/dev/null | tee "$R/gpurun_out/profiles/${TAG}_column_slice_time.jsonl"
python3 "$R/tools/column_slice_time.py" --workload powerlaw10m --world 1 2 4 8 2>/dev/null | tee "$R/gpurun_out/profiles/${TAG}_column_slice_time_powerlaw10m.jsonl"

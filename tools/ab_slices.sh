#!/bin/bash
# A/B of library variants (build/variants/libclane_hip_*.so) on the workloads the sub-wave kernels serve:
# one rank's column slice of config 3 at N = 8 / 4 / 2, config 2, config 4's shape.  Run from the repo root.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2; do
for lib in "$R"/build/variants/libclane_hip_*.so; do
  name=$(basename $lib .so | sed 's/libclane_hip_//')
  echo "== $name (round $round)"
  CLANE_HIP_LIB=$lib python3 "$R/tools/column_slice_time.py" --world 8 4 2 | python3 -c "
import sys, json
for ln in sys.stdin:
    j = json.loads(ln); print('  slice W=%d d=%d: %.3f ms  main %.3f hub %.3f split %.3f  build_P %.2f' % (j['world'], j['d_local'], j['ms_per_sweep'], j['kernels_ms']['main'], j['kernels_ms']['hub'], j['kernels_ms']['split'], j['build_P_ms']))"
  for w in rmat2m rmat200k powerlaw10m; do
    CLANE_HIP_LIB=$lib python3 "$R/bench.py" --workload $w --no-cpu-baseline --legs none --steps 30 --warmup 5 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
print('  $w: %.1f sweeps/s  %.3f ms  main %.3f  parity %.2e' % (j['value'], j['ms_per_step'], r['kernels'].get('spmm_update_kernel', {}).get('avg_launch_ms', 0), j['parity_rel_l2_vs_oracle_after_1_sweep']))"
  done
done
done

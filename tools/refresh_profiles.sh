#!/bin/bash
# Re-take the whole per-round profile set on the GPU box (everything profiles/README.md lists that depends on the K3
# kernels).  Needed whenever SweepEngine.kernel_config() changes: bench.py refuses PMC entries taken with another one.
#   tools/refresh_profiles.sh <tag> [part]      part: pmc | pmc1 | pmc2 | bench | all (default all); ~25 GPU-minutes in all
#   (one gpurun call is at most 20 minutes: pmc1 = config 3 and its column slices + config 2, pmc2 = config 4's shape + the
#   16M-vertex capacity run; copy gpurun_out/profiles/traffic.json into profiles/ before the bench part)
# Results land in gpurun_out/profiles/: copy them to profiles/ and commit.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"; TAG=${1:-r04}; PART=${2:-all}
P="$R/gpurun_out/profiles"; mkdir -p "$P"
cd "$R"
if [ "$PART" = pmc ] || [ "$PART" = pmc1 ] || [ "$PART" = all ]; then
  tools/profile_bench.sh rmat2m $TAG
  for N in 8 4 2; do
    CLANE_PROFILE_KEY=rmat2m_column_slice_of_$N tools/profile_bench.sh rmat2m $TAG --column-slice-of $N --steps 60
  done
  tools/profile_bench.sh rmat200k $TAG --steps 100
  tools/profile_bench.sh uniform2m $TAG                # the no-reuse anchor: traffic == algorithmic bytes
  cp "$P/traffic.json" "$R/profiles/traffic.json"      # the bench lines below quote it
fi
if [ "$PART" = pmc ] || [ "$PART" = pmc2 ] || [ "$PART" = all ]; then
  tools/profile_bench.sh powerlaw10m $TAG
  tools/profile_bench.sh rmat16m $TAG --steps 8 --warmup 2
  cp "$P/traffic.json" "$R/profiles/traffic.json"
fi
if [ "$PART" = bench ] || [ "$PART" = all ]; then
  tools/profile_slices.sh $TAG
  python3 bench.py                                                   > "$P/${TAG}_bench_rmat2m_n1.json"          2>/dev/null
  python3 bench.py --workload uniform2m                              > "$P/${TAG}_bench_uniform2m_n1.json"       2>/dev/null
  python3 bench.py --workload rmat200k --steps 200 --warmup 20       > "$P/${TAG}_bench_rmat200k_n1.json"        2>/dev/null
  python3 bench.py --workload powerlaw10m --steps 20 --warmup 5 --iterate > "$P/${TAG}_bench_powerlaw10m_n1.json" 2>/dev/null
  python3 bench.py --workload rmat16m --steps 10 --warmup 3          > "$P/${TAG}_bench_rmat16m_n1.json"         2>/dev/null
  python3 bench.py --steps 20 --warmup 5 --iterate --no-cpu-baseline --legs none > "$P/${TAG}_bench_iterate_rmat2m_n1.json"  2>/dev/null
  for f in rmat2m uniform2m rmat200k powerlaw10m rmat16m iterate_rmat2m; do
    python3 - "$P/${TAG}_bench_${f}_n1.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j["roofline"]
print(f"{sys.argv[1].split('/')[-1]}: {j['value']:.1f} sweeps/s  {j['ms_per_step']:.3f} ms  frac {r['frac']:.3f}  traffic "
      f"{'ok' if r['traffic'] else r['kernels'][r['kernel']].get('traffic_missing')}  parity {j.get('parity_rel_l2_vs_oracle_after_1_sweep')}")
PY
  done
fi

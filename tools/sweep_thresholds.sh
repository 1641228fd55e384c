#!/bin/bash
# GPU box: sweep the row-binning thresholds of the K3 pass (bench.py --long-threshold / --hub-threshold).
cd "$(dirname "$0")/.."
for spec in "32 32" "48 48" "64 64" "24 24" "16 16" "32 128" "32 256" "64 128" "96 96" "128 128"; do
  set -- $spec
  timeout -k 10 120 python bench.py --no-cpu-baseline --legs none --steps 30 --warmup 5 --long-threshold $1 --hub-threshold $2 2>/dev/null |
    python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
ks=' '.join('%s=%.3f' % (k.split('<')[-1][:8] if '<' in k else 'main', v['avg_launch_ms']) for k,v in r['kernels'].items())
print('T=$1 H=$2 step %.3f pass %.3f (%.0f GB/s) | %s | buildP %.1f' % (j['ms_per_step'], r['k3_pass']['ms'], r['k3_pass']['GBps'], ks, j['build_P_ms']))"
done

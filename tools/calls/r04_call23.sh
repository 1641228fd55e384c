for CT in default 128 512 1024; do
  if [ $CT = default ]; then F=""; else F="--class-threshold $CT"; fi
  python3 bench.py --workload powerlaw10m --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline --no-parity $F 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('CT=$CT', round(j['value'],1), round(j['ms_per_step'],4), {k:round(v['avg_launch_ms'],4) for k,v in r['kernels'].items()}, 'bP', round(j['build_P_ms'],3), r['kernel_config']['long_threshold'], r['kernel_config']['class_threshold'])"
done

mkdir -p gpurun_out/r04
for CT in default 0 32 64 128 256 512 1024; do
  if [ $CT = default ]; then F=""; else F="--class-threshold $CT"; fi
  python3 bench.py --workload rmat200k --steps 200 --warmup 20 --no-cpu-baseline --no-parity $F 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('CT=$CT', round(j['value'],1), round(j['ms_per_step'],4), {k:round(v['avg_launch_ms'],4) for k,v in r['kernels'].items()}, 'bP', round(j['build_P_ms'],3), r['kernel_config']['long_threshold'], r['kernel_config']['class_threshold'])"
done
for LT in 64 128 256 512; do
  python3 bench.py --workload rmat200k --steps 200 --warmup 20 --no-cpu-baseline --no-parity --long-threshold $LT 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('LT=$LT', round(j['value'],1), round(j['ms_per_step'],4), {k:round(v['avg_launch_ms'],4) for k,v in r['kernels'].items()}, 'bP', round(j['build_P_ms'],3), r['kernel_config']['long_threshold'], r['kernel_config']['class_threshold'])"
done

set -e
mkdir -p gpurun_out/r04 gpurun_out/profiles
cp profiles/traffic.json gpurun_out/profiles/traffic.json
tools/profile_bench.sh rmat200k r04 --steps 100 > gpurun_out/r04/profile_rmat200k.log 2>&1 || { tail -20 gpurun_out/r04/profile_rmat200k.log; exit 1; }
cp gpurun_out/profiles/traffic.json profiles/traffic.json
python3 bench.py --workload rmat200k --steps 200 --warmup 20 > gpurun_out/profiles/r04_bench_rmat200k_n1.json 2>/dev/null
python3 - <<'PY'
import json
j = json.loads(open('gpurun_out/profiles/r04_bench_rmat200k_n1.json').read().strip().splitlines()[-1]); r = j['roofline']
print('rmat200k', j['value'], j['ms_per_step'], r['frac'], r['traffic'], r['kernel_config']['class_threshold'], j['build_P_ms'], j['parity_rel_l2_vs_oracle_after_1_sweep'])
PY
python3 -m pytest tests -q -m gpu -x -k "rmat_200k or config2 or partitioned or class" > gpurun_out/r04/t_ct.log 2>&1 || { tail -30 gpurun_out/r04/t_ct.log; exit 1; }
tail -2 gpurun_out/r04/t_ct.log

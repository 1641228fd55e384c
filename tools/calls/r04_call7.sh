set -e
mkdir -p gpurun_out/r04
python3 tools/placement_probe.py --graph uniform --blocks 3 --out gpurun_out/r04/placement_probe_box3_default.jsonl > gpurun_out/r04/pp3a.log 2>&1
echo "[default allocator]"
python3 - <<'PY'
import json
for l in open('gpurun_out/r04/placement_probe_box3_default.jsonl'):
    r = json.loads(l); print(r['label'], r['ms_median'])
PY
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True python3 tools/placement_probe.py --graph uniform --blocks 3 --out gpurun_out/r04/placement_probe_box3_expandable.jsonl > gpurun_out/r04/pp3b.log 2>&1 || tail -5 gpurun_out/r04/pp3b.log
echo "[expandable segments]"
python3 - <<'PY'
import json
for l in open('gpurun_out/r04/placement_probe_box3_expandable.jsonl'):
    r = json.loads(l); print(r['label'], r['ms_median'], r['ptr']['Z0'])
PY
python3 bench.py --workload powerlaw10m --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_powerlaw10m_a.json 2> gpurun_out/r04/bench_powerlaw10m_a.err
python3 - <<'PY'
import json
j = json.loads(open('gpurun_out/r04/bench_powerlaw10m_a.json').read().strip().splitlines()[-1])
print('powerlaw10m', j['value'], j['ms_per_step'], j['build_P_ms'], {k: round(v['avg_launch_ms'], 3) for k, v in j['roofline']['kernels'].items()})
PY

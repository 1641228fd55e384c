set -e
mkdir -p gpurun_out/r04
python3 -m pytest tests -q -m gpu --durations=5 -x > gpurun_out/r04/gputests_6.log 2>&1 || { tail -60 gpurun_out/r04/gputests_6.log; exit 1; }
tail -9 gpurun_out/r04/gputests_6.log
python3 bench.py --iterate > gpurun_out/r04/bench_rmat2m_l1norms.json 2> gpurun_out/r04/bench_rmat2m_l1norms.err
python3 - <<'PY'
import json
j = json.loads(open('gpurun_out/r04/bench_rmat2m_l1norms.json').read().strip().splitlines()[-1])
print('rmat2m', j['value'], j['ms_per_step'], 'build_P', j['build_P_ms'], 'cold', j['build_P_cold_ms'], j['iterate']['wall_s'], j['roofline']['frac'], j['cpu_baseline_torch']['sample'][-120:])
PY

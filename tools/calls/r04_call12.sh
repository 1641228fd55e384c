set -e
mkdir -p gpurun_out/r04
python3 -m pytest tests -q -m gpu --durations=3 -x > gpurun_out/r04/gputests_7.log 2>&1 || { tail -60 gpurun_out/r04/gputests_7.log; exit 1; }
tail -7 gpurun_out/r04/gputests_7.log
python3 bench.py --workload rmat16m --steps 10 --warmup 3 > gpurun_out/profiles/r04_bench_rmat16m_n1.json 2> gpurun_out/r04/bench_rmat16m.err
python3 - <<'PY'
import json
j = json.loads(open('gpurun_out/profiles/r04_bench_rmat16m_n1.json').read().strip().splitlines()[-1])
print('rmat16m', j['value'], j['roofline']['frac'], j['cpu_baseline']['value'], j['cpu_baseline_torch']['value'], j['cpu_baseline_torch']['sample'][-220:])
PY
/usr/bin/time -v python3 bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
grep -E "Elapsed|Maximum resident" gpurun_out/r04/bench_default.err

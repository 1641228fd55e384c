set -e
mkdir -p gpurun_out/r04
python3 -m pytest tests -q -m gpu --durations=3 -x > gpurun_out/r04/gputests_9.log 2>&1 || { tail -60 gpurun_out/r04/gputests_9.log; exit 1; }
tail -7 gpurun_out/r04/gputests_9.log
for W in rmat2m powerlaw10m; do python3 tools/build_p_time.py --workload $W >> gpurun_out/r04/build_p_time.jsonl 2>> gpurun_out/r04/bpt.err || tail -5 gpurun_out/r04/bpt.err; done
cat gpurun_out/r04/build_p_time.jsonl | cut -c1-600
python3 bench.py --no-cpu-baseline > gpurun_out/r04/bench_softmax.json 2> gpurun_out/r04/bench_softmax.err
python3 - <<'PY'
import json
j = json.loads(open('gpurun_out/r04/bench_softmax.json').read().strip().splitlines()[-1])
print('rmat2m', j['value'], j['ms_per_step'], j['build_P_ms'], j['build_P_cold_ms'], j['parity_P_rel_l2_vs_oracle'])
PY

set -e
mkdir -p gpurun_out/r04 gpurun_out/profiles
python3 -m pytest tests -q -m gpu --durations=3 -x > gpurun_out/r04/gputests_8.log 2>&1 || { tail -60 gpurun_out/r04/gputests_8.log; exit 1; }
tail -7 gpurun_out/r04/gputests_8.log
python3 -c "import __graft_entry__ as g; g.smoke()"
python3 bench.py > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err
python3 - <<'PY'
import json
j = json.loads(open('gpurun_out/r04/bench_final.json').read().strip().splitlines()[-1])
print('final', j['value'], j['ms_per_step'], j['roofline']['frac'], j['build_P_ms'], j['build_P_cold_ms'], j['parity_rel_l2_vs_oracle_after_1_sweep'], j['cpu_baseline']['value'], j['cpu_baseline_torch']['value'])
PY

set -e
mkdir -p gpurun_out/r04
python3 -m pytest tests -q -m gpu --durations=8 -x > gpurun_out/r04/gputests_4.log 2>&1 || { tail -60 gpurun_out/r04/gputests_4.log; exit 1; }
tail -12 gpurun_out/r04/gputests_4.log
python3 tools/placement_probe.py --graph uniform --out gpurun_out/r04/placement_probe_box2.jsonl > gpurun_out/r04/placement_uniform_box2.log 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/r04/placement_probe_box2.jsonl'):
    r = json.loads(l); print(r['label'], r['ms_median'], r['ms_blocks'])
PY
python3 tools/threshold_robustness.py --out gpurun_out/r04/r04_threshold_robustness.md > gpurun_out/r04/robust.jsonl 2> gpurun_out/r04/robust.err || tail -20 gpurun_out/r04/robust.err
grep -E "^##|default|behind" gpurun_out/r04/r04_threshold_robustness.md | head -60

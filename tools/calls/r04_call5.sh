set -e
mkdir -p gpurun_out/r04
python3 -m pytest tests -q -m gpu --durations=15 -x > gpurun_out/r04/gputests_3.log 2>&1 || { tail -60 gpurun_out/r04/gputests_3.log; exit 1; }
tail -22 gpurun_out/r04/gputests_3.log
for C in 2 4; do
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 --exchange grid --grid-cols $C >> gpurun_out/r04/rank_compute_grid.jsonl 2>> gpurun_out/r04/rank_compute_grid.err
done
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 4 --exchange halo >> gpurun_out/r04/rank_compute_grid.jsonl 2>> gpurun_out/r04/rank_compute_grid.err
cat gpurun_out/r04/rank_compute_grid.jsonl | cut -c1-900

set -e
mkdir -p gpurun_out/r04
python3 -m pytest tests -q -m gpu --durations=5 -x > gpurun_out/r04/gputests_5.log 2>&1 || { tail -60 gpurun_out/r04/gputests_5.log; exit 1; }
tail -9 gpurun_out/r04/gputests_5.log
python3 -c "import __graft_entry__ as g; g.smoke()"
tools/refresh_profiles.sh r04 bench > gpurun_out/refresh_bench.log 2>&1 || { tail -30 gpurun_out/refresh_bench.log; exit 1; }
tail -12 gpurun_out/refresh_bench.log

set -e
mkdir -p gpurun_out/r04
python3 -m pytest tests -q -m gpu --durations=40 -x > gpurun_out/r04/gputests_1.log 2>&1 || { tail -60 gpurun_out/r04/gputests_1.log; exit 1; }
tail -50 gpurun_out/r04/gputests_1.log
tools/dram_counters.sh rmat2m r04 > gpurun_out/r04/dram_rmat2m.log 2>&1 || tail -20 gpurun_out/r04/dram_rmat2m.log
tools/dram_counters.sh uniform2m r04 > gpurun_out/r04/dram_uniform2m.log 2>&1 || tail -20 gpurun_out/r04/dram_uniform2m.log
echo "[dram] done"

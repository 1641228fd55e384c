set -e
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "integration_stub or partitioned_engine" > gpurun_out/r04/gputests_2a.log 2>&1 || { tail -60 gpurun_out/r04/gputests_2a.log; exit 1; }
tail -3 gpurun_out/r04/gputests_2a.log
python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_beyond_2_31.py -q -m gpu --durations=25 -x > gpurun_out/r04/gputests_2.log 2>&1 || { tail -60 gpurun_out/r04/gputests_2.log; exit 1; }
tail -32 gpurun_out/r04/gputests_2.log
tools/dram_counters.sh rmat2m r04 > gpurun_out/r04/dram_rmat2m.log 2>&1 || tail -20 gpurun_out/r04/dram_rmat2m.log
tools/dram_counters.sh uniform2m r04 > gpurun_out/r04/dram_uniform2m.log 2>&1 || tail -20 gpurun_out/r04/dram_uniform2m.log
echo "[dram] done"

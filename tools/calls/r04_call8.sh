set -e
mkdir -p gpurun_out/r04
python3 -m pytest tests -q -m gpu --durations=5 -x > gpurun_out/r04/gputests_5.log 2>&1 || { tail -60 gpurun_out/r04/gputests_5.log; exit 1; }
tail -9 gpurun_out/r04/gputests_5.log
python3 -c "import __graft_entry__ as g; g.smoke()"
python3 tools/rank_compute_time.py --workload rmat2m --world 8 --exchange halo > gpurun_out/r04/rank_compute_halo_mirror.jsonl 2> gpurun_out/r04/rank_compute_halo_mirror.err
python3 tools/rank_compute_time.py --workload rmat2m --world 8 --exchange halo_p2p >> gpurun_out/r04/rank_compute_halo_mirror.jsonl 2>> gpurun_out/r04/rank_compute_halo_mirror.err
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 --exchange halo >> gpurun_out/r04/rank_compute_halo_mirror.jsonl 2>> gpurun_out/r04/rank_compute_halo_mirror.err
cut -c1-420 gpurun_out/r04/rank_compute_halo_mirror.jsonl
tools/refresh_profiles.sh r04 bench > gpurun_out/refresh_bench.log 2>&1 || { tail -30 gpurun_out/refresh_bench.log; exit 1; }
tail -12 gpurun_out/refresh_bench.log

set -e
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "norms or lane_xor or class_pass_random or split_hub or spmm_update_vs or embedder_iterate or class_affine or partitioned_engine or lagged" > gpurun_out/r04/t_fused.log 2>&1 || { tail -40 gpurun_out/r04/t_fused.log; exit 1; }
tail -3 gpurun_out/r04/t_fused.log
python3 bench.py --iterate > gpurun_out/r04/bench_rmat2m_fused.json 2> gpurun_out/r04/bench_rmat2m_fused.err
echo "[2] headline bench done"
python3 bench.py --workload uniform2m > gpurun_out/r04/bench_uniform2m.json 2> gpurun_out/r04/bench_uniform2m.err
echo "[3] uniform bench done"
(rocm-smi --showmemorypartition --showcomputepartition --showmeminfo vram > gpurun_out/r04/smi.txt 2>&1 || true)

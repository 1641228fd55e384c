set -e
mkdir -p gpurun_out/r04
python3 bench.py --gpus 2 --backend gloo --share-gpu 2> gpurun_out/r04/bench_n2.err | grep '^{' > gpurun_out/r04/r04_bench_rmat2m_n2_gloo_shared_gpu.json
echo "[1] 2 ranks, config 3: $(python3 -c "import json;j=json.load(open('gpurun_out/r04/r04_bench_rmat2m_n2_gloo_shared_gpu.json'));print(j['value'], j['parity_rel_l2_vs_oracle_after_1_sweep'], j['north_star_literal'].get('parity_rel_l2_vs_oracle_after_1_sweep'), list(j.get('other_divisions',{}).keys()))")"
python3 bench.py --gpus 4 --backend gloo --share-gpu --workload rmat200k --steps 10 --warmup 2 --also-exchange allgather_all,allgather,halo,grid 2> gpurun_out/r04/bench_n4.err | grep '^{' > gpurun_out/r04/r04_bench_rmat200k_n4_gloo_shared_gpu.json
echo "[2] 4 ranks, config 2: $(python3 -c "import json;j=json.load(open('gpurun_out/r04/r04_bench_rmat200k_n4_gloo_shared_gpu.json'));print(j['value'], j['parity_rel_l2_vs_oracle_after_1_sweep'], {k:(v.get('parity_rel_l2_vs_oracle_after_1_sweep'), v.get('error')) for k,v in j.get('other_divisions',{}).items()})")"
python3 bench.py --rehearse-rccl 2> gpurun_out/r04/bench_rccl.err | grep '^{' > gpurun_out/r04/r04_bench_rehearse_rccl_rmat2m.json
echo "[3] rccl rehearsal: $(python3 -c "import json;j=json.load(open('gpurun_out/r04/r04_bench_rehearse_rccl_rmat2m.json'));print(j['value'], j['comm']['backend'], j['north_star_literal'].get('value'))")"
python3 tools/determinism_soak.py > gpurun_out/r04/determinism_soak.jsonl 2> gpurun_out/r04/determinism_soak.err
cat gpurun_out/r04/determinism_soak.jsonl
CLANE_BIG=1 python3 -m pytest tests/test_gpu_beyond_2_31.py -q -m gpu -s > gpurun_out/r04/big.log 2>&1 || { tail -30 gpurun_out/r04/big.log; exit 1; }
tail -5 gpurun_out/r04/big.log

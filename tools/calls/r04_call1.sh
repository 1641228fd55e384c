set -e
mkdir -p gpurun_out/r04
(rocprofv3 -L > gpurun_out/r04/counters_full.txt 2>&1 || true)
grep -i -E "TCC_EA|DRAM|MALL|HBM|TCC_REQ|TCC_HIT|TCC_MISS|FETCH_SIZE|WRITE_SIZE|TCC_BUBBLE|IOMMU" gpurun_out/r04/counters_full.txt | cut -c1-300 > gpurun_out/r04/counters_mem.txt || true
echo "[1] counters listed: $(wc -l < gpurun_out/r04/counters_full.txt) lines"
python3 bench.py --no-cpu-baseline > gpurun_out/r04/bench_rmat2m_pre.json 2> gpurun_out/r04/bench_rmat2m_pre.err
echo "[2] headline bench done"
tools/profile_bench.sh uniform2m r04 > gpurun_out/r04/profile_uniform2m.log 2>&1
echo "[3] uniform2m profile done"
python3 tools/placement_probe.py --graph uniform --out gpurun_out/r04/placement_probe.jsonl > gpurun_out/r04/placement_uniform.log 2>&1
echo "[4] placement uniform done"
python3 tools/placement_probe.py --graph star --out gpurun_out/r04/placement_probe.jsonl > gpurun_out/r04/placement_star.log 2>&1
echo "[5] placement star done"

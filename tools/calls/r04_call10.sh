set -e
mkdir -p gpurun_out/r04
python3 tools/fused_norms_ab.py --out gpurun_out/r04/fused_norms_ab.jsonl > gpurun_out/r04/fused_norms_ab.log 2>&1 || { tail -20 gpurun_out/r04/fused_norms_ab.log; exit 1; }
cat gpurun_out/r04/fused_norms_ab.jsonl
python3 tools/placement_probe.py --graph uniform --blocks 3 --contiguous --out gpurun_out/r04/placement_probe_contiguous.jsonl > gpurun_out/r04/pp_contig.log 2>&1 || { tail -20 gpurun_out/r04/pp_contig.log; exit 1; }
python3 - <<'PY'
import json
for l in open('gpurun_out/r04/placement_probe_contiguous.jsonl'):
    r = json.loads(l); print(r['label'], r['table_alloc'], r['ms_median'], r.get('table_alloc_note'))
PY
python3 bench.py --gpus 2 --backend gloo --share-gpu --steps 20 --warmup 5 2> gpurun_out/r04/bench_n2.err | grep '^{' > gpurun_out/r04/r04_bench_rmat2m_n2_gloo_shared_gpu.json
echo "[n2] $(python3 -c "import json;j=json.load(open('gpurun_out/r04/r04_bench_rmat2m_n2_gloo_shared_gpu.json'));print(j['value'], j['parity_rel_l2_vs_oracle_after_1_sweep'], j['north_star_literal'].get('parity_rel_l2_vs_oracle_after_1_sweep'), list(j.get('other_divisions',{}).keys()))")"

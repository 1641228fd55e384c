set -e
mkdir -p gpurun_out/r04 gpurun_out/profiles
python3 bench.py --workload rmat16m --steps 10 --warmup 3 > gpurun_out/profiles/r04_bench_rmat16m_n1.json 2> gpurun_out/r04/bench_rmat16m.err
python3 - <<'PY'
import json
j = json.loads(open('gpurun_out/profiles/r04_bench_rmat16m_n1.json').read().strip().splitlines()[-1])
print('rmat16m', j['value'], j['roofline']['frac'], j['cpu_baseline']['value'], j['cpu_baseline_torch']['value'], j['cpu_baseline_torch']['sample'][-220:])
PY
T0=$SECONDS
python3 bench.py > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
echo "default bench.py wall: $((SECONDS - T0)) s"
O=gpurun_out/r04/rank_compute_halo_variants.jsonl
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 --exchange halo >> $O 2>> gpurun_out/r04/rcv.err
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 --exchange halo --no-fused-pack >> $O 2>> gpurun_out/r04/rcv.err
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 --exchange halo --chunks 1 >> $O 2>> gpurun_out/r04/rcv.err
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 --exchange halo --no-overlap >> $O 2>> gpurun_out/r04/rcv.err
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 --exchange allgather_all >> $O 2>> gpurun_out/r04/rcv.err
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 --exchange allgather_all --chunks 1 >> $O 2>> gpurun_out/r04/rcv.err
python3 - <<'PY'
import json
for l in open('gpurun_out/r04/rank_compute_halo_variants.jsonl'):
    r = json.loads(l); print(r['exchange'], 'fused' if r['fused_pack'] else 'nofuse', r['kernels_ms_summed_over_chunks'], r['rank0_compute_ms_per_sweep'], r['table_rows'], r['E_loc'])
PY

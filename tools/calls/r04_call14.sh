set -e
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "mirror or partitioned_engine or halo_p2p or spmm_update_vs or random_shapes or class_pass_random" > gpurun_out/r04/t_mirror.log 2>&1 || { tail -40 gpurun_out/r04/t_mirror.log; exit 1; }
tail -3 gpurun_out/r04/t_mirror.log
O=gpurun_out/r04/rank_compute_halo_prefetch.jsonl
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 --exchange halo >> $O 2>> gpurun_out/r04/rcv2.err
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 --exchange halo --chunks 1 >> $O 2>> gpurun_out/r04/rcv2.err
python3 tools/rank_compute_time.py --workload powerlaw10m --world 8 --exchange halo_p2p >> $O 2>> gpurun_out/r04/rcv2.err
python3 tools/rank_compute_time.py --workload rmat2m --world 8 --exchange halo >> $O 2>> gpurun_out/r04/rcv2.err
python3 tools/rank_compute_time.py --workload rmat2m --world 8 --exchange halo --chunks 1 >> $O 2>> gpurun_out/r04/rcv2.err
python3 - <<'PY'
import json
for l in open('gpurun_out/r04/rank_compute_halo_prefetch.jsonl'):
    r = json.loads(l); print(r['exchange'], r['d_local'], r['kernels_ms_summed_over_chunks'], r['rank0_compute_ms_per_sweep'], r['table_rows'])
PY

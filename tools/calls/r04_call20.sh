mkdir -p gpurun_out/r04
python3 tools/column_slice_time.py --workload rmat200k --world 1 2 4 --class-threshold 32 64 128 256 512 --steps 100 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print('rmat200k W', r['world'], 'd', r['d_local'], 'CT', r['class_threshold'], 'T', r['long_threshold'], 'ms', r['ms_per_sweep'], 'bP', r['build_P_ms'], r['kernels_ms'])"
python3 tools/column_slice_time.py --workload rmat2m --world 8 --class-threshold 128 256 512 --steps 60 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print('rmat2m W', r['world'], 'd', r['d_local'], 'CT', r['class_threshold'], 'T', r['long_threshold'], 'ms', r['ms_per_sweep'], 'bP', r['build_P_ms'], r['kernels_ms'])"

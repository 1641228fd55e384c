python3 tools/column_slice_time.py --workload rmat200k256 --world 1 2 --class-threshold 32 64 128 256 --steps 100 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print('rmat200k256 W', r['world'], 'd', r['d_local'], 'CT', r['class_threshold'], 'T', r['long_threshold'], 'ms', r['ms_per_sweep'], 'bP', r['build_P_ms'], r['kernels_ms'])"

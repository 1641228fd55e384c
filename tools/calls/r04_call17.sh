set -e
mkdir -p gpurun_out/r04
R=$PWD
cd /tmp && export TMPDIR=/tmp
for W in rmat2m powerlaw10m; do
rm -rf /tmp/sm_$W
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sm_$W -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --blocks 1 --no-cpu-baseline --no-parity > /tmp/sm_$W.json 2> /tmp/sm_$W.err || { tail -5 /tmp/sm_$W.err; exit 1; }
F=$(find /tmp/sm_$W -name "*kernel_stats.csv" | head -1)
echo "== $W"; grep -E "edge_softmax|edge_score|row_sqnorm|degree_weighted|l1_distance" $F | awk -F, '{print $1, $2, $4}' | cut -c1-160
done
cd $R
for W in rmat2m powerlaw10m; do python3 tools/build_p_time.py --workload $W >> gpurun_out/r04/build_p_time.jsonl 2>> gpurun_out/r04/bpt.err || tail -5 gpurun_out/r04/bpt.err; done
python3 - <<'PY'
import json
for l in open('gpurun_out/r04/build_p_time.jsonl'):
    r = json.loads(l); print({k: v for k, v in r.items() if k.endswith(' ms') or k in ('workload', 'world')})
PY

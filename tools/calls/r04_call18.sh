set -e
mkdir -p gpurun_out/r04
R=$PWD
cd /tmp && export TMPDIR=/tmp
for W in rmat2m powerlaw10m; do
rm -rf /tmp/sm_$W
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sm_$W -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --blocks 1 --no-cpu-baseline --no-parity > /tmp/sm_$W.json 2> /tmp/sm_$W.err || { tail -5 /tmp/sm_$W.err; exit 1; }
python3 - $W <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/sm_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
print("==", sys.argv[1])
for r in csv.DictReader(open(f)):
    if "clane::" in r["Name"] and any(k in r["Name"] for k in ("softmax", "edge_score", "sqnorm", "l1_distance")):
        print(r["Name"].split("(")[0][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done

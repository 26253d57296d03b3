# per-kernel times of the kNN stages inside the eval step (one stream, rocprofv3 --stats): bash tools/knn_subkernels.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp
LPD_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o k -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train > /dev/null 2>&1
f=$(find /tmp/kp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:44]
    if "knn" in n:
        print(n, r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY

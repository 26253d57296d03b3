# start / end / gap of every kernel of one eval step (rocprofv3 kernel trace): SIDE=0|1 bash tools/step_timeline.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
LPD_SIDE_STREAM=${SIDE:-0} rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o g -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))) for r in csv.DictReader(open(f))))
starts = [i for i, r in enumerate(rows) if "morton" in r[2]]
a, b = starts[-12], starts[-11]
t0 = rows[a][0]
prev_end = rows[a - 1][1]
for s, e, n, q, g, w in rows[a - 3:b + 2]:
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:46]
    print("q%s %9.1f %9.1f  dur %7.1f  gap %6.1f  %s  grid %s wg %s" % (q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n, g, w))
    prev_end = max(prev_end, e)
PY

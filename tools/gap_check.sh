# kernel-busy time against wall time of the eval step on one stream: bash tools/gap_check.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gc
LPD_SIDE_STREAM=${SIDE:-0} rocprofv3 --kernel-trace --output-format csv -d /tmp/gc -o g -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train > /tmp/gc_line.json 2>/dev/null
python3 - <<'PY'
import csv, glob, json
f = glob.glob("/tmp/gc/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
# the timed region: the last 20 steps; find step boundaries by the morton sort kernel
starts = [i for i, r in enumerate(rows) if "morton" in r[2]]
print("morton launches", len(starts))
seg = rows[starts[-21]:starts[-1]]          # 20 whole steps
wall = seg[-1][1] - seg[0][0]
# union of the kernel intervals (two streams overlap)
busy, cur_s, cur_e, gaps, gap_at = 0, seg[0][0], seg[0][1], [], []
for i, (s_, e_, n_) in enumerate(seg[1:], 1):
    if s_ > cur_e:
        busy += cur_e - cur_s; gaps.append(s_ - cur_e); gap_at.append(i); cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
print("kernels per step %.1f  busy %.3f ms/step  wall %.3f ms/step  idle %.1f %%  median gap %.2f us  gaps > 5 us: %d per step" % (
    len(seg) / 20, busy / 20e6, wall / 20e6, 100 * (1 - busy / wall), (sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0.0), sum(g > 5000 for g in gaps) / 20))
big = sorted(((gaps[j], seg[gap_at[j] - 1][2][:50], seg[gap_at[j]][2][:50]) for j in range(len(gaps))), reverse=True)[:8]
for g, a, b in big: print("  gap %.1f us after %s before %s" % (g / 1e3, a, b))
print(open("/tmp/gc_line.json").read()[:120])
PY

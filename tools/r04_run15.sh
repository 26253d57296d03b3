#!/bin/bash
mkdir -p gpurun_out/r04o
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04o/prof -o p -- python3 /root/repo/tools/map16_bench.py > /root/repo/gpurun_out/r04o/bench.txt 2>&1
cd /root/repo
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r04o/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:24]:
        print(r["Name"][:100], r["Calls"], r["AverageNs"])
PY
rm -rf gpurun_out/r04o/prof

#!/bin/bash
mkdir -p gpurun_out/r04n
python tools/tn_bench.py all > gpurun_out/r04n/tn_f32.txt 2>&1
python tools/tn_bench.py all bf16 > gpurun_out/r04n/tn_bf16.txt 2>&1
python tools/map16_bench.py > gpurun_out/r04n/map16.txt 2>&1
cat gpurun_out/r04n/tn_f32.txt gpurun_out/r04n/tn_bf16.txt gpurun_out/r04n/map16.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r04n/prof -o p -- python3 /root/repo/tools/map16_bench.py > /dev/null 2>&1
cd /root/repo
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r04n/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY

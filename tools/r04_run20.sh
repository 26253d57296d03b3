#!/bin/bash
mkdir -p gpurun_out/r04t
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r04t/prof -o p -- python3 /root/repo/tools/train_kernels.py bf16 10 > /dev/null 2>&1
cd /root/repo
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r04t/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 10 / 1e3
    print("all kernels per step: %.0f us" % tot)
    for r in rows:
        n = r["Name"]
        if "anonymous namespace" in n and "at::native" not in n or n.startswith("_ZN12_GLOBAL") or "gemm_p8" in n or "knn" in n:
            continue
        print("%8.1f us/step  x%5.1f  %s" % (float(r["TotalDurationNs"]) / 10 / 1e3, int(r["Calls"]) / 10, n[:150]))
PY
rm -rf gpurun_out/r04t/prof

#!/bin/bash
# per-kernel durations of a python tool under rocprofv3:  tools/prof_kernels.sh <tag> <pattern> python-script args...
# (kernel trace + stats only; prints the rows of <tag>_kernel_stats.csv whose kernel name matches <pattern>)
tag=$1; pat=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
script=$1; shift; case $script in /*) ;; *) script=$R/$script;; esac
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $tag -- python3 $script "$@" > $O.out 2> $O.err
python3 - "$O" "$pat" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r["Name"]):
        print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  min {float(r["MinNs"]) / 1e3:8.1f}  max {float(r["MaxNs"]) / 1e3:8.1f}')
PY

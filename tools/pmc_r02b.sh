# one PMC pass over the eval step: bash tools/pmc_r02b.sh <tag> <counter> [<counter> ...]
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcb
tag=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LPD_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/$tag -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-train --no-cpu-baseline > /dev/null 2> $O/$tag.err
tail -2 $O/$tag.err
find $O -type f -size +4M -delete

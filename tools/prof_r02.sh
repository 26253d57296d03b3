set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r02 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/pmc_cfg2_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-train --no-cpu-baseline > /dev/null 2> $O/pmc_cfg2_$c.err
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/pmc_cfg5_$c -o p -- python3 $R/bench.py --points 16384 --k 64 --batch 64 --steps 2 --warmup 1 --no-train --no-cpu-baseline > /dev/null 2> $O/pmc_cfg5_$c.err
done
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_cfg5_TCC -o p -- python3 $R/bench.py --points 16384 --k 64 --batch 64 --steps 2 --warmup 1 --no-train --no-cpu-baseline > /dev/null 2> $O/pmc_cfg5_TCC.err
cd $R
python3 bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
python3 bench.py --points 16384 --k 64 --batch 64 --no-train --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_cfg5.json 2>> $O/bench.err
for d in $O/pmc_*; do [ -d $d ] && python3 tools/pmc_sum.py $d edge_gather > $d.summary.json; done
find $O -name "*.csv" | head -40
find $O -name "*kernel_stats*" -exec cp {} $O/ \;
# keep only small files
find $O -type f -size +3M -delete
ls -la $O | head -40

# round-3 profile set: bash tools/prof_r03.sh   (on the GPU box; results under gpurun_out/r03prof, summaries copied to profiles/ by hand)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r03 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/stats.err
pmc() {  # tag, counters...
  tag=$1; shift
  LPD_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/pmc_$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-train --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_$tag.err
}
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum
cd $R
for t in sq1 sq2 fetch write tcc; do python3 tools/pmc_kernels.py $O/pmc_$t > $O/pmc_$t.summary.txt; done
find $O -name "*kernel_stats*" -exec cp {} $O/ \;
find $O -type f -size +3M -delete
python3 bench.py > $O/bench_line.json 2> $O/bench.err
ls -la $O | head -30

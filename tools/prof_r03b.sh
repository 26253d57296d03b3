# round 3, second profile set (after the training-step work): bash tools/prof_r03b.sh   (GPU box; output gpurun_out/r03bprof)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03bprof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r03b -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/stats.err
find $O -name "*kernel_stats*" -exec cp {} $O/ \;
pmc() {  # tag, counters...
  tag=$1; shift
  rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/pmc_$tag -o p -- python3 $R/tools/train_profile.py lpdnet bf16 > /dev/null 2> $O/pmc_$tag.err
}
pmc sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $R
for t in sq1 sq2 fetch write; do python3 tools/pmc_kernels.py $O/pmc_$t > $O/pmc_$t.summary.txt; done
find $O -type f -size +3M -delete
python3 bench.py > $O/bench_line.json 2> $O/bench.err
ls -la $O | head -30

# a round's profile set (on the GPU box): bash tools/prof_round.sh <tag, e.g. r06>   -> gpurun_out/<tag>prof (summaries are copied to profiles/<tag>_* by hand)
#  * K-agg HBM traffic by PMC for configs[1] and configs[4] (tools/kagg_pmc.py -> profiles/kagg_pmc.json)
#  * rocprofv3 --kernel-trace --stats of the default bench (eval with two batches in flight + both train legs), of the same eval loop with one
#    batch in flight, of the configs[4] stress step and of the bf16 train step
#  * PMC passes (separate runs, --kernel-trace only) over the configs[4] eval step and over one bf16 train step:
#    SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / SQ_WAVE_CYCLES / SQ_WAIT_*, SQ_LDS_*, FETCH_SIZE, WRITE_SIZE
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}prof; mkdir -p $O; cd $R
python3 tools/kagg_pmc.py cfg2 cfg5 > $O/kagg_pmc.log 2>&1
cp profiles/kagg_pmc.json $O/kagg_pmc.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/stats.err
find $O/stats -name "*kernel_stats*" -exec cp {} $O/${TAG}_eval10_train10_kernel_stats.csv \;
# the same eval loop with ONE batch in flight (no train legs): every kernel alone on the chip -- the durations the roofline fractions are quoted on
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o one -- python3 $R/bench.py --steps 10 --warmup 3 --in-flight 1 --no-train --no-cpu-baseline --no-secondary > $O/bench_one_in_flight_under_rocprof.json 2> $O/stats1.err
find $O/stats1 -name "*kernel_stats*" -exec cp {} $O/${TAG}_eval10_one_in_flight_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5 -o c5 -- python3 $R/bench.py --batch 64 --points 16384 --k 64 --steps 5 --warmup 2 --no-train --no-cpu-baseline --no-secondary > $O/cfg5_under_rocprof.json 2> $O/stats5.err
find $O/stats5 -name "*kernel_stats*" -exec cp {} $O/${TAG}_cfg5_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $O/statst -o tb -- python3 $R/tools/train_kernels.py bf16 8 > /dev/null 2> $O/statst.err
find $O/statst -name "*kernel_stats*" -exec cp {} $O/${TAG}_train_bf16_kernel_stats.csv \;
pmc() {  # tag, output name, counters ... -- program args
  tag=$1; shift; cnt=(); while [ "$1" != "--" ]; do cnt+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --output-format csv --pmc "${cnt[@]}" -d $O/pmc_$tag -o p -- python3 "$@" > /dev/null 2> $O/pmc_$tag.err
  python3 $R/tools/pmc_kernels.py $O/pmc_$tag > $O/pmc_$tag.summary.txt
}
C5="$R/bench.py --batch 64 --points 16384 --k 64 --steps 2 --warmup 1 --no-train --no-cpu-baseline --no-secondary"
pmc c5_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $C5
pmc c5_sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- $C5
pmc c5_fetch FETCH_SIZE -- $C5
pmc c5_write WRITE_SIZE -- $C5
TB="$R/tools/train_profile.py lpdnet bf16"
pmc tb_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $TB
pmc tb_sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- $TB
pmc tb_fetch FETCH_SIZE -- $TB
pmc tb_write WRITE_SIZE -- $TB
cd $R
# the per-program tables need the per-dispatch counter files, which are too large to travel back: build them here, then drop the big files
python3 tools/pmc_table.py $O c5 > $O/${TAG}_pmc_cfg5.txt 2>&1
python3 tools/pmc_table.py $O tb > $O/${TAG}_pmc_train_bf16.txt 2>&1
find $O -type f -size +3M -delete
python3 tools/train_profile.py lpdnet bf16 > $O/train_profile_bf16.txt 2>&1
python3 tools/train_profile.py lpdnet f32 > $O/train_profile_f32.txt 2>&1
python3 bench.py > $O/bench_line.json 2> $O/bench.err; cp bench_detail.json $O/bench_detail.json
tail -n 5 $O/kagg_pmc.log; ls $O | head -40; head -c 300 $O/bench_line.json

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kaggw_floor; mkdir -p $O; cd $R
python3 tools/kaggw_bench.py 16 > $O/kaggw_bench_b16.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/lds -o p -- python3 $R/tools/kaggw_bench.py 16 > /dev/null 2> $O/lds.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS -d $O/sq -o p -- python3 $R/tools/kaggw_bench.py 16 > /dev/null 2> $O/sq.err
cd $R
python3 tools/pmc_kernels.py $O/lds edge_gather > $O/pmc_lds.txt
python3 tools/pmc_kernels.py $O/sq edge_gather > $O/pmc_sq.txt
find $O -type f -size +3M -delete
cat $O/kaggw_bench_b16.txt $O/pmc_lds.txt $O/pmc_sq.txt

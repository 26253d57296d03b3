R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_tn; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $O/$tag -o p -- python3 $R/tools/tn_bench.py "dW conv3" > /dev/null 2> $O/$tag.err
  cd $R; python3 tools/pmc_kernels.py $O/$tag gemm_tn; cd /tmp
done

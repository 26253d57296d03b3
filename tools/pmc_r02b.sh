# PMC passes over the eval step (no trace domains besides kernel-trace): SQ busy/wait split, instruction mix, L2, LDS.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcb
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  LPD_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $O/p$i -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-train --no-cpu-baseline > /dev/null 2> $O/p$i.err
  python3 $R/tools/pmc_sum.py $O/p$i > $O/p$i.json
  tail -2 $O/p$i.err
done
find $O -type f -size +1M -delete

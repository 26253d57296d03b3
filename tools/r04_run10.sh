R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
LPD_BF16_X1=1 LPD_TEST_VERBOSE=1 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k "cfg2_full or bf16" 2>&1 | grep -a "cfg2\|passed\|failed\|convergence" | cut -c1-420 > $O/x1.log
LPD_BF16_X1=1 python tools/train_profile.py lpdnet bf16 > $O/train_profile_bf16_x1.txt 2>&1
cat $O/x1.log; head -14 $O/train_profile_bf16_x1.txt

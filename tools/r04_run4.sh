R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "edge_mlp_train or gemm_act or x3w or gemm" 2>&1 | tail -30 > $O/ops.log
python -m pytest tests/test_train_gpu.py -m gpu -x -q 2>&1 | tail -30 > $O/train.log
LPD_TEST_VERBOSE=1 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k cfg2_full 2>&1 | grep -a "cfg2\|passed\|failed" > $O/cfg2.log
python tools/train_profile.py lpdnet bf16 > $O/train_profile_bf16.txt 2>&1
python tools/train_profile.py lpdnet f32 > $O/train_profile_f32.txt 2>&1
tail -8 $O/ops.log; tail -5 $O/train.log; cat $O/cfg2.log | cut -c1-400; head -16 $O/train_profile_bf16.txt; head -16 $O/train_profile_f32.txt

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04l; mkdir -p $O; cd $R
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py -m gpu -q -x 2>&1 | tail -8 > $O/tests.log
python tools/train_profile.py lpdnet bf16 > $O/tp_bf16.txt 2>&1
python tools/train_profile.py lpdnet f32 > $O/tp_f32.txt 2>&1
python tools/side_small.py 1 10 32 > $O/side.txt 2>&1
tail -4 $O/tests.log; sed -n 2p $O/tp_bf16.txt; grep "gemmx3w\[4096\|dw_smallk\|gemm\[4096" $O/tp_bf16.txt; sed -n 2p $O/tp_f32.txt; tail -3 $O/side.txt

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -k "gemm_act or stage_tensors" 2>&1 | grep -v "^E    +" | tail -60 > $O/t.log
cat $O/t.log

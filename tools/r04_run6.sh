R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
python tools/r04_dbg_bwd.py > $O/dbg_bwd.txt 2>&1
python -m pytest tests/test_ops_gpu.py -m gpu -q -k "gemm_bnbwd or gemm_act or edge_mlp_train" 2>&1 | tail -12 > $O/ops.log
cat $O/dbg_bwd.txt | tail -16; tail -12 $O/ops.log

R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
python -m pytest tests/test_ops_gpu.py -m gpu -q -k "windowed_kagg or gemm_act" 2>&1 | grep -v "^E    +" | tail -30 > $O/t.log
python tools/kaggw_bench.py 16 > $O/kaggw_bench.txt 2>&1
tail -5 $O/t.log; cat $O/kaggw_bench.txt

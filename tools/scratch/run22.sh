#!/bin/bash
mkdir -p gpurun_out/r04v
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "dw_sel or dg2 or fused" > gpurun_out/r04v/ops.log 2>&1
tail -5 gpurun_out/r04v/ops.log
timeout 600 python tools/train_profile.py lpdnet bf16 > gpurun_out/r04v/prof_bf16.txt 2>&1
grep "step\|dw_sel" gpurun_out/r04v/prof_bf16.txt
timeout 600 python tools/train_profile.py lpdnet f32 > gpurun_out/r04v/prof_f32.txt 2>&1
grep "step\|dw_sel" gpurun_out/r04v/prof_f32.txt
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu -s -k "cfg2_full_size" > gpurun_out/r04v/cfg2.log 2>&1
grep -i "desc\|loss\|grad\|passed\|failed" gpurun_out/r04v/cfg2.log | head -20

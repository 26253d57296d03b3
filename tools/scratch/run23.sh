#!/bin/bash
mkdir -p gpurun_out/r04v
LPD_TEST_VERBOSE=1 timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu -s -k "cfg2_full_size" > gpurun_out/r04v/cfg2.log 2>&1
grep "cfg2\|passed\|failed" gpurun_out/r04v/cfg2.log | head
LPD_MAP_BF16=0 LPD_SPLIT_BWD_BF16=0 LPD_TEST_VERBOSE=1 timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu -s -k "cfg2_full_size and bf16" > gpurun_out/r04v/cfg2_old.log 2>&1
grep "cfg2\|passed\|failed" gpurun_out/r04v/cfg2_old.log | head
timeout 600 python tools/train_profile.py lpdnet f32 > gpurun_out/r04v/prof_f32.txt 2>&1
grep "step\|dw_sel" gpurun_out/r04v/prof_f32.txt

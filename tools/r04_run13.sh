#!/bin/bash
# bf16 conv3 map: operator tests, train tests in bf16 storage, train profile
mkdir -p gpurun_out/r04m
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "bf16_map or gemm_tn or gemm_act or x3w_batched or stats or x3t or rows" > gpurun_out/r04m/ops.log 2>&1
tail -5 gpurun_out/r04m/ops.log
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "bf16 or cfg2" > gpurun_out/r04m/train.log 2>&1
tail -5 gpurun_out/r04m/train.log
timeout 600 python tools/train_profile.py lpdnet bf16 > gpurun_out/r04m/prof_bf16.txt 2>&1
head -45 gpurun_out/r04m/prof_bf16.txt

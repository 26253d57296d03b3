#!/bin/bash
mkdir -p gpurun_out/r04q
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "operand_transform or gemm_tn or bf16_map or gemm_act or batched" > gpurun_out/r04q/ops.log 2>&1
tail -5 gpurun_out/r04q/ops.log
timeout 1500 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "bf16 or cfg2" > gpurun_out/r04q/train.log 2>&1
tail -5 gpurun_out/r04q/train.log
timeout 600 python tools/train_profile.py lpdnet bf16 > gpurun_out/r04q/prof_bf16.txt 2>&1
head -8 gpurun_out/r04q/prof_bf16.txt; grep "gemm_tn\|gemmx3w" gpurun_out/r04q/prof_bf16.txt
timeout 600 python tools/train_profile.py lpdnet f32 > gpurun_out/r04q/prof_f32.txt 2>&1
head -8 gpurun_out/r04q/prof_f32.txt; grep "gemm_tn\|gemmx3w" gpurun_out/r04q/prof_f32.txt

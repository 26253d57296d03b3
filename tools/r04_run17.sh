#!/bin/bash
mkdir -p gpurun_out/r04q
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_tn or bf16_map" > gpurun_out/r04q/ops.log 2>&1
tail -3 gpurun_out/r04q/ops.log
python tools/tn_bench.py all 2>&1 | grep -v amdgpu
python tools/tn_bench.py all bf16 2>&1 | grep -v amdgpu
timeout 600 python tools/train_profile.py lpdnet bf16 > gpurun_out/r04q/prof_bf16.txt 2>&1
grep "step\|gemm_tn" gpurun_out/r04q/prof_bf16.txt
timeout 600 python tools/train_profile.py lpdnet f32 > gpurun_out/r04q/prof_f32.txt 2>&1
grep "step\|gemm_tn" gpurun_out/r04q/prof_f32.txt

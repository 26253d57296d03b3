#!/bin/bash
mkdir -p gpurun_out/r04p
python tools/map16_bench.py 2>&1 | head -3
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/r04p/ops.log 2>&1
tail -3 gpurun_out/r04p/ops.log
timeout 600 python tools/train_profile.py lpdnet bf16 > gpurun_out/r04p/prof_bf16.txt 2>&1
head -12 gpurun_out/r04p/prof_bf16.txt
timeout 600 python tools/train_profile.py lpdnet f32 > gpurun_out/r04p/prof_f32.txt 2>&1
head -12 gpurun_out/r04p/prof_f32.txt

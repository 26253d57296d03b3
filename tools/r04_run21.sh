#!/bin/bash
mkdir -p gpurun_out/r04u
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/r04u/gpu_tests.log 2>&1
tail -5 gpurun_out/r04u/gpu_tests.log

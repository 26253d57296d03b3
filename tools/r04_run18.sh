#!/bin/bash
# HBM traffic of the transposed-read weight-gradient kernel (FETCH_SIZE KiB, doubled for 16-byte loads on gfx950), normal and memory-only modes
mkdir -p gpurun_out/r04r
cd /tmp && export TMPDIR=/tmp
for d in 0 2; do
  export LPD_TN_TR_DBG=$d
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /root/repo/gpurun_out/r04r/f$d -o p -- python3 /root/repo/tools/tn_bench.py conv3 bf16 > /dev/null 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d /root/repo/gpurun_out/r04r/t$d -o p -- python3 /root/repo/tools/tn_bench.py conv3 bf16 > /dev/null 2>&1
done
cd /root/repo
for d in f0 t0 f2 t2; do echo $d; python tools/pmc_kernels.py gpurun_out/r04r/$d gemm_tn_tr; done
rm -rf gpurun_out/r04r

#!/bin/bash
mkdir -p gpurun_out/r04s
for side in 0 1; do
  LPD_TRAIN_SIDE=$side python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/r04s/bench_side$side.json 2> gpurun_out/r04s/bench_side$side.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04s/bench_side$side.json").read().strip().splitlines()[-1])
print("side=$side", "eval", d["value"], {k: v for k, v in d.items() if "train" in k})
PY
done
timeout 1800 python -m pytest tests/test_train_gpu.py -x -q -m gpu > gpurun_out/r04s/train.log 2>&1
tail -5 gpurun_out/r04s/train.log

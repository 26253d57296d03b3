"""A/B of the training step (bench.train_bench at configs[2]: 44 clouds, three fresh models per process):

    [LPD_DEBUG=...] python tools/train_ab.py [bf16|f32] [steps]

Run A and B in ONE gpurun call: the boxes of the pool differ by +-3 %."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lpd-net-pytorch_amd")]
import torch
import bench
storage = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
for rep in range(3):
    r = bench.train_bench(dev, None, 1, 0, 4096, steps, storage=storage)
    print(os.environ.get("LPD_DEBUG", ""), storage, r["ms_per_step"], r["losses"][:3], flush=True)

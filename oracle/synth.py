"""oracle/synth.py -- TEST INFRASTRUCTURE: closed-form synthetic weights and clouds.

Every value is a pure function of (tensor name, flat index), computed with 64-bit integer hashing
in numpy, so the generating script (tests/golden/make_golden.py, run against the reference in the
build container) and the tests / bench on the GPU box see bit-identical fp32 tensors without any
weights being committed (the model has 17.6 M parameters = 70 MB).

Not tied to torch's RNG or to nn.Module construction order.
"""
import numpy as np

_MASK = (1 << 64) - 1


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(name: str, n: int) -> np.ndarray:
    """n values in [-1, 1), float64, function of (name, index) only."""
    base = np.uint64(_fnv1a(name))
    with np.errstate(over="ignore"):
        ctr = base + np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)
    z = _splitmix64(ctr)
    # top 53 bits -> [0,1)
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return 2.0 * u - 1.0


def tensor_for(key: str, shape, *, gain: float = 1.0) -> np.ndarray:
    """Synthetic value of one state_dict entry (fp32; int64 for num_batches_tracked)."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    u = uniform(key, n)
    if leaf == "running_var":
        v = 0.5 + np.abs(u)
    elif leaf == "running_mean":
        v = 0.1 * u
    elif len(shape) >= 2:  # conv / linear / NetVLAD matrices
        if leaf in ("cluster_weights", "gating_weights"):
            fan_in = shape[0]                 # stored [in, out]
        elif leaf == "cluster_weights2":
            fan_in = shape[1]                 # [1, feature, cluster]
        elif leaf == "hidden1_weights":
            fan_in = max(shape[0] // 64, 1)   # reference scales by 1/sqrt(feature_size); 64 clusters
        else:
            fan_in = int(np.prod(shape[1:]))  # torch [out, in, ...]
        v = u * np.sqrt(3.0 / fan_in) * gain
    elif leaf == "weight":  # 1-D weight => BatchNorm gamma; ~28 % negative to exercise the min path
        v = 0.25 + 0.9 * u
    else:  # biases
        v = 0.1 * u
    return v.astype(np.float32).reshape(shape)


def state_dict_like(shapes: dict, *, gain: float = 1.0) -> dict:
    """shapes: {key: shape} -> {key: np.ndarray} with synthetic values."""
    return {k: tensor_for(k, s, gain=gain) for k, s in shapes.items()}


def cloud(seed: int, B: int, N: int, C: int = 3) -> np.ndarray:
    """[B, N, C] fp32 points, U[-1,1)^C, function of (seed, b, n, c) only."""
    out = np.empty((B, N, C), dtype=np.float32)
    for b in range(B):
        out[b] = uniform(f"cloud/{seed}/{b}", N * C).astype(np.float32).reshape(N, C)
    return out

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


def scene_cloud(seed: int, B: int, N: int) -> np.ndarray:
    """[B, N, 3] fp32 structured clouds: every cloud is a DIFFERENT arrangement of a noisy ground patch, 2-5 vertical wall segments
    and 2-4 compact blobs inside [-1, 1)^3 (a crude stand-in for the Oxford submaps the reference trains on: zero-centred,
    ground-removed street scenes), function of (seed, b, n) only.  Unlike `cloud` (44 draws of the SAME uniform distribution, whose
    descriptors are nearly equal, so that the head's train-mode BatchNorms over the batch rows divide by a vanishing variance and
    amplify rounding and kNN near-ties a hundredfold), the clouds of a batch differ from one another the way real submaps do."""
    out = np.empty((B, N, 3), dtype=np.float32)
    for b in range(B):
        par = uniform(f"scene/{seed}/{b}/par", 64)
        u = uniform(f"scene/{seed}/{b}/pts", N * 3).reshape(N, 3)
        walls = 2 + int((par[0] + 1.0) * 2.0) % 4                 # 2..5
        blobs = 2 + int((par[1] + 1.0) * 1.5) % 3                 # 2..4
        ground_frac = 0.25 + 0.15 * par[2]                        # 0.10 .. 0.40 of the points
        n_ground = int(ground_frac * N)
        n_blob = int((0.10 + 0.05 * par[3]) * N)
        pts = np.empty((N, 3), dtype=np.float64)
        # ground: a tilted patch with 2 cm-scale roughness
        gx, gy = par[4] * 0.2, par[5] * 0.2
        pts[:n_ground, 0] = u[:n_ground, 0] * 0.95
        pts[:n_ground, 1] = u[:n_ground, 1] * 0.95
        pts[:n_ground, 2] = -0.6 + 0.3 * par[6] + gx * pts[:n_ground, 0] + gy * pts[:n_ground, 1] + 0.02 * u[:n_ground, 2]
        # blobs
        lo = n_ground
        for j in range(blobs):
            hi = lo + n_blob // blobs if j < blobs - 1 else n_ground + n_blob
            c = par[8 + 3 * j: 11 + 3 * j] * np.array([0.8, 0.8, 0.4])
            r = 0.05 + 0.10 * (par[20 + j] + 1.0) * 0.5
            pts[lo:hi] = c + r * u[lo:hi]
            lo = hi
        # walls: vertical rectangles of random position, heading, length and height, 1 cm thick
        n_wall = N - lo
        for j in range(walls):
            hi = lo + n_wall // walls if j < walls - 1 else N
            cx, cy = par[24 + 2 * j] * 0.7, par[25 + 2 * j] * 0.7
            t = par[34 + j]                                        # heading by the rational parametrisation of the circle:
            ct, st = (1.0 - t * t) / (1.0 + t * t), 2.0 * t / (1.0 + t * t)   # + - * / only, bit-identical on every host
            half_len = 0.15 + 0.25 * (par[40 + j] + 1.0) * 0.5
            height = 0.3 + 0.5 * (par[46 + j] + 1.0) * 0.5
            s = u[lo:hi, 0] * half_len
            pts[lo:hi, 0] = cx + s * ct - 0.01 * u[lo:hi, 1] * st
            pts[lo:hi, 1] = cy + s * st + 0.01 * u[lo:hi, 1] * ct
            pts[lo:hi, 2] = -0.5 + (u[lo:hi, 2] + 1.0) * 0.5 * height
            lo = hi
        pts -= 0.5 * (pts.min(axis=0, keepdims=True) + pts.max(axis=0, keepdims=True))     # centre of the bounding box (exact ops)
        pts /= np.abs(pts).max() * 1.0001
        # a fixed pseudo-random permutation so that no stage sees the elements in blocks
        order = np.argsort(uniform(f"scene/{seed}/{b}/perm", N), kind="stable")
        out[b] = pts[order].astype(np.float32)
    return out

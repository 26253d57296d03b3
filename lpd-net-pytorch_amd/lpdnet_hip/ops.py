"""Tensor-level wrappers over the C-ABI (include/lpd_hip.h).  Forward-only building blocks;
autograd lives in autograd.py.  Every op requires CUDA fp32 tensors and raises otherwise --
there is deliberately no CPU or eager-PyTorch fallback.
"""
import ctypes
import os

import torch

from . import _debug, _lib

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3

class _PerThread(__import__("threading").local):
    """Measurement hooks, per host thread (a profile opened by one thread must not collect another thread's launches); read and
    written as module attributes (`ops.PROFILE = {}`): the module's class maps them onto this object."""
    # When set to a dict, every C-ABI call is bracketed by HIP events recorded on the launch stream
    # (torch's current stream): {label: [(start_event, end_event), ...]}.  Used by bench.py.
    PROFILE = None
    PROFILE_ONLY = None   # tuple of label prefixes: only those calls get events (90 event pairs per eval step cost 1.6 ms of host
                          # time -- more than the GPU needs for the step -- so a timed region brackets only what it reports)
    # Launch tape (engine.replay_eval): while a Tape is open on this thread every C-ABI call is also appended to it with its marshalled
    # arguments, every tensor whose address goes into a call is kept alive by it, and the zero-fills / stream joins that torch issues
    # between the calls are noted -- enough to re-issue the whole forward later without running any of the Python around the calls.
    TAPE = None


class Tape:
    CALL, ZERO, JOIN = 0, 1, 2

    def __init__(self):
        self.actions = []      # (CALL, fn, args) | (ZERO, tensor) | (JOIN, waiting stream, signalling stream, event)
        self.keep = []         # every tensor a recorded call points into


_TLS = _PerThread()


class _OpsModule(__import__("types").ModuleType):
    PROFILE = property(lambda self: _TLS.PROFILE, lambda self, v: setattr(_TLS, "PROFILE", v))
    PROFILE_ONLY = property(lambda self: _TLS.PROFILE_ONLY, lambda self, v: setattr(_TLS, "PROFILE_ONLY", v))


__import__("sys").modules[__name__].__class__ = _OpsModule


def _call(label, fn, *args):
    if _TLS.TAPE is not None:
        _TLS.TAPE.actions.append((Tape.CALL, fn, args))
    prof, only = _TLS.PROFILE, _TLS.PROFILE_ONLY
    if prof is None or (only is not None and not label.startswith(only)):
        rc = fn(*args)
        if rc != 0:
            _stat_ws_drop()
        _lib.check(rc, fn.__name__)
        return
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    rc = fn(*args)
    b.record()
    if rc != 0:
        _stat_ws_drop()
    _lib.check(rc, fn.__name__)
    prof.setdefault(label, []).append((a, b))


def _ptr(t):
    if t is None:
        return None
    if _TLS.TAPE is not None:
        _TLS.TAPE.keep.append(t)
    return ctypes.c_void_p(t.data_ptr())


def _zeros(shape, dtype, device):
    """torch.zeros for a workspace a kernel ACCUMULATES into: a launch of torch's, not ours -- noted on an open tape so that a replay
    re-zeroes the buffer."""
    t = torch.zeros(shape, dtype=dtype, device=device)
    if _TLS.TAPE is not None:
        _TLS.TAPE.actions.append((Tape.ZERO, t))
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """torch's current stream on the current device as a raw pointer.  torch.cuda.current_stream() builds a Stream object
    through three layers of Python (10 us a call, a quarter of the host time of an eval forward: measured with cProfile in round 2, HISTORY.md); the raw
    accessor is what torch's own code generators use."""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_STAT_WS = {}
_STAT_LOCK = __import__("threading").Lock()


def _stat_ws():
    """The `stat_ws` argument of the C-ABI's cross-workgroup reductions (include/lpd_hip.h, Conventions): a zero-filled
    lpd_stat_ws_bytes() buffer per (device, stream), owned here (torch's allocator), kept all-zero by the library between calls.
    Keyed by the raw stream handle: calls on one stream are ordered and share it; a recycled handle finds an all-zero buffer."""
    dev = _cur_device() if _cur_device is not None else torch.cuda.current_device()
    raw = _raw_stream(dev) if _raw_stream is not None else torch.cuda.current_stream().cuda_stream
    t = _STAT_WS.get((dev, raw))
    if t is None:
        n = int(_lib.load().lpd_stat_ws_bytes()) // 8
        t = torch.zeros(n, dtype=torch.float64, device=torch.device("cuda", dev))      # memset on the current stream: ordered
        if torch.cuda.is_current_stream_capturing():
            # graph-private memory: not cached beyond the capture (the memset is a graph node); the tensor stays referenced until this
            # thread's next call outside a capture, so the block cannot be handed out again in front of the launch that uses it
            keep = _TLS.__dict__.setdefault("capture_keep", [])
            keep.append(t)
            return ctypes.c_void_p(t.data_ptr())
        with _STAT_LOCK:
            t = _STAT_WS.setdefault((dev, raw), t)
    if _TLS.__dict__.get("capture_keep") and not torch.cuda.is_current_stream_capturing():
        _TLS.__dict__["capture_keep"] = []
    return ctypes.c_void_p(t.data_ptr())


def _stat_ws_drop():
    """An entry point returned an error: if it had launched its reduction before failing, the cached workspace of this (device, stream)
    is no longer all-zero and every later statistic on the stream would be silently wrong.  Forget it: the next call zero-fills a new one."""
    dev = _cur_device() if _cur_device is not None else torch.cuda.current_device()
    raw = _raw_stream(dev) if _raw_stream is not None else torch.cuda.current_stream().cuda_stream
    with _STAT_LOCK:
        _STAT_WS.pop((dev, raw), None)


def _req(t, name, dtype=torch.float32):
    if t is None:
        return
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise _lib.LpdHipError(
            f"{name}: tensor is on {t.device}; the LPD-Net HIP path runs on MI355X only (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def _rows(t, name):
    """2-D view with unit column stride -> (tensor, leading dim)."""
    _req(t, name)
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D tensor with contiguous rows, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0)


def _vec(t, name, n):
    if t is None:
        return None
    _req(t, name)
    if t.numel() != n:
        raise ValueError(f"{name}: expected {n} elements, got {t.numel()}")
    return t.contiguous()


KNN_IMPL = _debug.value("knn-impl", 0)   # A/B switch for benchmarking (0 = product kernel)


def knn_workspace_floats(B, C, N, k=20):
    """Workspace of lpd_knn in floats (squared norms, packed MFMA operand image, tile statistics of the best-first path)."""
    return int(_lib.load().lpd_knn_workspace_floats(B, C, N, k))


def knn(x_cm, k, impl=None):
    """x_cm [B,C,N] channel-major fp32 -> idx [B,N,k] int32 (reference util/lpdnet_model.py:317-326)."""
    if impl is None:
        impl = KNN_IMPL
    _req(x_cm, "x")
    if x_cm.dim() != 3:
        raise ValueError("knn: expected [B,C,N]")
    x_cm = x_cm.contiguous()
    B, C, N = x_cm.shape
    idx = torch.empty((B, N, k), dtype=torch.int32, device=x_cm.device)
    ws = torch.empty((knn_workspace_floats(B, C, N),), dtype=torch.float32, device=x_cm.device)
    lib = _lib.load()
    _call(f"knn[C={C},k={k}]", lib.lpd_knn, _ptr(x_cm), B, C, N, k, _ptr(idx), _ptr(ws), impl, _stream())
    return idx


def lpdnet_front(xyz, W1, s1, b1, W2, s2, b2, B, N, k, act=ACT_NONE, slope=0.01):
    """F0 = act(BN2(conv2(act(BN1(conv1(xyz)))))) [B*N, 64] of the LPD-Net trunk in one launch, exact fp32, together with the kNN
    operands of F0 (lpd_lpdnet_front).  Returns (F0, ws): pass ws to knn_prepared."""
    _req(xyz, "xyz")
    ldx = _rows(xyz, "xyz")
    M = xyz.shape[0]
    if M != B * N or N % 128:
        raise ValueError("lpdnet_front: rows != B*N or N % 128 != 0")
    W1 = W1.reshape(64, 3).contiguous()
    W2 = W2.reshape(64, 64).contiguous()
    f0 = torch.empty((M, 64), dtype=torch.float32, device=xyz.device)
    ws = torch.empty((knn_workspace_floats(B, 64, N),), dtype=torch.float32, device=xyz.device)
    lib = _lib.load()
    s1, b1, s2, b2 = _vec(s1, "s1", 64), _vec(b1, "b1", 64), _vec(s2, "s2", 64), _vec(b2, "b2", 64)      # named: they outlive the launch call
    _call("lpdnet_front", lib.lpd_lpdnet_front, _ptr(xyz), ldx, _ptr(W1), _ptr(s1), _ptr(b1), _ptr(W2),
          _ptr(s2), _ptr(b2), act, float(slope), _ptr(f0), B, N, k, _ptr(ws), _stream())
    return f0, ws


KNN_PM_PREPARED = 256


def knn_prepared(ws, B, N, k, impl=None):
    """kNN graph from the operands lpdnet_front left in ws (64 channels)."""
    if impl is None:
        impl = KNN_IMPL
    idx = torch.empty((B, N, k), dtype=torch.int32, device=ws.device)
    lib = _lib.load()
    _call(f"knn[C=64,k={k}]", lib.lpd_knn_pm, None, 64, B, 64, N, k, _ptr(idx), _ptr(ws), impl | KNN_PM_PREPARED, _stream())
    return idx


def knn_pm(x_pm, B, N, k, impl=None):
    """kNN on point-major rows x_pm [B*N, C] (C <= 64, k <= 32): same indices as knn() on the transposed tensor, without
    the transpose + pack round trip."""
    if impl is None:
        impl = KNN_IMPL
    ld = _rows(x_pm, "x_pm")
    M, C = x_pm.shape
    if M != B * N:
        raise ValueError("knn_pm: rows != B*N")
    idx = torch.empty((B, N, k), dtype=torch.int32, device=x_pm.device)
    ws = torch.empty((knn_workspace_floats(B, C, N),), dtype=torch.float32, device=x_pm.device)
    lib = _lib.load()
    _call(f"knn[C={C},k={k}]", lib.lpd_knn_pm, _ptr(x_pm), ld, B, C, N, k, _ptr(idx), _ptr(ws), impl, _stream())
    return idx


# Dense products run as split-bf16 ("bf16x3": three bf16 MFMA products per term, fp32 accumulate; ~2e-6 on the
# descriptors, csrc/lpd_gemm.hip) unless the call asks for exact fp32 (layers whose output feeds the kNN) or this switch
# is off (LPD_GEMM_FP32=1: every product on the f32-input MFMA, bit-for-bit the round-1 numerics).
GEMM_BF16X3 = __import__("os").environ.get("LPD_GEMM_FP32", "0") != "1"


class _ExactState(__import__("threading").local):   # per thread: nn.DataParallel runs one forward per device thread
    depth = 0


_EXACT = _ExactState()


# Train-mode FORWARD products: batch statistics over a handful of clouds (B = 6 in the step-0 fixtures) amplify GEMM rounding
# ~30x -- with split-bf16 products the train-mode descriptors of those fixtures sit 1.3e-4 from the fp64 oracle (the fp32 reference
# itself at 0.5e-4) -- so batches of fewer than TRAIN_FWD_X3_MIN_CLOUDS clouds keep the exact f32-input MFMA (at 16 clouds the flip-free gradient gate of 3e-3 still sees 4.8e-3).  From 32 clouds on
# (BASELINE configs[2]: 44) the head's BatchNorms average over enough rows: measured on the reference's B = 44, N = 4096 step-0
# fixture the split-bf16 forward stays as close to the reference's fp64 forward as the exact one (tests/test_train_gpu.py,
# cfg2 test), and the forward products of a step cost 2.2 ms instead of 4.8.  LPD_DEBUG=train-fwd-x3=1 / 0: always / never.
# Eval forward and every backward product use the fast form regardless.
_tfx = _debug.value("train-fwd-x3", "auto")
TRAIN_FWD_BF16X3 = "auto" if _tfx == "auto" else (_tfx == "1")
TRAIN_FWD_X3_MIN_CLOUDS = 32


class _FastState(__import__("threading").local):
    depth = 0


_FAST = _FastState()


# Measured on the bf16 training step (B = 44, N = 4096): one product instead of three changes the fp32-operand GEMMs by
# -0.4 ms of 18.6 (conv3 forward 0.70 -> 0.56 ms, its dX 0.57 -> 0.43; the k-major weight-gradient products not at all: they are
# bound by staging, not by the MFMA) and costs another factor ~2 on the loss error, so it is off unless LPD_DEBUG=bf16-x1=1.
BF16_SINGLE_PRODUCT = _debug.on("bf16-x1", False)


class bf16_gemm:
    """with ops.bf16_gemm(): (LPD_DEBUG=bf16-x1=1 only) split-bf16 products on fp32 operands run with ONE bf16 product per term
    (operands rounded to bf16, fp32 accumulation) instead of three.  exact_gemm() regions inside stay exact."""

    def __enter__(self):
        self.on = BF16_SINGLE_PRODUCT
        if self.on:
            _FAST.depth += 1

    def __exit__(self, *exc):
        if self.on:
            _FAST.depth -= 1
        return False


class exact_gemm:
    """with ops.exact_gemm(): every GEMM inside runs on the f32-input MFMA (the layers in front of the feature-space
    kNN: the neighbour indices must not depend on the GEMM precision switch)."""

    def __enter__(self):
        _EXACT.depth += 1

    def __exit__(self, *exc):
        _EXACT.depth -= 1
        return False


class train_forward_gemm(exact_gemm):
    """exact_gemm for the forward of the training autograd Functions on small batches (see TRAIN_FWD_BF16X3); clouds = B."""

    def __init__(self, clouds=None):
        self.clouds = clouds

    def __enter__(self):
        x3 = (self.clouds is not None and self.clouds >= TRAIN_FWD_X3_MIN_CLOUDS) if TRAIN_FWD_BF16X3 == "auto" else bool(TRAIN_FWD_BF16X3)
        self.on = not x3
        if self.on:
            super().__enter__()

    def __exit__(self, *exc):
        if self.on:
            super().__exit__(*exc)
        return False


_FRAG_CACHE = {}
_FRAG_LOCK = __import__("threading").Lock()     # nn.DataParallel-style callers: one forward per device thread
X3W_FORWARD = _debug.on("x3w-fwd")    # forward layers with K >= 256 on the prepared-fragment kernel
X3W_IMPL = _debug.value("x3w-impl", 0)         # lpd_gemm_x3w impl (0 = by shape); benchmarking only
X3W_BATCHED = _debug.on("x3w-batched")  # batched deep-reduction products with per-problem k-major weights on lpd_gemm_x3w_batched
X3T_PANELS = _debug.on("x3t")          # short-reduction panel-to-panel products on lpd_gemm_x3t
X3T_ROWS = _debug.on("x3t-rows")       # ... and row-major ones (K = 64 / 128) on lpd_gemm_x3t_rows


def _weight_frags(B2, b_kmajor, N, K):
    """hi/lo bf16 MFMA fragments of a weight matrix (lpd_gemm_prep_b).  Cached only for nn.Parameters (or views of one):
    their storage lives as long as the cache entry's reference and every update bumps `_version`; anything else (derived
    weights, activations) may reuse an address with a fresh version counter, so it is prepared per call (~5 us) -- unless
    the engine marks it `_lpd_stable` (a derived weight it caches itself: ONE tensor object per version of its sources,
    e.g. the stacked [neighbour ; centre] edge weights), which the cache entry then keeps alive like a parameter."""
    base = B2._base if B2._base is not None else B2
    cacheable = isinstance(base, torch.nn.Parameter) or bool(getattr(base, "_lpd_stable", False))
    key = (B2.data_ptr(), base._version, tuple(B2.shape), B2.stride(0), bool(b_kmajor)) if cacheable else None
    if cacheable:
        with _FRAG_LOCK:
            hit = _FRAG_CACHE.get(key)
        if hit is not None:
            consumer_sync(hit[2], hit[0])
            return hit[0]
    lib = _lib.load()
    frags = torch.empty((int(lib.lpd_gemm_prep_b_bytes(N, K)),), dtype=torch.uint8, device=B2.device)
    _call("gemm_prep_b", lib.lpd_gemm_prep_b, _ptr(B2), B2.stride(0), int(bool(b_kmajor)), N, K, _ptr(frags), _stream())
    if cacheable:
        mark = producer_mark()
        with _FRAG_LOCK:
            if len(_FRAG_CACHE) > 256:
                _FRAG_CACHE.clear()
            _FRAG_CACHE[key] = (frags, base, mark)     # the reference keeps the parameter's storage from being recycled under the key
    return frags


def producer_mark():
    """(raw stream, event) behind the launches that filled a cache entry: a later hit from ANOTHER stream (a second host thread
    running the same module on its own stream) must not read the entry before those launches are done."""
    if torch.cuda.is_current_stream_capturing():
        return None
    ev = torch.cuda.Event()
    ev.record()
    return (_stream().value, ev)


def consumer_sync(mark, *tensors):
    if mark is not None and mark[0] != _stream().value:
        cur = torch.cuda.current_stream()
        cur.wait_event(mark[1])
        for t in tensors:
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)


def gemm(A, B, *, a_kmajor=False, b_kmajor=True, bias=None, scale=None, shift=None, act=ACT_NONE, slope=0.01,
         out=None, splits=1, accumulate=False, exact=False, a_panels=False, out_panels=False, out_bf16=False, a_affine=None):
    """Single (2-D) or batched (3-D) GEMM with fused epilogue.

    out_bf16: the result is a bfloat16 tensor (built on the transposed short-reduction kernel only: lpd_gemm_x3t_rows).
    A may be bfloat16 rows for the batched per-problem-weight product (lpd_gemm_x3w_batched) -- the bf16-storage training mode's
    conv3-map tensors; every other combination raises.
    a_affine = (scale [K], shift [K], act, slope): A's rows are act(scale * A + shift) -- applied in the operand loader of the batched
    per-problem-weight product (nothing stored; bf16 rows: rounded to bf16 again); every other path raises.

    a_panels / out_panels: A / out are cloud-panel tensors [B, cols/8, N, 8] (the layout the cloud-resident K-agg kernel
    streams; see panels_empty / panels_to_rows) instead of row-major matrices.

    A: [M,K] (a_kmajor False) or [K,M] (True); B: [K,N] (b_kmajor True) or [N,K] (False).
    3-D inputs add a leading batch dim (must be contiguous in that dim ordering).
    """
    a16 = isinstance(A, torch.Tensor) and A.dtype == torch.bfloat16
    _req(A, "A", torch.bfloat16 if a16 else torch.float32)
    _req(B, "B")
    if a_panels or out_panels:
        return _gemm_panels(A, B, a_kmajor, b_kmajor, bias, scale, shift, act, slope, out, accumulate, exact, a_panels, out_panels)
    batched = A.dim() == 3
    if batched:
        if B.dim() != 3 or A.shape[0] != B.shape[0]:
            raise ValueError("gemm: batched A needs batched B with the same batch")
        nb = A.shape[0]
        A2, B2 = A[0], B[0]
        sA, sB = A.stride(0), B.stride(0)
    else:
        nb, A2, B2, sA, sB = 1, A, B, 0, 0
    if a16 and (A2.dim() != 2 or A2.stride(1) != 1):
        raise ValueError("gemm: bf16 A needs contiguous rows")
    lda, ldb = A2.stride(0) if a16 else _rows(A2, "A"), _rows(B2, "B")
    M, K = (A2.shape[1], A2.shape[0]) if a_kmajor else (A2.shape[0], A2.shape[1])
    Kb, N = (B2.shape[0], B2.shape[1]) if b_kmajor else (B2.shape[1], B2.shape[0])
    if K != Kb:
        raise ValueError(f"gemm: inner dims differ ({K} vs {Kb})")
    if out is None:
        out = torch.empty((nb, M, N) if batched else (M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=A.device)
    _req(out, "out", torch.bfloat16 if out_bf16 else torch.float32)
    O2 = out[0] if batched else out
    if O2.dim() != 2 or O2.stride(1) != 1:
        raise ValueError("gemm: out needs contiguous rows")
    ldc = O2.stride(0)
    if O2.shape[0] != M or O2.shape[1] != N:
        raise ValueError("gemm: out has the wrong shape")
    sC = out.stride(0) if batched else 0
    ws = None
    if splits > 1:
        ws = torch.empty((nb * splits * M * N,), dtype=torch.float32, device=A.device)
    bias, scale, shift = _vec(bias, "bias", N), _vec(scale, "scale", N), _vec(shift, "shift", N)
    lib = _lib.load()
    # weight-shaped B (a layer's weights in either layout; dX = dY W of the backward pass) and a deep reduction:
    # fragments of B prepared once, B never staged through LDS (lpd_gemm_x3w).  Measured (a one-off script of round 2, HISTORY.md 3.2; M = 131072):
    # conv3 512 -> 1024 660 -> 505 us, dX 1024 -> 512 433 us; at K <= 128 the generic kernel is as fast or faster
    # (SN1 projection 142 vs 149 us, DG1 projection 52 vs 58 us), and k-major weights transpose in registers there.
    # a short reduction (K = 64 / 128) over many rows: the transposed product lpd_gemm_x3t_rows (data rows as the MFMA's B operand, one
    # barrier, float4 stores) -- the 128 x 128 block kernel spends it in barriers and 4-byte stores (SN1 projection of the training step 211 us)
    if (GEMM_BF16X3 and X3T_ROWS and not exact and _EXACT.depth == 0 and (_FAST.depth == 0 or out_bf16) and not a_kmajor and splits == 1 and not a16
            and a_affine is None and not accumulate and nb * M >= 16384 and nb <= 65535 and lib.lpd_gemm_x3t_rows_applies(M, N, K, act, lda, ldc)
            and A.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0 and sA % 4 == 0 and sC % (8 if out_bf16 else 4) == 0
            and (not out_bf16 or ldc % 8 == 0)):
        if batched:      # per-problem weights (the NetVLAD backward's [a | dA0] . [dVraw_b | Wc]^T): their fragments, every call
            fb = int(lib.lpd_gemm_prep_b_bytes(N, K))
            frags = torch.empty((nb * fb,), dtype=torch.uint8, device=A.device)
            _call("gemm_prep_b", lib.lpd_gemm_prep_b_batch, _ptr(B), ldb, int(bool(b_kmajor)), N, K, nb, sB, _ptr(frags), _stream())
        else:
            fb, frags = 0, _weight_frags(B, b_kmajor, N, K)
        _call(f"gemmx3t[{M}x{N}x{K}]" + (f"x{nb}" if batched else ""), lib.lpd_gemm_x3t_rows, _ptr(A), lda, _ptr(frags), _ptr(out), ldc,
              int(bool(out_bf16)), M, N, K, _ptr(bias), _ptr(scale), _ptr(shift), act, float(slope), nb, sA, sC, fb, _stream())
        return out
    if out_bf16:
        raise ValueError(f"gemm: a bf16 result is built on the transposed short-reduction kernel only (M={M}, N={N}, K={K}, batch {nb})")
    if (GEMM_BF16X3 and X3W_BATCHED and batched and not exact and _EXACT.depth == 0 and not a_kmajor and b_kmajor and splits == 1 and not accumulate
            and bias is None and scale is None and act == ACT_NONE and M % 128 == 0 and nb * M >= 16384 and 64 <= N <= 128 and K >= 256
            and N * K <= (1 << 22) and sA == M * lda and sC == M * ldc and nb * M < (1 << 31) and (not a16 or (K % 32 == 0 and lda % 4 == 0))):
        # per-problem k-major weights over consecutive row ranges of one row-major A (NetVLAD backward: dA[b] = x[b] . dV[b]): the
        # prepared-fragment kernel with a fragment set per problem; the generic batched kernel ran this at 2.6 TB/s of the 738-MB operand
        fb = int(lib.lpd_gemm_prep_b_bytes(N, K))
        frags = torch.empty((nb * fb,), dtype=torch.uint8, device=A.device)
        _call("gemm_prep_b", lib.lpd_gemm_prep_b_batch, _ptr(B), ldb, 1, N, K, nb, sB, _ptr(frags), _stream())
        if a_affine is not None:
            a_sc, a_sh, a_act, a_slope = _vec(a_affine[0], "a_affine scale", K), _vec(a_affine[1], "a_affine shift", K), a_affine[2], a_affine[3]
        else:
            a_sc = a_sh = None
            a_act, a_slope = ACT_NONE, 0.0
        _call(f"gemmx3w[{M}x{N}x{K}]x{nb}", lib.lpd_gemm_x3w_batched, _ptr(A), lda, int(a16), _ptr(frags), fb, M, _ptr(out), ldc, nb * M, N, K,
              _ptr(a_sc), _ptr(a_sh), a_act, float(a_slope), 16 if (_FAST.depth > 0 and not a16 and a_affine is None) else 0, _stream())
        return out
    if a_affine is not None:
        raise ValueError(f"gemm: an operand transform is built for the batched per-problem-weight product only (M={M}, N={N}, K={K}, batch {nb})")
    if a16:
        raise ValueError(f"gemm: bf16 rows as A are built for the batched per-problem-weight product only (M={M}, N={N}, K={K}, batch {nb})")
    if (GEMM_BF16X3 and not exact and _EXACT.depth == 0 and not a_kmajor and not batched and splits == 1
            and M >= 1024 and N >= 64 and N * K <= (1 << 22)
            and ((b_kmajor and K >= 128) or (X3W_FORWARD and K >= 256))):
        frags = _weight_frags(B, b_kmajor, N, K)
        fast = _FAST.depth > 0
        _call(f"gemmx{'1' if fast else '3'}w[{M}x{N}x{K}]", lib.lpd_gemm_x3w, _ptr(A), lda, _ptr(frags), _ptr(out), ldc, M, N, K, _ptr(bias), _ptr(scale),
              _ptr(shift), act, float(slope), int(bool(accumulate)), 0, 0, 0, 0, X3W_IMPL | (16 if fast else 0), _stream())
        return out
    # split-bf16 where it is faster (measured, tools/gemm_bench.py): outputs of at least 128 x 128 with a row-major A
    # or with both operands k-major (weight gradients); skinny outputs (per-cloud rows, 64 clusters) stay on the f32-input MFMA
    # (the batched k-major pooling product act^T x, 64 clusters wide: 1 % of the eval step faster in split form)
    x3 = (GEMM_BF16X3 and not exact and _EXACT.depth == 0 and M >= 128 and (not a_kmajor or b_kmajor)
          and N >= (64 if (batched and a_kmajor and b_kmajor) else 128))
    x1 = x3 and _FAST.depth > 0
    _call(f"gemm{('x1' if x1 else 'x3') if x3 else ''}[{M}x{N}x{K}]", (lib.lpd_gemm_bf16x1 if x1 else lib.lpd_gemm_bf16x3) if x3 else lib.lpd_gemm,
          _ptr(A), _ptr(B), _ptr(out), M, N, K, lda, ldb, ldc, int(a_kmajor), int(b_kmajor), nb,
          sA, sB, sC, splits, _ptr(ws), _ptr(bias), _ptr(scale), _ptr(shift), act, float(slope),
          int(bool(accumulate)), 0, 0, 0, 0, _stream())
    return out


PANEL_PAD_ROWS = 8   # 256 B between consecutive panels: a power-of-two panel stride would put them all on the same HBM channels


def panels_empty(B, N, C, device):
    """Uninitialised CLOUD-PANEL activation buffer, logically [B, C/8, N, 8]: element (cloud b, point n, channel c) at
    [b, c // 8, n, c % 8].  A channel range c0:c1 (multiples of 8) is the view buf[:, c0//8:c1//8].  Panels are padded to
    N + PANEL_PAD_ROWS rows in memory (the returned tensor is the [:, :, :N] view)."""
    return torch.empty((B, C // 8, N + PANEL_PAD_ROWS, 8), dtype=torch.float32, device=device)[:, :, :N]


def panels_to_rows(P):
    """cloud-panel [B, C/8, N, 8] -> row-major [B*N, C] copy (diagnostics / tests)."""
    Bc, Pn, N, _ = P.shape
    return P.permute(0, 2, 1, 3).reshape(Bc * N, Pn * 8).contiguous()


def rows_to_panels(X, B):
    """row-major [B*N, C] -> cloud-panel [B, C/8, N, 8] copy (tests)."""
    M, C = X.shape
    out = panels_empty(B, M // B, C, X.device)
    out.copy_(X.reshape(B, M // B, C // 8, 8).permute(0, 2, 1, 3))
    return out


def split_panels_empty(B, N, C, device):
    """Uninitialised SPLIT cloud-panel buffer: the two bf16 planes (hi = bf16(x), lo = bf16(x - hi)) of a [B, C/8, N, 8] activation
    tensor, as one tensor [2, B, C/8, N(+pad), 8] (the [:, :, :, :N] view is returned): what lpd_gemm_p8 reads."""
    return torch.empty((2, B, C // 8, N + PANEL_PAD_ROWS, 8), dtype=torch.bfloat16, device=device)[:, :, :, :N]


def split_panels(P, out=None):
    """fp32 cloud panels [B, C/8, N, 8] -> split bf16 planes [2, B, C/8, N, 8] (lpd_split_panels)."""
    if not _is_panels(P):
        raise ValueError("split_panels: expected a cloud-panel [B, C/8, N, 8] tensor")
    Bc, Pn, N, _ = P.shape
    if out is None:
        out = split_panels_empty(Bc, N, Pn * 8, P.device)
    lib = _lib.load()
    _call("split_panels", lib.lpd_split_panels, _ptr(P), P.stride(0), P.stride(1) // 8, _ptr(out[0]), _ptr(out[1]), out.stride(1),
          out.stride(2) // 8, Bc, Pn, N, _stream())
    return out


def _is_split(t):
    """a (panel-range view of a) split cloud-panel buffer: [2, B, P, N, 8] bf16"""
    return (t is not None and t.dim() == 5 and t.dtype == torch.bfloat16 and t.shape[0] == 2 and t.shape[4] == 8 and t.stride(4) == 1
            and t.stride(3) == 8 and t.stride(2) % 8 == 0 and t.stride(2) >= t.shape[3] * 8)


def split_to_rows(S):
    """split cloud panels [2, B, C/8, N, 8] -> row-major fp32 [B*N, C] (hi + lo; diagnostics / tests)."""
    x = S[0].float() + S[1].float()
    Bc, Pn, N, _ = x.shape
    return x.permute(0, 2, 1, 3).reshape(Bc * N, Pn * 8).contiguous()


P8_IMPL = _debug.value("p8-impl", 0)


def gemm_x3t_split(S, W, *, out=None):
    """S [2, B, K/8, N, 8] split planes x W [Nout, K] -> fp32 cloud panels [B, Nout/8, N, 8] (the bare product: the neighbour / centre
    projections of the split SN1 edge convolution from the x2 block as the fused edge MLP writes it for conv3)."""
    _req(W, "W")
    if not _is_split(S):
        raise ValueError("gemm_x3t_split: expected split cloud panels [2, B, K/8, N, 8] (bf16)")
    _, Bc, Pn, Np, _ = S.shape
    M, K, N = Bc * Np, Pn * 8, W.shape[0]
    if out is None:
        out = panels_empty(Bc, Np, N, S.device)
    lib = _lib.load()
    if (W.shape[1] != K or not _is_panels(out) or out.shape[1] * 8 != N or out.shape[2] != Np or out.stride(1) != S.stride(2)
            or not lib.lpd_gemm_x3t_applies(M, N, K, 0, S.stride(1), out.stride(0), Np)):
        raise ValueError(f"gemm_x3t_split: shape not built (M={M}, N={N}, K={K}, points per cloud {Np})")
    frags = _weight_frags(W, False, N, K)
    _call(f"gemmx3t[{M}x{N}x{K}]", lib.lpd_gemm_x3ts, _ptr(S[0]), S.stride(0), _ptr(frags), _ptr(out), M, N, K, None, None, None, 0, 0.0,
          S.stride(1), out.stride(0), Np, out.stride(1) // 8, _stream())
    return out


def gemm_p8(S, W, *, scale=None, shift=None, act=ACT_NONE, slope=0.01, out=None, out_panels=False, assign_w=None):
    """act(scale * (A W^T) + shift) for split cloud-panel activations S [2, B, K/8, N, 8] (split_panels / the producers' split
    epilogues) and a weight W [Nout, K] (torch conv layout): lpd_gemm_p8.  out: row-major [B*N, Nout] or fp32 cloud panels.
    The kernel adds a bias only: `scale` is folded into W here (per call -- callers with a fixed scale pass the folded weight,
    engine.conv3_folded, whose fragments are cached).
    assign_w [Nout, 64] (NetVLAD cluster_weights): also returns parts [Nout/256, B*N, 64], the per-column-block partial products
    out[:, 256 j : 256 j + 256] @ assign_w[256 j : 256 j + 256] computed in the same launch -> (out, parts)."""
    _req(W, "W")
    if not _is_split(S):
        raise ValueError("gemm_p8: expected split cloud panels [2, B, K/8, N, 8] (bf16)")
    _, Bc, Pn, Np, _ = S.shape
    M, K, N = Bc * Np, Pn * 8, W.shape[0]
    lib = _lib.load()
    if W.dim() != 2 or W.shape[1] != K or not lib.lpd_gemm_p8_applies(M, N, K, Np):
        raise ValueError(f"gemm_p8: shape not built (M={M}, N={N}, K={K}, points per cloud {Np})")
    if out is None:
        out = panels_empty(Bc, Np, N, S.device) if out_panels else torch.empty((M, N), dtype=torch.float32, device=S.device)
    if out_panels:
        if not _is_panels(out) or out.shape[1] * 8 != N or out.shape[2] != Np or out.shape[0] != Bc:
            raise ValueError("gemm_p8: out_panels expects a cloud-panel [B, Nout/8, N, 8] tensor")
        ldc, c_cloud, c_ld = 8, out.stride(0), out.stride(1) // 8
    else:
        _req(out, "out")
        if out.shape[0] != M or out.shape[1] != N:
            raise ValueError("gemm_p8: out has the wrong shape")
        ldc, c_cloud, c_ld = _rows(out, "out"), 0, 0
    scale, shift = _vec(scale, "scale", N), _vec(shift, "shift", N)
    if scale is not None:
        W = (W * scale.unsqueeze(1)).contiguous()
    if act == ACT_LEAKY and not 0.0 <= slope <= 1.0:
        raise ValueError("gemm_p8: LeakyReLU slope outside [0, 1]")
    frags = _weight_frags(W, False, N, K)
    if assign_w is not None:
        _req(assign_w, "assign_w")
        if assign_w.dim() != 2 or tuple(assign_w.shape) != (N, 64) or assign_w.stride(1) != 1:
            raise ValueError(f"gemm_p8: assign_w must be [{N}, 64] with unit column stride")
        frags2 = _weight_frags(assign_w, True, 64, N)
        parts = torch.empty((N // 256, M, 64), dtype=torch.float32, device=S.device)
        _call(f"gemm_p8+assign[{M}x{N}x{K}]", lib.lpd_gemm_p8_fused, _ptr(S[0]), _ptr(S[1]), S.stride(1), S.stride(2) // 8, _ptr(frags),
              _ptr(out), ldc, c_cloud, c_ld, M, N, K, Np, _ptr(shift), act, float(slope), _ptr(frags2), _ptr(parts), parts.stride(0), _stream())
        return out, parts
    _call(f"gemm_p8[{M}x{N}x{K}]", lib.lpd_gemm_p8, _ptr(S[0]), _ptr(S[1]), S.stride(1), S.stride(2) // 8, _ptr(frags), _ptr(out), ldc,
          c_cloud, c_ld, M, N, K, Np, _ptr(shift), act, float(slope), P8_IMPL, _stream())
    return out


def _is_panels(t):
    """a (view of a) cloud-panel buffer: [B, P, N, 8] with strides (anything, panel_ld * 8 >= N * 8, 8, 1)"""
    return (t is not None and t.dim() == 4 and t.shape[3] == 8 and t.stride(3) == 1 and t.stride(2) == 8
            and t.stride(1) % 8 == 0 and t.stride(1) >= t.shape[2] * 8)


def _panel_ld(*tensors):
    """rows per panel in memory; all cloud-panel operands of one call must agree"""
    lds = {t.stride(1) // 8 for t in tensors if t is not None and t.dim() == 4}
    if len(lds) > 1:
        raise ValueError("cloud-panel operands of one call must share the panel stride")
    return lds.pop() if lds else 0


def _gemm_panels(A, B, a_kmajor, b_kmajor, bias, scale, shift, act, slope, out, accumulate, exact, a_panels, out_panels):
    if a_kmajor or B.dim() != 2:
        raise ValueError("gemm: cloud-panel operands need a_kmajor=False and a 2-D B")
    a_cloud = c_cloud = 0
    if a_panels:
        if not _is_panels(A):
            raise ValueError("gemm: a_panels expects a cloud-panel [B, K/8, N, 8] tensor")
        nc, Np = A.shape[0], A.shape[2]
        M, K, lda, a_cloud = nc * Np, A.shape[1] * 8, 8, A.stride(0)
    else:
        lda = _rows(A, "A")
        M, K = A.shape
        nc = Np = None
    ldb = _rows(B, "B")
    Kb, N = (B.shape[0], B.shape[1]) if b_kmajor else (B.shape[1], B.shape[0])
    if K != Kb:
        raise ValueError(f"gemm: inner dims differ ({K} vs {Kb})")
    if out_panels:
        if out is None:
            if nc is None:
                raise ValueError("gemm: out_panels without a_panels needs an explicit cloud-panel out")
            out = panels_empty(nc, Np, N, A.device)
        if not _is_panels(out) or out.shape[1] * 8 != N or out.shape[0] * out.shape[2] != M:
            raise ValueError("gemm: out_panels expects a cloud-panel [B, N/8, Np, 8] tensor")
        if Np is not None and out.shape[2] != Np:
            raise ValueError("gemm: A and out disagree on the points per cloud")
        Np = out.shape[2]
        ldc, c_cloud = 8, out.stride(0)
    else:
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=A.device)
        ldc = _rows(out, "out")
    if K % 32 or Np % 128:
        raise ValueError("gemm: cloud-panel operands need K % 32 == 0 and points per cloud % 128 == 0")
    _req(out, "out")
    bias, scale, shift = _vec(bias, "bias", N), _vec(scale, "scale", N), _vec(shift, "shift", N)
    lib = _lib.load()
    x3 = GEMM_BF16X3 and not exact and _EXACT.depth == 0 and N >= 128 and M >= 128
    pld = _panel_ld(A if a_panels else None, out if out_panels else None)
    if (x3 and X3T_PANELS and not accumulate and M >= 1024
            and lib.lpd_gemm_x3t_applies(M, N, K, act, a_cloud, c_cloud, Np)):
        # short reduction, panels in and out (the SN1 projection): the transposed product, no LDS (164 -> see DESIGN.md 9.6a)
        frags = _weight_frags(B, b_kmajor, N, K)
        _call(f"gemmx3t[{M}x{N}x{K}]", lib.lpd_gemm_x3t, _ptr(A), _ptr(frags), _ptr(out), M, N, K, _ptr(bias), _ptr(scale), _ptr(shift),
              act, float(slope), a_cloud, c_cloud, Np, pld, _stream())
        return out
    if x3 and X3W_FORWARD and M >= 1024 and K >= 256 and N * K <= (1 << 22):
        frags = _weight_frags(B, b_kmajor, N, K)
        _call(f"gemmx3w[{M}x{N}x{K}]", lib.lpd_gemm_x3w, _ptr(A), lda, _ptr(frags), _ptr(out), ldc, M, N, K, _ptr(bias), _ptr(scale),
              _ptr(shift), act, float(slope), int(bool(accumulate)), a_cloud, c_cloud, Np, pld, X3W_IMPL, _stream())
        return out
    _call(f"gemm{'x3' if x3 else ''}[{M}x{N}x{K}]", lib.lpd_gemm_bf16x3 if x3 else lib.lpd_gemm, _ptr(A), _ptr(B), _ptr(out), M, N, K, lda, ldb,
          ldc, 0, int(b_kmajor), 1, 0, 0, 0, 1, None, _ptr(bias), _ptr(scale), _ptr(shift), act, float(slope), int(bool(accumulate)),
          a_cloud, c_cloud, Np, pld, _stream())
    return out


def linear(x, w, *, bias=None, scale=None, shift=None, act=ACT_NONE, slope=0.01, out=None, splits=None, exact=False):
    """y = act((x @ w.T + bias) * scale + shift); x [M,K] rows, w [N,K] (torch Linear/Conv1x1 layout).

    splits None: the per-cloud fully-connected layers (M = B rows, K >= 512: T-Net fc1/fc2) run split-K over 128-deep
    slices -- one block per output tile would leave the chip idle, and the short partial sums are also more accurate."""
    ldx = _rows(x, "x")
    _req(w, "w")
    w = w.reshape(w.shape[0], -1).contiguous()
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"linear: weight has {w.shape[1]} inputs, x has {K}")
    if K <= 8:
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=x.device)
        ldy = _rows(out, "out")
        bias, scale, shift = _vec(bias, "bias", N), _vec(scale, "scale", N), _vec(shift, "shift", N)
        lib = _lib.load()
        _call("linear_smallk", lib.lpd_linear_smallk, _ptr(x), ldx, _ptr(w), K, 1, 0, 0, _ptr(out), ldy, M, N, K, _ptr(bias),
                                         _ptr(scale), _ptr(shift), act, float(slope), _stream())
        return out
    if splits is None:
        splits = K // 128 if (M <= 128 and K >= 512) else 1
    return gemm(x, w, a_kmajor=False, b_kmajor=False, bias=bias, scale=scale, shift=shift, act=act, slope=slope,
                out=out, splits=splits, exact=exact)


def apply_transform(x, trans, rows_per_cloud, out=None):
    """Per-cloud alignment product y[m] = x[m] @ trans[m // rows_per_cloud]; x [M,K], trans [B,K,K].

    K <= 8 -> small-K kernel; otherwise the batched MFMA GEMM (K % 32 == 0).
    (util/lpdnet_model.py:229,240; util/PointNetVlad.py:209,223)
    """
    _rows(x, "x")
    _req(trans, "trans")
    trans = trans.contiguous()
    M, K = x.shape
    Bn = trans.shape[0]
    if trans.shape[1] != K or trans.shape[2] != K or Bn * rows_per_cloud != M:
        raise ValueError("apply_transform: shape mismatch")
    if K <= 8:
        if out is None:
            out = torch.empty((M, K), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        # y[m][n] = sum_c x[m][c] * trans[b][c][n]  => W(n,c) at b*K*K + n*1 + c*K
        _call("linear_smallk", lib.lpd_linear_smallk, _ptr(x), x.stride(0), _ptr(trans), 1, K, K * K, rows_per_cloud, _ptr(out),
                                         out.stride(0), M, K, K, None, None, None, ACT_NONE, 0.0, _stream())
        return out
    if not x.is_contiguous():
        raise ValueError("apply_transform: x must be contiguous for the batched GEMM")
    y = gemm(x.view(Bn, rows_per_cloud, K), trans, a_kmajor=False, b_kmajor=True)
    return y.view(M, K)


# bench.py's roofline line names the kernel behind each K-agg wrapper
KAGG_KERNEL_NAMES = {"edge_gather_max16": "edge_gather_max_cloud16p_kernel (persistent workgroups, LDS-resident cloud slice)",
                     "edge_gather_max": "edge_gather_max_kernel (direct gather, one part-wavefront per point)",
                     "edge_gather_maxw": "edge_gather_max_window_kernel (Z-order window of P rows in LDS, misses from L2)"}


def edge_gather_max(P, Q, idx, N, *, scale=None, shift=None, act=ACT_NONE, slope=0.01, out=None):
    """K-agg: out[m] = act(scale * (sel_t P[nbr(m,t)] + Q[m]) + shift); idx [M,k] int32 local indices."""
    ldp = _rows(P, "P")
    ldq = _rows(Q, "Q") if Q is not None else 0
    _req(idx, "idx", torch.int32)
    idx = idx.reshape(-1, idx.shape[-1]).contiguous()
    M, C = P.shape
    k = idx.shape[1]
    if idx.shape[0] != M:
        raise ValueError("edge_gather_max: idx rows != P rows")
    if out is None:
        out = torch.empty((M, C), dtype=torch.float32, device=P.device)
    ldo = _rows(out, "out")
    scale, shift = _vec(scale, "scale", C), _vec(shift, "shift", C)
    lib = _lib.load()
    _call(f"edge_gather_max[C={C}]", lib.lpd_edge_gather_max, _ptr(P), ldp, _ptr(Q), ldq, _ptr(idx), _ptr(out), ldo, _ptr(scale),
                                       _ptr(shift), M, N, C, k, act, float(slope), _stream())
    return out


def pack_idx16(idx):
    """int32 kNN indices [..., k=20] -> the blocked uint16 copy the cloud-resident K-agg reads (opaque int16 tensor)."""
    _req(idx, "idx", torch.int32)
    k = idx.shape[-1]
    idx = idx.reshape(-1, k).contiguous()
    M = idx.shape[0]
    out = torch.empty(((M + 31) // 32 * 32, k), dtype=torch.int16, device=idx.device)
    lib = _lib.load()
    _call("pack_idx16", lib.lpd_pack_idx16, _ptr(idx), _ptr(out), M, k, _stream())
    return out


def edge_gather_max16(P, Q, idx16, N, *, scale=None, shift=None, act=ACT_NONE, slope=0.01, out=None):
    """Cloud-resident K-agg (k = 20, N <= 4096): same result as edge_gather_max; idx16 from pack_idx16.
    P, Q and out are row-major [M, C] (2-D, column slices allowed) or cloud-panel views [B, C/8, N, 8] (4-D): with panels
    a block's 8-channel slice is one contiguous 32*N-byte run."""
    _req(idx16, "idx16", torch.int16)
    pan_p, pan_q = P.dim() == 4, Q is not None and Q.dim() == 4
    M, C = (P.shape[0] * P.shape[2], P.shape[1] * 8) if pan_p else P.shape
    k = idx16.shape[1]
    if idx16.dim() != 2 or idx16.shape[0] != (M + 31) // 32 * 32 or not idx16.is_contiguous():
        raise ValueError("edge_gather_max16: idx16 must come from pack_idx16 of this graph")
    if out is None:
        out = torch.empty((M, C), dtype=torch.float32, device=P.device)
    if _is_split(out):           # split bf16 planes for lpd_gemm_p8 (same bytes as the fp32 panels)
        if out.shape[2] * 8 != C or out.shape[3] != N or out.shape[1] * N != M:
            raise ValueError("edge_gather_max16: split out must be a [2, B, C/8, N, 8] view")
        for t, pan, name in ((P, pan_p, "P"), (Q, pan_q, "Q")):
            _req(t, name)
            if t is not None and pan and (not _is_panels(t) or t.shape[1] * 8 != C or t.shape[2] != N or t.shape[0] * N != M
                                          or t.stride(1) != out.stride(2)):
                raise ValueError(f"edge_gather_max16: cloud-panel {name} must be a [B, C/8, N, 8] view with the panel stride of out")
        scale, shift = _vec(scale, "scale", C), _vec(shift, "shift", C)
        lib = _lib.load()
        _call(f"edge_gather_max16[C={C}]", lib.lpd_edge_gather_max16s, _ptr(P), 8 if pan_p else _rows(P, "P"), _ptr(Q),
              0 if Q is None else (8 if pan_q else _rows(Q, "Q")), _ptr(idx16), _ptr(out[0]), out.stride(0), _ptr(scale), _ptr(shift),
              M, N, C, k, act, float(slope), P.stride(0) if pan_p else 0, Q.stride(0) if pan_q else 0, out.stride(1),
              out.stride(2) // 8, _stream())
        return out
    pan_o = out.dim() == 4
    for t, pan, name in ((P, pan_p, "P"), (Q, pan_q, "Q"), (out, pan_o, "out")):
        _req(t, name)
        if t is not None and pan and (not _is_panels(t) or t.shape[1] * 8 != C or t.shape[2] != N or t.shape[0] * N != M):
            raise ValueError(f"edge_gather_max16: cloud-panel {name} must be a [B, C/8, N, 8] view")
    ldp = 8 if pan_p else _rows(P, "P")
    ldq = 0 if Q is None else (8 if pan_q else _rows(Q, "Q"))
    ldo = 8 if pan_o else _rows(out, "out")
    scale, shift = _vec(scale, "scale", C), _vec(shift, "shift", C)
    lib = _lib.load()
    _call(f"edge_gather_max16[C={C}]", lib.lpd_edge_gather_max16, _ptr(P), ldp, _ptr(Q), ldq, _ptr(idx16), _ptr(out), ldo,
          _ptr(scale), _ptr(shift), M, N, C, k, act, float(slope), P.stride(0) if pan_p else 0, Q.stride(0) if pan_q else 0,
          out.stride(0) if pan_o else 0, _panel_ld(P, Q, out), _stream())
    return out


KAGGW_PERMUTE = _debug.on("kaggw-permute")   # windowed K-agg: every list with its out-of-window neighbours first


def pack_idx16w(idx, N=None):
    """int32 kNN indices [..., k] (k in {20, 32, 64}) -> the blocked RAW uint16 copy the windowed K-agg reads.  N (points per cloud;
    taken from a [B, N, k] tensor): every point's list is partitioned, the neighbours outside the point's 4095-row window first -- the
    miss phase of edge_gather_maxw then ends after the longest miss list of a wave instead of walking all k / 4 index quads (same
    sets, same result; include/lpd_hip.h lpd_pack_idx16w)."""
    _req(idx, "idx", torch.int32)
    k = idx.shape[-1]
    if N is None and idx.dim() == 3:
        N = idx.shape[1]
    idx = idx.reshape(-1, k).contiguous()
    M = idx.shape[0]
    out = torch.empty(((M + 31) // 32 * 32, k), dtype=torch.int16, device=idx.device)
    lib = _lib.load()
    _call("pack_idx16w", lib.lpd_pack_idx16w, _ptr(idx), _ptr(out), M, k, int(N) if (N and KAGGW_PERMUTE) else 0, _stream())
    return out


def edge_gather_maxw(P, Q, idx16, N, *, scale=None, shift=None, act=ACT_NONE, slope=0.01, out=None):
    """Windowed K-agg for large clouds (include/lpd_hip.h lpd_edge_gather_maxw): same result as edge_gather_max; idx16 from
    pack_idx16w.  P, Q, out row-major [M, C] (column slices allowed) or cloud-panel views [B, C/8, N, 8]."""
    _req(idx16, "idx16", torch.int16)
    pan_p, pan_q = P.dim() == 4, Q is not None and Q.dim() == 4
    M, C = (P.shape[0] * P.shape[2], P.shape[1] * 8) if pan_p else P.shape
    k = idx16.shape[1]
    if idx16.dim() != 2 or idx16.shape[0] != (M + 31) // 32 * 32 or not idx16.is_contiguous():
        raise ValueError("edge_gather_maxw: idx16 must come from pack_idx16w of this graph")
    if out is None:
        out = torch.empty((M, C), dtype=torch.float32, device=P.device)
    if _is_split(out):
        # the split-plane epilogue exists in the cloud-resident kernel only, which reads pack_idx16's byte offsets; idx16 here holds
        # pack_idx16w's raw indices (same tensor type and shape): refuse instead of computing with the wrong encoding
        raise ValueError("edge_gather_maxw: split bf16 output planes are not built for the windowed kernel "
                         "(use edge_gather_max16 with pack_idx16 for N <= 4096, k = 20, or an fp32 output)")
    pan_o = out.dim() == 4
    for t, pan, name in ((P, pan_p, "P"), (Q, pan_q, "Q"), (out, pan_o, "out")):
        _req(t, name)
        if t is not None and pan and (not _is_panels(t) or t.shape[1] * 8 != C or t.shape[2] != N or t.shape[0] * N != M):
            raise ValueError(f"edge_gather_maxw: cloud-panel {name} must be a [B, C/8, N, 8] view")
    ldp = 8 if pan_p else _rows(P, "P")
    ldq = 0 if Q is None else (8 if pan_q else _rows(Q, "Q"))
    ldo = 8 if pan_o else _rows(out, "out")
    scale, shift = _vec(scale, "scale", C), _vec(shift, "shift", C)
    lib = _lib.load()
    _call(f"edge_gather_maxw[C={C}]", lib.lpd_edge_gather_maxw, _ptr(P), ldp, _ptr(Q), ldq, _ptr(idx16), _ptr(out), ldo,
          _ptr(scale), _ptr(shift), M, N, C, k, act, float(slope), P.stride(0) if pan_p else 0, Q.stride(0) if pan_q else 0,
          out.stride(0) if pan_o else 0, _panel_ld(P, Q, out), _stream())
    return out


def edge_mlp(P, Q, idx, N, s1, b1, W2, s2, b2, *, act=ACT_LEAKY, slope=0.01, out=None, exact=False, x1_out=None):
    """(docstring below)  `out` may be a cloud-panel view [B, CO/8, N, 8].  x1_out (split planes like `out`, 128 -> 128 only): also
    x1 = max over k of the stage-1 activation -- the DG1-stage K-agg -- from the same launch (edge_mlp_x1_applies)."""
    return _edge_mlp(P, Q, idx, N, s1, b1, W2, s2, b2, act, slope, out, exact, x1_out)


EDGE_MLP_X1 = _debug.on("edge-mlp-x1")      # eval: the DG1-stage K-agg rides in the fused edge MLP (off: its own launch, lpd_edge_gather_max16)


def edge_mlp_x1_applies(M, N, CM, CO, exact=False):
    """can edge_mlp(..., x1_out=...) run?  (split planes, split-bf16 products, 128 -> 128 channels, 32-point blocks inside one cloud)"""
    return bool(EDGE_MLP_X1 and GEMM_BF16X3 and not exact and _EXACT.depth == 0 and CM == 128 and CO == 128 and M % 32 == 0 and N % 64 == 0)


def _edge_mlp(P, Q, idx, N, s1, b1, W2, s2, b2, act, slope, out, exact, x1_out=None):
    """Fused DG1-activation -> DG2 conv -> BN -> act -> max over k (include/lpd_hip.h lpd_edge_mlp)."""
    ldp = _rows(P, "P")
    ldq = _rows(Q, "Q") if Q is not None else 0
    _req(idx, "idx", torch.int32)
    idx = idx.reshape(-1, idx.shape[-1]).contiguous()
    M, CM = P.shape
    k = idx.shape[1]
    _req(W2, "W2")
    W2 = W2.reshape(W2.shape[0], -1).contiguous()
    CO = W2.shape[0]
    if W2.shape[1] != CM:
        raise ValueError("edge_mlp: W2 input width != P width")
    if out is None:
        out = torch.empty((M, CO), dtype=torch.float32, device=P.device)
    out_cloud = 0
    if _is_split(out):
        if out.shape[2] * 8 != CO or out.shape[1] * out.shape[3] != M or out.shape[3] != N:
            raise ValueError("edge_mlp: split out must be a [2, B, CO/8, N, 8] view")
        if not (GEMM_BF16X3 and not exact and _EXACT.depth == 0):
            raise ValueError("edge_mlp: split output is written by the split-bf16 kernel only")
        s1, b1 = _vec(s1, "s1", CM), _vec(b1, "b1", CM)
        s2, b2 = _vec(s2, "s2", CO), _vec(b2, "b2", CO)
        lib = _lib.load()
        if x1_out is not None:
            if (not _is_split(x1_out) or tuple(x1_out.shape) != tuple(out.shape) or tuple(x1_out.stride()) != tuple(out.stride())
                    or not edge_mlp_x1_applies(M, N, CM, CO, exact)):
                raise ValueError("edge_mlp: x1_out must be split planes with the shape and strides of `out` (128 -> 128, M % 32 == 0, N % 64 == 0)")
            _call(f"edge_mlpx3+x1[{CM}->{CO}]", lib.lpd_edge_mlp_x1_bf16x3s, _ptr(P), ldp, _ptr(Q), ldq, _ptr(idx), _ptr(s1), _ptr(b1), _ptr(W2),
                  _ptr(s2), _ptr(b2), _ptr(out[0]), _ptr(x1_out[0]), out.stride(0), M, N, k, act, float(slope), out.stride(1), out.stride(2) // 8,
                  _stream())
            return out
        _call(f"edge_mlpx3[{CM}->{CO}]", lib.lpd_edge_mlp_bf16x3s, _ptr(P), ldp, _ptr(Q), ldq, _ptr(idx), _ptr(s1), _ptr(b1), _ptr(W2), _ptr(s2),
              _ptr(b2), _ptr(out[0]), out.stride(0), M, N, CM, CO, k, act, float(slope), out.stride(1), out.stride(2) // 8, _stream())
        return out
    if x1_out is not None:
        raise ValueError("edge_mlp: x1_out is written by the split-plane form only")
    if out.dim() == 4:
        if not _is_panels(out) or out.shape[1] * 8 != CO or out.shape[0] * out.shape[2] != M or out.shape[2] != N:
            raise ValueError("edge_mlp: cloud-panel out must be a [B, CO/8, N, 8] view")
        _req(out, "out")
        ldo, out_cloud = 8, out.stride(0)
    else:
        ldo = _rows(out, "out")
    s1, b1 = _vec(s1, "s1", CM), _vec(b1, "b1", CM)
    s2, b2 = _vec(s2, "s2", CO), _vec(b2, "b2", CO)
    lib = _lib.load()
    x3 = GEMM_BF16X3 and not exact and _EXACT.depth == 0
    _call(f"edge_mlp{'x3' if x3 else ''}[{CM}->{CO}]", lib.lpd_edge_mlp_bf16x3 if x3 else lib.lpd_edge_mlp, _ptr(P), ldp, _ptr(Q), ldq, _ptr(idx), _ptr(s1), _ptr(b1), _ptr(W2), _ptr(s2),
                                _ptr(b2), _ptr(out), ldo, M, N, CM, CO, k, act, float(slope), out_cloud, _panel_ld(out), _stream())
    return out


def transpose(x):
    """[batch,R,C] -> [batch,C,R] (contiguous)."""
    _req(x, "x")
    if x.dim() != 3:
        raise ValueError("transpose: expected a 3-D tensor")
    x = x.contiguous()
    nb, R, C = x.shape
    out = torch.empty((nb, C, R), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _call("transpose", lib.lpd_transpose, _ptr(x), _ptr(out), nb, R, C, C, R, R * C, R * C, _stream())
    return out


def softmax_affine(x, scale=None, shift=None, out=None, colsum_rows=None):
    """softmax(scale * x + shift) over the (<= 64) columns of every row.

    colsum_rows = N: the rows are clouds of N consecutive points (N % 16 == 0); also returns the [B, 2*n] workspace of
    vlad_finalize with the per-cloud column sums (NetVLAD's a_sum) in its first n columns -> (out, ws)."""
    _rows(x, "x")
    x = x.contiguous()
    rows, n = x.shape
    if out is None:
        out = torch.empty_like(x)
    scale, shift = _vec(scale, "scale", n), _vec(shift, "shift", n)
    lib = _lib.load()
    if colsum_rows is None:
        _call("softmax_affine", lib.lpd_softmax_affine, _ptr(x), _ptr(out), rows, n, _ptr(scale), _ptr(shift), 0, None, 0, _stream())
        return out
    if colsum_rows % 16 or rows % colsum_rows:
        raise ValueError("softmax_affine: colsum_rows must be a multiple of 16 that divides the row count")
    ws = _zeros((rows // colsum_rows, 2 * n), torch.float32, x.device)
    _call("softmax_affine", lib.lpd_softmax_affine, _ptr(x), _ptr(out), rows, n, _ptr(scale), _ptr(shift), colsum_rows, _ptr(ws), 2 * n,
          _stream())
    return out, ws


def softmax_affine_parts(parts, scale=None, shift=None, colsum_rows=None):
    """softmax(scale * sum_j parts[j] + shift) over the 64 columns (parts [P, rows, 64]: the partial assignment products of
    gemm_p8(..., assign_w=...)), with the per-cloud column sums like softmax_affine(colsum_rows=N) -> (out [rows, 64], ws)."""
    _req(parts, "parts")
    if parts.dim() != 3 or parts.shape[2] != 64 or not parts[0].is_contiguous():
        raise ValueError("softmax_affine_parts: expected [P, rows, 64]")
    P, rows, n = parts.shape
    if colsum_rows is None or colsum_rows % 64 or rows % colsum_rows:
        raise ValueError("softmax_affine_parts: colsum_rows must be a multiple of 64 that divides the row count")
    out = torch.empty((rows, n), dtype=torch.float32, device=parts.device)
    scale, shift = _vec(scale, "scale", n), _vec(shift, "shift", n)
    ws = _zeros((rows // colsum_rows, 2 * n), torch.float32, parts.device)
    lib = _lib.load()
    _call("softmax_affine", lib.lpd_softmax_affine_parts, _ptr(parts), P, parts.stride(0), _ptr(out), rows, _ptr(scale), _ptr(shift),
          colsum_rows, _ptr(ws), 2 * n, _stream())
    return out, ws


def vlad_finalize(vraw, act, cw2, out=None, aux=None, ws=None):
    """vraw [B,F,KC], act [B,N,KC], cw2 [F,KC] -> [B,F*KC] normalised VLAD.

    out: optional preallocated [>=B, F*KC] buffer (rows beyond B untouched); aux: optional dict that
    receives asum [B,KC], inv_c [B,KC], inv_g [B] for the backward pass; ws: the workspace returned by
    softmax_affine(..., colsum_rows=N) (a_sum already accumulated: no pass over act)."""
    _req(vraw, "vraw"), _req(act, "act"), _req(cw2, "cw2")
    vraw, act, cw2 = vraw.contiguous(), act.contiguous(), cw2.contiguous()
    B, F, KC = vraw.shape
    N = act.shape[1]
    if out is None:
        out = torch.empty((B, F * KC), dtype=torch.float32, device=vraw.device)
    a1 = a2 = a3 = None
    if aux is not None:
        a1 = torch.empty((B, KC), dtype=torch.float32, device=vraw.device)
        a2 = torch.empty((B, KC), dtype=torch.float32, device=vraw.device)
        a3 = torch.empty((B,), dtype=torch.float32, device=vraw.device)
        aux.update(asum=a1, inv_c=a2, inv_g=a3)
    ready = ws is not None
    if ready:
        if ws.shape != (B, 2 * KC) or not ws.is_contiguous():
            raise ValueError("vlad_finalize: ws must be the [B, 2*KC] workspace of softmax_affine(colsum_rows=N)")
    else:
        ws = torch.empty((B, 2 * KC), dtype=torch.float32, device=vraw.device)
    lib = _lib.load()
    _call("vlad_finalize", lib.lpd_vlad_finalize, _ptr(vraw), _ptr(act), _ptr(cw2), _ptr(out), _ptr(ws), _ptr(a1), _ptr(a2),
          _ptr(a3), B, N, F, KC, int(ready), _stream())
    return out


def colmax(x, B, N):
    """x [B*N, C] rows -> per-cloud max [B, C]."""
    ldi = _rows(x, "x")
    C = x.shape[1]
    if x.shape[0] != B * N:
        raise ValueError("colmax: rows != B*N")
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _call("colmax", lib.lpd_colmax, _ptr(x), ldi, _ptr(out), B, N, C, _stream())
    return out


def gating(h, Wg, bias=None, scale=None, shift=None):
    """Context gating: h * sigmoid((h @ Wg + bias) * scale + shift); h [B,D] rows, Wg [D,D] (k-major)."""
    ldh, ldw = _rows(h, "h"), _rows(Wg, "Wg")
    B, D = h.shape
    if Wg.shape != (D, D):
        raise ValueError("gating: Wg must be [D, D]")
    out = torch.empty((B, D), dtype=torch.float32, device=h.device)
    bias, scale, shift = _vec(bias, "bias", D), _vec(scale, "scale", D), _vec(shift, "shift", D)
    lib = _lib.load()
    _call("gating", lib.lpd_gating, _ptr(h), ldh, _ptr(Wg), ldw, _ptr(bias), _ptr(scale), _ptr(shift), _ptr(out), D, B, D, _stream())
    return out


def mul(a, b):
    _req(a, "a"), _req(b, "b")
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    lib = _lib.load()
    _call("mul", lib.lpd_mul, _ptr(a), _ptr(b), _ptr(out), a.numel(), _stream())
    return out


def morton_sort(x, want_perm=False):
    """x [B,1,N,3] or [B,N,3] -> points reordered along a Z-order curve per cloud (same shape); N <= 16384."""
    _req(x, "x")
    shape = x.shape
    x3 = x.reshape(-1, shape[-2], 3).contiguous()
    B, N = x3.shape[0], x3.shape[1]
    out = torch.empty_like(x3)
    perm = torch.empty((B, N), dtype=torch.int32, device=x.device) if want_perm else None
    lib = _lib.load()
    _call("morton_sort", lib.lpd_morton_sort, _ptr(x3), _ptr(out), _ptr(perm), B, N, _stream())
    out = out.view(shape)
    return (out, perm) if want_perm else out


# ------------------------------------------------------------------------------------------------
# training-path wrappers (csrc/lpd_train.hip)
# ------------------------------------------------------------------------------------------------
class BNStats:
    """Per-channel batch statistics of one BatchNorm application (all fp32 [C] device tensors)."""
    __slots__ = ("scale", "shift", "mean", "invstd", "count")

    def __init__(self, scale, shift, mean, invstd, count):
        self.scale, self.shift, self.mean, self.invstd, self.count = scale, shift, mean, invstd, count


def bn_train_stats(X, bn, rows=None):
    """Batch statistics of X [R, C] rows (first `rows` rows) for nn.BatchNorm `bn` in train mode; updates
    bn.running_mean / running_var / num_batches_tracked like torch does."""
    ld = _rows(X, "X")
    R = X.shape[0] if rows is None else rows
    C = X.shape[1]
    dev = X.device
    sums = torch.empty((2, C), dtype=torch.float64, device=dev)
    lib = _lib.load()
    _call("colstats", lib.lpd_colstats, _ptr(X), ld, R, C, _ptr(sums[0]), _ptr(sums[1]), _stat_ws(), _stream())
    return _bn_finalize(sums, R, C, bn)


STATS_IN_GEMM = _debug.on("gemm-stats")


STAT_CMAX = 1024      # columns of the statistics workspace (csrc/lpd_common.h LPD_STAT_CMAX): lpd_gemm_x3w_stats refuses wider layers


def linear_bn_stats_fused_applies(M, N, K):
    return (STATS_IN_GEMM and GEMM_BF16X3 and _EXACT.depth == 0 and _FAST.depth == 0 and X3W_FORWARD
            and M >= 1024 and 64 <= N <= STAT_CMAX and K >= 128 and N * K <= (1 << 22) and (K >= 256 or N >= 128))


def linear_bn_stats(x, w, bn, bias=None, out_bf16=False):
    """(y, BNStats): y = x @ w.T (+ bias) raw and the train-mode statistics of `bn` over its rows (running stats updated).  Where the
    product runs on the prepared-fragment split-bf16 kernel (the policy of ops.gemm), the column sums come out of its epilogue
    (lpd_gemm_x3w_stats) instead of a second pass over y (lpd_colstats); otherwise linear + bn_train_stats.
    out_bf16: y is stored as bfloat16 (the statistics are those of the fp32 accumulators); needs the fused kernel (N % 32 == 0).
    x may be bfloat16 rows (with out_bf16, N >= 256, K % 32 == 0): the rows are the hi image, two products per term."""
    x16 = x.dtype == torch.bfloat16
    if x16:
        _req(x, "x", torch.bfloat16)
        if x.dim() != 2 or x.stride(1) != 1 or x.stride(0) % 4 != 0 or not out_bf16:
            raise ValueError("linear_bn_stats: bf16 rows need unit column stride and a bf16 result")
        ldx = x.stride(0)
    else:
        ldx = _rows(x, "x")
    _req(w, "w")
    w2 = w.reshape(w.shape[0], -1)
    M, K = x.shape
    N = w2.shape[0]
    if (linear_bn_stats_fused_applies(M, N, K) and w2.is_contiguous() and w2.shape[1] == K and (not out_bf16 or N % 32 == 0)
            and (not x16 or (N >= 256 and K % 32 == 0))):
        bias = _vec(bias, "bias", N)
        frags = _weight_frags(w2, False, N, K)
        y = torch.empty((M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
        sums = torch.empty((2, N), dtype=torch.float64, device=x.device)
        lib = _lib.load()
        _call(f"gemmx3w+stats[{M}x{N}x{K}]", lib.lpd_gemm_x3w_stats, _ptr(x), ldx, _ptr(frags), _ptr(y), N, int(bool(out_bf16)) | (2 if x16 else 0), M, N, K, _ptr(bias),
              _ptr(sums[0]), _ptr(sums[1]), X3W_IMPL, _stat_ws(), _stream())
        return y, _bn_finalize(sums, M, N, bn)
    if out_bf16 or x16:
        raise ValueError(f"linear_bn_stats: bf16 operands need the fused kernel (M={M}, N={N}, K={K})")
    y = linear(x, w, bias=bias)
    return y, bn_train_stats(y, bn)


ASSIGN_ACT = _debug.on("assign-act")     # train mode: bn3 affine + act inside the NetVLAD assignment product's loader


def gemm_act_applies(M, N, K):
    return (ASSIGN_ACT and GEMM_BF16X3 and _EXACT.depth == 0 and X3W_FORWARD and M >= 1024 and 64 <= N <= 128 and K >= 256
            and N * K <= (1 << 22))


def gemm_act(x, w_kn, a_scale, a_shift, act, slope, out_bf16=False, store=True):
    """(x_act, c): x_act = act(a_scale * x + a_shift) (rows [M, K]: the BatchNorm affine + activation of the layer in front, applied in
    the product's operand loader and stored on the way) and c = x_act @ w_kn for a k-major weight [K, N], N <= 128
    (include/lpd_hip.h lpd_gemm_x3w_act: util/lpdnet_model.py:262 -> util/PointNetVlad.py:48).
    x may be bfloat16 rows (K % 32 == 0); out_bf16: x_act is stored as bfloat16 and the product takes the stored (rounded) values.
    store=False: x_act is not written (None is returned for it) -- its consumers apply the same transform in their own loaders
    (gemm_tn / gemm with a_affine); with out_bf16 the product still takes the values rounded to bfloat16."""
    x16 = x.dtype == torch.bfloat16
    _req(x, "x", torch.bfloat16 if x16 else torch.float32)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("gemm_act: x needs contiguous rows")
    ldx = x.stride(0)
    _req(w_kn, "w_kn")
    M, K = x.shape
    N = w_kn.shape[1]
    if w_kn.shape[0] != K or not gemm_act_applies(M, N, K):
        raise ValueError(f"gemm_act: shape not built (M={M}, N={N}, K={K})")
    a_scale, a_shift = _vec(a_scale, "a_scale", K), _vec(a_shift, "a_shift", K)
    frags = _weight_frags(w_kn, True, N, K)
    if x16 and (K % 32 != 0 or ldx % 4 != 0):
        raise ValueError("gemm_act: bf16 rows need K % 32 == 0")
    x_act = torch.empty((M, K), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device) if store else None
    c = torch.empty((M, N), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _call(f"gemmx3w+act[{M}x{N}x{K}]", lib.lpd_gemm_x3w_act, _ptr(x), ldx, _ptr(frags), _ptr(c), N, M, N, K, None, _ptr(a_scale), _ptr(a_shift),
          act, float(slope), _ptr(x_act), K, int(x16) | (2 if out_bf16 else 0), 16 if _FAST.depth > 0 else 0, _stream())
    return x_act, c


def gemm_bf16a(A16, W, b_kmajor=True, out=None, accumulate=False):
    """out [M, N] fp32 (+)= A16 @ W for bfloat16 rows A16 [M, K] (K % 32 == 0) and a weight W ([K, N] k-major or [N, K]): the rows are the hi
    image of the split product, two MFMA products against the split weight (lpd_gemm_x3w_bf16a).  dX = dY W of the bf16-storage training
    mode's conv3 backward (util/lpdnet_model.py:262)."""
    _req(A16, "A16", torch.bfloat16)
    _req(W, "W")
    if A16.dim() != 2 or A16.stride(1) != 1 or A16.stride(0) % 4 != 0:
        raise ValueError("gemm_bf16a: A16 needs contiguous rows")
    M, K = A16.shape
    Kb, N = (W.shape[0], W.shape[1]) if b_kmajor else (W.shape[1], W.shape[0])
    if K != Kb or K % 32 != 0 or N * K > (1 << 22) or not GEMM_BF16X3:
        raise ValueError(f"gemm_bf16a: shape not built (M={M}, N={N}, K={K}, inner {Kb})")
    if out is None:
        if accumulate:
            raise ValueError("gemm_bf16a: accumulate needs out")
        out = torch.empty((M, N), dtype=torch.float32, device=A16.device)
    ldc = _rows(out, "out")
    frags = _weight_frags(W, b_kmajor, N, K)
    lib = _lib.load()
    _call(f"gemmx2w[{M}x{N}x{K}]", lib.lpd_gemm_x3w_bf16a, _ptr(A16), A16.stride(0), _ptr(frags), _ptr(out), ldc, M, N, K, int(bool(accumulate)),
          X3W_IMPL, _stream())
    return out


class _BatchCounts(__import__("threading").local):
    pending = None


_BATCH_COUNTS = _BatchCounts()


class deferred_batch_counts:
    """with ops.deferred_batch_counts(): the `num_batches_tracked += 1` of every train-mode BatchNorm inside is queued and applied by ONE
    launch on exit (torch._foreach_add_) instead of one 5-us launch per layer; layers with momentum=None read the counter and are
    updated at once.  Nested uses flush at the outermost exit."""

    def __enter__(self):
        self.outer = _BATCH_COUNTS.pending is None
        if self.outer:
            _BATCH_COUNTS.pending = []
        return self

    def __exit__(self, *exc):
        if self.outer:
            pend, _BATCH_COUNTS.pending = _BATCH_COUNTS.pending, None
            if pend:
                torch._foreach_add_(pend, 1)
        return False


def _bn_finalize(sums, R, C, bn):
    """fp64 column sums / sums of squares over R rows -> BNStats (+ running-stat update)."""
    out = torch.empty((4, C), dtype=torch.float32, device=sums.device)
    lib = _lib.load()
    track = bn.track_running_stats and bn.running_mean is not None
    if R <= 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got {R} row(s) of {C} channels")   # torch's rule
    if bn.momentum is None:      # torch: cumulative moving average, factor 1 / num_batches_tracked (after the increment)
        momentum = 1.0 / (int(bn.num_batches_tracked) + 1) if track else 0.0
    else:
        momentum = bn.momentum
    _call("bn_finalize", lib.lpd_bn_finalize, _ptr(sums[0]), _ptr(sums[1]), float(R), C, _ptr(bn.weight), _ptr(bn.bias),
          _ptr(bn.running_mean) if track else None, _ptr(bn.running_var) if track else None, float(momentum),
          float(bn.eps), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _stream())
    if track:
        pend = getattr(_BATCH_COUNTS, "pending", None)
        if pend is not None and bn.momentum is not None:
            pend.append(bn.num_batches_tracked)      # one launch for all layers of the step (deferred_batch_counts)
        else:
            bn.num_batches_tracked += 1
        from . import engine
        engine._CACHES.pop(bn, None)             # the folded eval-mode affine (engine.bn_affine) is stale now
    return BNStats(out[0], out[1], out[2], out[3], R)


def colsum(X, rows=None):
    """fp32 [C] column sums over the first `rows` rows (fp64 accumulation; bias gradients)."""
    ld = _rows(X, "X")
    R = X.shape[0] if rows is None else rows
    C = X.shape[1]
    sums = torch.empty((2, C), dtype=torch.float64, device=X.device)
    lib = _lib.load()
    _call("colstats", lib.lpd_colstats, _ptr(X), ld, R, C, _ptr(sums[0]), _ptr(sums[1]), _stat_ws(), _stream())
    return sums[0].float()


def affine_act(X, scale, shift, act=ACT_NONE, slope=0.01, out=None, rows=None, out16=None, only16=False):
    """out = act(scale * X + shift) over the (first `rows`) rows; out16: a bfloat16 tensor (column slices allowed) that receives a copy
    of the result rows rounded to bf16 (lpd_affine_act2); only16: nothing but that copy is written (returns out16)."""
    ldx = _rows(X, "X")
    R = X.shape[0] if rows is None else rows
    C = X.shape[1]
    if only16 and out16 is None:
        raise ValueError("affine_act: only16 needs out16")
    if out is None and not only16:
        out = torch.empty((X.shape[0], C), dtype=torch.float32, device=X.device)
    ldy = 0 if only16 else _rows(out, "out")
    lib = _lib.load()
    if out16 is not None:
        _req(out16, "out16", torch.bfloat16)
        if out16.dim() != 2 or out16.stride(1) != 1 or out16.shape[1] != C or out16.shape[0] < R or out16.stride(0) % 4 != 0:
            raise ValueError("affine_act: out16 must be bf16 rows of the result's shape")
        _call("affine_act", lib.lpd_affine_act2, _ptr(X), ldx, None if only16 else _ptr(out), ldy, _ptr(out16), out16.stride(0), R, C, _ptr(scale),
              _ptr(shift), act, float(slope), _stream())
        return out16 if only16 else out
    _call("affine_act", lib.lpd_affine_act, _ptr(X), ldx, _ptr(out), ldy, R, C, _ptr(scale), _ptr(shift), act, float(slope),
          _stream())
    return out


def bn_act_bwd(dY, X, st, act=ACT_NONE, slope=0.01, out=None, rows=None):
    """Backward of Y = act(BN_train(X)) (st: BNStats) or of Y = act(X) (st None).
    Returns (dX, dgamma fp32 [C], dbeta fp32 [C]).  out may alias dY (in-place)."""
    lddy, ldx = _rows(dY, "dY"), _rows(X, "X")
    R = X.shape[0] if rows is None else rows
    C = X.shape[1]
    if out is None:
        out = torch.empty((dY.shape[0], C), dtype=torch.float32, device=X.device)
    lddx = _rows(out, "out")
    red = torch.empty((2, C), dtype=torch.float64, device=X.device)
    lib = _lib.load()
    has_bn = st is not None
    _call("bn_act_bwd", lib.lpd_bn_act_bwd, _ptr(dY), lddy, _ptr(X), ldx, _ptr(out), lddx, R, C,
          _ptr(st.scale) if has_bn else None, _ptr(st.shift) if has_bn else None, _ptr(st.mean) if has_bn else None,
          _ptr(st.invstd) if has_bn else None, act, float(slope), int(has_bn), _ptr(red[0]), _ptr(red[1]), _stat_ws(), _stream())
    redf = red.float()
    return out, redf[1], redf[0]


def bn_act_bwd_bf16(dY, X, st, act=ACT_NONE, slope=0.01, out=None):
    """bn_act_bwd on bfloat16 tensors (dY, X, the result: [R, C] rows, C a power of two; lpd_bn_act_bwd_bf16): the arithmetic of the fp32
    kernels on the widened values, one rounding on the way out.  out may alias dY."""
    for t, n in ((dY, "dY"), (X, "X"), (out, "out")):
        _req(t, n, torch.bfloat16)
        if t is not None and (t.dim() != 2 or t.stride(1) != 1):
            raise ValueError(f"bn_act_bwd_bf16: {n} needs contiguous rows")
    R, C = X.shape
    if out is None:
        out = torch.empty((R, C), dtype=torch.bfloat16, device=X.device)
    red = torch.empty((2, C), dtype=torch.float64, device=X.device)
    lib = _lib.load()
    has_bn = st is not None
    _call("bn_act_bwd_bf16", lib.lpd_bn_act_bwd_bf16, _ptr(dY), dY.stride(0), _ptr(X), X.stride(0), _ptr(out), out.stride(0), R, C,
          _ptr(st.scale) if has_bn else None, _ptr(st.shift) if has_bn else None, _ptr(st.mean) if has_bn else None,
          _ptr(st.invstd) if has_bn else None, act, float(slope), int(has_bn), _ptr(red[0]), _ptr(red[1]), _stat_ws(), _stream())
    redf = red.float()
    return out, redf[1], redf[0]


def edge_build(P, Q, idx, N, bn=None):
    """U[(i,t)] = P[nbr(i,t)] + Q[i].  With `bn`: also the train-mode statistics of U for that BatchNorm, accumulated while
    the rows are written (returns (U, BNStats)) -- no second pass over the edge tensor."""
    ldp = _rows(P, "P")
    ldq = _rows(Q, "Q") if Q is not None else 0
    idx = idx.reshape(-1, idx.shape[-1])
    M, C = P.shape
    k = idx.shape[1]
    U = torch.empty((M * k, C), dtype=torch.float32, device=P.device)
    sums = torch.empty((2, C), dtype=torch.float64, device=P.device) if bn is not None else None
    lib = _lib.load()
    _call(f"edge_build[C={C}]", lib.lpd_edge_build, _ptr(P), ldp, _ptr(Q), ldq, _ptr(idx), _ptr(U), M, N, C, k,
          _ptr(sums[0]) if bn is not None else None, _ptr(sums[1]) if bn is not None else None, _stat_ws(), _stream())
    if bn is None:
        return U
    return U, _bn_finalize(sums, M * k, C, bn)


def group_max(X, k, scale, shift, act, slope, out, keep_sel=False):
    """out [M,C] = act(scale * sel_t X[(i,t)] + shift), -> arg [M,C] uint8; keep_sel: -> (arg, xsel [M,C]) with the raw selected values
    (what edge_bn_bwd(..., xsel=) reads instead of gathering them from X again)."""
    ldx, ldo = _rows(X, "X"), _rows(out, "out")
    M, C = out.shape
    if X.shape[0] != M * k or X.shape[1] != C:
        raise ValueError("group_max: shape mismatch")
    arg = torch.empty((M, C), dtype=torch.uint8, device=X.device)
    lib = _lib.load()
    if keep_sel:
        xsel = torch.empty((M, C), dtype=torch.float32, device=X.device)
        _call(f"group_max[C={C}]", lib.lpd_group_max_sel, _ptr(X), ldx, k, _ptr(scale), _ptr(shift), act, float(slope), _ptr(out), ldo,
              _ptr(arg), _ptr(xsel), C, M, C, _stream())
        return arg, xsel
    _call(f"group_max[C={C}]", lib.lpd_group_max, _ptr(X), ldx, k, _ptr(scale), _ptr(shift), act, float(slope), _ptr(out), ldo,
          _ptr(arg), M, C, _stream())
    return arg


def group_max_bwd(dOut, arg, k, dX=None, accumulate=False):
    ldo = _rows(dOut, "dOut")
    M, C = arg.shape
    if dX is None:
        dX = torch.empty((M * k, C), dtype=torch.float32, device=dOut.device)
        accumulate = False
    lib = _lib.load()
    _call(f"group_max_bwd[C={C}]", lib.lpd_group_max_bwd, _ptr(dOut), ldo, _ptr(arg), k, _ptr(dX), M, C, int(accumulate),
          _stream())
    return dX


def post_consts(post_bn, act, slope):
    """What the backward of a stage that kept POST-activation values (edge_mlp_train's Y1e = act(gamma xhat + beta)) needs to recover
    xhat = (pre - beta) / gamma, pre = y or y / ns: (beta, 1 / gamma, 1 / ns), SNAPSHOT at the time of the call -- autograd takes it in
    the forward, so a parameter that changes between forward and backward cannot give an inconsistent xhat (beta is copied).
    A channel whose gamma is exactly 0 has no recoverable xhat (its Y1e is the constant act(beta)): 1 / gamma := 0 there, i.e. that
    channel's dgamma comes out 0 instead of sum(dpre xhat) -- the one case in which this path differs from the U1-storing chain
    (LPD_DEBUG=edge-mlp-train=0), which a model with an exactly-zero BatchNorm scale should use."""
    if act not in (ACT_NONE, ACT_LEAKY) or (act == ACT_LEAKY and not 0.0 < slope <= 1.0):
        raise ValueError("post-activation edge tensors need an invertible activation (none / LeakyReLU with 0 < slope <= 1)")
    g = post_bn.weight.detach()
    rg = torch.where(g != 0, 1.0 / g, torch.zeros_like(g)).contiguous()
    return post_bn.bias.detach().clone().contiguous(), rg, (1.0 if act == ACT_NONE else 1.0 / float(slope))


def _post_consts(st, post_bn, act, slope):
    """post_bn: the BatchNorm module (constants read now) or the triple post_consts() returned at forward time"""
    if isinstance(post_bn, tuple):
        return post_bn
    return post_consts(post_bn, act, slope)


def edge_bn_bwd(dOut, arg, k, X, st, act, slope, dense=None, dQ=None, xsel=None, post_bn=None):
    """Fused backward through max-over-k + activation + train-mode BatchNorm on a materialised edge tensor X [M*k, C].
    dOut [M, C] (view allowed): gradient of the group-max output; dense [M*k, C]: optional dense gradient on the
    post-activation edges (overwritten with the result).  Returns (dX [M*k, C], dgamma, dbeta); fills dQ if given.
    xsel [M, C] (arg-max-only form): the raw selected values the forward kept (group_max(keep_sel=True)) -- same result.
    post_bn: X holds the POST-activation values act(BN(U)) (edge_mlp_train's Y1e) and this is the BatchNorm module."""
    ldo = _rows(dOut, "dOut")
    M, C = arg.shape
    dX = dense if dense is not None else torch.empty((M * k, C), dtype=torch.float32, device=X.device)
    ldq = _rows(dQ, "dQ") if dQ is not None else 0
    red = torch.empty((2, C), dtype=torch.float64, device=X.device)
    lib = _lib.load()
    if xsel is not None:
        if dense is not None or dQ is not None:
            raise ValueError("edge_bn_bwd: xsel belongs to the arg-max-only form (no dense gradient, no dQ)")
        _req(xsel, "xsel")
        _call(f"edge_bn_bwd[C={C}]", lib.lpd_edge_bn_bwd_sel, _ptr(dOut), ldo, _ptr(arg), _ptr(X), _ptr(xsel), _rows(xsel, "xsel"),
              _ptr(dX), k, M, C, _ptr(st.scale), _ptr(st.shift), _ptr(st.mean), _ptr(st.invstd), act, float(slope), _ptr(red[0]),
              _ptr(red[1]), _stat_ws(), _stream())
        redf = red.float()
        return dX, redf[1], redf[0]
    mean, invstd, inv_ns = st.mean, st.invstd, 0.0
    if post_bn is not None:
        mean, invstd, inv_ns = _post_consts(st, post_bn, act, slope)
    _call(f"edge_bn_bwd[C={C}]", lib.lpd_edge_bn_bwd, _ptr(dOut), ldo, _ptr(arg), _ptr(dense), _ptr(X), _ptr(dX), _ptr(dQ), ldq, k,
          M, C, _ptr(st.scale), _ptr(st.shift), _ptr(mean), _ptr(invstd), act, float(slope), float(inv_ns), _ptr(red[0]), _ptr(red[1]),
          _stat_ws(), _stream())
    redf = red.float()
    return dX, redf[1], redf[0]


def group_sum(dU, k, out):
    ldq = _rows(out, "out")
    M, C = out.shape
    lib = _lib.load()
    _call(f"group_sum[C={C}]", lib.lpd_group_sum, _ptr(dU), k, _ptr(out), ldq, M, C, _stream())
    return out


def scatter_add_rows(dU, idx, dP, N):
    """dP[nbr(i,t)] += dU[(i,t)]; dP must be zero-initialised (or hold a partial sum)."""
    ldp = _rows(dP, "dP")
    idx = idx.reshape(-1, idx.shape[-1])
    M, C = dP.shape
    k = idx.shape[1]
    lib = _lib.load()
    _call(f"scatter_add_rows[C={C}]", lib.lpd_scatter_add_rows, _ptr(dU), _ptr(idx), _ptr(dP), ldp, M, N, k, C, _stream())
    return dP


class GraphT:
    """Transposed kNN graph (CSR) of one index tensor: rowptr [M+1], edges [M*k]."""
    __slots__ = ("rowptr", "edges", "M", "k")

    def __init__(self, idx, N):
        _req(idx, "idx", torch.int32)
        idx = idx.reshape(-1, idx.shape[-1]).contiguous()
        self.M, self.k = idx.shape
        dev = idx.device
        self.rowptr = torch.empty((self.M + 1,), dtype=torch.int32, device=dev)
        self.edges = torch.empty((self.M * self.k,), dtype=torch.int32, device=dev)
        ws = torch.empty((2 * self.M,), dtype=torch.int32, device=dev)
        lib = _lib.load()
        _call("graph_transpose", lib.lpd_graph_transpose, _ptr(idx), self.M, N, self.k, _ptr(self.rowptr), _ptr(self.edges), _ptr(ws),
              _stream())


def gather_sum_rows(dU, graph, dP, accumulate=False):
    """dP[j] (+)= sum over incoming edges of dU rows (backward of the neighbour gather, no float atomics)."""
    _req(dU, "dU")
    ldp = _rows(dP, "dP")
    M, C = dP.shape
    if not dU.is_contiguous() or dU.shape[0] != graph.M * graph.k or dU.shape[1] != C or M != graph.M:
        raise ValueError("gather_sum_rows: dU must be contiguous [M*k, C] matching the graph")
    lib = _lib.load()
    _call(f"gather_sum_rows[C={C}]", lib.lpd_gather_sum_rows, _ptr(dU), _ptr(graph.rowptr), _ptr(graph.edges), _ptr(dP), ldp, M, C,
          int(bool(accumulate)), _stream())
    return dP


# ------------------------------------------------------------------------------------------------
# training path, second generation (csrc/lpd_train2.hip): split-form edge stage, bf16-storage edge tensors
# ------------------------------------------------------------------------------------------------
def _bf16_rows(t, name, C=None):
    _req(t, name, torch.bfloat16)
    if t.dim() != 2 or not t.is_contiguous() or (C is not None and t.shape[1] != C):
        raise ValueError(f"{name}: expected a contiguous 2-D bfloat16 tensor" + (f" with {C} columns" if C else ""))


def edge_split_fwd(P, Q, idx, N, bn):
    """Train-mode split-form edge stage, forward half (include/lpd_hip.h lpd_edge_split_fwd): one gather pass over the graph
    -> (S [M,C] = sum_t P[nbr], usel [M,C] = sel_t P[nbr] + Q, arg [M,C] uint8, BNStats of U = P[nbr] + Q over all M*k edges);
    the [M*k, C] edge tensor is never built.  Updates bn's running statistics like torch."""
    ldp, ldq = _rows(P, "P"), _rows(Q, "Q")
    _req(idx, "idx", torch.int32)
    idx = idx.reshape(-1, idx.shape[-1]).contiguous()
    M, C = P.shape
    k = idx.shape[1]
    dev = P.device
    S = torch.empty((M, C), dtype=torch.float32, device=dev)
    usel = torch.empty((M, C), dtype=torch.float32, device=dev)
    arg = torch.empty((M, C), dtype=torch.uint8, device=dev)
    sums = torch.empty((2, C), dtype=torch.float64, device=dev)
    lib = _lib.load()
    if (lib.lpd_edge_split_fwd16_applies(N, C, k) and P.data_ptr() % 16 == 0 and Q.data_ptr() % 16 == 0
            and bn.weight.data_ptr() % 16 == 0):
        # cloud-resident slices (the eval K-agg kernel's organisation): the k neighbour rows come from LDS, not through L2
        # (a NAMED tensor: a temporary handed over as a raw pointer is freed before the launch, and the next allocation in this
        #  argument list -- the zero-filled statistics workspace of a stream's first call -- was seen to land on it)
        idx16 = pack_idx16(idx)
        _call(f"edge_split_fwd[C={C}]", lib.lpd_edge_split_fwd16, _ptr(P), ldp, _ptr(Q), ldq, _ptr(idx16), _ptr(bn.weight),
              _ptr(S), _ptr(usel), _ptr(arg), M, N, C, k, _ptr(sums[0]), _ptr(sums[1]), _stat_ws(), _stream())
        return S, usel, arg, _bn_finalize(sums, M * k, C, bn)
    _call(f"edge_split_fwd[C={C}]", lib.lpd_edge_split_fwd, _ptr(P), ldp, _ptr(Q), ldq, _ptr(idx), _ptr(bn.weight), _ptr(S), _ptr(usel),
          _ptr(arg), M, N, C, k, _ptr(sums[0]), _ptr(sums[1]), _stat_ws(), _stream())
    return S, usel, arg, _bn_finalize(sums, M * k, C, bn)


SPLIT_BWD_BF16 = _debug.on("split-bwd-bf16")   # bf16 storage: the SN1 backward gathers bf16 rows of G and Q


def edge_split_bwd(dOut, usel, arg, S, P, Q, graph, st, act, slope, k, dP, dQ, half=False):
    """Backward half: dOut [M,C] (view allowed) -> fills dP, dQ ([M,C] views), returns (dgamma, dbeta) fp32.
    half (bf16 storage, C = 256): the rows gathered over the transposed graph are bf16 copies (lpd_edge_split_bwd)."""
    out16 = dP.dtype == torch.bfloat16      # (bf16 storage: the gradient rows of the projection as bf16, with half)
    if out16:
        for t_, n_ in ((dP, "dP"), (dQ, "dQ")):
            _req(t_, n_, torch.bfloat16)
            if t_.dim() != 2 or t_.stride(1) != 1 or t_.stride(0) % 4 != 0:
                raise ValueError(f"edge_split_bwd: {n_} must be bf16 rows")
        if not (half and SPLIT_BWD_BF16 and usel.shape[1] == 256):
            raise ValueError("edge_split_bwd: bf16 dP / dQ go with the bf16 gather rows (half=True, C = 256)")
        ldo, ldp, ldq, lddp, lddq = _rows(dOut, "dOut"), _rows(P, "P"), _rows(Q, "Q"), dP.stride(0), dQ.stride(0)
    else:
        ldo, ldp, ldq, lddp, lddq = _rows(dOut, "dOut"), _rows(P, "P"), _rows(Q, "Q"), _rows(dP, "dP"), _rows(dQ, "dQ")
    M, C = usel.shape
    G = torch.empty((M, C), dtype=torch.float32, device=usel.device)
    red = torch.empty((2, C), dtype=torch.float64, device=usel.device)
    lib = _lib.load()
    _call(f"edge_split_bwd[C={C}]", lib.lpd_edge_split_bwd, _ptr(dOut), ldo, _ptr(usel), _ptr(arg), _ptr(S), _ptr(P), ldp, _ptr(Q), ldq,
          _ptr(graph.rowptr), _ptr(graph.edges), _ptr(G), _ptr(dP), lddp, _ptr(dQ), lddq, M, C, k, _ptr(st.scale), _ptr(st.shift),
          _ptr(st.mean), _ptr(st.invstd), act, float(slope), (3 if out16 else 1) if (half and SPLIT_BWD_BF16 and C == 256) else 0, _ptr(red[0]), _ptr(red[1]),
          _stat_ws(), _stream())
    redf = red.float()
    return redf[1], redf[0]


def edge_build_bf16(P, Q, idx, N, bn):
    """edge_build with bf16 storage: (U [M*k, C] bfloat16, BNStats of the stored values)."""
    ldp = _rows(P, "P")
    ldq = _rows(Q, "Q") if Q is not None else 0
    idx = idx.reshape(-1, idx.shape[-1]).contiguous()
    M, C = P.shape
    k = idx.shape[1]
    U = torch.empty((M * k, C), dtype=torch.bfloat16, device=P.device)
    sums = torch.empty((2, C), dtype=torch.float64, device=P.device)
    lib = _lib.load()
    _call(f"edge_build_bf16[C={C}]", lib.lpd_edge_build_bf16, _ptr(P), ldp, _ptr(Q), ldq, _ptr(idx), _ptr(U), M, N, C, k, _ptr(sums[0]),
          _ptr(sums[1]), _stat_ws(), _stream())
    return U, _bn_finalize(sums, M * k, C, bn)


def edge_act_max(U, k, st, act, slope, out):
    """One pass over U (fp32 [M*k, C]): -> (Y = act(BN(U)) [M*k, C], arg uint8 [M, C]); out [M, C] receives max_k of the same values
    (what group_max + affine_act compute with two reads of U)."""
    _req(U, "U")
    ldo = _rows(out, "out")
    M, C = out.shape
    if U.shape != (M * k, C) or not U.is_contiguous():
        raise ValueError("edge_act_max: shape mismatch")
    Y = torch.empty_like(U)
    arg = torch.empty((M, C), dtype=torch.uint8, device=U.device)
    lib = _lib.load()
    _call(f"edge_act_max[C={C}]", lib.lpd_edge_act_max, _ptr(U), k, _ptr(st.scale), _ptr(st.shift), act, float(slope), _ptr(Y),
          _ptr(out), ldo, _ptr(arg), M, C, _stream())
    return Y, arg


def edge_act_max_bf16(U, k, st, act, slope, out):
    """One pass over U (bf16): -> (Y = act(BN(U)) bf16 [M*k, C], arg uint8 [M, C]); out [M, C] receives max_k of the same values."""
    _bf16_rows(U, "U")
    ldo = _rows(out, "out")
    M, C = out.shape
    if U.shape != (M * k, C):
        raise ValueError("edge_act_max_bf16: shape mismatch")
    Y = torch.empty_like(U)
    arg = torch.empty((M, C), dtype=torch.uint8, device=U.device)
    lib = _lib.load()
    _call(f"edge_act_max_bf16[C={C}]", lib.lpd_edge_act_max_bf16, _ptr(U), k, _ptr(st.scale), _ptr(st.shift), act, float(slope), _ptr(Y),
          _ptr(out), ldo, _ptr(arg), M, C, _stream())
    return Y, arg


def group_sel_stats_bf16(Z, k, bn):
    """One pass over the raw conv output Z (bf16 [M*k, C]): -> (sel [M,C] raw selected values, arg, BNStats of Z)."""
    _bf16_rows(Z, "Z")
    C = Z.shape[1]
    M = Z.shape[0] // k
    sel = torch.empty((M, C), dtype=torch.float32, device=Z.device)
    arg = torch.empty((M, C), dtype=torch.uint8, device=Z.device)
    sums = torch.empty((2, C), dtype=torch.float64, device=Z.device)
    lib = _lib.load()
    _call(f"group_sel_stats_bf16[C={C}]", lib.lpd_group_sel_stats_bf16, _ptr(Z), k, _ptr(bn.weight), _ptr(sel), C, _ptr(arg), M, C,
          _ptr(sums[0]), _ptr(sums[1]), _stat_ws(), _stream())
    return sel, arg, _bn_finalize(sums, M * k, C, bn)


def edge_bn_bwd_bf16(dOut, arg, k, X, st, act, slope, dense=None, dQ=None, xsel=None, post_bn=None):
    """edge_bn_bwd on bf16 tensors: -> (dX bf16 [M*k, C] (aliases `dense` when given), dgamma, dbeta).  xsel [M, C] fp32: the raw
    selected values of group_sel_stats_bf16 (arg-max-only form)."""
    ldo = _rows(dOut, "dOut")
    M, C = arg.shape
    _bf16_rows(X, "X", C)
    if dense is not None:
        _bf16_rows(dense, "dense", C)
    dX = dense if dense is not None else torch.empty((M * k, C), dtype=torch.bfloat16, device=X.device)
    ldq = _rows(dQ, "dQ") if dQ is not None else 0
    red = torch.empty((2, C), dtype=torch.float64, device=X.device)
    lib = _lib.load()
    if xsel is not None:
        if dense is not None or dQ is not None:
            raise ValueError("edge_bn_bwd_bf16: xsel belongs to the arg-max-only form (no dense gradient, no dQ)")
        _req(xsel, "xsel")
        _call(f"edge_bn_bwd_bf16[C={C}]", lib.lpd_edge_bn_bwd_bf16_sel, _ptr(dOut), ldo, _ptr(arg), _ptr(X), _ptr(xsel),
              _rows(xsel, "xsel"), _ptr(dX), k, M, C, _ptr(st.scale), _ptr(st.shift), _ptr(st.mean), _ptr(st.invstd), act, float(slope),
              _ptr(red[0]), _ptr(red[1]), _stat_ws(), _stream())
        redf = red.float()
        return dX, redf[1], redf[0]
    mean, invstd, inv_ns = st.mean, st.invstd, 0.0
    if post_bn is not None:
        mean, invstd, inv_ns = _post_consts(st, post_bn, act, slope)
    _call(f"edge_bn_bwd_bf16[C={C}]", lib.lpd_edge_bn_bwd_bf16, _ptr(dOut), ldo, _ptr(arg), _ptr(dense), _ptr(X), _ptr(dX), _ptr(dQ), ldq,
          k, M, C, _ptr(st.scale), _ptr(st.shift), _ptr(mean), _ptr(invstd), act, float(slope), float(inv_ns), _ptr(red[0]), _ptr(red[1]),
          _stat_ws(), _stream())
    redf = red.float()
    return dX, redf[1], redf[0]


EDGE_MLP_TRAIN = _debug.on("edge-mlp-train")      # train-mode DG1 -> DG2 stage in one launch (lpd_edge_mlp_train)


EDGE_MLP_TRAIN_BWD = _debug.on("edge-mlp-train-bwd")   # ... and its backward in two launches (lpd_edge_mlp_train_bwd)


def edge_mlp_train_applies(M, N, k, C, act, slope):
    return (EDGE_MLP_TRAIN and C == 128 and M % 64 == 0 and N % 64 == 0 and 0 < k <= 255 and GEMM_BF16X3 and _EXACT.depth == 0
            and (act == ACT_NONE or (act == ACT_LEAKY and 0.0 < slope <= 1.0)))


Z_BF16 = _debug.on("z-bf16")      # fp32 storage mode: the stored Z of the DG2 stage as bf16 (see lpd_edge_mlp_train)


EDGE_NOZ = _debug.on("edge-noz")     # bf16 storage: the DG2 output Z is not stored; its backward term is Y1e K (lpd_edge_mlp_train_bwd)


def edge_mlp_train(P, Q, idx, N, scale1, shift1, W2, bn2, act, slope, bf16, z_bf16=None, store_z=True):
    """Train-mode DG1 -> DG2 stage in one launch (include/lpd_hip.h lpd_edge_mlp_train): -> (Y1e [E,128] (bf16 or fp32), Z [E,128]
    (bf16 unless z_bf16 is False; None with store_z=False), zsel [M,128] raw selected values, arg2 [M,128] uint8, BNStats of Z with
    bn2's running statistics updated)."""
    z_bf16 = bool(bf16 or (Z_BF16 if z_bf16 is None else z_bf16))
    ldp, ldq = _rows(P, "P"), _rows(Q, "Q")
    _req(idx, "idx", torch.int32)
    idx = idx.reshape(-1, idx.shape[-1]).contiguous()
    M, C = P.shape
    k = idx.shape[1]
    _req(W2, "W2")
    if idx.shape[0] != M or tuple(W2.shape) != (128, 128) or not W2.is_contiguous() or C != 128:
        raise ValueError("edge_mlp_train: shape mismatch (128 -> 128 channels)")
    dt = torch.bfloat16 if bf16 else torch.float32
    Y = torch.empty((M * k, 128), dtype=dt, device=P.device)
    Z = torch.empty((M * k, 128), dtype=torch.bfloat16 if z_bf16 else torch.float32, device=P.device) if store_z else None
    zsel = torch.empty((M, 128), dtype=torch.float32, device=P.device)
    arg2 = torch.empty((M, 128), dtype=torch.uint8, device=P.device)
    sums = torch.empty((2, 128), dtype=torch.float64, device=P.device)
    scale1, shift1 = _vec(scale1, "scale1", 128), _vec(shift1, "shift1", 128)
    lib = _lib.load()
    _call(f"edge_mlp_train[{'bf16' if bf16 else 'f32'}]", lib.lpd_edge_mlp_train, _ptr(P), ldp, _ptr(Q), ldq, _ptr(idx), _ptr(scale1),
          _ptr(shift1), _ptr(W2), _ptr(bn2.weight), _ptr(Y), _ptr(Z), int(bool(bf16)), int(z_bf16), _ptr(zsel), 128, _ptr(arg2), _ptr(sums[0]),
          _ptr(sums[1]), M, N, k, act, float(slope), _stat_ws(), _stream())
    return Y, Z, zsel, arg2, _bn_finalize(sums, M * k, 128, bn2)


def edge_mlp_train_bwd(Z, arg2, dpre2, W2, st2, red2, Y1e, arg1, dx1, bn1, k, act, slope):
    """Backward of edge_mlp_train, dense part (include/lpd_hip.h lpd_edge_mlp_train_bwd): -> (G [E,128] like Y1e: the gradient in front of
    BatchNorm1, gsum [M,128] = its sums over the k slots of a point, red1 [2,128] fp64 = (dbeta1, dgamma1)).  Z None: the form that
    does not need the stored DG2 output (edge_mlp_train(store_z=False))."""
    M, C = arg2.shape
    bf16, z_bf16 = Y1e.dtype == torch.bfloat16, (Z is None or Z.dtype == torch.bfloat16)
    for t, name in ((Z, "Z"), (Y1e, "Y1e")):
        if t is None:
            continue
        if t.dtype not in (torch.bfloat16, torch.float32) or tuple(t.shape) != (M * k, 128) or not t.is_contiguous() or not t.is_cuda:
            raise ValueError(f"edge_mlp_train_bwd: {name} must be a contiguous [M k, 128] bf16 / fp32 tensor")
    if (bf16 and not z_bf16) or dpre2.dtype != Y1e.dtype or tuple(dpre2.shape) != (M, 128) or not dpre2.is_contiguous():
        raise ValueError("edge_mlp_train_bwd: dpre2 must be [M, 128] of Y1e's type (bn_sel_bwd_reduce); bf16 Y1e goes with bf16 Z")
    _req(W2, "W2")
    if C != 128 or tuple(W2.shape) != (128, 128) or not W2.is_contiguous() or M % 32 != 0:
        raise ValueError("edge_mlp_train_bwd: 128 -> 128 channels, M % 32 == 0")
    lddx1 = _rows(dx1, "dx1")
    beta1, rgamma1, inv_ns = _post_consts(None, bn1, act, slope)      # bn1: the module, or post_consts(bn1, ...) taken at forward time
    G = torch.empty_like(Y1e)
    gsum = torch.empty((M, 128), dtype=torch.float32, device=Y1e.device)
    red1 = torch.empty((2, 128), dtype=torch.float64, device=Y1e.device)
    kws = torch.empty((128 * 128 + 128,), dtype=torch.float32, device=Y1e.device) if Z is None else None
    lib = _lib.load()
    _call(f"edge_mlp_train_bwd[{'bf16' if bf16 else 'f32'}]", lib.lpd_edge_mlp_train_bwd, _ptr(Z), _ptr(arg2), _ptr(dpre2), _ptr(W2),
          _ptr(st2.scale), _ptr(st2.mean), _ptr(st2.invstd), _ptr(red2[0]), _ptr(red2[1]), _ptr(Y1e), _ptr(arg1), _ptr(dx1), lddx1,
          _ptr(beta1), _ptr(rgamma1), int(bf16), int(z_bf16), _ptr(G), _ptr(gsum), _ptr(red1[0]), _ptr(red1[1]), M, k, act, float(slope), float(inv_ns),
          _ptr(kws), _stat_ws(), _stream())
    return G, gsum, red1


def edge_dense_bwd_apply(G, gsum, S, P, Q, graph, st, red, k, dP, dQ):
    """dP, dQ of the edge tensor U = P[nbr] + Q behind a train-mode BatchNorm whose incoming gradient G [M k, C] is dense, in closed
    form from one gather pass over the transposed graph (include/lpd_hip.h lpd_edge_dense_bwd_apply)."""
    M, C = gsum.shape
    ldp, ldq, lddp, lddq = _rows(P, "P"), _rows(Q, "Q"), _rows(dP, "dP"), _rows(dQ, "dQ")
    bf16 = G.dtype == torch.bfloat16
    if tuple(G.shape) != (M * k, C) or not G.is_contiguous():
        raise ValueError("edge_dense_bwd_apply: G must be a contiguous [M k, C] tensor")
    lib = _lib.load()
    _call(f"edge_dense_bwd_apply[C={C}]", lib.lpd_edge_dense_bwd_apply, _ptr(G), int(bf16), _ptr(gsum), _ptr(S), _ptr(P), ldp, _ptr(Q), ldq,
          _ptr(graph.rowptr), _ptr(graph.edges), _ptr(dP), lddp, _ptr(dQ), lddq, M, C, k, _ptr(st.scale), _ptr(st.mean), _ptr(st.invstd),
          _ptr(red[0]), _ptr(red[1]), _stream())


def dg2_bwd_fused_applies(M, k, C):
    """The DG2 backward without the dZ tensor (csrc/lpd_train3.hip) covers 128 channels, 16 <= k <= 255, whole 128-row chunks."""
    return DG2_BWD_FUSED and C == 128 and 16 <= k <= 255 and (M * k) % 128 == 0


def bn_sel_bwd_reduce(dOut, xsel, st, act, slope, dtype=torch.bfloat16):
    """Arg-max-only BatchNorm backward, reduction part: -> (dpre [M, C] (bf16, or fp32 for the fp32 storage mode) = dOut * act'(pre),
    red fp64 [2, C] = (sum dpre, sum dpre xhat)); xsel: the raw selected values of the forward."""
    ldo = _rows(dOut, "dOut")
    _req(xsel, "xsel")
    M, C = xsel.shape
    dpre = torch.empty((M, C), dtype=dtype, device=xsel.device)
    red = torch.empty((2, C), dtype=torch.float64, device=xsel.device)
    lib = _lib.load()
    _call(f"bn_sel_bwd_reduce[C={C}]", lib.lpd_bn_sel_bwd_reduce if dtype == torch.bfloat16 else lib.lpd_bn_sel_bwd_reduce_f32, _ptr(dOut),
          ldo, _ptr(xsel), _rows(xsel, "xsel"), M, C, _ptr(st.scale), _ptr(st.shift), _ptr(st.mean), _ptr(st.invstd), act, float(slope),
          _ptr(dpre), _ptr(red[0]), _ptr(red[1]), _stat_ws(), _stream())
    return dpre, red


def edge_dw_sel_f32(Y, arg, dpre, k, W2, st, red):
    """edge_dw_sel_bf16 for the fp32 storage mode: Y [M*k, 128] fp32, dpre [M, 128] fp32 (split-bf16 products)."""
    _req(Y, "Y"), _req(dpre, "dpre"), _req(W2, "W2")
    if Y.shape[1] != 128 or not Y.is_contiguous() or not dpre.is_contiguous():
        raise ValueError("edge_dw_sel_f32: contiguous [*, 128] tensors expected")
    M = arg.shape[0]
    lib = _lib.load()
    ws = torch.empty((int(lib.lpd_edge_dw_sel_bf16_ws_bytes(M * k)),), dtype=torch.uint8, device=Y.device)
    dW2 = torch.empty((128, 128), dtype=torch.float32, device=Y.device)
    _call(f"edge_dw_sel_f32[{M * k}]", lib.lpd_edge_dw_sel_f32, _ptr(Y), _ptr(arg), _ptr(dpre), k, M, _ptr(W2), W2.stride(0), _ptr(st.scale),
          _ptr(st.mean), _ptr(st.invstd), _ptr(red[0]), _ptr(red[1]), _ptr(dW2), _ptr(ws), _stream())
    return dW2


def gemm_f32s_bnbwd(Z, arg, dpre, k, W2, st, red):
    """gemm_bf16s_bnbwd for the fp32 storage mode: dY [M*k, 128] fp32 = dZ W2, dZ generated from fp32 Z in the operand loader."""
    _req(Z, "Z"), _req(dpre, "dpre"), _req(W2, "W2")
    if Z.shape[1] != 128 or not Z.is_contiguous() or not dpre.is_contiguous():
        raise ValueError("gemm_f32s_bnbwd: contiguous [*, 128] tensors expected")
    M = arg.shape[0]
    dY = torch.empty((M * k, 128), dtype=torch.float32, device=Z.device)
    lib = _lib.load()
    _call(f"gemm_f32s_bnbwd[{M * k}x128x128]", lib.lpd_gemm_f32s_bnbwd, _ptr(Z), _ptr(arg), _ptr(dpre), k, M, _ptr(W2), W2.stride(0),
          _ptr(st.scale), _ptr(st.mean), _ptr(st.invstd), _ptr(red[0]), _ptr(red[1]), _ptr(dY), _stream())
    return dY


def edge_dw_sel_bf16(Y, arg, dpre16, k, W2, st, red):
    """dW2 [128, 128] = dZ^T Y of the DG2 stage from one pass over Y [M*k, 128] bf16 (no dZ tensor): see lpd_edge_dw_sel_bf16."""
    _bf16_rows(Y, "Y", 128), _bf16_rows(dpre16, "dpre16", 128)
    _req(W2, "W2")
    M = arg.shape[0]
    lib = _lib.load()
    ws = torch.empty((int(lib.lpd_edge_dw_sel_bf16_ws_bytes(M * k)),), dtype=torch.uint8, device=Y.device)
    dW2 = torch.empty((128, 128), dtype=torch.float32, device=Y.device)
    _call(f"edge_dw_sel_bf16[{M * k}]", lib.lpd_edge_dw_sel_bf16, _ptr(Y), _ptr(arg), _ptr(dpre16), k, M, _ptr(W2), W2.stride(0),
          _ptr(st.scale), _ptr(st.mean), _ptr(st.invstd), _ptr(red[0]), _ptr(red[1]), _ptr(dW2), _ptr(ws), _stream())
    return dW2


def gemm_bf16s_bnbwd(Z, arg, dpre16, k, W2, st, red):
    """dY [M*k, 128] bf16 = dZ W2 with dZ (BatchNorm backward of the arg-max gradient) generated in the operand loader."""
    _bf16_rows(Z, "Z", 128), _bf16_rows(dpre16, "dpre16", 128)
    _req(W2, "W2")
    M = arg.shape[0]
    dY = torch.empty((M * k, 128), dtype=torch.bfloat16, device=Z.device)
    lib = _lib.load()
    _call(f"gemm_bf16s_bnbwd[{M * k}x128x128]", lib.lpd_gemm_bf16s_bnbwd, _ptr(Z), _ptr(arg), _ptr(dpre16), k, M, _ptr(W2), W2.stride(0),
          _ptr(st.scale), _ptr(st.mean), _ptr(st.invstd), _ptr(red[0]), _ptr(red[1]), _ptr(dY), _stream())
    return dY


def gather_sum_rows_bf16(dU, graph, dP, accumulate=False):
    ldp = _rows(dP, "dP")
    M, C = dP.shape
    _bf16_rows(dU, "dU", C)
    if dU.shape[0] != graph.M * graph.k or M != graph.M:
        raise ValueError("gather_sum_rows_bf16: dU must be [M*k, C] matching the graph")
    lib = _lib.load()
    _call(f"gather_sum_rows_bf16[C={C}]", lib.lpd_gather_sum_rows_bf16, _ptr(dU), _ptr(graph.rowptr), _ptr(graph.edges), _ptr(dP), ldp, M,
          C, int(bool(accumulate)), _stream())
    return dP


def gemm_bf16s(A, W, b_kmajor=False):
    """C [M,N] (bf16) = A [M,K] (bf16) x W: W [N,K] (b_kmajor False: torch conv weight) or [K,N] (True), fp32, split hi + lo."""
    _bf16_rows(A, "A")
    _req(W, "W")
    W = W.reshape(W.shape[0], -1).contiguous()
    M, K = A.shape
    N = W.shape[1] if b_kmajor else W.shape[0]
    if (W.shape[0] if b_kmajor else W.shape[1]) != K:
        raise ValueError("gemm_bf16s: inner dims differ")
    C = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    lib = _lib.load()
    _call(f"gemm_bf16s[{M}x{N}x{K}]", lib.lpd_gemm_bf16s, _ptr(A), _ptr(W), W.stride(0), int(bool(b_kmajor)), _ptr(C), M, N, K, _stream())
    return C


def gemm_tn_bf16(A, B):
    """dW [KA,KB] (fp32) = A^T B over the rows; A [M,KA], B [M,KB] bf16."""
    _bf16_rows(A, "A"), _bf16_rows(B, "B")
    M, KA = A.shape
    KB = B.shape[1]
    if B.shape[0] != M:
        raise ValueError("gemm_tn_bf16: row counts differ")
    lib = _lib.load()
    ws = torch.empty((int(lib.lpd_gemm_tn_bf16_ws_floats(M, KA, KB)),), dtype=torch.float32, device=A.device)
    dW = torch.empty((KA, KB), dtype=torch.float32, device=A.device)
    _call(f"gemm_tn_bf16[{KA}x{KB}x{M}]", lib.lpd_gemm_tn_bf16, _ptr(A), _ptr(B), _ptr(dW), _ptr(ws), M, KA, KB, _stream())
    return dW


DG2_BWD_FUSED = _debug.on("dg2-bwd-fused")    # bf16 storage: DG2 backward without the dZ tensor (lpd_train3.hip)
GEMM_TN = _debug.on("gemm-tn")      # weight gradients on the register-transposing kernel (lpd_gemm_tn)


FEAT_IN_LOADER = _debug.on("feat-in-loader")   # train mode: no activated conv3 map -- its consumers transform the raw one


def feat_in_loader_applies(B, N, E, K):
    """the NetVLAD head can run on the trunk's RAW last-layer output (BatchNorm affine + activation in the loaders of the assignment,
    pooling, dA and assignment-weight-gradient products): shapes all four kernels are built for"""
    return (FEAT_IN_LOADER and GEMM_TN and GEMM_BF16X3 and X3W_BATCHED and _EXACT.depth == 0 and _FAST.depth == 0 and K == 64 and E % 256 == 0
            and E >= 256 and E * K <= (1 << 22) and N % 128 == 0 and N >= 2048 and B * N >= 16384 and B * N < (1 << 31))


def gemm_tn_act_applies(A, B):
    """the operand transform of gemm_tn(a_affine=...) is built for these operands"""
    M, KA, KB = A.shape[-2], A.shape[-1], B.shape[-1]
    return (GEMM_TN and GEMM_BF16X3 and _EXACT.depth == 0 and KA % 256 == 0 and KB % 64 == 0 and KB % 128 != 0 and M % 32 == 0 and M >= 2048
            and A.stride(-1) == 1 and B.stride(-1) == 1 and A.stride(-2) % (8 if A.dtype == torch.bfloat16 else 4) == 0 and B.stride(-2) % 4 == 0
            and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def gemm_tn(A, B, rows=None, a_affine=None):
    """dW [KA, KB] = A^T B over the (first `rows`) rows; A [M, KA], B [M, KB] fp32 row-major (column slices allowed);
    KA % 128 == 0, KB % 64 == 0.  Split-bf16 (three products): the accuracy of lpd_gemm_bf16x3.  A may be bfloat16 rows (they are
    the hi image: two products).  a_affine = (scale [KA], shift [KA], act, slope): A's rows are act(scale * A + shift), applied where
    they are staged (lpd_gemm_tn_act; gemm_tn_act_applies)."""
    a16 = A.dtype == torch.bfloat16
    b16 = B.dtype == torch.bfloat16      # (with a bf16 A, KA % 256 == 0, KB % 256 == 0: one product per term)
    if a16:
        _req(A, "A", torch.bfloat16)
        if A.stride(-1) != 1 or A.stride(-2) % 8 != 0:
            raise ValueError("gemm_tn: bf16 A needs contiguous rows, leading dim % 8 == 0")
    if b16:
        _req(B, "B", torch.bfloat16)
        if not a16 or B.dim() != 2 or B.stride(1) != 1 or B.stride(0) % 8 != 0 or a_affine is not None:
            raise ValueError("gemm_tn: bf16 B goes with a bf16 A (2-D, leading dim % 8 == 0, no operand transform)")
    if A.dim() == 3:      # batched: A [nb, M, KA], B [nb, M, KB] -> [nb, KA, KB]
        nb = A.shape[0]
        lda, ldb = A.stride(1) if a16 else _rows(A[0], "A"), _rows(B[0], "B")
        M, KA, KB, sA, sB = A.shape[1], A.shape[2], B.shape[2], A.stride(0), B.stride(0)
        if B.shape[0] != nb or B.shape[1] != M:
            raise ValueError("gemm_tn: batched operands must agree on batch and rows")
        shape = (nb, KA, KB)
    else:
        nb, sA, sB = 1, 0, 0
        lda, ldb = A.stride(0) if a16 else _rows(A, "A"), B.stride(0) if b16 else _rows(B, "B")
        M = A.shape[0] if rows is None else rows
        KA, KB = A.shape[1], B.shape[1]
        shape = (KA, KB)
    lib = _lib.load()
    dW = torch.empty(shape, dtype=torch.float32, device=A.device)
    if a_affine is not None:
        if rows is not None or not gemm_tn_act_applies(A, B):
            raise ValueError(f"gemm_tn: the operand transform is not built for these operands (M={M}, KA={KA}, KB={KB})")
        a_sc, a_sh = _vec(a_affine[0], "a_affine scale", KA), _vec(a_affine[1], "a_affine shift", KA)
        ws = torch.empty((int(lib.lpd_gemm_tn_act_ws_floats(M, KA, KB, nb, int(a16))),), dtype=torch.float32, device=A.device)
        _call(f"gemm_tn+act[{KA}x{KB}x{M}]" + (f"x{nb}" if nb > 1 else ""), lib.lpd_gemm_tn_act, _ptr(A), lda, _ptr(B), ldb, _ptr(dW), _ptr(ws), M,
              KA, KB, nb, sA, sB, int(a16), _ptr(a_sc), _ptr(a_sh), a_affine[2], float(a_affine[3]), _stream())
        return dW
    ws = torch.empty((int(lib.lpd_gemm_tn_ws_floats(M, KA, KB, nb)),), dtype=torch.float32, device=A.device)
    _call(f"gemm_tn[{KA}x{KB}x{M}]" + (f"x{nb}" if nb > 1 else ""), lib.lpd_gemm_tn, _ptr(A), lda, _ptr(B), ldb, _ptr(dW), _ptr(ws), M, KA,
          KB, nb, sA, sB, int(a16) | (2 if b16 else 0), _stream())
    return dW


def gemm_tn_applies(A, B, rows):
    return (GEMM_TN and GEMM_BF16X3 and _EXACT.depth == 0 and A.dim() == 2 and B.dim() == 2 and A.shape[1] % 128 == 0
            and B.shape[1] % 64 == 0 and rows >= 4096 and A.stride(1) == 1 and B.stride(1) == 1
            and A.stride(0) % (8 if A.dtype == torch.bfloat16 else 4) == 0
            and B.stride(0) % 4 == 0 and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def pool_tn(feat3, act3):
    """NetVLAD residual pooling act^T x per cloud (util/PointNetVlad.py:64-67) -> [B, E, K]: feat3 [B, N, E], act3 [B, N, K].
    The register-transposing kernel where it applies (split-bf16 products allowed, E % 128 == 0, K % 64 == 0), else the generic
    batched k-major product."""
    B, N, E = feat3.shape
    K = act3.shape[2]
    if (GEMM_TN and GEMM_BF16X3 and _EXACT.depth == 0 and E % 128 == 0 and K % 64 == 0 and N >= 256 and feat3.is_contiguous()
            and act3.is_contiguous()):
        return gemm_tn(feat3, act3)
    return None


def dw_smallk(dY, X):
    lddy, ldx = _rows(dY, "dY"), _rows(X, "X")
    M, Co = dY.shape
    Kin = X.shape[1]
    dW = torch.empty((Co, Kin), dtype=torch.float32, device=dY.device)
    lib = _lib.load()
    _call("dw_smallk", lib.lpd_dw_smallk, _ptr(dY), lddy, _ptr(X), ldx, M, Co, Kin, _ptr(dW), _stream())
    return dW


def colmax_arg(x, B, N):
    """x [B*N, C] rows -> (per-cloud max [B, C], arg-max row inside the cloud int32 [B, C])."""
    ld = _rows(x, "x")
    C = x.shape[1]
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    arg = torch.empty((B, C), dtype=torch.int32, device=x.device)
    lib = _lib.load()
    _call("colmax_arg", lib.lpd_colmax_arg, _ptr(x), ld, _ptr(out), _ptr(arg), B, N, C, _stream())
    return out, arg


def colmax_bwd(dOut, arg, N):
    """-> dense [B*N, C] gradient, zero except at the arg-max rows."""
    B, C = arg.shape
    dIn = torch.zeros((B * N, C), dtype=torch.float32, device=dOut.device)
    lib = _lib.load()
    dOut = dOut.contiguous()
    _call("colmax_bwd", lib.lpd_colmax_bwd, _ptr(dOut), _ptr(arg), _ptr(dIn), C, B, N, C, _stream())
    return dIn


def cloud_outer(X, dY, B, N):
    """dT[b] = sum_m X[m]^T dY[m] over each cloud; X, dY [B*N, KD] rows, KD <= 8 -> [B, KD, KD]."""
    ldx, ldy = _rows(X, "X"), _rows(dY, "dY")
    KD = X.shape[1]
    dT = torch.empty((B, KD, KD), dtype=torch.float32, device=X.device)
    lib = _lib.load()
    _call("cloud_outer", lib.lpd_cloud_outer, _ptr(X), ldx, _ptr(dY), ldy, _ptr(dT), B, N, KD, _stream())
    return dT


def softmax_bwd(A, dA, dasum, rows_per_cloud):
    rows, n = A.shape
    dS = torch.empty_like(A)
    lib = _lib.load()
    _call("softmax_bwd", lib.lpd_softmax_bwd, _ptr(A), _ptr(dA), _ptr(dasum), _ptr(dS), rows, n, rows_per_cloud, _stream())
    return dS


def vlad_finalize_bwd(dOut, v, aux, cw2, B, F, KC):
    dev = dOut.device
    dVraw = torch.empty((B, F, KC), dtype=torch.float32, device=dev)
    dasum = torch.empty((B, KC), dtype=torch.float32, device=dev)
    dcw2 = torch.empty((F, KC), dtype=torch.float32, device=dev)
    lib = _lib.load()
    _call("vlad_finalize_bwd", lib.lpd_vlad_finalize_bwd, _ptr(dOut), _ptr(v), _ptr(aux["inv_c"]), _ptr(aux["inv_g"]),
          _ptr(aux["asum"]), _ptr(cw2), _ptr(dVraw), _ptr(dasum), _ptr(dcw2), B, F, KC, _stream())
    return dVraw, dasum, dcw2


def retrieval_topk(Q, D, k):
    """k nearest database descriptors D [ndb, dim] of every query Q [nq, dim] (squared L2, ascending; ties -> lower index):
    -> (idx int32 [nq, k], dist fp32 [nq, k]).  The score matrix Q D^T runs on the exact f32-input MFMA GEMM."""
    ldq, ldd = _rows(Q, "Q"), _rows(D, "D")
    nq, dim = Q.shape
    ndb = D.shape[0]
    if D.shape[1] != dim or k > ndb:
        raise ValueError("retrieval_topk: Q and D must share the descriptor size and k <= len(D)")
    S = gemm(Q, D, a_kmajor=False, b_kmajor=False, exact=True)
    idx = torch.empty((nq, k), dtype=torch.int32, device=Q.device)
    dist = torch.empty((nq, k), dtype=torch.float32, device=Q.device)
    ws = torch.empty((nq + ndb,), dtype=torch.float32, device=Q.device)
    lib = _lib.load()
    _call("retrieval_topk", lib.lpd_retrieval_topk, _ptr(S), _ptr(Q), ldq, _ptr(D), ldd, nq, ndb, dim, k, _ptr(idx), _ptr(dist), _ptr(ws),
          _stream())
    return idx, dist


def hard_negatives(table, Q, cand, k):
    """For every query row Q[b]: the k rows of `table` nearest to it among cand[b] (int32 [bq, nc] row numbers), nearest first
    -> (positions into cand[b] int32 [bq, k], squared distances [bq, k]).  One launch for the whole batch."""
    ldt, ldq = _rows(table, "table"), _rows(Q, "Q")
    _req(cand, "cand", torch.int32)
    if cand.dim() != 2 or cand.shape[0] != Q.shape[0] or table.shape[1] != Q.shape[1]:
        raise ValueError("hard_negatives: cand must be [bq, nc] and table / Q must share the descriptor size")
    cand = cand.contiguous()
    bq, nc = cand.shape
    pos = torch.empty((bq, k), dtype=torch.int32, device=Q.device)
    dist = torch.empty((bq, k), dtype=torch.float32, device=Q.device)
    lib = _lib.load()
    _call("hard_negatives", lib.lpd_hard_negatives, _ptr(table), ldt, _ptr(Q), ldq, _ptr(cand), bq, nc, Q.shape[1], k, _ptr(pos), _ptr(dist),
          _stream())
    return pos, dist


def f64_to_f32(x64, out=None):
    """float64 CUDA tensor -> float32 (same shape), round to nearest even."""
    _req(x64, "x", torch.float64)
    x64 = x64.contiguous()
    if out is None:
        out = torch.empty(x64.shape, dtype=torch.float32, device=x64.device)
    lib = _lib.load()
    _call("f64_to_f32", lib.lpd_f64_to_f32, _ptr(x64), _ptr(out), x64.numel(), _stream())
    return out

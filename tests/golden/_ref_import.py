"""Import the (read-only, untrusted) reference hot-path modules in THIS container only.

Used solely by tests/golden/make_golden.py to *generate* fixtures; never imported by
tests, bench.py or smoke (the GPU box has no /root/reference).

Two import-time shims, zero edits to reference files:
  1. a stub `pynvml` module (util/lpdnet_model.py:10 -> util/gpu_mem_track.py:3),
  2. the module-global `torch` of util.lpdnet_model is rebound to a proxy whose
     `.device('cuda')` returns the CPU device (lpdnet_model.py:123,307,338 hard-code CUDA).
"""
import sys
import types

REF_ROOT = "/root/reference"


class _TorchProxy:
    def __init__(self, torch):
        object.__setattr__(self, "_t", torch)

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_t"), name)

    def device(self, *a, **k):
        t = object.__getattribute__(self, "_t")
        return t.device("cpu")


def load_reference():
    import torch
    if "pynvml" not in sys.modules:
        sys.modules["pynvml"] = types.ModuleType("pynvml")
    # make sure OUR util/loss packages are not shadowing the reference's
    for name in list(sys.modules):
        if name == "util" or name.startswith("util.") or name == "loss" or name.startswith("loss."):
            del sys.modules[name]
    sys.path.insert(0, REF_ROOT)
    try:
        import util.lpdnet_model as ref_lpd
        import util.PointNetVlad as ref_pnv
        import loss.pointnetvlad_loss as ref_loss
    finally:
        sys.path.remove(REF_ROOT)
    ref_lpd.torch = _TorchProxy(torch)
    mods = types.SimpleNamespace(lpd=ref_lpd, pnv=ref_pnv, loss=ref_loss)
    # detach from sys.modules so later `import util...` resolves to this repo's package
    for name in list(sys.modules):
        if name == "util" or name.startswith("util.") or name == "loss" or name.startswith("loss."):
            del sys.modules[name]
    return mods

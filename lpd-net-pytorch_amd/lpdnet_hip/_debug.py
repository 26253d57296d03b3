"""A/B ablation switches of the package: ONE environment variable,

    LPD_DEBUG="no-p8,knn-impl=4,x3w-kc=64"

a comma-separated list of `name` (on), `no-name` (off) and `name=value` tokens; README.md lists the names.  Everything here exists for
timing comparisons and for tests that exercise a superseded path -- the product behaviour never depends on LPD_DEBUG being set.  (Rounds
1-5 grew 37 separate LPD_* variables; round 6 folded them into this list.  The few switches a deployment may want are their own
variables: LPD_HIP_LIB, LPD_GEMM_FP32, LPD_EVAL_CHUNK, LPD_SIDE_STREAM, LPD_REPLAY.)  The C side reads the same variable (lpd_debug()
in csrc/lpd_abi.hip)."""
import os

_TOKENS = {}
for _t in os.environ.get("LPD_DEBUG", "").split(","):
    _t = _t.strip().lower()
    if not _t:
        continue
    if "=" in _t:
        _k, _v = _t.split("=", 1)
        _TOKENS[_k.strip()] = _v.strip()
    elif _t.startswith("no-"):
        _TOKENS[_t[3:]] = "0"
    else:
        _TOKENS[_t] = "1"


def on(name, default=True):
    """is the feature `name` on?  `no-name` / `name=0` switch it off, `name` / `name=1` on, otherwise `default`"""
    v = _TOKENS.get(name)
    return default if v is None else v not in ("0", "off", "false")


def value(name, default):
    """`name=value` from LPD_DEBUG (int when the default is an int), else `default`"""
    v = _TOKENS.get(name)
    if v is None:
        return default
    return int(v) if isinstance(default, int) and not isinstance(default, bool) else v

"""The C-ABI boundary as a C caller sees it: tests/abi_smoke.c is compiled (by __graft_entry__.build()) against
include/lpd_hip.h and linked with liblpd_hip.so.  On CPU only the symbol/size calls run; on the GPU box it drives
lpd_gemm (row-major and cloud-panel operands), lpd_gemm_bf16x3 and lpd_knn through the header's prototypes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "abi_smoke")


def _ensure_built():
    if not os.path.exists(BIN):
        from lpdnet_hip import _build
        _build.build()
        _build.build_abi_smoke()
    return BIN


def test_definitions_are_compiled_against_the_public_header():
    """csrc/lpd_common.h includes include/lpd_hip.h, so a prototype that differs from its extern "C" definition is a compile
    error in build(); this guards the include itself."""
    text = open(os.path.join(ROOT, "lpd-net-pytorch_amd", "csrc", "lpd_common.h")).read()
    assert '#include "../../include/lpd_hip.h"' in text
    for f in os.listdir(os.path.join(ROOT, "lpd-net-pytorch_amd", "csrc")):
        if f.endswith(".hip"):
            assert '#include "lpd_common.h"' in open(os.path.join(ROOT, "lpd-net-pytorch_amd", "csrc", f)).read(), f


def test_c_caller_links_and_resolves_symbols():
    r = subprocess.run([_ensure_built(), "--symbols"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "symbols OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_caller_runs_the_kernels_through_the_header(cuda):
    r = subprocess.run([_ensure_built()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "abi_smoke OK" in r.stdout, r.stdout + r.stderr

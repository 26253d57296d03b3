"""Build liblpd_hip.so (gfx950) in-tree with hipcc.  No torch dependency: pure HIP runtime.

Used by __graft_entry__.build(); the resulting .so travels to the GPU box with the repo snapshot.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB_PATH = os.path.join(HERE, "liblpd_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")
ARCH = "gfx950"
REPO = os.path.dirname(os.path.dirname(HERE))
PUBLIC_HEADER = os.path.join(REPO, "include", "lpd_hip.h")      # every csrc/*.hip is compiled against it (lpd_common.h)
ABI_SMOKE_SRC = os.path.join(REPO, "tests", "abi_smoke.c")
ABI_SMOKE_BIN = os.path.join(REPO, "tests", "abi_smoke")
# -ffp-contract=off: the kNN arithmetic contract distinguishes fused from non-fused operations
# (HIP's __fmul_rn/__fadd_rn are plain operators and DO get contracted otherwise); explicit fmaf /
# MFMA are unaffected.
# -fno-slp-vectorize (round 6): the SLP vectoriser packs adjacent scalar fp32 adds / multiplies into v_pk_add_f32 / v_pk_mul_f32 with
# op_sel half-selects.  One such chain (the squared norms in lpd_front.hip) returned WRONG sums whenever an MFMA-heavy kernel of another
# HIP stream was co-resident on the SIMD (profiles/r06_concurrency_packed_f32.txt); packed fp32 beside MFMAs is an anti-lever for speed
# as well (MI355X_MICROARCH.md).  Packed operations that are written as such (float2 __builtin_elementwise_fma) are unaffected.
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-value"]
# measurement builds only (e.g. LPD_EXTRA_FLAGS=-DLPD_P8_BENCH for tools/p8_bench.py's timing-only conv3 variants); part of the
# content hash, so switching it rebuilds
FLAGS += os.environ.get("LPD_EXTRA_FLAGS", "").split()


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build liblpd_hip.so")
    return exe


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path, headers):
    h = hashlib.sha256()
    for p in [path] + headers:
        with open(p, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link liblpd_hip.so. Incremental by content hash."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [PUBLIC_HEADER]
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        stamp = obj + ".sha"
        dig = _digest(src, headers)
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.exists(stamp)
                and open(stamp).read() == dig):
            continue
        jobs.append((src, obj, stamp, dig))

    def compile_one(job):
        src, obj, stamp, dig = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        with open(stamp, "w") as fh:
            fh.write(dig)
        return src

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for done in ex.map(compile_one, jobs):
                if verbose:
                    print("compiled", os.path.basename(done))
    if jobs or not os.path.exists(LIB_PATH):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB_PATH


def build_abi_smoke(werror=False):
    """gcc tests/abi_smoke.c against include/lpd_hip.h + liblpd_hip.so: a plain C caller of the C-ABI (run on the GPU box by
    tests/test_abi.py; `--symbols` runs without a GPU).  A test artefact: -Werror only when the test asks for it."""
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-O1", "-std=c11", "-Wall"] + (["-Werror"] if werror else []) + [ABI_SMOKE_SRC, f"-I{rocm}/include", f"-L{HERE}", "-llpd_hip",
           f"-L{rocm}/lib", "-lamdhip64", "-lm", "-Wl,-rpath,$ORIGIN/../lpd-net-pytorch_amd/lpdnet_hip",
           f"-Wl,-rpath,{rocm}/lib", "-o", ABI_SMOKE_BIN]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"gcc failed for {ABI_SMOKE_SRC}:\n{r.stderr}")
    return ABI_SMOKE_BIN


if __name__ == "__main__":
    print(build(verbose=True))

"""HBM traffic of the K-agg launches by PMC, written to profiles/kagg_pmc.json with a hash of the kernel's source:

    python3 tools/kagg_pmc.py [cfg2] [cfg5]            (on the GPU box; default: cfg2)

Per MI355X_MICROARCH.md (HBM / rocprofv3): one `rocprofv3 --kernel-trace --pmc <counter>` pass per counter group (FETCH_SIZE,
WRITE_SIZE, TCC_HIT_sum + TCC_MISS_sum), no other trace domain; FETCH_SIZE / WRITE_SIZE are KiB summed over the XCDs, and on
gfx950 FETCH_SIZE counts half of the bytes of 16-byte-per-lane reads, so it is doubled.  The profiled command is bench.py itself
(3 steps, one stream), i.e. the benched binary and workload; the SN1-stage launches are the dispatches whose kernel name matches and
the second one of each forward (the DG1 stage launches the same template first); every pass is averaged over the same last launches.  This script makes no HIP call itself: it only spawns rocprofv3 with python3
right behind the `--`.  bench.py copies `hbm_bytes_per_launch` into `roofline.traffic` only while
sha256(csrc/lpd_edge.hip + csrc/lpd_edge_win.hip) still equals the recorded `kernel_source_sha256`.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lpd-net-pytorch_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "kagg_pmc")
# per_forward: launches of the matching kernel template in one forward (cfg2: the DG1 stage, two items per CU, runs the non-persistent
# template `edge_gather_max_cloud16_kernel`, so the persistent one is the SN1 stage alone; cfg5: both stages run the window kernel)
CFG = {"cfg2": dict(batch=32, points=4096, k=20, bench_key="edge_gather_max16[C=256]", match="edge_gather_max_cloud16p_kernel", args=[],
                    per_forward=1),
       "cfg5": dict(batch=64, points=16384, k=64, bench_key="edge_gather_maxw[C=256]", match="edge_gather_max_window_kernel",
                    args=["--batch", "64", "--points", "16384", "--k", "64"], per_forward=2)}


def kernel_source_sha256():
    h = hashlib.sha256()
    for f in ("lpd_edge.hip", "lpd_edge_win.hip"):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def one_pass(tag, counters, bench_args):
    d = os.path.join(OUT, tag)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, LPD_SIDE_STREAM="0", TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "--pmc"] + counters + ["-d", d, "-o", "p", "--", "python3",
           os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-train", "--no-cpu-baseline", "--no-secondary"] + bench_args
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(f"rocprofv3 pass {tag} failed:\n{r.stderr[-2000:]}")
    per, meta = collections.defaultdict(float), {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            meta[row["Dispatch_Id"]] = (row["Kernel_Name"], int(row["Grid_Size"]))
    shutil.rmtree(d, ignore_errors=True)
    return per, meta


def sn1_dispatches(meta, match, per_forward):
    """Dispatch ids of the SN1-stage (C = 256) launches, deterministically: on one stream a forward launches the matching template
    `per_forward` times, the DG1 stage (C = 128) first and the SN1 stage last, so the matching dispatches in id order form groups of
    that size and the SN1 launch is the last of each group.  Where the stages have different grids the grid decides instead."""
    ids = sorted((d for d, (n, g) in meta.items() if match + "<" in n or n.endswith(match)), key=int)
    if not ids:
        raise SystemExit(f"no dispatch of {match} in the counter pass")
    grids = {meta[d][1] for d in ids}
    if per_forward > 1 and len(grids) > 1:
        return [d for d in ids if meta[d][1] == max(grids)]
    if len(ids) % per_forward:
        raise SystemExit(f"{len(ids)} dispatches of {match}: expected {per_forward} per forward")
    return ids[per_forward - 1::per_forward]


def mean_for(per, meta, match, counter, last, per_forward):
    """mean of `counter` over the LAST `last` SN1-stage launches of the pass (the same positions of the program in every pass: the
    timed steps and what follows them, not the clock-settling forwards whose number differs from run to run)"""
    ids = sn1_dispatches(meta, match, per_forward)[-last:]
    vals = [per[(d, counter)] for d in ids]
    return sum(vals) / len(vals), len(vals)


def main():
    which = [a for a in sys.argv[1:] if a in CFG] or ["cfg2"]
    path = os.path.join(ROOT, "profiles", "kagg_pmc.json")
    rec = json.load(open(path)) if os.path.exists(path) else {"runs": []}
    sha = kernel_source_sha256()
    for name in which:
        c = CFG[name]
        p_f, p_w, p_t = (one_pass(name + "_fetch", ["FETCH_SIZE"], c["args"]), one_pass(name + "_write", ["WRITE_SIZE"], c["args"]),
                         one_pass(name + "_tcc", ["TCC_HIT_sum", "TCC_MISS_sum"], c["args"]))
        pf = c["per_forward"]
        last = min(len(sn1_dispatches(p[1], c["match"], pf)) for p in (p_f, p_w, p_t))      # the same launch positions in all three passes
        last = min(last, 8)
        fetch, n_f = mean_for(*p_f, c["match"], "FETCH_SIZE", last, pf)
        write, n_w = mean_for(*p_w, c["match"], "WRITE_SIZE", last, pf)
        hit, _ = mean_for(*p_t, c["match"], "TCC_HIT_sum", last, pf)
        miss, n_t = mean_for(*p_t, c["match"], "TCC_MISS_sum", last, pf)
        pts = c["batch"] * c["points"]
        alg = (3 * 256 * 4 + 2 * c["k"]) * pts      # the kernel reads its indices as uint16 (pack_idx16)
        hbm = int(round((2.0 * fetch + write) * 1024))
        run = {"batch": c["batch"], "points": c["points"], "k": c["k"], "bench_key": c["bench_key"],
               "kernel": f"{c['match']} (SN1 stage, C=256: the second launch of this template in each forward)",
               "FETCH_SIZE_KiB": round(fetch, 2), "WRITE_SIZE_KiB": round(write, 2), "hbm_bytes_per_launch": hbm,
               "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round(hbm / alg, 4),
               "TCC_HIT_sum": round(hit, 1), "TCC_MISS_sum": round(miss, 1), "l2_hit_rate": round(hit / max(hit + miss, 1.0), 3),
               "kernel_source_sha256": sha,
               "source": f"tools/kagg_pmc.py {name} (rocprofv3 --pmc passes over bench.py --steps 3: FETCH_SIZE {n_f}, WRITE_SIZE {n_w}, "
                         f"TCC {n_t} launches averaged; FETCH_SIZE doubled per MI355X_MICROARCH.md)"}
        rec["runs"].insert(0, run)      # bench.py takes the first matching record
        print(json.dumps(run, indent=1))
    json.dump(rec, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()

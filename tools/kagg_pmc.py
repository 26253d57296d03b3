"""HBM traffic of the K-agg launches by PMC, written to profiles/kagg_pmc.json with a hash of the kernel's source:

    python3 tools/kagg_pmc.py [cfg2] [cfg5]            (on the GPU box; default: cfg2)

Per MI355X_MICROARCH.md (HBM / rocprofv3): one `rocprofv3 --kernel-trace --pmc <counter>` pass per counter group (FETCH_SIZE,
WRITE_SIZE, TCC_HIT_sum + TCC_MISS_sum), no other trace domain; FETCH_SIZE / WRITE_SIZE are KiB summed over the XCDs, and on
gfx950 FETCH_SIZE counts half of the bytes of 16-byte-per-lane reads, so it is doubled.  The profiled command is bench.py itself
(3 steps, one stream), i.e. the benched binary and workload; the SN1-stage launches are the dispatches whose kernel name matches and
whose grid is the largest of the two K-agg stages.  This script makes no HIP call itself: it only spawns rocprofv3 with python3
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
CFG = {"cfg2": dict(batch=32, points=4096, k=20, bench_key="edge_gather_max16[C=256]", match="edge_gather_max_cloud16p_kernel", args=[]),
       "cfg5": dict(batch=64, points=16384, k=64, bench_key="edge_gather_maxw[C=256]", match="edge_gather_max_window_kernel",
                    args=["--batch", "64", "--points", "16384", "--k", "64"])}


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


def mean_for(per, meta, match, counter):
    grids = [g for (n, g) in meta.values() if match in n]
    if not grids:
        raise SystemExit(f"no dispatch of {match} in the counter pass")
    vals = [v for (d, c), v in per.items() if c == counter and match in meta[d][0]]
    # both K-agg stages run the same kernel template; C = 256 moves twice the bytes of C = 128: keep the upper half by value
    vals.sort()
    top = vals[len(vals) // 2:] if len(set(grids)) == 1 else [v for (d, c), v in per.items()
                                                              if c == counter and match in meta[d][0] and meta[d][1] == max(grids)]
    return sum(top) / len(top), len(top)


def main():
    which = [a for a in sys.argv[1:] if a in CFG] or ["cfg2"]
    path = os.path.join(ROOT, "profiles", "kagg_pmc.json")
    rec = json.load(open(path)) if os.path.exists(path) else {"runs": []}
    sha = kernel_source_sha256()
    for name in which:
        c = CFG[name]
        fetch, n_f = mean_for(*one_pass(name + "_fetch", ["FETCH_SIZE"], c["args"]), c["match"], "FETCH_SIZE")
        write, n_w = mean_for(*one_pass(name + "_write", ["WRITE_SIZE"], c["args"]), c["match"], "WRITE_SIZE")
        per, meta = one_pass(name + "_tcc", ["TCC_HIT_sum", "TCC_MISS_sum"], c["args"])
        hit, _ = mean_for(per, meta, c["match"], "TCC_HIT_sum")
        miss, n_t = mean_for(per, meta, c["match"], "TCC_MISS_sum")
        pts = c["batch"] * c["points"]
        alg = (3 * 256 * 4 + 4 * c["k"]) * pts
        hbm = int(round((2.0 * fetch + write) * 1024))
        run = {"batch": c["batch"], "points": c["points"], "k": c["k"], "bench_key": c["bench_key"],
               "kernel": f"{c['match']} (SN1 stage, C=256; the upper half by counter value of the dispatches of this template)",
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

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the LPD-Net global-descriptor hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): LPD-Net (featnet=lpdnet, emb_dims=1024, k=20, no T-Nets) eval
forward, N=4096 points, eval_batch_size=32 clouds per step per GPU; synthetic U[-1,1)^3 clouds
already resident in HBM; random-init weights of the architecture.  One "step" = one forward of one
32-cloud batch -> 32 global descriptors.  Multi-GPU = one process per GPU, each rank embeds its own
shard of clouds (the path shards by cloud, no data-path collective): weak scaling.

`--gpus N` with N > 1 and no RANK in the environment: this process is a LAUNCHER -- it touches no GPU, starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` as a child process (one rank per
GPU over RCCL), passes rank 0's JSON line through and exits with the children's status.  Launched by torch.distributed.run
directly (RANK set) it is a worker.

Prints ONE JSON line on rank 0.  `roofline` is for the kNN-aggregation kernel BASELINE.json names
(edge_gather_max on the SN1 stage): algorithmic bytes = 3152 B/point / average launch
time measured with HIP events on the launch stream inside the timed region.  `cpu_baseline` is the
oracle (torch-CPU restatement of the reference path) timed on this box's host cores on a bounded
sample; it is a reported baseline, not the optimisation target.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lpd-net-pytorch_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
KAGG_ROW_BYTES = 3 * 256 * 4   # C=256 fp32, split form: P row + Q row + out row; + KAGG_IDX_BYTES * k for the indices
KAGG_IDX_BYTES = 2             # the K-agg kernels read uint16 indices (lpd_pack_idx16 / lpd_pack_idx16w), not the int32 of SURVEY 8d
MFMA_BF16_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 MFMA
VALU_F32_PEAK_TF = 157.3       # fp32 vector peak (= f32-input MFMA peak): the bound of the exact-fp32 kNN distance arithmetic
# SQ_VALU_MFMA_BUSY_CYCLES share of the kNN walk kernels, from the committed PMC table (not collected inside a bench run)
KNN_MFMA_BUSY = {64: {"value": 0.29, "source": "profiles/r05_pmc_train_bf16_final.txt"}, 3: {"value": 0.05, "source": "profiles/r05_pmc_train_bf16_final.txt"}}
LINE_MAX_BYTES = 4096          # the driver parses the final stdout line: it stays below this; everything else goes to bench_detail.json


def kernel_rooflines(kern, batch, points, k, split_bf16, knn_visits=None):
    """Roofline entries of the step's DOMINANT kernels (the K-agg the headline `roofline` names is ~5 % of the step): algorithmic FLOPs of
    SURVEY.md 8(d) per launch / HIP-event time of the launch (this run's `kernels` table) / the peak of the unit that bounds it.
    `executed` = the FLOPs the matrix cores actually issue (three bf16 products per term on the split-bf16 path)."""
    M = batch * points
    out = {}

    def add(name, key, flops, peak, bound, note, mult=1.0):
        if key not in kern:
            return
        t = kern[key]["avg_us"] * 1e-6
        ach = flops / t / 1e12
        out[name] = {"key": key, "bound": bound, "avg_launch_us": kern[key]["avg_us"], "algorithmic_flops_per_launch": int(flops),
                     "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                     **({"executed_frac": round(mult * ach / peak, 4)} if mult != 1.0 else {}), "note": note}
    x3 = 3.0 if split_bf16 else 1.0
    pk = MFMA_BF16_PEAK_TF if split_bf16 else VALU_F32_PEAK_TF
    how = "three bf16 MFMA products per term (split-bf16), fp32 accumulate" if split_bf16 else "f32-input MFMA (exact fp32)"
    key = next((k_ for k_ in kern if k_.startswith("gemm_p8")), None) or next((k_ for k_ in kern if k_.endswith(f"[{M}x1024x512]")), None)
    if key:
        fused = "+assign" in key
        add("conv3" + (" + NetVLAD assignment" if fused else ""), key, 2.0 * M * 512 * 1024 + (2.0 * M * 1024 * 64 if fused else 0.0), pk, "mfma",
            "SURVEY 8d K-mlp conv3 (512 -> 1024)" + (" + the assignment product x . cluster_weights (PointNetVlad.py:48) in the same launch" if fused else "") + "; " + how, x3)
    key = next((k_ for k_ in kern if k_.startswith("edge_mlp")), None)
    if key:
        add("edge MLP (DG1 act -> DG2 conv -> max over k)", key, 2.0 * M * k * 128 * 128, pk, "mfma",
            "SURVEY 8d K-agg DG1->DG2 fused: 2 N k 128 128 per cloud; " + how, x3)
    for C, what in ((64, "feature-space kNN"), (3, "xyz kNN")):
        key = f"knn[C={C},k={k}]"
        if key not in kern:
            continue
        # The search is exact fp32 (bit-exact indices forbid reduced precision), so its distance tiles are bound by the fp32 vector /
        # f32-input MFMA peak.  `frac` prices the FLOPs the search EXECUTES: query tiles x candidate tiles actually multiplied x
        # 32 x 32 x 2 C (the visit rate is measured on this run's own operands by the kernel's statistics variant, knn_visit_stats),
        # over the time of ALL launches of the search (bound pass, launch order, walk) -- at most 1 by construction.  What the
        # best-first walk saves against multiplying the full N x N matrix is a separate field, not a roofline fraction.
        t = kern[key]["avg_us"] * 1e-6
        full = float(batch) * (2.0 * points * points * C + 3.0 * points * points)       # SURVEY 8d K-knn
        nt = (points + 31) // 32
        ent = {"key": key, "bound": "fp32 vector / f32-input MFMA", "avg_launch_us": kern[key]["avg_us"], "peak": VALU_F32_PEAK_TF, "unit": "TFLOP/s",
               "bruteforce_flops_per_launch": int(full),
               "speedup_vs_bruteforce_at_peak": round(full / t / 1e12 / VALU_F32_PEAK_TF, 3),
               "mfma_busy": KNN_MFMA_BUSY.get(C)}
        v = (knn_visits or {}).get(C)
        if v is not None:
            v = min(float(v), float(nt))
            execd = float(batch) * nt * v * 32 * 32 * 2 * (4 if C <= 4 else C)      # three coordinates run as two channel pairs
            ach = execd / t / 1e12
            ent.update({"tiles_visited_per_query_tile": round(v, 2), "tiles_per_cloud": nt, "executed_flops_per_launch": int(execd),
                        "achieved": round(ach, 2), "frac": round(min(ach / VALU_F32_PEAK_TF, 1.0), 4)})
        else:
            ent.update({"achieved": None, "frac": None})
        out[what] = ent
    return out


def knn_visit_stats(model, x):
    """Candidate tiles the best-first kNN walk multiplies per query tile, measured on THIS run's operands: one hooked forward hands
    out F0, then the search's statistics variant (impl 5: per-wave counters instead of indices, tools/knn7_stats.py) runs on the
    feature rows and on the Z-ordered coordinates.  {64: tiles, 3: tiles}; a search that has no statistics variant is left out."""
    from lpdnet_hip import engine, ops
    out = {}
    try:
        B, N = x.shape[0], x.shape[2]
        k = model.emb_nn.k
        engine.DEBUG_AUX = {}
        with torch.no_grad():
            model(x)
        f0 = engine.DEBUG_AUX.get("F0")
        engine.DEBUG_AUX = None
        xs = ops.morton_sort(x) if engine.MORTON_ORDER else x
        for C, rows in ((64, f0), (3, xs.reshape(B * N, 3))):
            if rows is None:
                continue
            try:        # (a large batch runs as slices: the hook then holds the last slice's rows)
                raw = ops.knn(ops.transpose(rows.reshape(-1, N, C)[:32].contiguous()), k, impl=5).view(-1, 32, k)
                out[C] = float(raw[:, 0, 0].float().mean().item())       # first query of each tile carries the wave's counters
            except Exception:
                pass
    except Exception:
        pass
    finally:
        try:
            engine.DEBUG_AUX = None
        except Exception:
            pass
    return out


def _pick(d, keys):
    return {k_: d[k_] for k_ in keys if isinstance(d, dict) and k_ in d and d[k_] is not None}


def compact_line(full, detail_name="bench_detail.json"):
    """The ONE stdout line the driver parses, cut down to what the contract names (< LINE_MAX_BYTES): headline, `roofline`,
    `cpu_baseline`, the train steps, one number per secondary record and per dominant kernel.  Kernel tables, per-step arrays,
    notes and the full secondary records stay in `full`, which main() writes to bench_detail.json and prints to stderr."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dry_run", "higher_is_better", "scaling"))
    line["vs_baseline"] = full.get("vs_baseline")
    line.update(_pick(full, ("dtype", "data")))
    line["data"] = str(line.get("data", "synthetic"))[:80]
    cfg = full.get("config") or {}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:200], **_pick(cfg, ("clouds_per_step_per_gpu", "num_points", "batches_in_flight")),
                      "parallelism": str(cfg.get("parallelism", ""))[:80], "arithmetic": str(cfg.get("arithmetic", ""))[:200]}
    if isinstance(full.get("one_batch_in_flight"), dict):
        line["one_batch_in_flight"] = _pick(full["one_batch_in_flight"], ("value", "ms_per_step"))
    if full.get("descriptors_per_s_per_rank") and full.get("n_gpus", 1) > 1:
        line["descriptors_per_s_per_rank"] = full["descriptors_per_s_per_rank"]
    roof = full.get("roofline")
    if roof:
        r = _pick(roof, ("bound", "achieved", "peak", "unit", "frac", "frac_direct_form", "algorithmic_bytes_per_launch", "avg_launch_us"))
        r["kernel"] = str(roof.get("kernel", ""))[:160]
        r["traffic"] = roof.get("traffic")
        if isinstance(roof.get("stage"), dict):
            r["stage_frac_direct_form"] = roof["stage"].get("frac_direct_form")
        line["roofline"] = r
    else:
        line["roofline"] = None
    rk = full.get("roofline_kernels") or {}
    if rk:
        line["roofline_kernels"] = {name[:24]: _pick(e, ("bound", "avg_launch_us", "frac", "executed_frac", "speedup_vs_bruteforce_at_peak"))
                                    for name, e in rk.items()}
        for e in line["roofline_kernels"].values():
            if "bound" in e:
                e["bound"] = "mfma" if "mfma" == e["bound"] else "fp32"
    cb = full.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind"))
        c["sample"] = str(cb.get("sample_short") or cb.get("sample", ""))[:120]
        if isinstance(cb.get("torch_cpu_cross_check"), dict):
            c["torch_cpu_value"] = cb["torch_cpu_cross_check"].get("value")
        line["cpu_baseline"] = c
    if "parity_norm_rel_vs_oracle" in full:
        line["parity_norm_rel_vs_oracle"] = full["parity_norm_rel_vs_oracle"]
    for name in ("train", "train_bf16"):
        t = full.get(name)
        if t:
            e = _pick(t, ("value", "unit", "ms_per_step", "dtype", "clouds_per_s", "steps"))
            if isinstance(t.get("cpu_baseline"), dict):
                e["cpu_clouds_per_s"] = t["cpu_baseline"].get("clouds_per_s")
                if isinstance(t["cpu_baseline"].get("cfg2_batch"), dict):      # the same 44-cloud step on the host cores
                    e["cpu_steps_per_s_same_batch"] = t["cpu_baseline"]["cfg2_batch"].get("steps_per_s")
            if isinstance(t.get("exchange"), dict):
                e["exchange"] = _pick(t["exchange"], ("allreduce_ms_per_step_isolated", "allreduce_busbw_GBps", "allreduce_exposed_ms_per_step"))
            line[name] = e
    sec = full.get("secondary") or {}
    if sec:
        short = {}
        for name, rec in sec.items():
            if not isinstance(rec, dict):
                continue
            if "ms_per_step" in rec:
                e = _pick(rec, ("value", "ms_per_step"))
                if isinstance(rec.get("roofline"), dict):
                    e["kagg_frac"] = rec["roofline"].get("frac")
                short[name[:40]] = e
            else:                       # one level of nesting (operating points, storage variants): ms per step only
                short[name[:40]] = {str(k_)[:16]: v_.get("ms_per_step") for k_, v_ in rec.items() if isinstance(v_, dict)}
        line["secondary"] = short
    line["detail"] = detail_name
    # size guard: the optional blocks go first, the contract's keys never
    for drop in ("secondary", "roofline_kernels", "descriptors_per_s_per_rank"):
        if len(json.dumps(line)) < LINE_MAX_BYTES - 64:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) >= LINE_MAX_BYTES:
        line["config"] = {"workload": line["config"]["workload"][:120]}
        line["data"] = "synthetic"
    return line


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-seconds", type=float, default=1.0,
                    help="untimed forwards in front of the warm-up steps until the device has been busy this long (clock ramp of an idle box)")
    ap.add_argument("--settle-max-seconds", type=float, default=12.0,
                    help="... and at most this long, until two consecutive windows of 64 forwards agree within 1.5 %%")
    ap.add_argument("--batch", type=int, default=32, help="clouds per step per GPU (eval_batch_size)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="eval batches in flight per GPU (lpdnet_hip.harness.BatchPipeline, what harness.get_latent_vectors uses): consecutive "
                         "steps are enqueued on this many HIP streams; 1 = one after the other (also measured, reported beside the headline)")
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--k", type=int, default=20, help="neighbours per point (reference hard-codes 20; configs[4] uses 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary quadruplet train-step measurement")
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--no-train-bf16", action="store_true", help="skip the bf16-storage train-step measurement (configs[2] as stated)")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL, one GPU per rank (the measurement).  gloo = DRY RUN of the multi-rank path on however many GPUs are "
                         "visible (ranks share them: RCCL refuses two ranks on one device): launcher, exchange block and per-rank "
                         "reporting execute, the line is marked dry_run and its timings mean nothing")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary eval records (configs[1] at 128 clouds/step, configs[4]: N=16384, k=64, 64 clouds/step)")
    return ap.parse_args()


def visible_gpus():
    """GPUs of this node counted WITHOUT the HIP runtime (the launcher must not initialise it): the kfd topology in sysfs lists one
    node per agent, GPUs are the ones with SIMDs.  None when the topology is not visible (the ranks then report what they see)."""
    try:
        top = "/sys/class/kfd/kfd/topology/nodes"
        return sum(1 for d in os.listdir(top)
                   if any(ln.split()[0] == "simd_count" and int(ln.split()[1]) > 0 for ln in open(os.path.join(top, d, "properties"))))
    except (OSError, ValueError, IndexError):
        return None


def launch_ranks(args):
    """--gpus N > 1 without RANK: start the N ranks as a CHILD process tree (never exec / re-use a process that has touched
    the GPU; this parent makes no HIP call at all: the GPUs are counted from the kfd topology in sysfs)."""
    import socket
    import subprocess
    n_vis = visible_gpus()
    if n_vis and n_vis < args.gpus and args.dist_backend == "nccl":      # 0 / None: the topology is hidden (container): the ranks report what they see
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_vis} GPU(s) visible on this node")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def host_cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(model, points, seconds_target=20.0):
    """The reference path on this box's host cores, on a bounded sample of the same workload:
      * primary ("port"): the plain-C restatement of the whole eval path (oracle/lpd_forward.c: the reference's own formulation,
        gcc -O3 -march=native + OpenMP, one cloud per thread), one cloud per available core, median of 3 timed runs;
      * cross-check: the torch-CPU oracle (ATen: MKL / oneDNN) at the fastest of a few thread counts, median of 3.
    Both are checked against each other; the returned reference descriptors are the torch oracle's."""
    import ctypes
    import tempfile
    import numpy as np
    from oracle import lpd_oracle as orc  # checker-only import, cpu_baseline leg
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(4321)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # ---- C restatement, one cloud per core ----
    native = orc.build_c_oracle_native(os.path.join(tempfile.gettempdir(), f"liblpd_oracle_native_{os.getpid()}.so"))
    lib = ctypes.CDLL(native)
    # one cloud per OpenMP thread.  Two thread counts are tried -- every hardware thread of the host, and half of them (one per
    # physical core of an SMT host: on the pool's 2 x 64-core EPYC 9575F 128 threads embed 21 clouds/s, 256 threads 15) -- with
    # one timed run each after a warm-up run; the better one gets two more runs and its median of 3 is the value
    nA = max(1, min(avail, 256))
    xc = torch.rand((nA, 1, points, 3), generator=g) * 2 - 1

    def run(n):
        t0 = time.time()
        d, u = orc.forward_lpdnet_c(sd, xc[:n], k=model.emb_nn.k, threads=n, lib=lib)
        return time.time() - t0, d, u
    cand = [nA] + ([nA // 2] if nA >= 64 else [])
    run(cand[-1])                     # warm-up (page faults, weights into cache)
    first = {n: run(n) for n in cand}
    nC = max(cand, key=lambda n: n / first[n][0])
    times = [first[nC][0]] + [run(nC)[0] for _ in range(2)]
    tc = sorted(times)[1]
    dc, used = first[nC][1], first[nC][2]
    alt = None
    for n in cand:
        if n != nC:
            alt = {"threads": n, "clouds": n, "value": round(n / first[n][0], 3), "sample": "one timed run after the warm-up run"}
    # ---- torch-CPU oracle (cross-check and the reference descriptors for the parity figure) ----
    Bs = 4
    x = xc[:Bs].clone()
    best = None
    budget_t0 = time.time()
    with torch.no_grad():
        for threads in sorted({min(avail, t) for t in (16, 32, 64)}):
            torch.set_num_threads(threads)
            ref = orc.pointnetvlad_forward(sd, x, featnet="lpdnet", train=False, k=model.emb_nn.k)   # warm-up
            ts = []
            for _ in range(3):
                t0 = time.time()
                orc.pointnetvlad_forward(sd, x, featnet="lpdnet", train=False, k=model.emb_nn.k)
                ts.append(time.time() - t0)
            t = sorted(ts)[1]
            if best is None or t < best[0]:
                best = (t, threads)
            if time.time() - budget_t0 > seconds_target:
                break
    t, threads = best
    agree = float((np.abs(dc[:Bs] - ref.numpy()).max(1) / np.abs(ref.numpy()).max(1)).max())
    return {"value": round(nC / tc, 3), "unit": "descriptors/s", "cores": used, "kind": "port",
            "host_cpu": host_cpu_model(), "host_cores": avail,
            "sample_short": f"{nC} clouds x {points} pts, plain-C port of the reference path (OpenMP, {used} threads), median of {len(times)}",
            "sample": f"eval forward of {nC} clouds x {points} pts by the plain-C restatement of the reference path (oracle/lpd_forward.c, "
                      f"gcc -O3 -march=native, OpenMP: one cloud per thread, {used} threads of {avail} host cores), median of {len(times)} after a warm-up run",
            "torch_cpu_cross_check": {"value": round(Bs / t, 3), "unit": "descriptors/s", "cores": threads,
                                      "sample": f"torch-CPU oracle (ATen), {Bs} clouds, median of 3 at the fastest of the tried thread counts"},
            "other_thread_count": alt,
            "c_vs_torch_oracle_norm_rel": float(f"{agree:.3e}")}, x, ref, threads


def cpu_train_baseline(points, threads, seconds_target=25.0):
    """the quadruplet train step (forward in train mode, lazy quadruplet loss, backward, Adam) of the
    reference path on this box's host cores = the torch-CPU oracle (ATen + autograd, the reference's own formulation) at the
    thread count the eval cross-check found fastest.  B = 6 clouds (bq=1, P=2, Ng=2; BASELINE.md's 0.175 step/s row), median of
    3 after a warm-up step; then ONE step at bq=1, P=2, Ng=18 (B = 22, half of configs[2]'s batch) if the budget allows, so that
    the per-cloud scaling is on record (B = 44 needs ~35 GB and ~45 s per step in this formulation)."""
    from oracle import lpd_oracle as orc
    torch.set_num_threads(threads)
    sd0 = orc.synthetic_state("lpdnet", num_points=points)
    names = [k for k, v in sd0.items() if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var"))]

    def run(bq, P, Ng, reps):
        B = bq * (1 + P + Ng + 1)
        g = torch.Generator().manual_seed(4242)
        x = torch.rand((B, 1, points, 3), generator=g) * 2 - 1
        sd = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd0.items()}
        opt = torch.optim.Adam([sd[k] for k in names], lr=1e-5)
        ts = []
        for it in range(max(reps, 0) + 1):
            t0 = time.time()
            opt.zero_grad(set_to_none=True)
            d = orc.pointnetvlad_forward(sd, x, featnet="lpdnet", train=True, new_stats={})
            q, p, n, o = torch.split(d.view(bq, -1, 256), [1, P, Ng, 1], dim=1)
            loss = orc.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
            loss.backward()
            opt.step()
            if it or reps == 0:                      # reps == 0: the one (cold) step is the sample
                ts.append(time.time() - t0)
            elif reps > 1 and time.time() - t0 > seconds_target / 3:
                reps = 1
        return B, sorted(ts)[len(ts) // 2], len(ts)
    t_all = time.time()
    B, t6, n6 = run(1, 2, 2, 3)
    rec = {"value": round(1.0 / t6, 4), "unit": "steps/s", "cores": threads, "kind": "port", "clouds_per_step": B,
           "clouds_per_s": round(B / t6, 3),
           "compare": f"per cloud: this step holds {B} clouds, the GPU step 44 -- divide clouds_per_s by clouds_per_s, not steps/s by steps/s",
           "sample": f"torch-CPU oracle (ATen + autograd: the reference's formulation), quadruplet train step bq=1 P=2 Ng=2 -> {B} clouds "
                     f"x {points} pts, forward + lazy quadruplet loss + backward + Adam, {threads} threads, median of {n6} after a warm-up step"}
    try:
        free_gb = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2**30
    except (ValueError, OSError):
        free_gb = 0.0
    if time.time() - t_all < seconds_target * 0.6 and free_gb > 48 and 22.0 / B * t6 * 2 < seconds_target:
        B2, t22, _ = run(1, 2, 18, 1)
        rec["half_cfg2_batch"] = {"clouds_per_step": B2, "seconds_per_step": round(t22, 3), "clouds_per_s": round(B2 / t22, 3),
                                  "sample": "one timed step after a warm-up step, bq=1 P=2 Ng=18"}
    # ... and configs[2]'s own batch (bq=2, P=2, Ng=18 -> 44 clouds, ~35 GB in this formulation) where the host has the memory and the
    # step is predicted to stay under a minute: ONE step, not warmed up (its first-touch page faults included) -- the same 44 clouds
    # the GPU step holds, so that steps/s compare directly
    if free_gb > 96 and 44.0 / B * t6 < 60.0:
        t0 = time.time()
        try:
            B44, t44, _ = run(2, 2, 18, 0)
        except (RuntimeError, MemoryError, IndexError):
            B44, t44 = 0, 0.0
        if B44:
            rec["cfg2_batch"] = {"clouds_per_step": B44, "seconds_per_step": round(t44, 3), "steps_per_s": round(1.0 / t44, 4),
                                 "clouds_per_s": round(B44 / t44, 3), "sample": "ONE step, not warmed up, bq=2 P=2 Ng=18 (the GPU step's batch)"}
    return rec


def _time_steps(step, first, n, dist, dev):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    # (the caller has switched the cyclic garbage collector off in front of its warm-up steps: a full collection of this process's
    #  heap was caught stalling ONE timed step's enqueue for 82 ms -- gpu_ms 78 of that step, per_step below -- in one run of five;
    #  collecting HERE would leave the GPU idle for tens of ms right in front of the timed region, which costs its clocks:
    #  1.92 ms per eval step in a 20-step region against 1.83 in a 200-step one)
    t0 = time.perf_counter()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = []
    out = []
    evs[0].record()
    for i in range(n):
        th = time.perf_counter()
        out.append(step(first + i))
        host.append(round(1e3 * (time.perf_counter() - th), 2))
        evs[i + 1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    _time_steps.last = {"gpu_ms": [round(evs[i].elapsed_time(evs[i + 1]), 2) for i in range(n)], "host_enqueue_ms": host}
    if dist is not None:
        el = _dist_max(dist, dev, el)
    return el, out


def _dist_max(dist, dev, value):
    """max over the ranks of a host scalar (gloo has no GPU all_gather / barrier: CPU tensors there)"""
    t = torch.tensor([value], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def train_bench(dev, dist, world, rank, points, steps, warmup=4, storage="f32", tuple_shape=(2, 2, 18), label=None, featnet="lpdnet"):
    """Secondary metric of BASELINE.json: quadruplet train-steps/s (configs[2]: bq=2, P=2, Ng=18 -> 44 clouds/rank,
    lazy quadruplet loss, Adam), data-parallel across ranks with the RCCL gradient all-reduce (configs[3]).
    Four untimed steps first: the leg starts from an emptied allocator cache (12 GiB of blocks to re-create), and with two
    warm-up steps about one run in five still met device allocations inside its five timed steps (22 ms instead of 15.6)."""
    from util.PointNetVlad import PointNetVlad
    import loss.pointnetvlad_loss as L
    from lpdnet_hip import autograd
    bq, P, Ng = tuple_shape
    B = bq * (1 + P + Ng + 1)
    torch.manual_seed(1234)
    model = PointNetVlad(num_points=points, featnet=featnet, emb_dims=1024, output_dim=256).to(dev).train()
    net = model
    if dist is not None:
        from lpdnet_hip.parallel import GradAllReduce
        net = GradAllReduce(model)
    # torch's own fused multi-tensor Adam (one launch over all parameters; the default foreach form is seven passes, 0.26 ms of the step)
    opt = torch.optim.Adam(model.parameters(), lr=1e-5, fused=True)
    gen = torch.Generator().manual_seed(777 + rank)
    # a fresh tuple batch every step (2 MB each, resident in HBM): a fixed batch is memorised within a few Adam
    # steps and the hinge goes inactive (loss exactly 0), which would make the gradients trivial
    nbatch = steps + warmup
    clouds = [(torch.rand((B, 1, points, 3), generator=gen) * 2 - 1).to(dev) for _ in range(nbatch)]

    def make_step(fwd):
        def step(i):
            opt.zero_grad(set_to_none=True)
            out = fwd(clouds[i % nbatch]).view(bq, -1, 256)
            q, p, n, o = torch.split(out, [1, P, Ng, 1], dim=1)
            loss = L.quadruplet_loss(q, p, n, o, 0.5, 0.2, use_min=True, lazy=True, ignore_zero_loss=False)
            loss.backward()
            opt.step()
            return loss
        return step
    prev_storage = autograd.set_train_storage(storage)
    try:
        step = make_step(net)
        gc.collect()
        gc.disable()                                    # until the timed steps are over (see _time_steps)
        for i in range(warmup):
            step(i)
        torch.cuda.reset_peak_memory_stats(dev)
        el, losses = _time_steps(step, warmup, steps, dist, dev)
        gc.enable()
        per_step = getattr(_time_steps, "last", None)       # stream-event time and host enqueue time of every timed step
        losses = [round(float(l.item()), 4) for l in losses]
        peak = torch.cuda.max_memory_allocated(dev) / 2**30
        comm = None
        if dist is not None and world > 1:
            # (a) the step without its exchange (bare model, no hooks fire): what the all-reduce costs that backward does not hide
            net.enabled = False
            el0, _ = _time_steps(step, 0, steps, dist, dev)
            net.enabled = True
            # (b) the step's two all-reduces by themselves on an idle GPU (67 MB hidden1_weights gradient + the 3.3 MB bucket)
            big = torch.zeros_like(model.net_vlad.hidden1_weights)
            small = torch.zeros((sum(p.numel() for p in model.parameters()) - big.numel(),), device=dev)
            for _ in range(2):
                dist.all_reduce(big), dist.all_reduce(small)
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                dist.all_reduce(big), dist.all_reduce(small)
            e1.record()
            torch.cuda.synchronize()
            ar_ms = e0.elapsed_time(e1) / reps
            ar_max = _dist_max(dist, dev, ar_ms)
            nbytes = 4 * (big.numel() + small.numel())
            comm = {"allreduce_ms_per_step_isolated": round(ar_max, 3), "gradient_bytes": nbytes,
                    "allreduce_busbw_GBps": round(2 * (world - 1) / world * nbytes / (ar_max * 1e-3) / 1e9, 1),
                    "ms_per_step_without_exchange": round(1e3 * el0 / steps, 2),
                    "allreduce_exposed_ms_per_step": round(1e3 * (el - el0) / steps, 3),
                    "backend": "nccl (RCCL)" if dist.get_backend() == "nccl" else dist.get_backend() + " (dry run: not a measurement)", "buckets": "hidden1_weights (67 MB) from its post-accumulate hook, overlapped "
                    "with the trunk's backward; the other 3.3 MB flattened into one bucket at the end of backward"}
    finally:
        autograd.set_train_storage(prev_storage)
    return {"metric": "quadruplet train-steps/sec", "value": round(steps / el, 3), "unit": "steps/s",
            "tuples_per_s": round(bq * world * steps / el, 3), "clouds_per_s": round(B * world * steps / el, 1),
            "ms_per_step": round(1e3 * el / steps, 2), "steps": steps,
            "config": (label or f"BASELINE configs[{2 if world == 1 else 3}]") + f": featnet={featnet}, bq={bq} P={P} Ng={Ng} -> {B} clouds/rank, N={points}, lazy quadruplet, "
                      f"Adam (torch fused), {storage} storage; x{world} ranks data-parallel (per-rank BN, gradient all-reduce)",
            "dtype": storage, "losses": losses, "peak_hbm_gib": round(peak, 2), "exchange": comm, "per_step": per_step}


def secondary_eval(dev, points, k, batch, steps, warmup=2, featnet="lpdnet", exact=False, kernels=False, kagg_events=True):
    """One more eval-forward measurement on rank 0's GPU (world 1 only), reported INSIDE the JSON line next to the headline
    workload: same model family, random-init weights, clouds resident in HBM; the K-agg launches of the SN1 stage are bracketed
    by HIP events inside the timed region, like the headline's.  featnet: the trunk ('lpdnetorigin' = the reference's argparse
    default, util/initPara.py:74).  exact: every product on the f32-input MFMA (what LPD_GEMM_FP32=1 selects) instead of split-bf16.
    kernels: add the per-op table (a separate untimed pass) and the dominant kernels' roofline entries."""
    from lpdnet_hip import ops
    from util.PointNetVlad import PointNetVlad
    torch.manual_seed(1234)
    model = PointNetVlad(num_points=points, featnet=featnet, emb_dims=1024, output_dim=256)
    model.emb_nn.k = k
    model = model.to(dev).eval()
    gen = torch.Generator().manual_seed(4321)
    clouds = [(torch.rand((batch, 1, points, 3), generator=gen) * 2 - 1).to(dev) for _ in range(2)]
    was_x3 = ops.GEMM_BF16X3
    if exact:
        ops.GEMM_BF16X3 = False      # the module switch LPD_GEMM_FP32=1 sets at import
    try:
        with torch.no_grad():
            gc.collect()
            gc.disable()                                    # until the timed steps are over (see _time_steps)
            for i in range(warmup + 1):
                model(clouds[i % 2])
            torch.cuda.synchronize()
            # (kagg_events=False: no event hooks in the timed loop -- a hooked forward never takes the launch tape of small batches)
            ops.PROFILE, ops.PROFILE_ONLY = ({}, ("edge_gather_max",)) if kagg_events else (None, None)
            t0 = time.perf_counter()
            for i in range(steps):
                model(clouds[i % 2])
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            gc.enable()
            kern = kernel_table(ops.PROFILE or {})
            kern_all = None
            if kernels:
                from lpdnet_hip import engine as _eng
                ops.PROFILE, ops.PROFILE_ONLY = {}, None
                _eng._SIDE_FORCE.mode = False          # one stream: clean per-op durations
                try:
                    for i in range(3):
                        model(clouds[i % 2])
                    torch.cuda.synchronize()
                finally:
                    _eng._SIDE_FORCE.mode = None
                kern_all = kernel_table(ops.PROFILE)
                kern_all.update(kern)
            ops.PROFILE, ops.PROFILE_ONLY = None, None
        split = ops.GEMM_BF16X3
    finally:
        ops.GEMM_BF16X3 = was_x3
        ops.PROFILE, ops.PROFILE_ONLY = None, None
        gc.enable()
    rec = {"metric": "global descriptors/sec", "value": round(batch * steps / el, 2), "unit": "descriptors/s",
           "ms_per_step": round(1e3 * el / steps, 3), "steps": steps, "warmup": warmup + 1,
           "config": {"featnet": featnet, "num_points": points, "k": k, "clouds_per_step": batch,
                      "arithmetic": "every product on the f32-input MFMA (exact fp32; LPD_GEMM_FP32=1)" if exact else "split-bf16 products, fp32 accumulate"}}
    key = next((k_ for k_ in kern if k_.startswith("edge_gather_max") and k_.endswith("[C=256]")), None)
    if key is not None:
        t_s = kern[key]["avg_us"] * 1e-6
        per_step = max(1, round(kern[key]["launches"] / steps))      # a large batch runs as slices (engine.EVAL_CHUNK): one launch per slice
        pts = batch * points // per_step
        alg = (KAGG_ROW_BYTES + KAGG_IDX_BYTES * k) * pts
        alg_direct = (128 * 4 + 256 * 4 + 4 * k) * pts
        rec["config"]["clouds_per_launch"] = batch // per_step
        rec["roofline"] = {"kernel": f"{ops.KAGG_KERNEL_NAMES.get(key.split('[')[0], key)}, SN1 stage, C=256, k={k}", "bound": "hbm",
                           "achieved": round(alg / t_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(alg / t_s / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_us": kern[key]["avg_us"],
                           "algorithmic_bytes_per_launch": alg, "frac_direct_form": round(alg_direct / t_s / 1e9 / HBM_PEAK_GBS, 4)}
    if kern_all is not None:
        rec["kernels"] = kern_all
        if featnet == "lpdnet":
            rec["roofline_kernels"] = kernel_rooflines(kern_all, batch, points, k, split, knn_visit_stats(model, clouds[0]))
    del model, clouds
    torch.cuda.empty_cache()
    return rec


def kernel_table(prof):
    kern = {}
    for name, evs in prof.items():
        ms = [a.elapsed_time(b) for a, b in evs]
        kern[name] = {"launches": len(ms), "avg_us": round(1e3 * sum(ms) / len(ms), 2)}
    return kern


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))            # launcher: no GPU call in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" in os.environ and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback for the product path)")
    # stdout carries exactly ONE line (the JSON): libraries that print banners to fd 1 (RCCL prints its version block when the
    # first communicator is created) are sent to stderr; the line itself goes to the saved descriptor
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    dry_run = args.dist_backend != "nccl"
    if dry_run:                               # ranks share the visible GPUs (one on a gpurun box)
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run: one process per GPU over RCCL
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if dry_run:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from lpdnet_hip import engine, ops
    from util.PointNetVlad import PointNetVlad

    torch.manual_seed(1234)  # reference util/initPara.py:86
    model = PointNetVlad(num_points=args.points, featnet="lpdnet", emb_dims=1024, output_dim=256)
    # non-trivial BatchNorm statistics so that eval-mode BN is a real affine (random-init keeps 0/1)
    g = torch.Generator().manual_seed(99)
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
            m.weight.data.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
            m.bias.data.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    model.emb_nn.k = args.k
    model = model.to(dev).eval()

    gen = torch.Generator().manual_seed(1234 + rank)
    nbuf = 2
    clouds = [(torch.rand((args.batch, 1, args.points, 3), generator=gen) * 2 - 1).to(dev) for _ in range(nbuf)]

    # The product's way of embedding a sequence of batches (evaluate.py:96-159 -> harness.get_latent_vectors): BatchPipeline keeps
    # `--in-flight` consecutive batches on the GPU, batch i on HIP stream i % in_flight.  Same launches, same descriptors; the stages of
    # two batches (latency-bound searches, MFMA-bound products, HBM-bound gathers) fill each other's gaps.
    from lpdnet_hip import harness
    pipe = harness.BatchPipeline(model, max(1, args.in_flight), dev)

    def step(i):
        return pipe.submit(clouds[i % nbuf])

    def step_single(i):
        with torch.no_grad():
            return model(clouds[i % nbuf])

    # Settling phase, before the W warm-up steps of the contract: a box that has been idle runs its first ~second of kernels
    # below its steady clocks (first process on a fresh box: 3.0-3.2 ms per step for a whole 20-step region, 2.1 ms in the
    # process that follows).
    t_settle = time.perf_counter()
    n_settle = 0
    for _ in range(4):                                  # allocator, fragment caches, clocks
        step(n_settle)
        n_settle += 1
    while n_settle < 6 or time.perf_counter() - t_settle < args.settle_seconds:
        step(n_settle)
        n_settle += 1
        if n_settle % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    # ... and until the step time has stopped moving: windows of 64 forwards, two in a row within 1.5 % of each other, at most
    # --settle-max-seconds in all (a fresh box has been seen at 2.0 ms/step in its first process's timed region, 1.83 in the next)
    prev = None
    while time.perf_counter() - t_settle < args.settle_max_seconds:
        tw = time.perf_counter()
        for _ in range(64):
            step(n_settle)
            n_settle += 1
        torch.cuda.synchronize()
        cur = (time.perf_counter() - tw) / 64
        if prev is not None and abs(cur - prev) <= 0.015 * prev:
            break
        prev = cur
    settle_s = time.perf_counter() - t_settle
    gc.collect()
    gc.disable()                                        # until the timed steps are over (see _time_steps)
    for i in range(8):                                  # the collection left the GPU idle: a few forwards in front of the W warm-up steps
        step(i)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()

    # ---- the timed region of the contract: EXACTLY K steps between barrier + synchronize ----
    # (no event hooks here: with several batches in flight an event pair around a launch also spans the time the launch waits for CUs
    #  behind the other batch's kernels -- 150 us where rocprofv3 sees a 105-us kernel -- so the roofline kernel is timed in the second
    #  region below, one batch in flight, where a launch has the chip to itself: 93 us by events, 91 us by rocprofv3,
    #  profiles/r06_eval10_one_in_flight_kernel_stats.csv)
    pipe.join()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    pipe.join()
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # ---- a second timed region with ONE batch in flight (the latency view), its K-agg launches (the roofline kernel) bracketed by HIP
    # events on the launch stream.  Only those: bracketing all ~45 launches of a step costs 1.6 ms of host time per step (event creation +
    # two records per call) and made the host, not the GPU, the bottleneck of the timed region in some runs.
    for i in range(4):
        step_single(i)
    ops.PROFILE, ops.PROFILE_ONLY = {}, ("edge_gather_max",)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(args.steps):
        step_single(i)
    torch.cuda.synchronize()
    elapsed_single = time.perf_counter() - t1
    gc.enable()
    prof = ops.PROFILE
    step = step_single                                   # (the per-op pass and the parity forward below: plain forwards)
    # the per-op table of the detail record comes from a separate, untimed pass with every launch bracketed (HIP events on the launch
    # stream) and the second stream switched off, so the entries are clean per-op durations
    ops.PROFILE, ops.PROFILE_ONLY = {}, None
    engine._SIDE_FORCE.mode = False          # this pass on ONE stream: clean per-op durations (the timed region runs the product's two)
    try:
        for i in range(min(args.steps, 5) + 1):
            step(i)
        torch.cuda.synchronize()
    finally:
        engine._SIDE_FORCE.mode = None
    prof_all = ops.PROFILE
    prof_all.update({k_: v for k_, v in prof.items()})      # K-agg entries: the timed region's own measurements
    ops.PROFILE = None
    per_rank = [round(args.batch * args.steps / my_elapsed, 1)]
    if dist is not None:
        elapsed = _dist_max(dist, dev, elapsed)
        mine = torch.tensor([my_elapsed], device="cpu" if dry_run else dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [round(args.batch * args.steps / float(x.item()), 1) for x in allr]

    total_desc = args.batch * args.steps * world
    value = total_desc / elapsed

    kern = kernel_table(prof_all)
    roof = None
    key, kname = "edge_gather_max16[C=256]", "edge_gather_max_cloud16p_kernel (persistent workgroups, LDS-resident cloud slice)"
    if key not in kern:     # k != 20 or N > 4096
        key = next((k_ for k_ in kern if k_.startswith("edge_gather_max") and k_.endswith("[C=256]")), None)
        kname = ops.KAGG_KERNEL_NAMES.get(key.split("[")[0], key) if key else None
    if key in kern:
        t_s = kern[key]["avg_us"] * 1e-6
        pts = args.batch * args.points
        alg = (KAGG_ROW_BYTES + KAGG_IDX_BYTES * args.k) * pts
        alg_direct = (128 * 4 + 256 * 4 + 4 * args.k) * pts      # SURVEY 8d direct form: x2 row in, indices, x3 row out
        ach = alg / t_s / 1e9
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "kagg_pmc.json")
        if os.path.exists(pmc):          # PMC passes are kept per workload: {"runs": [{batch, points, k, bench_key, hbm_bytes_per_launch}]}
            try:
                import hashlib
                h = hashlib.sha256()
                for f_ in ("lpd_edge.hip", "lpd_edge_win.hip"):      # where the K-agg kernels live (tools/kagg_pmc.py stamps the same hash)
                    h.update(open(os.path.join(ROOT, "lpd-net-pytorch_amd", "csrc", f_), "rb").read())
                sha = h.hexdigest()
                rec = json.load(open(pmc))
                for r in rec.get("runs", [rec]):
                    if (r.get("batch", 32), r.get("points", 4096), r.get("k", 20)) == (args.batch, args.points, args.k) \
                            and r.get("bench_key") == key:      # the newest matching record comes first
                        if r.get("kernel_source_sha256") == sha:
                            traffic = r.get("hbm_bytes_per_launch")
                            traffic_source = ("profiles/kagg_pmc.json: rocprofv3 --pmc passes (tools/kagg_pmc.py) over this command on this "
                                              "workload with THIS kernel source (sha256 of csrc/lpd_edge*.hip matches); FETCH_SIZE doubled per "
                                              "MI355X_MICROARCH.md; not collected inside this run: " + str(r.get("source")))
                        else:
                            traffic_source = ("none: the newest PMC record in profiles/kagg_pmc.json was taken with a different kernel source "
                                              "(hash mismatch) -- re-run tools/kagg_pmc.py; its figure was "
                                              + str(r.get("hbm_bytes_per_launch")) + " B per launch")
                        break
            except Exception:
                traffic = None
        roof = {"kernel": f"{kname}, SN1 stage, C=256, k={args.k}; launch duration with one batch in flight (the kernel alone on the chip)",
                "bound": "hbm", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg, "avg_launch_us": kern[key]["avg_us"],
                "numerator": "split form: P row + Q row + out row + uint16 indices (2 k bytes) per point (the bytes this kernel reads and "
                             "writes once; P and Q exist because the SN1 convolution is split, DESIGN.md); frac_direct_form keeps SURVEY 8d's "
                             "1616 B/pt (x2 row, int32 indices, x3 row)",
                "frac_direct_form": round(alg_direct / t_s / 1e9 / HBM_PEAK_GBS, 4),
                "direct_form_bytes_per_launch": alg_direct}
        if args.in_flight > 1:
            roof["beside_another_batch"] = ("inside the headline's timed region (two batches in flight) the same launch shares the chip: "
                                            "rocprofv3 of this command averages 105 us = 0.49 (profiles/r06_eval10_train10_kernel_stats.csv)")
        # the SN1 stage as a whole: projection GEMM + K-agg for the stage's own inputs/outputs (x2 in, indices, x3 out)
        pk = next((k_ for k_ in kern if k_.startswith("gemm") and k_.endswith(f"[{pts}x512x128]")), None)
        if pk is not None:
            t_stage = (kern[pk]["avg_us"] + kern[key]["avg_us"]) * 1e-6
            roof["stage"] = {"kernels": [pk, key], "us": round(t_stage * 1e6, 1),
                             "frac_direct_form": round(alg_direct / t_stage / 1e9 / HBM_PEAK_GBS, 4)}

    secondary = None
    if world == 1 and not args.no_secondary and (args.points, args.k, args.batch) == (4096, 20, 32):
        # configs[1] "run both" batch sizes (evaluate.py:101-102 multiplies eval_batch_size by 1+P+Ng), and the
        # stress configuration configs[4] with its K-agg roofline, so that neither is builder-run only
        # (128 clouds run as four slices, two of them in flight: an event pair around a launch would span its wait for CUs -- no K-agg figure here)
        secondary = {"configs[1] at 128 clouds/step": secondary_eval(dev, 4096, 20, 128, 10, kagg_events=False),
                     "configs[4] stress (N=16384, k=64, 64 clouds/step)": secondary_eval(dev, 16384, 64, 64, 5, kernels=True),
                     # the headline step with every product on the f32-input MFMA (exact fp32) beside the split-bf16 headline
                     "configs[1] exact fp32 products (LPD_GEMM_FP32=1)": secondary_eval(dev, 4096, 20, 32, 10, exact=True, kernels=True),
                     # the reference's argparse-default trunk (util/initPara.py:74; model at util/lpdnet_model.py:68-114)
                     "featnet=lpdnetorigin eval, 32 clouds/step": secondary_eval(dev, 4096, 20, 32, 10, featnet="lpdnetorigin", kernels=True)}
        # the reference's OWN operating points (util/data.py:117-133 embeds one cloud at a time, util/initPara.py:32-43 gives an eval
        # batch of 6 x (1 + 1 + 2) = 24 and a train batch of bq=2, P=1, Ng=2 -> 10 clouds): small batches, where launch counts and
        # the 256-workgroup persistent kernels matter more than bandwidth
        secondary["reference operating points (eval, N=4096, k=20)"] = {
            f"{b} clouds/step": {kk: vv for kk, vv in secondary_eval(dev, 4096, 20, b, 60, warmup=5, kagg_events=False).items()
                                 if kk in ("value", "unit", "ms_per_step", "steps")}
            for b in (1, 2, 6, 10, 24)}
    train = None
    if not args.no_train:
        del out
        torch.cuda.empty_cache()
        train = train_bench(dev, dist, world, rank, args.points, args.train_steps)
        train_bf16 = None
        from lpdnet_hip import autograd as _ag
        if not args.no_train_bf16 and "bf16" in _ag.TRAIN_STORAGES:
            torch.cuda.empty_cache()
            train_bf16 = train_bench(dev, dist, world, rank, args.points, args.train_steps, storage="bf16")
        if world == 1 and not args.no_secondary and secondary is not None:
            torch.cuda.empty_cache()       # the reference's default trunk at configs[2]'s batch
            secondary["featnet=lpdnetorigin train (bq=2, P=2, Ng=18 -> 44 clouds)"] = {
                kk: vv for kk, vv in train_bench(dev, None, 1, 0, args.points, max(3, args.train_steps // 2), featnet="lpdnetorigin",
                                                 label="reference default trunk (util/initPara.py:74)").items()
                if kk in ("value", "unit", "ms_per_step", "steps", "config", "losses", "peak_hbm_gib")}
            torch.cuda.empty_cache()       # the reference's default train batch (initPara.py:32-43): bq=2, P=1, Ng=2 -> 10 clouds
            secondary["reference default train batch (bq=2, P=1, Ng=2 -> 10 clouds)"] = {
                st: {kk: vv for kk, vv in train_bench(dev, None, 1, 0, args.points, args.train_steps, storage=st, tuple_shape=(2, 1, 2),
                                                      label="reference defaults (util/initPara.py:32-43)").items()
                     if kk in ("value", "unit", "ms_per_step", "steps", "config", "losses", "peak_hbm_gib")}
                for st in ("f32", "bf16")}

    knn_visits = knn_visit_stats(model, clouds[0]) if rank == 0 else None
    if rank == 0:
        line = {
            "metric": "global descriptors/sec (4096-pt clouds)", "value": round(value, 2), "unit": "descriptors/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            **({"dry_run": f"--dist-backend {args.dist_backend}: {world} ranks on {torch.cuda.device_count()} GPU(s); exercises the launcher, the "
                           "exchange block and the per-rank reporting -- NOT a measurement"} if dry_run else {}),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (U[-1,1)^3 clouds resident in HBM, seed 1234+rank; random-init weights, randomised BN statistics)",
            "config": {"workload": ("BASELINE configs[1]" if (args.points, args.k) == (4096, 20) else "BASELINE configs[4] (stress)") +
                                   ": LPD-Net (featnet=lpdnet, emb_dims=1024, no T-Nets) eval forward, "
                                   f"N={args.points}, k={args.k}, eval_batch_size={args.batch} clouds/step/GPU",
                       "clouds_per_step_per_gpu": args.batch, "num_points": args.points, "batches_in_flight": max(1, args.in_flight),
                       "parallelism": f"shard-by-cloud x{world} (one process per GPU, no data-path collective)",
                       "hip_streams": engine.side_stream_report(dev),
                       "settle": f"{n_settle} untimed forwards ({settle_s:.1f} s: until two windows of 64 agree within 1.5 %) before the warm-up steps",
                       "arithmetic": ("fp32 tensors; kNN distances and every layer in front of the feature-space kNN exact fp32; large dense "
                                      "products as 3-product split-bf16 MFMA with fp32 accumulation (DESIGN.md 3.2)"
                                      if ops.GEMM_BF16X3 else "fp32 tensors, every product on the f32-input MFMA / fp32 FMA")},
            "descriptors_per_s_per_rank": per_rank,
            "one_batch_in_flight": {"value": round(args.batch * args.steps / elapsed_single, 2), "unit": "descriptors/s per GPU",
                                    "ms_per_step": round(1e3 * elapsed_single / args.steps, 3),
                                    "note": "the same K steps one after the other on one stream pair (rank 0): the latency of one batch; "
                                            "the roofline kernel's HIP-event time comes from this region"},
            "roofline": roof, "roofline_kernels": kernel_rooflines(kern, args.batch, args.points, args.k, ops.GEMM_BF16X3, knn_visits),
            "kernels": kern, "train": train,
        }
        if secondary is not None:
            line["secondary"] = secondary
        if train is not None and train_bf16 is not None:
            line["train_bf16"] = train_bf16
        if world == 1 and not args.no_cpu_baseline:
            base, xs, ref, best_threads = cpu_baseline(model, args.points)
            with torch.no_grad():
                got = model(xs.to(dev)).cpu()
            rel = ((got - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).max().item()
            line["cpu_baseline"] = base
            line["parity_norm_rel_vs_oracle"] = float(f"{rel:.3e}")
            if train is not None:
                tb = cpu_train_baseline(args.points, best_threads)
                train["cpu_baseline"] = tb
                if train_bf16 is not None:
                    train_bf16["cpu_baseline"] = tb
        # the full record (kernel tables, per-step arrays, secondary records, notes) goes to a side file and to stderr; stdout
        # carries ONE line below LINE_MAX_BYTES (the driver parses it: a 20 KB line was not parseable)
        detail = json.dumps(line)
        detail_name = "bench_detail.json" if world == 1 else f"bench_detail_n{world}.json"
        for d_ in (ROOT, __import__("tempfile").gettempdir()):
            try:
                with open(os.path.join(d_, detail_name), "w") as f_:
                    f_.write(detail + "\n")
                break
            except OSError:
                continue
        sys.stderr.write("bench detail: " + detail + "\n")
        sys.stderr.flush()
        short = json.dumps(compact_line(line, detail_name))
        assert len(short) < LINE_MAX_BYTES, len(short)
        line_out.write(short + "\n")
        line_out.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

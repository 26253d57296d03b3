"""Per-kernel table of one profiled program from the PMC passes of tools/prof_round.sh:
    python tools/pmc_table.py <gpurun_out/r05prof> <tag prefix, e.g. tb or c5> [min total us]
For every (kernel, grid): dispatches per run, mean duration (the kernel trace of the FETCH pass: durations under counter collection, a few
per cent above an unprofiled run), HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB, summed over the XCDs; FETCH_SIZE counts half of the bytes of
16-byte-per-lane reads on gfx950: MI355X_MICROARCH.md), the achieved HBM rate and its fraction of 8 TB/s, the matrix pipe's busy share
= SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), LDS bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE,
and the wave-time split (issuing / issue-stalled / parked = SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES)."""
import collections, csv, glob, os, sys
root, tag = sys.argv[1], sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0


def clean(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:60]


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, f"pmc_{tag}_{sub}", "**", "*counter_collection.csv"), recursive=True):
        per, names = collections.defaultdict(float), {}
        for row in csv.DictReader(open(f)):
            per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = (clean(row["Kernel_Name"]), int(row["Grid_Size"]))
        for (d, c), v in per.items():
            acc[names[d]][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def durations(sub):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, f"pmc_{tag}_{sub}", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            g = int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]) * int(row["Grid_Size_Z"])
            acc[(clean(row["Kernel_Name"]), g)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return acc


fetch, write, sq1, sq2 = counters("fetch"), counters("write"), counters("sq1"), counters("sq2")
dur = durations("fetch")
rows = []
for key, ds in dur.items():
    us = sum(ds) / len(ds)
    if us * len(ds) < min_us:
        continue
    f, w = fetch.get(key, {}).get("FETCH_SIZE"), write.get(key, {}).get("WRITE_SIZE")
    hbm = (2 * f + w) * 1024 if f is not None and w is not None else None
    s1, s2 = sq1.get(key, {}), sq2.get(key, {})
    mfma = s2.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1.0, 128.0 * s2.get("GRBM_GUI_ACTIVE", 0.0)) if s2 else None
    lds = s2.get("SQ_LDS_BANK_CONFLICT", 0.0) / s2["SQ_LDS_IDX_ACTIVE"] if s2.get("SQ_LDS_IDX_ACTIVE") else None
    wc = s1.get("SQ_WAVE_CYCLES")
    split = tuple(s1.get(c, 0.0) / wc for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")) if wc else None
    rows.append((us * len(ds), key, len(ds), us, hbm, mfma, lds, split))
rows.sort(reverse=True)
print(f"# {root} pmc_{tag}_*: kernel | grid (threads) | launches | mean us | HBM MB (2 FETCH + WRITE) | TB/s | of 8 TB/s | MFMA busy | LDS conflict share | issuing / issue-stalled / parked")
for tot, (name, grid), n, us, hbm, mfma, lds, split in rows:
    h = f"{hbm / 1e6:9.1f} {hbm / us / 1e6:5.2f} {hbm / us / 1e6 / 8:5.2f}" if hbm is not None else "        -     -     -"
    m = f"{mfma:5.2f}" if mfma is not None else "    -"
    l = f"{lds:5.2f}" if lds is not None else "    -"
    sp = "%.2f / %.2f / %.2f" % split if split else "-"
    print(f"{name:60s} {grid:9d} x{n:<3d} {us:8.1f} {h} {m} {l}  {sp}")

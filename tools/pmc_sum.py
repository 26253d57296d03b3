"""Aggregate rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch.
usage: python tools/pmc_sum.py <dir> [substring filter]"""
import csv, glob, os, sys, collections, json
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for row in csv.DictReader(open(f)):
        key = (f, row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[key] += float(row["Counter_Value"])
        names[(f, row["Dispatch_Id"])] = row["Kernel_Name"]
    for (ff, d, c), v in per_dispatch.items():
        kn = names[(ff, d)]
        if flt in kn:
            acc[kn.split("(")[0][:90]][c].append(v)
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": len(next(iter(cs.values())))} for k, cs in acc.items()}
print(json.dumps(out, indent=1))

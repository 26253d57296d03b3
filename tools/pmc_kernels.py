"""Mean PMC counter values per dispatch, grouped by (kernel name, grid): python tools/pmc_kernels.py <dir> [substring ...]"""
import csv, collections, glob, os, sys
root, subs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per, names = collections.defaultdict(float), {}
    for row in csv.DictReader(open(f)):
        per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        names[row["Dispatch_Id"]] = (row["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:48], row["Grid_Size"])
    for (d, c), v in per.items():
        acc[names[d]][c].append(v)
for k, cs in sorted(acc.items()):
    if subs and not any(s in k[0] for s in subs):
        continue
    print(k[0], k[1], {c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(cs.items())}, "x%d" % len(next(iter(cs.values()))))

"""Aggregate rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            short = name.split("(")[0].replace("void ", "")
            agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(agg):
    if not any(t in k for t in ("fwd", "bwd", "dw_", "reduce")):
        continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-28s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))

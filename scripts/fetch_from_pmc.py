"""HBM bytes per launch of the dW kernels from a rocprofv3 `--pmc FETCH_SIZE WRITE_SIZE` directory (gfx950: FETCH_SIZE counts
half the bytes of wide reads, MI355X_MICROARCH.md; units of 1 KB):   python scripts/fetch_from_pmc.py DIR [DIR ...]"""
import collections, csv, glob, os, sys

for root in sys.argv[1:]:
    val = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE") and "dw_" in r["Kernel_Name"]:
                val[(r["Kernel_Name"].split("(")[0].replace("void ", "")[:44], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    per = collections.defaultdict(list)
    for (name, _), c in val.items():
        per[name].append((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e9)
    print(os.path.basename(root))
    for name in sorted(per):
        v = sorted(per[name])
        print("  %-44s n=%3d  HBM traffic %.2f GB per launch (median; max %.2f)" % (name, len(v), v[len(v) // 2], v[-1]))

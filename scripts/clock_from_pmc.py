"""Clock a kernel ran at, from a rocprofv3 `--pmc GRBM_GUI_ACTIVE --kernel-trace` directory:
GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration.   python scripts/clock_from_pmc.py DIR [DIR ...]"""
import collections, csv, glob, os, sys

for root in sys.argv[1:]:
    dur, cyc = collections.defaultdict(list), collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[(r["Dispatch_Id"])] = (r["Kernel_Name"].split("(")[0].replace("void ", ""), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
                name, ns = dur[r["Dispatch_Id"]]
                if ns > 2e5:      # the 360 000-point launches (the boundary / value-mode launches are microseconds)
                    cyc[name].append((float(r["Counter_Value"]) / 8.0 / ns, ns * 1e-6))
    print(os.path.basename(root))
    for k in sorted(cyc):
        if any(t in k for t in ("fwd", "bwd", "dw_")):
            g = sorted(v[0] for v in cyc[k]); m = sorted(v[1] for v in cyc[k])
            print("  %-40s n=%3d  clock %.3f GHz (median)  %.3f ms (median, profiled)" % (k[:40], len(g), g[len(g) // 2], m[len(m) // 2]))

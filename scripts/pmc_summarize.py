"""Aggregate rocprofv3 --pmc counter_collection CSVs (one directory per pass, scripts/pmc_profile.sh):
per kernel name, the mean counter value per dispatch - printed as text, and with --json written as the
machine-readable record bench.py quotes its `traffic` / `matrix_pipe_busy_pmc` from:

    python scripts/pmc_summarize.py gpurun_out/pmc_<tag> [--json profiles/r03_pmc.json --precision bf16x3
                                                            --layers 6 --hidden 256 --points 360000]

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced streaming reads; WRITE_SIZE is exact).  Matrix-pipe busy =
SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs).  The JSON carries the content hash of
nsfnet_amd/csrc/ it was measured on; bench.py emits null for any other build."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def collect(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                short = name.split("(")[0].replace("void ", "")
                agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
                try:      # dispatch duration where the CSV carries timestamps: the clock the kernel ran at
                    dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                    if dur > 0 and row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                        agg[short]["_clock_ghz"].append(float(row["Counter_Value"]) / 8.0 / dur)
                except (KeyError, ValueError):
                    pass
    return agg


def plan_kernel_names(precision, layers, hidden, points):
    """The (forward, reverse sweep, dW) kernel families a residual plan of this shape launches, asked of the library
    itself (pinn_plan_kernel; plan creation needs no device).  Only THESE kernels get the shape's label: the same
    rocprof pass also sees the boundary / entropy-net launches of other plans (2052 points, value mode), whose
    counters must not be quoted for the 360 000-point launch."""
    import ctypes
    from nsfnet_amd import _lib
    from nsfnet_amd.engine import PRECISIONS
    lib = _lib.load()
    net, plan = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.pinn_net_create(3, layers, hidden, ctypes.byref(net)), "pinn_net_create")
    _lib.check(lib.pinn_net_set_precision(net, *[PRECISIONS[precision]] * 3), "pinn_net_set_precision")
    _lib.check(lib.pinn_plan_create(net, points, 4, ctypes.byref(plan)), "pinn_plan_create")
    names = [(lib.pinn_plan_kernel(plan, k) or b"").decode() for k in (0, 1, 2)]
    lib.pinn_plan_destroy(plan); lib.pinn_net_destroy(net)
    return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--json")
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--points", type=int, default=360000)
    ap.add_argument("--config", type=int, default=None, help="take layers / hidden / points from bench.py's BASELINE config N")
    args = ap.parse_args()
    if args.config:
        import bench
        c = bench.CONFIGS[args.config]
        args.layers, args.hidden, args.points = c["layers"], c["hidden"], c["grid"][0] * c["grid"][1]
    agg = collect(args.root)
    entries = []
    for k in sorted(agg):
        if not any(t in k for t in ("fwd", "bwd", "dw_", "reduce")):
            continue
        print(k)
        mean = {c: sum(v) / len(v) for c, v in agg[k].items()}
        for c, v in sorted(agg[k].items()):
            print("   %-28s n=%3d mean=%.6g" % (c, len(v), mean[c]))
        clock = mean.pop("_clock_ghz", None)
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            traffic = (2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0
            busy = None
            if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and mean.get("GRBM_GUI_ACTIVE", 0) > 0:
                busy = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (mean["GRBM_GUI_ACTIVE"] / 8.0)
            print("   %-28s %.4g GB   matrix-pipe busy %s" % ("=> HBM bytes / launch", traffic / 1e9,
                                                               "n/a" if busy is None else "%.3f" % busy))
            if clock:
                print("   %-28s %.3f GHz (GRBM_GUI_ACTIVE / 8 / dispatch time)" % ("=> clock held", clock))
            entries.append(dict(kernel=k, kernel_short=k.split("<")[0], traffic_bytes=traffic, mfma_busy=busy,
                                clock_ghz=clock, counters=mean))
    if args.json:
        import bench
        # one entry per kernel family the collocation plan launches: the instantiation that moves the most bytes
        # (residual mode, NS = 4).  Kernels of other plans in the same pass keep no entry.
        wanted = set(plan_kernel_names(args.precision, args.layers, args.hidden, args.points))
        best = {}
        for e in entries:
            if e["kernel_short"] not in wanted:
                continue
            if e["kernel_short"] not in best or e["traffic_bytes"] > best[e["kernel_short"]]["traffic_bytes"]:
                best[e["kernel_short"]] = e
        try:
            doc = json.load(open(args.json))
        except Exception:
            doc = {}
        h = bench.csrc_hash()
        if doc.get("csrc_hash") != h:
            doc = dict(csrc_hash=h, entries=[])
        try:
            doc["git"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True,
                                        text=True).stdout.strip()
        except Exception:
            pass
        doc["formula"] = "traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES/1024/(GRBM_GUI_ACTIVE/8)"
        keep = [e for e in doc["entries"] if not (e["precision"] == args.precision and e["layers"] == args.layers and
                                                  e["hidden"] == args.hidden and e["points"] == args.points)]
        for e in best.values():
            e.update(precision=args.precision, layers=args.layers, hidden=args.hidden, points=args.points)
            keep.append(e)
        doc["entries"] = keep
        json.dump(doc, open(args.json, "w"), indent=1, sort_keys=True)
        print("wrote %s (%d entries, csrc %s)" % (args.json, len(keep), h))


if __name__ == "__main__":
    main()

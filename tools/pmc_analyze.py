#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc ... --kernel-trace --output-format csv directory: per kernel name, mean counters."""
import collections, csv, glob, sys
d = sys.argv[1]
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
tr = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kt))}
agg = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    key = (r["Kernel_Name"].split("(")[0][-48:], r["Grid_Size"])
    a = agg.setdefault(key, {"n": {}, "v": collections.defaultdict(float), "dur": {}, })
    a["v"][r["Counter_Name"]] += float(r["Counter_Value"])
    a["n"][r["Dispatch_Id"]] = 1
    t = tr.get(r["Dispatch_Id"])
    if t: a["dur"][r["Dispatch_Id"]] = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
for (name, grid), a in agg.items():
    n = len(a["n"]); dur = sum(a["dur"].values()) / max(len(a["dur"]), 1)
    print("%-50s grid %-9s n=%d dur %.0f us" % (name, grid, n, dur))
    print("    " + "  ".join("%s=%.4g" % (k, v / n) for k, v in sorted(a["v"].items())))

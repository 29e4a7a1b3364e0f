#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc run:  python tools/pmc_kernels.py <dir> [substring]"""
import collections
import csv
import glob
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    if sub not in k:
        continue
    a = agg.setdefault(k, collections.OrderedDict(n=set()))
    a["n"].add(r["Dispatch_Id"])
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, a in agg.items():
    n = len(a.pop("n"))
    print(k, "launches", n)
    for c, v in a.items():
        print("   %-28s %16.0f  per launch %14.0f" % (c, v, v / n))

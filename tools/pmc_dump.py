#!/usr/bin/env python3
"""Print per-dispatch PMC counters of one rocprofv3 --pmc run: python tools/pmc_dump.py <dir> [kernel substring]"""
import collections, csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
by = collections.OrderedDict()
for r in rows:
    k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60])
    by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in by.items():
    if sub in k[1]:
        gui = v.get("GRBM_GUI_ACTIVE", 0) / 8
        extra = ""
        if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            extra = " mfma_util=%.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / gui)
        print(k[0], k[1], " ".join("%s=%.4g" % (n, x) for n, x in sorted(v.items())), extra)

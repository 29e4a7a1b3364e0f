#!/usr/bin/env python3
"""Per-kernel-family PMC table over ONE full bench step (the last of the run).

    python tools/summarize_step_pmc.py <sq_dir> <fetch_dir> <write_dir> [out.csv] [key=value ...]
(key=value pairs -- batch= height= width= arch= dtype= commit= -- are written to <out.csv>.meta.json: bench.py takes
``roofline.traffic`` from the table only when they match its own workload, and records the commit.)
Each directory = one `rocprofv3 --pmc ... --output-format csv -d <dir> -- python3 bench.py --steps 1 --warmup 1
--no-cpu-baseline --no-roofline` pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ... / FETCH_SIZE / WRITE_SIZE)."""
import collections
import csv
import glob
import sys


def load(d):
    rows = list(csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])))
    by = collections.OrderedDict()
    for r in rows:
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:64])
        by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return by


def family(name):
    fam = name.split("<")[0]
    args = [a.strip() for a in name.split("<")[1].split(">")[0].split(",")] if "<" in name else []
    if "wino" in fam and args:                        # keep the tile shape: same spelling as bench.py's table
        fam += "<%s>" % args[0]
    elif fam == "conv_bf16_kernel" and len(args) >= 5:    # <KS, S, KC, NPF, NA, FUSE, P3, F16, X3> -> bench.py's <KS,S,NA[,heads]>
        if len(args) > 8 and args[8] in ("true", "1"):     # split-operand instantiation: bench.py's conv_x3_kernel
            fam = "conv_x3_kernel"
        fam += "<%s,%s,%s%s>" % (args[0], args[1], args[4], ",heads" if len(args) > 5 and args[5] == "1" else "")
    elif fam == "dcn_x3_kernel" and len(args) >= 2:       # <MT, TW, ACT> -> <MT,TW>
        fam += "<%s,%s>" % (args[0], args[1])
    elif fam == "dcn_bf16_kernel" and args:               # <NA, R, PR> -> <NA>
        fam += "<%s>" % args[0]
    return fam


def last_step(by):
    """Dispatches of the last step: everything from the second-to-last stem launch on.  The passes are separate
    runs and the side streams interleave differently in each, so they are summed per kernel family, never paired."""
    keys = list(by.keys())
    stems = [i for i, k in enumerate(keys) if k[1].startswith("stem_")]
    # one stem launch per frame (f32 / un-fused bf16: stem_c16_kernel, bf16: stem_l0_bf16_kernel)
    per_step = 2 if len(stems) >= 2 and any(("c16" in keys[i][1] or "stem_l0" in keys[i][1]) for i in stems) else 1
    start = stems[-per_step]
    return [(k[1], by[k]) for k in keys[start:]]


def main():
    sq, fe, wr = (last_step(load(d)) for d in sys.argv[1:4])
    agg = collections.OrderedDict()
    for name, v in sq:
        a = agg.setdefault(family(name), dict(n=0, mfma=0.0, gui=0.0, fetch=0.0, write=0.0))
        a["n"] += 1
        a["mfma"] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        a["gui"] += v.get("GRBM_GUI_ACTIVE", 0) / 8
    for name, v in fe:
        if family(name) in agg:
            agg[family(name)]["fetch"] += v.get("FETCH_SIZE", 0)
    for name, v in wr:
        if family(name) in agg:
            agg[family(name)]["write"] += v.get("WRITE_SIZE", 0)
    lines = ["kernel,launches,cycles_per_XCD,mfma_pipe_util,FETCH_SIZE_MB_raw,WRITE_SIZE_MB_raw,HBM_GBs_at_2GHz_raw"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["gui"]):
        if a["gui"] < 2e4 or not name.strip():
            continue
        t = a["gui"] / 2.0e9
        lines.append("\"%s\",%d,%.0f,%.3f,%.1f,%.1f,%.0f" % (            # quoted: bf16 kernel families carry commas
            name, a["n"], a["gui"], a["mfma"] / 1024 / a["gui"], a["fetch"] * 1024 / 1e6, a["write"] * 1024 / 1e6,
            (a["fetch"] + a["write"]) * 1024 / 1e9 / t))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt + "\n")
        import json
        meta = {}
        for kv in sys.argv[5:]:
            k, v = kv.split("=", 1)
            meta[k] = int(v) if v.isdigit() else v
        json.dump(meta, open(sys.argv[4] + ".meta.json", "w"))


if __name__ == "__main__":
    main()

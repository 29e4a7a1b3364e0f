#!/usr/bin/env python3
"""Per-kernel-family PMC table over ONE full bench step (the last of the run).

    python tools/summarize_step_pmc.py <sq_dir> <fetch_dir> <write_dir> [out.csv]
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
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48])
        by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return by


def main():
    sq, fe, wr = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
    keys, fk, wk = list(sq.keys()), list(fe.keys()), list(wr.keys())
    start = [i for i, k in enumerate(keys) if k[1].startswith("stem_c16")][-2]      # first kernel of the last step
    agg = collections.OrderedDict()
    for idx in range(start, len(keys)):
        k, v = keys[idx], sq[keys[idx]]
        a = agg.setdefault(k[1].split("<")[0], dict(n=0, mfma=0.0, gui=0.0, fetch=0.0, write=0.0))
        a["n"] += 1
        a["mfma"] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        a["gui"] += v.get("GRBM_GUI_ACTIVE", 0) / 8
        if idx < len(fk) and fk[idx][1] == k[1]:
            a["fetch"] += fe[fk[idx]].get("FETCH_SIZE", 0)
        if idx < len(wk) and wk[idx][1] == k[1]:
            a["write"] += wr[wk[idx]].get("WRITE_SIZE", 0)
    lines = ["kernel,launches,cycles_per_XCD,mfma_pipe_util,FETCH_SIZE_MB_raw,WRITE_SIZE_MB_raw,HBM_GBs_at_2GHz"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["gui"]):
        if a["gui"] < 2e4 or not name.strip():
            continue
        t = a["gui"] / 2.0e9
        lines.append("%s,%d,%.0f,%.3f,%.1f,%.1f,%.0f" % (
            name, a["n"], a["gui"], a["mfma"] / 1024 / a["gui"], a["fetch"] * 1024 / 1e6, a["write"] * 1024 / 1e6,
            (a["fetch"] + a["write"]) * 1024 / 1e9 / t))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt + "\n")


if __name__ == "__main__":
    main()

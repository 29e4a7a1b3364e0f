#!/usr/bin/env python3
"""Condense the three rocprofv3 --pmc passes over tools/conv_bench.py into profiles/<tag>_pmc_layer_shapes.csv.

    python tools/summarize_pmc.py <fetch_dir> <write_dir> <sq_dir> <out.csv>
Each directory holds one *_counter_collection.csv (rocprofv3 --output-format csv)."""
import collections
import csv
import glob
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_bench import SHAPES          # noqa: E402  (same order as the launches in the trace)

NAMES = [sh[0] for sh in SHAPES]


def load(d):
    rows = list(csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])))
    by = collections.OrderedDict()
    for r in rows:
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:70])
        by.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [(k, v) for k, v in by.items() if "conv" in k[1] or "dcn" in k[1]]


def main():
    f, w, m = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
    with open(sys.argv[4], "w") as o:
        o.write("shape,kernel,FETCH_SIZE_KiB_raw,WRITE_SIZE_KiB_raw,SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE,"
                "mfma_pipe_util,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_WAVE_CYCLES,SQ_WAIT_INST_ANY,SQ_WAIT_ANY\n")
        for i, nm in enumerate(NAMES):                      # second launch of each shape (first = warm-up)
            (ka, a), (_, b), (_, c) = f[2 * i + 1], w[2 * i + 1], m[2 * i + 1]
            gui = c["GRBM_GUI_ACTIVE"] / 8
            o.write('%s,"%s",%.0f,%.0f,%.0f,%.0f,%.4f,%.0f,%.0f,%.0f,%.0f,%.0f\n' % (
                nm, ka[1], a["FETCH_SIZE"], b["WRITE_SIZE"], c["SQ_VALU_MFMA_BUSY_CYCLES"], c["GRBM_GUI_ACTIVE"],
                c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / gui, c["SQ_LDS_BANK_CONFLICT"], c["SQ_LDS_IDX_ACTIVE"],
                c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"], c["SQ_WAIT_ANY"]))
    print(open(sys.argv[4]).read())


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py with module-level tuning knobs overridden (tools only):

    python tools/knob_bench.py ops_bf16.HEADS_NPF=2 [ops.DCN_MT=2 ...] -- [bench.py arguments]"""
import importlib
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
args = sys.argv[1:]
rest = args[args.index("--") + 1:] if "--" in args else []
for kv in (args[:args.index("--")] if "--" in args else args):
    k, v = kv.split("=", 1)
    parts = k.split(".")                                  # module[.Class].attribute
    obj = importlib.import_module("sgtapose_amd." + parts[0])
    for a in parts[1:-1]:
        obj = getattr(obj, a)
    setattr(obj, parts[-1], eval(v))
sys.argv = ["bench.py"] + rest
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")

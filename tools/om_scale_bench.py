#!/usr/bin/env python3
"""bench.py --breakdown with the DCN sampling offsets scaled (tools only): how much time the far-sample paths cost.

    python tools/om_scale_bench.py <scale> [f32|bf16]"""
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import sgtapose_amd.ops as o  # noqa: E402
import sgtapose_amd.ops_bf16 as b  # noqa: E402

b.DEBUG_OM_SCALE = o.DEBUG_OM_SCALE = float(sys.argv[1])
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
sys.argv = ["bench.py", "--dtype", dt, "--breakdown", "--no-cpu-baseline"]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")

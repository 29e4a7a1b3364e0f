import sys, os, subprocess
sys.path.insert(0, "/root/repo")
import sgtapose_amd.ops_bf16 as b
b.DEBUG_OM_SCALE = float(sys.argv[1])
sys.argv = ["bench.py", "--dtype", "bf16", "--breakdown", "--no-cpu-baseline"]
import runpy
runpy.run_path("/root/repo/bench.py", run_name="__main__")

#!/usr/bin/env python3
"""Sampling-offset statistics of every DCN layer of the benched bf16 step (GPU box only, tools): how many samples leave
the +-R window of dcn_bf16.hip."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import sgtapose_amd.ops_bf16 as b  # noqa: E402
from sgtapose_amd.detector import BatchedDetector  # noqa: E402
from sgtapose_amd.synthetic import make_inputs  # noqa: E402

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev)
model.engine_dtype = "bf16"
eng = model.engine()
eng.use_side_stream = False
det = BatchedDetector(model)
inp = make_inputs(32, 480, 640, seed=317)
dev_in = [inp[k].to(dev) for k in bench.ARGS]
orig = b.dcn_v2


def hooked(x, om, pc, out, act=b.ACT_RELU):
    o = om.base[..., :18].float()
    a = o.abs()
    far = lambda r: float((a > r).any(-1).float().mean())                 # noqa: E731  pixels with at least one far tap
    fs = lambda r: float((a.view(*a.shape[:-1], 9, 2) > r).any(-1).float().mean())   # noqa: E731  far samples
    print("dcn %4d->%3d %3dx%3d  offset std %.2f  far samples: >5: %.3f  >6: %.3f  >10: %.3f  >20: %.3f   pixels with a far tap (>6): %.3f" % (
        pc.Cin, pc.Cout, x.H, x.W, float(o.std()), fs(5), fs(6), fs(10), fs(20), far(6)))
    return orig(x, om, pc, out, act)


b.dcn_v2 = hooked
import sgtapose_amd.engine_bf16 as e  # noqa: E402
e.bops.dcn_v2 = hooked
det.process_device(*dev_in, static=True)
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Un-profiled stage timeline of one step (GPU box only): HIP events recorded on the stream each stage is enqueued on,
third of three back-to-back steps (the host is then ahead of the GPU, as in bench.py).

    python tools/step_events.py [--dtype bf16|f32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from sgtapose_amd.detector import BatchedDetector  # noqa: E402
from sgtapose_amd.synthetic import make_inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--early-levels", default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")
model, sd = bench.build_model(dev)
if a.dtype == "bf16":
    model.engine_dtype = "bf16"
eng = model.engine()
if a.early_levels is not None:
    eng.early_levels = tuple(int(v) for v in a.early_levels.split(",") if v != "")
det = BatchedDetector(model)
inp = make_inputs(32, 480, 640, seed=317)
dev_in = [inp[k].to(dev) for k in bench.ARGS]
marks = []


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.current_stream())
    marks.append((name, ev))


def wrap(obj, name, label=None):
    fn = getattr(obj, name)

    def w(*args, **kw):
        mark((label or name) + ":begin")
        r = fn(*args, **kw)
        mark((label or name) + ":end")
        return r
    setattr(obj, name, w)


for n in ("backbone", "temporal", "temporal_tokens", "temporal_scatter", "neck", "heads"):
    wrap(eng, n)
for _ in range(5):
    det.process_device(*dev_in, static=True)
torch.cuda.synchronize()
for k in range(3):
    if k == 2:
        marks.clear()
        mark("step:begin")
    out, dets = det.process_device(*dev_in, static=True)
mark("step:end")
torch.cuda.synchronize()
t0 = marks[0][1]
for name, ev in marks:
    print("%9.3f ms  %s" % (t0.elapsed_time(ev), name))

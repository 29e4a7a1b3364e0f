#!/usr/bin/env python3
"""Step time of the f32x3 engine at the benched workload (B = 32, 480x640) under its engine switches:

    python tools/x3_variants.py

neck_parts (neck + heads as batch slices on streams), split_parts (backbone levels 2-5 as batch slices), fuse_stem.
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    from sgtapose_amd import _lib
    if os.environ.get("SGTA_LIB"):                 # A/B builds of the kernel library
        _lib.LIB_PATH = os.environ["SGTA_LIB"]
    only = sys.argv[1] if len(sys.argv) > 1 else None
    from bench import ARGS, build_model
    from sgtapose_amd.synthetic import make_inputs
    dev = torch.device("cuda", 0)
    m, _ = build_model(dev)
    m.engine_dtype = "f32x3"
    inp = make_inputs(32, 480, 640, seed=317)
    dev_in = [inp[k].to(dev) for k in ARGS]
    eng = m.engine()
    ref = None
    for name, kw in (("default", {}), ("neck_parts=2", {"neck_parts": 2}), ("neck_parts=4", {"neck_parts": 4}),
                     ("split_parts=1", {"split_parts": 1}), ("split_parts=4", {"split_parts": 4}),
                     ("fuse_stem=False", {"fuse_stem": False})):
        if only and name != only:
            continue
        saved = {k: getattr(eng, k) for k in kw}
        for k, v in kw.items():
            setattr(eng, k, v)
        for _ in range(4):
            out = eng.forward(*dev_in, static=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(15):
            out = eng.forward(*dev_in, static=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 15 * 1e3
        hm = out["hm"].clone()
        if ref is None:
            ref = hm
        print("%-18s %.3f ms/step  %.1f pairs/s   max |hm - default| %.2e" % (name, ms, 32e3 / ms, float((hm - ref).abs().max())), flush=True)
        for k, v in saved.items():
            setattr(eng, k, v)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A/B: the B = 32 batch step launched eagerly vs replayed from a HIP graph (torch.cuda.CUDAGraph), optionally with the neck
and heads as batch slices on two streams (EngineBF16.neck_parts) -- does removing the host from the step buy GPU time?

    python tools/graph_step_bench.py [--dtype f16] [--batch 32] [--steps 30]
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402

ARGS = ("x", "pre_img", "pre_hm", "repro_hm", "pre_hm_cls", "repro_hm_cls")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    args = ap.parse_args()
    from util import golden_state_dict
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
    from sgtapose_amd.synthetic import make_inputs
    dev = torch.device("cuda", 0)
    m = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(), default_opt())
    m.load_state_dict(golden_state_dict(0), strict=True)
    m = m.to(dev).eval()
    m.engine_dtype = args.dtype
    det = BatchedDetector(m)
    inp = make_inputs(args.batch, args.height, args.width, seed=317)
    dev_in = [inp[k].to(dev) for k in ARGS]
    eng = m.engine()

    def step():
        return det.process_device(*dev_in, static=True)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n, 1e3 * enq / n

    for parts in (1, 2, 4):
        if hasattr(eng, "neck_parts"):
            eng.neck_parts = parts
        elif parts > 1:
            break
        for _ in range(3):
            step()
        ms, enq = timed(step, args.steps)
        out, dets = step()
        ref = {k: v.clone() for k, v in out.items()}
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        g.replay()
        torch.cuda.synchronize()
        same = all(torch.equal(ref[k], out[k]) for k in ref)
        gms, genq = timed(g.replay, args.steps)
        print("%s B=%d neck_parts=%d: eager %.3f ms/step (host enqueue %.2f)  graph %.3f ms/step (host %.2f)  replay == eager: %s"
              % (args.dtype, args.batch, parts, ms, enq, gms, genq, same), flush=True)


if __name__ == "__main__":
    main()

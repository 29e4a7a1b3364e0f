#!/usr/bin/env python3
"""Per-layer timing of the deformable convs of the neck at the benched workload (B = 32, 480x640): the fp32 LDS-window kernel
(sgta_dcnv2_lds_f32) beside the split-operand kernel (sgta_dcnv2_x3) at both work-group widths.

    python tools/dcn_x3_bench.py [--om-scale 2.0] [--reps 10]

Offsets are normal with the given standard deviation in pixels (the synthetic network's offset convs produce ~1 px at
120x160 and ~4 px at 15x20; samples further than 6 rows / 8 columns from the tile take the global-memory fallback).
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

# (name, launches per step, Cin, Cout, H, W)
LAYERS = [
    ("dla_up.ida_0.proj_1", 1, 512, 256, 15, 20),
    ("dla_up.ida_0.node_1", 1, 256, 256, 30, 40),
    ("dla_up.ida_1.proj_1/2", 2, 256, 128, 30, 40),
    ("dla_up.ida_1.node_1/2", 2, 128, 128, 60, 80),
    ("ida_2 / ida_up proj 128->64", 4, 128, 64, 60, 80),
    ("ida_up.proj_2", 1, 256, 64, 30, 40),
    ("nodes 64->64", 5, 64, 64, 120, 160),
]


def timed(fn, reps):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--om-scale", type=float, default=2.0)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--zero-input", action="store_true", help="all-zero feature maps: the same launches without the data's power draw")
    args = ap.parse_args()
    from sgtapose_amd import _lib, ops, ops_x3 as xops
    if os.environ.get("SGTA_LIB"):                 # ablation builds (tools/dcn_x3_ablate.sh)
        _lib.LIB_PATH = os.environ["SGTA_LIB"]
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    B = args.batch
    tot = {"f32": 0.0, "x3": 0.0}
    print("%-30s %9s %9s %9s %9s" % ("layer", "f32 ms", "x3 mt=2", "x3 mt=4", "TFLOP/s(best x3)"))
    for name, cnt, cin, cout, H, W in LAYERS:
        x = torch.randn(B, cin, H, W, generator=g).to(dev)
        if args.zero_input:
            x.zero_()
        w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        om = (torch.randn(B, 27, H, W, generator=g) * args.om_scale).to(dev)
        pc = ops.PackedConv(w.to(dev), None, tap_major=True)
        pk = xops.PackedDCNX3(pc)
        out = torch.empty(B, cout, H, W, device=dev)
        t32 = timed(lambda: ops.dcn_v2(x, om, pc, out, ops.ACT_RELU), args.reps)
        t2 = timed(lambda: xops.dcn_v2(x, om, pk, out, ops.ACT_RELU, mt=2), args.reps)
        t4 = timed(lambda: xops.dcn_v2(x, om, pk, out, ops.ACT_RELU, mt=4), args.reps) if cout % 128 == 0 else float("nan")
        best = min(t2, t4) if t4 == t4 else t2
        fl = 2.0 * B * H * W * cout * cin * 9
        print("%-30s %9.4f %9.4f %9.4f %9.1f   x%d per step, default tile %d" % (name, t32, t2, t4, fl / best / 1e9, cnt,
                                                                              pc.dcn_tile(H, W, B)[0]), flush=True)
        tot["f32"] += cnt * t32
        tot["x3"] += cnt * best
    print("sum over the step's 16 launches: fp32 %.3f ms, split-operand (best width) %.3f ms" % (tot["f32"], tot["x3"]))


if __name__ == "__main__":
    main()

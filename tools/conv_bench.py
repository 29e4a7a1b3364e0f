#!/usr/bin/env python3
"""Kernel-level timing of the conv / DCN layer shapes of the B=32 480x640 workload (GPU box only).

    python tools/conv_bench.py [--reps 5] [--only NAME] [--tile T]
Prints per-shape ms and algorithmic TFLOP/s; interleaves variants in one process (guide rule 24)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sgtapose_amd import ops  # noqa: E402

# name, kind, Cin, Cout, ks, stride, H, W, images
SHAPES = [
    ("stem_img", "conv", 3, 16, 7, 1, 480, 640, 32),
    ("stem_hm", "conv", 1, 16, 7, 1, 480, 640, 32),
    ("level0", "conv", 16, 16, 3, 1, 480, 640, 64),
    ("level1", "conv", 16, 32, 3, 2, 480, 640, 64),
    ("l2_s2", "conv", 32, 64, 3, 2, 240, 320, 64),
    ("l3_s2", "conv", 64, 128, 3, 2, 120, 160, 64),
    ("l5_s2", "conv", 256, 512, 3, 2, 30, 40, 64),
    ("l2_3x3", "conv", 64, 64, 3, 1, 120, 160, 64),
    ("l3_3x3", "conv", 128, 128, 3, 1, 60, 80, 64),
    ("l4_3x3", "conv", 256, 256, 3, 1, 30, 40, 64),
    ("l5_3x3", "conv", 512, 512, 3, 1, 15, 20, 64),
    ("l3_root", "conv", 448, 128, 1, 1, 60, 80, 64),
    ("l2_root", "conv", 128, 64, 1, 1, 120, 160, 64),
    ("l5_root", "conv", 1280, 512, 1, 1, 15, 20, 64),
    ("head_hm", "conv", 256, 7, 1, 1, 120, 160, 32),
    ("heads0", "conv", 64, 768, 3, 1, 120, 160, 32),
    ("om_64", "conv", 64, 27, 3, 1, 120, 160, 32),
    ("om_128", "conv", 128, 27, 3, 1, 60, 80, 32),
    ("om_256", "conv", 256, 27, 3, 1, 30, 40, 32),
    ("om_512", "conv", 512, 27, 3, 1, 15, 20, 32),
    ("dcn_64_64", "dcn", 64, 64, 3, 1, 120, 160, 32),
    ("dcn_128_64", "dcn", 128, 64, 3, 1, 60, 80, 32),
    ("dcn_128_128", "dcn", 128, 128, 3, 1, 60, 80, 32),
    ("dcn_512_256", "dcn", 512, 256, 3, 1, 15, 20, 32),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--tiles", default="0")
    ap.add_argument("--generic", action="store_true", help="force the implicit-GEMM gather kernel")
    ap.add_argument("--no-wino4", action="store_true", help="Winograd F(2x2,3x3) everywhere (no F(4x4,3x3))")
    ap.add_argument("--no-wino", action="store_true", help="direct (LDS-range) 3x3 kernel instead of Winograd")
    a = ap.parse_args()
    dev = "cuda:0"
    ops.USE_PATCH_KERNEL = not a.generic
    ops.USE_CONV1X1 = not a.generic
    ops.USE_S2_KERNEL = not a.generic
    ops.USE_WINOGRAD = not (a.generic or a.no_wino)
    ops.USE_WINO4 = not a.no_wino4
    ops.USE_DCN_LDS_KERNEL = not a.generic
    tiles = [int(t) for t in a.tiles.split(",")]
    for name, kind, ci, co, ks, st, H, W, N in SHAPES:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        g = torch.Generator().manual_seed(1)
        x = torch.randn(N, ci, H, W, device=dev)
        w = (torch.randn(co, ci, ks, ks, generator=g) / (ci * ks * ks) ** 0.5).to(dev)
        Ho, Wo = H // st, W // st
        out = torch.empty(N, co, Ho, Wo, device=dev)
        if kind == "dcn":
            pc = ops.PackedConv(w, None, tap_major=True)
            om = torch.randn(N, 27, H, W, device=dev) * float(os.environ.get("OM_SCALE", "1.0"))
            def run(t, x=x, om=om, pc=pc, out=out):
                if a.generic:
                    return ops.dcn_v2(x, om, pc, out, ops.ACT_RELU, tile=t)
                return ops.dcn_v2(x, om, pc, out, ops.ACT_RELU, tile=0)
        else:
            pc = ops.PackedConv(w, None, stride=st)
            run = lambda t: ops.conv2d(x, pc, out, ops.ACT_RELU, tile=t)
        flops = pc.flops(N * Ho * Wo)
        res = {}
        for t in tiles:
            try:
                run(t)
            except Exception as e:       # tile not valid for this Cout
                res[t] = None
                continue
            res[t] = []
        torch.cuda.synchronize()
        for _ in range(a.reps):
            for t in tiles:
                if res[t] is None:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(t)
                e1.record()
                torch.cuda.synchronize()
                res[t].append(e0.elapsed_time(e1))
        line = "%-12s %s %4d->%-4d k%d s%d %3dx%-3d N=%d " % (name, kind, ci, co, ks, st, H, W, N)
        for t in tiles:
            if res[t]:
                ms = sorted(res[t])[len(res[t]) // 2]
                line += "| tile%d %7.3f ms %6.1f TF " % (t, ms, flops / ms / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()

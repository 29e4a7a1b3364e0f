#!/usr/bin/env python3
"""Per-layer timing of the 16-bit convolution kernel (sgta_conv_bf16 / sgta_conv_f16) on the layer shapes of the benched
workload (B = 32 frame pairs: 64 images through the backbone, 32 through the neck / heads), and the measured cost of
emulating fp32 on the same kernel by operand splitting:

    python tools/conv16_bench.py [--dtype f16] [--x3]

--x3: every layer again with THREE TIMES the input channels.  An fp32 convolution emulated with 16-bit operands (x = x_hi +
x_lo, w = w_hi + w_lo, three products x_hi w_hi + x_lo w_hi + x_hi w_lo accumulated in fp32: "bf16x3") is exactly a 16-bit
convolution over the channel concatenation [x_hi, x_lo, x_hi] with weights [w_hi, w_hi, w_lo] -- same kernel, 3x the chunks,
2x the activation bytes -- so its time is measured here without writing the emulated kernel.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

# (name, count per step, N images, Cin, Cout, H, W, ks, stride)
LAYERS = [
    ("level0 (in the fused stem kernel: as a plain conv here)", 1, 64, 16, 16, 480, 640, 3, 1),
    ("level1", 1, 64, 16, 32, 480, 640, 3, 2),
    ("level2 3x3 s2", 1, 64, 32, 64, 240, 320, 3, 2),
    ("level2 3x3", 3, 64, 64, 64, 120, 160, 3, 1),
    ("level2 root 1x1", 1, 64, 128, 64, 120, 160, 1, 1),
    ("level3 3x3 s2", 1, 64, 64, 128, 120, 160, 3, 2),
    ("level3 3x3", 7, 64, 128, 128, 60, 80, 3, 1),
    ("level3 root 1x1 (448)", 1, 64, 448, 128, 60, 80, 1, 1),
    ("level4 3x3 s2", 1, 64, 128, 256, 60, 80, 3, 2),
    ("level4 3x3", 7, 64, 256, 256, 30, 40, 3, 1),
    ("level4 root 1x1 (896)", 1, 64, 896, 256, 30, 40, 1, 1),
    ("level5 3x3 s2", 1, 64, 256, 512, 30, 40, 3, 2),
    ("level5 3x3", 3, 64, 512, 512, 15, 20, 3, 1),
    ("level5 root 1x1 (1280)", 1, 64, 1280, 512, 15, 20, 1, 1),
    ("offset conv 64ch @120x160", 7, 32, 64, 32, 120, 160, 3, 1),
    ("offset conv 128ch @60x80", 5, 32, 128, 32, 60, 80, 3, 1),
    ("offset conv 256ch @30x40", 3, 32, 256, 32, 30, 40, 3, 1),
    ("heads 3x3 64->768 (fused kernel in the engine: plain conv here)", 1, 32, 64, 768, 120, 160, 3, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16"])
    ap.add_argument("--x3", action="store_true")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    from sgtapose_amd import _lib, ops_bf16 as bops
    if os.environ.get("SGTA_LIB"):                 # ablation builds (-DCB_ABLATE=k: 1 no window fetch, 2 no weight fetch)
        _lib.LIB_PATH = os.environ["SGTA_LIB"]
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    tot = {1: 0.0, 3: 0.0}
    print("%-62s %5s %9s %9s %s" % ("layer", "x", "ms/launch", "TFLOP/s", "(direct-form FLOPs of the fp32 layer)"))
    for name, cnt, N, cin, cout, H, W, ks, stride in LAYERS:
        for mult in ((1, 3) if args.x3 else (1,)):
            ci = cin * mult
            if ci % (16 if ks == 3 else 32):
                ci = -(-ci // 32) * 32
            w = torch.randn(cout, ci, ks, ks, generator=g) * (1.0 / (ci * ks * ks) ** 0.5)
            pc = bops.PackedConvBF16(w.to(dev), None, None, stride=stride, dtype=dt)
            x = bops.ViewH(torch.randn(N, ci // 16, H, W, 16, device=dev).to(dt))
            Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
            yv = y = bops.ViewH(torch.empty(N, cout // 16, Ho, Wo, 16, dtype=dt, device=dev))
            for _ in range(2):
                bops.conv(x, pc, yv, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                bops.conv(x, pc, yv, 1)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            fl = 2.0 * N * Ho * Wo * cout * cin * ks * ks            # the fp32 layer's algorithmic FLOPs
            print("%-62s x%-4d %9.4f %9.1f  x%d per step" % (name, mult, ms, fl / ms / 1e9, cnt), flush=True)
            tot[mult] += ms * cnt
            del x, y, pc
    print("sum over the step's conv launches: %.3f ms" % tot[1] + (", with 3x channels (emulated fp32): %.3f ms" % tot[3] if args.x3 else ""))


if __name__ == "__main__":
    main()

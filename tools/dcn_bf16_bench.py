#!/usr/bin/env python3
"""Micro-benchmark of sgta_dcnv2_fwd_bf16 / sgta_conv_bf16 on the layer shapes of the bf16 step (GPU box only).
    python tools/dcn_bf16_bench.py [dcn|conv]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sgtapose_amd import _lib  # noqa: E402
if os.environ.get("SGTA_LIB"):                 # ablation builds (tools/dcn_ablate.sh)
    _lib.LIB_PATH = os.environ["SGTA_LIB"]
from sgtapose_amd import ops_bf16 as bops  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def dcn():
    if 'MARGIN' in os.environ:
        bops.DCN_MARGIN = int(os.environ['MARGIN'])
    g = torch.Generator().manual_seed(0)
    for (N, cin, cout, H, W) in ((32, 64, 64, 120, 160), (32, 128, 64, 60, 80), (32, 256, 64, 30, 40), (32, 128, 128, 60, 80),
                                 (32, 256, 256, 30, 40), (32, 512, 256, 15, 20)):
        x = torch.randn(N, cin // 16, H, W, 16, generator=g).to(BF).to(DEV)
        w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        pc = bops.PackedDCNBF16(w.to(DEV), torch.zeros(cout, device=DEV), na=(int(os.environ['NA']) if 'NA' in os.environ else None))
        out = torch.empty(N, cout // 16, H, W, 16, dtype=BF, device=DEV)
        for name, scale in (("zero", 0.0), ("s1", 1.0), ("s3", 3.0), ("s10", 10.0)):
            om = torch.zeros(N, H, W, 32, device=DEV)
            om[..., :27] = torch.randn(N, H, W, 27, generator=g).to(DEV) * scale
            ms = timeit(lambda: bops.dcn_v2(bops.ViewH(x), bops.ViewH(om), pc, bops.ViewH(out)))
            fl = 2.0 * N * H * W * cin * cout * 9
            print("dcn %4d->%3d %3dx%3d offsets %-4s %7.3f ms  %6.1f TFLOP/s" % (cin, cout, H, W, name, ms, fl / ms / 1e9), flush=True)


def conv():
    g = torch.Generator().manual_seed(0)
    shapes = ((64, 16, 16, 480, 640, 3, 1), (64, 64, 64, 120, 160, 3, 1), (64, 128, 128, 60, 80, 3, 1), (64, 256, 256, 30, 40, 3, 1),
              (64, 512, 512, 15, 20, 3, 1), (32, 64, 768, 120, 160, 3, 1), (32, 64, 32, 120, 160, 3, 1), (64, 128, 64, 120, 160, 1, 1),
              (64, 448, 128, 60, 80, 1, 1), (64, 896, 256, 30, 40, 1, 1), (64, 1280, 512, 15, 20, 1, 1), (64, 64, 128, 120, 160, 3, 2))
    for (N, cin, cout, H, W, ks, st) in shapes:
        x = torch.randn(N, cin // 16, H, W, 16, generator=g).to(BF).to(DEV)
        w = torch.randn(cout, cin, ks, ks, generator=g) * 0.05
        pc = bops.PackedConvBF16(w.to(DEV), torch.zeros(cout, device=DEV), stride=st, na=(int(os.environ['NA']) if 'NA' in os.environ else None))
        Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
        out = torch.empty(N, cout // 16, Ho, Wo, 16, dtype=BF, device=DEV)
        for npf in ((1,) if st == 2 else (2, 1)):
            bops.FORCE_NPF = npf
            ms = timeit(lambda: bops.conv(bops.ViewH(x), pc, bops.ViewH(out), 1))
            bops.FORCE_NPF = None
            fl = 2.0 * N * Ho * Wo * cin * cout * ks * ks
            by = 2.0 * (N * H * W * cin + N * Ho * Wo * cout)
            print("conv%d s%d %4d->%3d %3dx%3d npf %d na %d  %7.3f ms  %7.1f TFLOP/s %6.0f GB/s" % (
                ks, st, cin, cout, H, W, npf, pc.na, ms, fl / ms / 1e9, by / ms / 1e6), flush=True)


if __name__ == "__main__":
    (conv if (len(sys.argv) > 1 and sys.argv[1] == "conv") else dcn)()

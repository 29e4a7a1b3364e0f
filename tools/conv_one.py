#!/usr/bin/env python3
"""One 16-bit convolution layer, a few launches (the program profiled under rocprofv3 --pmc when a single kernel is studied):

    python tools/conv_one.py [N Cin Cout H W ks stride reps]      default: the level-3 3x3 layer of the benched workload
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def main():
    a = [int(v) for v in sys.argv[1:]]
    N, cin, cout, H, W, ks, stride, reps = a + [64, 128, 128, 60, 80, 3, 1, 50][len(a):]
    from sgtapose_amd import _lib, ops_bf16 as bops
    if os.environ.get("SGTA_LIB"):
        _lib.LIB_PATH = os.environ["SGTA_LIB"]
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    dt = torch.float16
    w = torch.randn(cout, cin, ks, ks, generator=g) * (1.0 / (cin * ks * ks) ** 0.5)
    pc = bops.PackedConvBF16(w.to(dev), None, None, stride=stride, dtype=dt)
    x = bops.ViewH(torch.randn(N, cin // 16, H, W, 16, device=dev).to(dt))
    if os.environ.get("CONV_ONE_ZERO_INPUT"):      # the same launches on an all-zero input: what the DATA costs (power, clocks)
        x.base.zero_()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = bops.ViewH(torch.empty(N, cout // 16, Ho, Wo, 16, dtype=dt, device=dev))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bops.conv(x, pc, y, 1)
    e0.record()
    for _ in range(reps):
        bops.conv(x, pc, y, 1)
    e1.record()
    torch.cuda.synchronize()
    print("%.4f ms/launch" % (e0.elapsed_time(e1) / reps))


if __name__ == "__main__":
    main()

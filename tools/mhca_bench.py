#!/usr/bin/env python3
"""Timing of the fused attention core at the three transformer levels (GPU box only): bias as [heads, n, n] (1),
re-packed into 16 x 16 tiles (2), and without bias (0)."""
import sys, torch
sys.path.insert(0, ".")
from sgtapose_amd import ops
dev = "cuda:0"
B = 32
for (lvl, n, C, hid) in [(0, 1183, 16, 32), (1, 343, 32, 64), (2, 63, 64, 128)]:
    M = B * n
    q = torch.randn(M, hid, device=dev); kv = torch.randn(M, 2 * hid, device=dev)
    pos = torch.randn(8, n, n, device=dev) * 0.02
    out = torch.empty(M, hid, device=dev)
    tiles = ops.PosTiles(pos)
    for use_pos in (1, 2, 0):
        for _ in range(3):
            ops.mhca(q, kv[:, :hid], kv[:, hid:], (None, pos, tiles)[use_pos], out, B, n, 8, hid // 8)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.mhca(q, kv[:, :hid], kv[:, hid:], (None, pos, tiles)[use_pos], out, B, n, 8, hid // 8)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        flops = 4.0 * B * 8 * n * n * (hid // 8)
        print("level %d n=%4d d=%2d bias=%d : %.3f ms  %.1f TFLOP/s  bias stream %.0f GB/s" % (
            lvl, n, hid // 8, use_pos, ms, flops / ms / 1e9, (B * 8 * n * n * 4 / ms / 1e6) if use_pos else 0))

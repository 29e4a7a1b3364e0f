import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from sgtapose_amd import ops
DEV = "cuda:0"
for (cin, cout, H, W) in [(16, 32, 24, 40), (16, 32, 16, 16), (16, 32, 4, 64), (16, 32, 8, 32), (16, 32, 32, 8), (24, 32, 32, 8), (16, 64, 32, 8)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.zeros(cout)
    ref = F.conv2d(x, w, bias, padding=1)
    pc = ops.PackedConv(w.to(DEV), bias.to(DEV))
    out = torch.zeros(ref.shape, device=DEV)
    ops.conv2d(x.to(DEV), pc, out, ops.ACT_NONE)
    e = (out.cpu() - ref).abs()
    print(cin, cout, H, W, "tile", pc.wino_tile(H, W), "max err", float(e.max()))
    bad = (e > 1e-3)
    print(" bad rows:", sorted(set(bad.nonzero()[:, 2].tolist()))[:40])
    print(" bad cols:", sorted(set(bad.nonzero()[:, 3].tolist()))[:40])
    print(" bad couts:", sorted(set(bad.nonzero()[:, 1].tolist()))[:40])

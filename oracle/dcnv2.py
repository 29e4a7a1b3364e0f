"""Oracle: modulated deformable convolution v2 (DCNv2), 3x3 / stride 1 / pad 1 / dil 1 / 1 group.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  **Parity unpinned**: the operator's source is the
un-vendored lbin/DCNv2 (branch pytorch_1.9) that the reference builds from a git clone
(reference README.md:21-28; import dla.py:21-25; the only constructor call is dla.py:545 with
kernel (3,3), stride 1, padding 1, dilation 1, deformable_groups 1; forward call dla.py:548).
What follows restates that project's published algorithm (its ``dcn_v2.py::DCN.forward`` and the
``modulated_deformable_im2col`` + GEMM pair):

    om        = conv3x3(x; conv_offset_mask)                       27 channels
    o1,o2,m   = chunk(om, 3);  offset = cat(o1, o2) == om[:, :18];  mask = sigmoid(om[:, 18:])
    tap k = ky*3+kx :  dy = offset[2k],  dx = offset[2k+1]
    py = h - 1 + ky + dy ;  px = w - 1 + kx + dx
    col[c,k,h,w] = mask[k,h,w] * bilinear(x[c], py, px)    if -1 < py < H and -1 < px < W else 0
    y[o,h,w]  = bias[o] + sum_{c,k} weight[o,c,ky,kx] * col[c,k,h,w]

bilinear: corners (y0,x0)=(floor py, floor px) .. (y0+1,x0+1); a corner contributes only when it
lies inside the image; weights (1-ly)(1-lx), (1-ly)lx, ly(1-lx), ly*lx, summed in that order.
"""
import torch
import torch.nn.functional as F


def dcn_v2_im2col(x, offset, mask):
    """x [B,C,H,W] f32, offset [B,18,H,W], mask [B,9,H,W] (already sigmoid-ed) -> col [B,C,9,H,W]."""
    B, C, H, W = x.shape
    dev = x.device
    hh = torch.arange(H, device=dev, dtype=torch.float32).view(1, H, 1)
    ww = torch.arange(W, device=dev, dtype=torch.float32).view(1, 1, W)
    xf = x.reshape(B, C, H * W)
    cols = []
    for k in range(9):
        ky, kx = divmod(k, 3)
        py = hh + float(ky - 1) + offset[:, 2 * k]          # [B,H,W]
        px = ww + float(kx - 1) + offset[:, 2 * k + 1]
        inside = (py > -1) & (px > -1) & (py < H) & (px < W)
        y0 = torch.floor(py)
        x0 = torch.floor(px)
        ly = py - y0
        lx = px - x0
        hy = 1 - ly
        hx = 1 - lx
        y0i = y0.long()
        x0i = x0.long()
        y1i = y0i + 1
        x1i = x0i + 1
        acc = None
        for (yi, xi, wgt, ok) in (
            (y0i, x0i, hy * hx, (y0i >= 0) & (x0i >= 0)),
            (y0i, x1i, hy * lx, (y0i >= 0) & (x1i <= W - 1)),
            (y1i, x0i, ly * hx, (y1i <= H - 1) & (x0i >= 0)),
            (y1i, x1i, ly * lx, (y1i <= H - 1) & (x1i <= W - 1)),
        ):
            ok = ok & inside
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).view(B, 1, H * W).expand(B, C, H * W)
            v = torch.gather(xf, 2, idx).view(B, C, H, W)
            v = torch.where(ok.view(B, 1, H, W), v, torch.zeros((), dtype=x.dtype, device=dev))
            term = wgt.view(B, 1, H, W) * v
            acc = term if acc is None else acc + term
        cols.append(acc * mask[:, k].view(B, 1, H, W))
    return torch.stack(cols, dim=2)


def dcn_v2_forward(x, weight, bias, om_weight, om_bias):
    """Full DCN module forward: x [B,Ci,H,W] -> [B,Co,H,W] (no BN / ReLU; those are DeformConv's)."""
    om = F.conv2d(x, om_weight, om_bias, stride=1, padding=1)
    offset = om[:, :18]
    mask = torch.sigmoid(om[:, 18:27])
    col = dcn_v2_im2col(x, offset, mask)                       # [B,Ci,9,H,W]
    B, Ci, _, H, W = col.shape
    Co = weight.shape[0]
    y = torch.matmul(weight.reshape(Co, Ci * 9), col.reshape(B, Ci * 9, H * W))
    return y.view(B, Co, H, W) + bias.view(1, Co, 1, 1)


class OracleDCN(torch.nn.Module):
    """nn.Module with the lbin/DCNv2 ``DCN`` constructor signature and parameter names; used by
    tests/golden/make_golden.py as the reference network's DCN (the real extension is absent)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding,
                 dilation=1, deformable_groups=1):
        super().__init__()
        ks = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        assert tuple(ks) == (3, 3) and stride == 1 and padding == 1 and dilation == 1
        assert deformable_groups == 1
        self.weight = torch.nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        self.conv_offset_mask = torch.nn.Conv2d(in_channels, 27, 3, stride=1, padding=1, bias=True)
        stdv = 1.0 / (in_channels * 9) ** 0.5
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()

    def forward(self, x):
        return dcn_v2_forward(x, self.weight, self.bias,
                              self.conv_offset_mask.weight, self.conv_offset_mask.bias)

"""``DCN``: drop-in for ``from .DCNv2.dcn_v2 import DCN`` (reference dla.py:21-25).

Same constructor signature and parameter names as the external lbin/DCNv2 module the reference
builds from a git clone (README.md:21-28): ``weight [Co,Ci,3,3]``, ``bias [Co]``,
``conv_offset_mask.{weight [27,Ci,3,3], bias [27]}`` (zero-initialised).  ``forward`` runs the fused
HIP kernels (offset/mask conv as an implicit GEMM, then gather+GEMM) -- there is no PyTorch fallback.
Only the configuration the reference uses is supported (3x3, stride 1, pad 1, dil 1, 1 group;
dla.py:545).
"""
import math

import torch
from torch import nn


class DCN(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        ks = tuple(kernel_size) if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        if ks != (3, 3) or stride != 1 or padding != 1 or dilation != 1 or deformable_groups != 1:
            raise ValueError("DCN: only kernel 3x3, stride 1, padding 1, dilation 1, deformable_groups 1 "
                             "(the reference's only configuration, dla.py:545) is implemented")
        if in_channels % 16:
            raise ValueError("DCN: in_channels must be a multiple of 16")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.conv_offset_mask = nn.Conv2d(in_channels, 27, kernel_size=3, stride=1, padding=1, bias=True)
        stdv = 1.0 / math.sqrt(in_channels * 9)
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()
        self._packed = None

    def _pack(self):
        from . import ops
        key = (self.weight._version, self.bias._version, self.conv_offset_mask.weight._version,
               self.conv_offset_mask.bias._version, self.weight.device)
        if self._packed is None or self._packed[0] != key:
            om = ops.PackedConv(self.conv_offset_mask.weight, self.conv_offset_mask.bias)
            main = ops.PackedConv(self.weight, self.bias, tap_major=True)
            self._packed = (key, om, main)
        return self._packed[1], self._packed[2]

    def forward(self, x):
        from . import ops
        x = x.contiguous().float()
        B, _, H, W = x.shape
        pc_om, pc = self._pack()
        om = torch.empty(B, 27, H, W, dtype=torch.float32, device=x.device)
        out = torch.empty(B, self.out_channels, H, W, dtype=torch.float32, device=x.device)
        ops.conv2d(x, pc_om, om)
        ops.dcn_v2(x, om, pc, out)
        return out

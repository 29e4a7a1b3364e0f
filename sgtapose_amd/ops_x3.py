"""Python faces of the split-operand ("x3") kernels: fp32 emulated on the 16-bit matrix pipe.

An activation is TWO IEEE-half planes of the blocked channels-last layout, value = hi + lo with hi = half(value) and lo =
half(value - hi) (22 significant bits), held in ONE allocation [2, N, C/16, H, W, 16]; weights are split the same way and
every product is x_hi w_hi + x_hi w_lo + x_lo w_hi accumulated in fp32.  For the convolution kernel that is a 16-bit
convolution over three times the channel chunks (conv_bf16.hip, X3): the matrix work of three 16-bit convs (2500 / 3 = 830
TFLOP/s of fp32-grade products at peak against the f32 MFMA's 157).  The fused stems + level 0 (stem_l0_bf16.hip) and the
deformable conv (dcn_x3.hip, fp32 NCHW in and out) have split-operand kernels of their own.  torch is plumbing only.
"""
import torch

from . import _lib
from .ops import ACT_NONE, fold_bn, _stream
from .ops_bf16 import F16, ViewH, PackedConvBF16, PackedStemL0BF16, _swizzle_rows, HEADS_NPF

OUT_X3_BLOCKED, OUT_F32_NCHW, OUT_F32_NHWC = 0, 1, 2


def split16(t):
    """float tensor -> (hi, lo) IEEE-half parts, hi + lo == t to 22 bits (clamped to the half range)."""
    t = t.float().clamp(-65504.0, 65504.0)
    hi = t.to(F16)
    return hi, (t - hi.float()).to(F16)


class ViewX(ViewH):
    """A ViewH over the hi plane of an x3 tensor [2, N, C/16, H, W, 16]; ``lo_delta`` = elements to the lo plane."""
    __slots__ = ("lo_delta",)

    def __init__(self, base, c0=0, C=None, lo_delta=None):
        super().__init__(base, c0, C)
        assert base.dtype == F16 and lo_delta is not None and lo_delta > 0 and lo_delta % 8 == 0
        self.lo_delta = int(lo_delta)

    @classmethod
    def alloc(cls, N, C, H, W, device):
        full = torch.empty((2, N, C // 16, H, W, 16), dtype=F16, device=device)
        return cls.of(full)

    @classmethod
    def of(cls, full):
        """full: the [2, N, C/16, H, W, 16] allocation."""
        assert full.dim() == 6 and full.shape[0] == 2 and full.is_contiguous()
        return cls(full[0], 0, None, full[0].numel())

    def slice(self, c0, C):
        return ViewX(self.base, self.c0 + c0, C, self.lo_delta)

    def images(self, lo, hi):
        return ViewX(self.base[lo:hi], self.c0, self.C, self.lo_delta)

    @property
    def lo(self):
        """The lo plane as a plain ViewH (tests)."""
        flat = torch.as_strided(self.base, self.base.shape, self.base.stride(), self.base.storage_offset() + self.lo_delta)
        return ViewH(flat, self.c0, self.C)

    def nchw(self):
        return ViewH.nchw(self) + self.lo.nchw()

    # bytes addressable from the view's first element to the end of the lo plane's last image
    nbytes = property(lambda s: (s.lo_delta + s.base.numel() - (s.c0 * s.H * s.W)) * 2)


def x3_from_nchw(t, device):
    """float [N, C, H, W] -> ViewX (tests, boundary conversions on the host)."""
    N, C, H, W = t.shape
    blk = t.reshape(N, C // 16, 16, H, W).permute(0, 1, 3, 4, 2).contiguous()
    hi, lo = split16(blk)
    return ViewX.of(torch.stack([hi, lo]).contiguous().to(device))


class PackedConvX3(PackedConvBF16):
    """Weights of sgta_conv_x3: the folded fp32 weights split into IEEE-half hi / lo parts and packed over 3 Cin virtual channels
    (per kc-channel chunk [w_hi, w_lo, w_hi]) in PackedConvBF16's LDS-image layout."""

    def __init__(self, weight, bias=None, bn=None, stride=1, cout_pad_to=None, na=None):
        w, b = fold_bn(weight.detach().float(), None if bias is None else bias.detach().float(), bn)
        co, ci, ks, _ = w.shape
        kc = 16 if ks == 3 else 32
        if ci % kc:
            raise _lib.SgtaHipError("x3 conv needs Cin %% %d == 0, got %d" % (kc, ci))
        hi, lo = split16(w)
        hi, lo = hi.float(), lo.float()
        nch = ci // kc
        v = torch.stack([hi.view(co, nch, kc, ks, ks), lo.view(co, nch, kc, ks, ks), hi.view(co, nch, kc, ks, ks)], 2)
        wv = v.reshape(co, 3 * ci, ks, ks)                      # [co][chunk][part][kc] flattened: virtual channel order
        super().__init__(wv, b, None, stride=stride, cout_pad_to=cout_pad_to, na=na, dtype=F16)
        self.Cin_real = ci
        self.w_fold = w                                         # tests: the fp32 weights the split represents

    def flops(self, M):
        return 2.0 * M * self.Cout * self.Cin_real * self.ks * self.ks


def conv(x, pc, out, act=ACT_NONE, res=None, out_mode=OUT_X3_BLOCKED, nsig=0):
    """out = act(conv(x) + bias [+ res]) at fp32 accuracy.  x / res: ViewX; out: ViewX, or a ViewH of an fp32 NHWC tensor
    (out_mode = OUT_F32_NHWC), or a contiguous fp32 NCHW tensor (OUT_F32_NCHW, sigmoid on channels < nsig)."""
    assert isinstance(x, ViewX) and x.C == pc.Cin_real
    Ho, Wo = (x.H - 1) // pc.stride + 1, (x.W - 1) // pc.stride + 1
    cout = pc.Cout
    ylo = 0
    if out_mode == OUT_F32_NCHW:
        assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (x.N, cout, Ho, Wo)
        yptr, ldy = out.data_ptr(), 0
    elif out_mode == OUT_F32_NHWC:
        assert out.base.dtype == torch.float32 and (out.N, out.H, out.W, out.C) == (x.N, Ho, Wo, cout) and not out.blocked
        yptr, ldy = out.ptr, out.ld
    else:
        assert isinstance(out, ViewX) and (out.N, out.H, out.W, out.C) == (x.N, Ho, Wo, cout)
        yptr, ldy, ylo = out.ptr, out.bs, out.lo_delta
    rp, ldr, rlo = None, 0, 0
    if res is not None:
        assert isinstance(res, ViewX) and (res.N, res.H, res.W, res.C) == (x.N, Ho, Wo, cout)
        rp, ldr, rlo = res.ptr, res.bs, res.lo_delta
    meta = None
    if _lib.PROFILE is not None:
        M = x.N * Ho * Wo
        meta = {"kernel": "conv_x3_kernel<%d,%d,%d>" % (pc.ks, pc.stride, pc.na), "flops": pc.flops(M),
                "exec_flops": 3.0 * pc.flops(M), "peak_tflops": 2500.0,
                "bytes": 4.0 * (pc.Cin_real * x.N * x.H * x.W + pc.Cout * pc.Cin_real * pc.ks ** 2) + 4.0 * cout * M}
    npf = pc.tile_npf(x.N, Ho, Wo)
    _lib.call("sgta_conv_x3", x.ptr, x.nbytes, x.bs, x.lo_delta, pc.wt.data_ptr(), pc.wt.numel() * 2, pc.bias.data_ptr(),
              rp, ldr, rlo, yptr, ldy, ylo, x.N, x.H, x.W, pc.Cin_real, cout, pc.ks, pc.stride, pc.kc, pc.na, npf,
              pc.tile_log_tw(x.N, Ho, Wo, npf), act, out_mode, nsig, _stream(), meta=meta)
    return out


class PackedHeadsX3:
    """Weights of sgta_heads_x3: ``pc1`` = PackedConvX3 of the concatenated 3x3 convs (na = 4), the block-diagonal 1x1 weights
    of all heads as hi / lo A fragments of the second product.  heads: [(w3x3, b3x3, w1x1, b1x1), ...] in output order."""

    def __init__(self, heads):
        w1 = torch.cat([h[0].detach().float() for h in heads], 0)
        b1 = torch.cat([h[1].detach().float() for h in heads], 0)
        self.pc1 = PackedConvX3(w1, b1)
        chid = w1.shape[0]
        self.rows = [h[2].shape[0] for h in heads]
        if self.pc1.na != 4 or chid % 128 or len(heads) > 4 or sum(self.rows) > 16:
            raise _lib.SgtaHipError("fused heads need Chid % 128 == 0, <= 4 heads and <= 16 output rows")
        dev = w1.device
        W2 = torch.zeros(32, chid, device=dev)
        b2 = torch.zeros(16, device=dev)
        r0 = c0 = 0
        for (w3, _, w, b) in heads:
            n, ch = w.shape[0], w3.shape[0]
            W2[r0:r0 + n, c0:c0 + ch] = w.detach().float().reshape(n, ch)
            b2[r0:r0 + n] = b.detach().float()
            r0, c0 = r0 + n, c0 + ch
        lane = torch.arange(64, device=dev)
        m, h = lane & 31, lane >> 5
        j = torch.arange(8, device=dev)
        frag = torch.zeros(chid // 128, 4, 2, 64, 8, device=dev)
        for ct in range(chid // 128):
            for i in range(4):
                for t in range(2):
                    ch = 128 * ct + 32 * i + 8 * (2 * t + (j[None, :] >> 2)) + 4 * h[:, None] + (j[None, :] & 3)
                    frag[ct, i, t] = W2[m[:, None].expand(64, 8), ch]
        hi, lo = split16(frag)
        self.w2, self.w2_lo = hi.contiguous(), lo.contiguous()
        self.b2 = b2.contiguous()
        self.Chid = chid


def heads_fused(x, pk, outs, nsig=0):
    assert isinstance(x, ViewX) and x.C == pk.pc1.Cin_real and len(outs) == len(pk.rows)
    for o, n in zip(outs, pk.rows):
        assert o.dtype == torch.float32 and o.is_contiguous() and tuple(o.shape) == (x.N, n, x.H, x.W)
    meta = None
    if _lib.PROFILE is not None:
        M = x.N * x.H * x.W
        fl = pk.pc1.flops(M) + 2.0 * M * pk.Chid * sum(pk.rows)
        meta = {"kernel": "conv_x3_kernel<3,1,4,heads>", "flops": fl, "exec_flops": 3.0 * (pk.pc1.flops(M) + 2.0 * M * pk.Chid * 32), "peak_tflops": 2500.0,
                "bytes": 4.0 * pk.pc1.Cin_real * M + 4.0 * sum(pk.rows) * M + 4.0 * pk.Chid * pk.pc1.Cin_real * 9}
    ptrs = [o.data_ptr() for o in outs] + [None] * (4 - len(outs))
    rows = list(pk.rows) + [0] * (4 - len(outs))
    _lib.call("sgta_heads_x3", x.ptr, x.nbytes, x.bs, x.lo_delta, pk.pc1.wt.data_ptr(), pk.pc1.wt.numel() * 2,
              pk.pc1.bias.data_ptr(), pk.w2.data_ptr(), pk.w2_lo.data_ptr(), pk.b2.data_ptr(), *ptrs, *rows, x.N, x.H, x.W,
              pk.pc1.Cin_real, pk.Chid, HEADS_NPF, pk.pc1.tile_log_tw(x.N, x.H, x.W, HEADS_NPF), nsig, _stream(), meta=meta)
    return outs


class PackedDCNX3:
    """Weights of sgta_dcnv2_x3, made from the fp32 DCN pack (ops.PackedConv, tap_major): the folded fp32 weights split into
    IEEE-half hi / lo parts, [n_ct][Cin/8][5 k-steps][hi, lo][BM = 32*mt][16] with k = 8*(tap & 1) + (channel % 8), tap = 2*step +
    (k >> 3) -- the tenth tap is zero.  One image per work-group width (mt = 2 / 4), built on first use."""

    def __init__(self, pc):
        w = pc._w_folded
        if w.shape[2] != 3 or w.shape[1] % 8:
            raise _lib.SgtaHipError("x3 deformable conv needs a 3x3 kernel and Cin % 8 == 0")
        self.pc, self.Cin, self.Cout = pc, w.shape[1], w.shape[0]
        self._packs = {}

    def packed(self, mt):
        if mt not in self._packs:
            w, co, ci, bm = self.pc._w_folded, self.Cout, self.Cin, 32 * mt
            n_ct = -(-co // bm)
            w10 = torch.zeros(n_ct * bm, ci, 10, device=w.device)
            w10[:co, :, :9] = w.reshape(co, ci, 9)
            parts = [p.view(n_ct, bm, ci // 8, 8, 5, 2).permute(0, 2, 4, 1, 5, 3).reshape(n_ct, ci // 8, 5, bm, 16)
                     for p in split16(w10)]
            bias = torch.zeros(n_ct * bm, device=w.device)
            bias[:co] = self.pc.bias[:co]
            self._packs[mt] = (torch.stack(parts, 3).contiguous(), bias)
        return self._packs[mt]

    def usable(self, x):
        return x.W % 4 == 0 and x.bs % 4 == 0 and x.ptr % 16 == 0 and x.W >= 4


def dcn_v2(x, om, pk, out, act=ACT_NONE, mt=None):
    """Fused modulated deformable conv on fp32 NCHW views at fp32 accuracy, the product on the 16-bit matrix pipe.  ``om``: the
    raw 27-channel offset / mask conv output.  x.W % 4 == 0, 16-byte aligned rows (``pk.usable``)."""
    from . import ops
    x, om, out = ops.as_view(x), ops.as_view(om), ops.as_view(out)
    assert om.C == 27 and x.C == pk.Cin and out.C == pk.Cout and pk.usable(x)
    code, _ = pk.pc.dcn_tile(x.H, x.W, x.B)
    if mt is None:                                  # 128 output channels per work-group wherever the layer has them: every
        mt = 4 if pk.Cout % 128 == 0 else 2         # gathered, blended and split sample then feeds 12 MFMAs instead of 6
    wt, bias = pk.packed(mt)
    meta = None
    if _lib.PROFILE is not None:
        M = x.B * x.H * x.W
        fl = 2.0 * M * pk.Cout * pk.Cin * 9
        meta = {"kernel": "dcn_x3_kernel<%d,%d>" % (mt, (32, 16, 8)[code % 10]), "flops": fl, "exec_flops": 3.0 * fl * 10 / 9,
                "peak_tflops": 2500.0, "bytes": 4.0 * M * (pk.Cin + 27 + pk.Cout) + 4.0 * pk.Cout * pk.Cin * 9}
    _lib.call("sgta_dcnv2_x3", x.ptr, x.bs, om.ptr, om.bs, wt.data_ptr(), wt.numel() * 2, bias.data_ptr(), out.ptr, out.bs,
              x.B, pk.Cin, x.H, x.W, pk.Cout, mt, 5 - code % 10, act, _stream(), meta=meta)
    return out


class PackedStemL0X3(PackedStemL0BF16):
    """A fragments of sgta_stem_level0_x3: PackedStemL0BF16's fragments of the folded fp32 weights, split into hi / lo halves."""

    def __init__(self, w_img, bn_img, w_hm, bn_hm, w_l0, bn_l0):
        super().__init__(w_img, bn_img, w_hm, bn_hm, w_l0, bn_l0, dtype=F16)
        hi, lo = split16(self.frag_stem_f32)
        self.w_stem, self.w_stem_lo = hi.contiguous(), lo.contiguous()
        hi, lo = split16(self.frag_l0_f32)
        self.w_l0, self.w_l0_lo = hi.contiguous(), lo.contiguous()


def stem_level0(img, hm, pk, out):
    """out = level0(relu(bn(conv7x7(img))) + relu(bn(conv7x7(hm)))) in one kernel at fp32 accuracy; img / hm: fp32 NCHW tensors at
    the API boundary; out: a 16-channel ViewX (two half planes) or an fp32 NCHW ops.View."""
    assert img.dtype == torch.float32 and hm.dtype == torch.float32 and img.is_contiguous() and hm.is_contiguous()
    B, _, H, W = img.shape
    if isinstance(out, ViewX):
        assert (out.N, out.H, out.W, out.C) == (B, H, W, 16) and out.base.shape[1] == 1
        yptr, ybs, ylo, f32 = out.ptr, out.bs, out.lo_delta, 0
    else:
        assert (out.B, out.C, out.H, out.W) == (B, 16, H, W)
        yptr, ybs, ylo, f32 = out.ptr, out.bs, 0, 1
    meta = None
    if _lib.PROFILE is not None:
        M = B * H * W
        meta = {"kernel": "stem_l0_x3_kernel", "flops": pk.flops(M), "exec_flops": 3.0 * 2.0 * M * 16 * (14 * 32 + 5 * 32),
                "peak_tflops": 2500.0, "bytes": M * (16.0 + 64.0)}
    _lib.call("sgta_stem_level0_x3", img.data_ptr(), 3 * H * W, hm.data_ptr(), H * W, pk.w_stem.data_ptr(), pk.w_stem_lo.data_ptr(),
              pk.b_stem.data_ptr(), pk.w_l0.data_ptr(), pk.w_l0_lo.data_ptr(), pk.b_l0.data_ptr(), yptr, ybs, ylo, f32, B, H, W,
              _stream(), meta=meta)
    return out


def maxpool2(x, out):
    assert isinstance(x, ViewX) and isinstance(out, ViewX) and out.C == x.C
    _lib.call("sgta_maxpool2_x3", x.ptr, x.bs, x.lo_delta, out.ptr, out.bs, out.lo_delta, x.N, x.C, x.H, x.W, _stream())
    return out


def window_gather(feats, fid, tok, col0=0):
    B, n = fid.shape
    assert isinstance(feats, ViewX) and tok.dtype == torch.float32 and tok.stride(-1) == 1
    _lib.call("sgta_window_gather_x3", feats.ptr, feats.bs, feats.lo_delta, fid.data_ptr(), tok.data_ptr(), B, feats.C, feats.H,
              feats.W, n, tok.stride(0), col0, _stream())
    return tok


def window_scatter(tok, fid, win, feats):
    B, n = fid.shape
    assert isinstance(feats, ViewX) and tok.dtype == torch.float32 and tok.stride(-1) == 1 and tok.shape[-1] == feats.C
    _lib.call("sgta_window_scatter_x3", tok.data_ptr(), tok.stride(0), fid.data_ptr(), win.data_ptr(), feats.ptr, feats.bs,
              feats.lo_delta, B, feats.C, feats.H, feats.W, n, _stream())
    return feats


def from_nchw(x, out):
    """x: ops.View (fp32 NCHW, may be a channel slice) -> out: ViewX of the same shape."""
    assert isinstance(out, ViewX) and (out.N, out.C, out.H, out.W) == (x.B, x.C, x.H, x.W)
    _lib.call("sgta_nchw_to_x3", x.ptr, x.bs, out.ptr, out.bs, out.lo_delta, x.B, x.C, x.H, x.W, _stream())
    return out


def to_nchw(x, out):
    """x: ViewX -> out: ops.View (fp32 NCHW) of the same shape: exact."""
    assert isinstance(x, ViewX) and (x.N, x.C, x.H, x.W) == (out.B, out.C, out.H, out.W)
    _lib.call("sgta_x3_to_nchw", x.ptr, x.bs, x.lo_delta, out.ptr, out.bs, x.N, x.C, x.H, x.W, _stream())
    return out

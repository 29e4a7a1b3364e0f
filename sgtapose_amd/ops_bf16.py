"""Python faces of the bf16 data path (BASELINE configs[2]): bf16 blocked channels-last feature maps ([N, C/16, H, W, 16]) and bf16 weights on
v_mfma_f32_32x32x16_bf16, fp32 accumulation; everything that leaves the bf16 domain (DCN offsets / masks, head outputs,
attention tokens, decode) is fp32.  torch is plumbing only; every function enqueues hand-written HIP kernels.
"""
import torch

from . import _lib
from .ops import ACT_NONE, ACT_RELU, fold_bn, _stream

OUT_BF16_BLOCKED, OUT_F32_NCHW, OUT_F32_NHWC = 0, 1, 2
HEADS_NPF = 2               # 512-position tiles in the fused head kernel
DCN_WIDE_MAX_HW = 60 * 80   # maps up to this size use the +-10-pixel window with patch slots (levels 3..5 at 480x640)
DCN_MARGIN = None           # tests / tools: force the DCN window margin (6 or 10)
DEBUG_OM_SCALE = None       # tools: multiply the DCN sampling offsets (never set in the product)
FORCE_NPF = None            # tools / tests: force the conv tile size (1: 256 positions, 2: 512)
BF16 = torch.bfloat16
F16 = torch.float16
# The 16-bit data path exists in two element formats with the same layouts and kernels: bfloat16 (C ABI ``sgta_*_bf16``) and
# IEEE half (``sgta_*_f16``: 11 significant bits instead of 8 -- an eighth of bf16's rounding error -- and a range of
# +-65504, clamped on store).  Activations carry their format in their torch dtype; packed weights in ``.dtype``.
_SUFFIX = {BF16: "bf16", F16: "f16"}
_PACK_DTYPE = [BF16]


class pack_dtype:
    """Context manager: the element format the Packed* classes below convert weights to (an engine packs inside it)."""

    def __init__(self, dtype):
        assert dtype in _SUFFIX
        self.dtype = dtype

    def __enter__(self):
        _PACK_DTYPE.append(self.dtype)

    def __exit__(self, *exc):
        _PACK_DTYPE.pop()


def _to16(t, dtype):
    return (t.clamp(-65504.0, 65504.0) if dtype == F16 else t).to(dtype)


def _sfx(t):
    try:
        return _SUFFIX[t.dtype]
    except KeyError:
        raise _lib.SgtaHipError("16-bit data path: expected a bfloat16 or float16 tensor, got %s" % t.dtype)


class ViewH:
    """A bfloat16 activation in the BLOCKED channels-last layout [N, C/16, H, W, 16] that may be a range of 16-channel
    blocks of a wider (concatenation) buffer; also wraps the fp32 channels-last tensors of the path ([N, H, W, C], e.g.
    the DCN offset maps), for which only N / H / W / C / ld / ptr are meaningful."""
    __slots__ = ("base", "c0", "C")

    def __init__(self, base, c0=0, C=None):
        if not (base.is_cuda and base.is_contiguous() and base.dim() in (4, 5)):
            raise _lib.SgtaHipError("expected a contiguous [N,C/16,H,W,16] (or fp32 [N,H,W,C]) tensor on the GPU")
        self.base = base
        ctot = base.shape[1] * 16 if base.dim() == 5 else base.shape[3]
        self.c0 = c0
        self.C = ctot - c0 if C is None else C
        assert 0 <= c0 and c0 + self.C <= ctot
        if base.dim() == 5:
            assert base.shape[4] == 16 and c0 % 16 == 0 and self.C % 16 == 0, "views are ranges of 16-channel blocks"

    blocked = property(lambda s: s.base.dim() == 5)
    N = property(lambda s: s.base.shape[0])
    H = property(lambda s: s.base.shape[2] if s.base.dim() == 5 else s.base.shape[1])
    W = property(lambda s: s.base.shape[3] if s.base.dim() == 5 else s.base.shape[2])
    ld = property(lambda s: s.base.shape[3])                                   # fp32 NHWC tensors only: channels per pixel
    bs = property(lambda s: s.base[0].numel())                                 # elements per image of the underlying buffer
    ptr = property(lambda s: s.base.data_ptr() + (s.c0 * s.H * s.W if s.base.dim() == 5 else s.c0) * s.base.element_size())
    nbytes = property(lambda s: (s.base.numel() - (s.c0 * s.H * s.W if s.base.dim() == 5 else s.c0)) * s.base.element_size())

    def slice(self, c0, C):
        return ViewH(self.base, self.c0 + c0, C)

    def images(self, lo, hi):
        """Sub-batch [lo, hi) (a view of the same memory)."""
        return ViewH(self.base[lo:hi], self.c0, self.C)

    def nchw(self):
        """float32 [N, C, H, W] copy (tests / debugging)."""
        b = self.base[:, self.c0 // 16:(self.c0 + self.C) // 16]
        return b.permute(0, 1, 4, 2, 3).reshape(self.N, self.C, self.H, self.W).float()


def blocked_from_nchw(t, device=None, dtype=BF16):
    """float [N, C, H, W] (C % 16 == 0) -> 16-bit blocked [N, C/16, H, W, 16] (tests, boundary conversions)."""
    N, C, H, W = t.shape
    out = _to16(t.reshape(N, C // 16, 16, H, W).permute(0, 1, 3, 4, 2).contiguous(), dtype)
    return out.to(device) if device is not None else out


def empty_blocked(N, C, H, W, device, dtype=BF16):
    return torch.empty((N, C // 16, H, W, 16), dtype=dtype, device=device)


def _swizzle_rows(img, KC):
    """img [..., rows, KC] -> the LDS image of conv_bf16.hip / dcn_bf16.hip: 16-byte pieces (8 k) of row ``r`` stored at
    piece ^ ((r >> log_r) & (P - 1)) (conflict-free ds_read_b128 fragment reads)."""
    P = KC // 8
    log_r = {16: 3, 32: 2, 64: 1}[KC]
    rows = img.shape[-2]
    v = img.reshape(*img.shape[:-1], P, 8)
    r = torch.arange(rows, device=img.device)
    src = (torch.arange(P, device=img.device)[None, :] ^ ((r[:, None] >> log_r) & (P - 1)))       # [rows, P]
    idx = src.view(*([1] * (v.dim() - 3)), rows, P, 1).expand_as(v)
    return torch.gather(v, -2, idx).reshape(img.shape)


class PackedConvBF16:
    """bf16 weights of one conv in the [cout tile][chunk][tap][TC][KC] LDS-image layout of sgta_conv_bf16, BN folded."""

    def __init__(self, weight, bias=None, bn=None, stride=1, cout_pad_to=None, na=None, dtype=None):
        self.dtype = dtype = _PACK_DTYPE[-1] if dtype is None else dtype
        w, b = fold_bn(weight.detach().float(), None if bias is None else bias.detach().float(), bn)
        co, ci, ks, _ = w.shape
        if cout_pad_to is not None and cout_pad_to > co:                 # e.g. the 27-channel offset / mask conv -> 32
            w = torch.cat([w, w.new_zeros(cout_pad_to - co, ci, ks, ks)], 0)
            b = torch.cat([b, b.new_zeros(cout_pad_to - co)], 0)
            co = cout_pad_to
        assert ks in (1, 3) and stride in (1, 2) and not (ks == 1 and stride == 2)
        self.Cout, self.Cin, self.ks, self.stride = co, ci, ks, stride
        self.kc = 16 if ks == 3 else 32
        if ci % self.kc:
            raise _lib.SgtaHipError("bf16 conv needs Cin %% %d == 0, got %d" % (self.kc, ci))
        if na is None:
            na = 1 if co <= 32 else (2 if co <= 64 else 4)
        if stride == 2:
            na = min(na, 2)
        self.na, self.TC = na, 32 * na
        n_ct = -(-co // self.TC)
        taps, nch = ks * ks, ci // self.kc
        wp = w.new_zeros(n_ct * self.TC, ci, taps)
        wp[:co] = w.reshape(co, ci, taps)
        img = wp.view(n_ct, self.TC, nch, self.kc, taps).permute(0, 2, 4, 1, 3).contiguous()    # [ct][chunk][tap][TC][KC]
        self.wt = _to16(_swizzle_rows(img, self.kc), dtype).contiguous()
        bb = b.new_zeros(n_ct * self.TC)
        bb[:co] = b
        self.bias = bb.float().contiguous()
        self.w_ref = _to16(w, dtype)              # tests: the rounded weights, [Cout, Cin, ks, ks]

    def tile_npf(self, N, Ho, Wo):
        """32-position fragments per wave: 2 (512-position tiles) unless the launch would leave the chip's 256 CUs a short
        last round (small maps: the tail of a 1.2-round launch costs 40 %); stride 2 is built for 1."""
        if FORCE_NPF is not None:
            return 1 if self.stride == 2 else FORCE_NPF
        if self.ks == 3 and self.stride == 2:
            return 1
        if self.ks == 3 and self.na == 4:
            # three-deep pipeline (conv_bf16.hip P3): 256-position tiles keep the rings at 80 KB = two work-groups per CU
            return 1
        n_ct = -(-self.Cout // self.TC)

        def rounds(npf):
            ltw = self.tile_log_tw(N, Ho, Wo, npf)
            tw, tr = 1 << ltw, (256 * npf) >> ltw
            blocks = -(-Wo // tw) * -(-(N * (Ho + 1)) // tr) * n_ct
            return -(-blocks // 256) * npf, blocks
        r2, b2 = rounds(2)
        r1, _ = rounds(1)
        return 1 if (r1 < r2 and b2 < 4 * 256) else 2

    def tile_log_tw(self, N, Ho, Wo, npf=None):
        """log2 of the tile width that wastes the fewest positions (ties: the wider tile = longer contiguous DMA rows)."""
        if npf is None:
            npf = 1 if (self.ks == 3 and self.stride == 2) else 2
        tp = 256 * npf
        best = None
        for ltw in (6, 5, 4, 3):
            tw = 1 << ltw
            tr = tp // tw
            work = (-(-Wo // tw) * tw) * (-(-(N * (Ho + 1)) // tr) * tr)
            if best is None or work < best[1]:
                best = (ltw, work)
        return best[0]

    def flops(self, M):
        return 2.0 * M * self.Cout * self.Cin * self.ks * self.ks


def conv(x, pc, out, act=ACT_NONE, res=None, out_mode=OUT_BF16_BLOCKED, nsig=0, cout=None):
    """out = act(conv(x) + bias [+ res]).  x: ViewH (bf16); out: ViewH (bf16 blocked / f32 NHWC) or a contiguous f32 NCHW
    tensor (out_mode = OUT_F32_NCHW, sigmoid on channels < nsig); res: ViewH bf16 at the output resolution."""
    assert x.base.dtype == pc.dtype and x.C == pc.Cin
    Ho, Wo = (x.H - 1) // pc.stride + 1, (x.W - 1) // pc.stride + 1
    cout = pc.Cout if cout is None else cout
    assert x.blocked
    if out_mode == OUT_F32_NCHW:
        assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (x.N, cout, Ho, Wo)
        yptr, ldy = out.data_ptr(), 0
    else:
        assert out.base.dtype == (pc.dtype if out_mode == OUT_BF16_BLOCKED else torch.float32)
        assert (out.N, out.H, out.W, out.C) == (x.N, Ho, Wo, cout)
        yptr, ldy = (out.ptr, out.bs) if out_mode == OUT_BF16_BLOCKED else (out.ptr, out.ld)
        assert out.blocked == (out_mode == OUT_BF16_BLOCKED)
    rp, ldr = None, 0
    if res is not None:
        assert res.base.dtype == pc.dtype and res.blocked and (res.N, res.H, res.W, res.C) == (x.N, Ho, Wo, cout)
        rp, ldr = res.ptr, res.bs
    meta = None
    if _lib.PROFILE is not None:
        M = x.N * Ho * Wo
        meta = {"kernel": "conv_bf16_kernel<%d,%d,%d>" % (pc.ks, pc.stride, pc.na), "flops": pc.flops(M),
                "exec_flops": pc.flops(M),
                "bytes": 2.0 * (pc.Cin * x.N * x.H * x.W + pc.Cout * pc.Cin * pc.ks ** 2) + (2.0 if out_mode == 0 else 4.0) * cout * M}
    npf = pc.tile_npf(x.N, Ho, Wo)
    _lib.call("sgta_conv_" + _SUFFIX[pc.dtype], x.ptr, x.nbytes, x.bs, pc.wt.data_ptr(), pc.wt.numel() * 2, pc.bias.data_ptr(),
              rp, ldr, yptr, ldy, x.N, x.H, x.W, pc.Cin, cout, pc.ks, pc.stride, pc.kc, pc.na, npf,
              pc.tile_log_tw(x.N, Ho, Wo, npf), act, out_mode, nsig, _stream(), meta=meta)
    return out


class PackedHeadsBF16:
    """Weights of sgta_heads_bf16: ``pc1`` = PackedConvBF16 of the concatenated 3x3 convs (na = 4) and the block-diagonal 1x1
    weights of all heads as A fragments of the second product.  heads: [(w3x3, b3x3, w1x1, b1x1), ...] in output order."""

    def __init__(self, heads, dtype=None):
        self.dtype = dtype = _PACK_DTYPE[-1] if dtype is None else dtype
        w1 = torch.cat([h[0].detach().float() for h in heads], 0)
        b1 = torch.cat([h[1].detach().float() for h in heads], 0)
        self.pc1 = PackedConvBF16(w1, b1, dtype=dtype)
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
                    ch = 128 * ct + 32 * i + 8 * (2 * t + (j[None, :] >> 2)) + 4 * h[:, None] + (j[None, :] & 3)     # [64, 8]
                    frag[ct, i, t] = W2[m[:, None].expand(64, 8), ch]
        self.w2 = _to16(frag, dtype).contiguous()
        self.b2 = b2.contiguous()
        self.w2_ref = _to16(W2[:16], dtype)       # tests: [16, Chid] rounded
        self.Chid = chid


def heads_fused(x, pk, outs, nsig=0):
    """outs[i] (f32 NCHW [N, rows_i, H, W]) = head i of the fused head kernel on x (ViewH bf16)."""
    assert x.blocked and x.base.dtype == pk.dtype and x.C == pk.pc1.Cin and len(outs) == len(pk.rows)
    for o, n in zip(outs, pk.rows):
        assert o.dtype == torch.float32 and o.is_contiguous() and tuple(o.shape) == (x.N, n, x.H, x.W)
    meta = None
    if _lib.PROFILE is not None:
        M = x.N * x.H * x.W
        fl = pk.pc1.flops(M) + 2.0 * M * pk.Chid * sum(pk.rows)
        meta = {"kernel": "conv_bf16_kernel<3,1,4,heads>", "flops": fl, "exec_flops": pk.pc1.flops(M) + 2.0 * M * pk.Chid * 32,
                "bytes": 2.0 * pk.pc1.Cin * M + 4.0 * sum(pk.rows) * M + 2.0 * pk.Chid * pk.pc1.Cin * 9}
    npf = HEADS_NPF if FORCE_NPF is None else FORCE_NPF
    ptrs = [o.data_ptr() for o in outs] + [None] * (4 - len(outs))
    rows = list(pk.rows) + [0] * (4 - len(outs))
    _lib.call("sgta_heads_" + _SUFFIX[pk.dtype], x.ptr, x.nbytes, x.bs, pk.pc1.wt.data_ptr(), pk.pc1.wt.numel() * 2, pk.pc1.bias.data_ptr(),
              pk.w2.data_ptr(), pk.b2.data_ptr(), *ptrs, *rows, x.N, x.H, x.W, pk.pc1.Cin, pk.Chid, npf,
              pk.pc1.tile_log_tw(x.N, x.H, x.W, npf), nsig, _stream(), meta=meta)
    return outs


class PackedDCNBF16:
    """fp16 weights of one DCN main product in sgta_dcnv2_fwd_bf16's layout [cout tile][Cin/16][tap][TC][16] (the 3x3
    layout of sgta_conv_bf16), BN folded."""

    def __init__(self, weight, bias=None, bn=None, na=None):
        w, b = fold_bn(weight.detach().float(), None if bias is None else bias.detach().float(), bn)
        co, ci, ks, _ = w.shape
        assert ks == 3
        if ci % 16 or co % 8:
            raise _lib.SgtaHipError("bf16 DCN needs Cin %% 16 == 0 and Cout %% 8 == 0, got %d -> %d" % (ci, co))
        self.Cout, self.Cin = co, ci
        self.na = (2 if co <= 64 else 4) if na is None else na
        self.TC = 32 * self.na
        n_ct = -(-co // self.TC)
        nch = ci // 16
        wp = w.new_zeros(n_ct * self.TC, ci, 9)
        wp[:co] = w.reshape(co, ci, 9)
        img = wp.view(n_ct, self.TC, nch, 16, 9).permute(0, 2, 4, 1, 3).contiguous()      # [ct][chunk][tap][TC][16]
        # fp16 weights (v_mfma_f32_32x32x16_f16: the blended samples are fp16, dcn_bf16.hip), clamped to the fp16 range
        self.wt = _swizzle_rows(img, 16).clamp(-65504.0, 65504.0).to(torch.float16).contiguous()
        bb = b.new_zeros(n_ct * self.TC)
        bb[:co] = b
        self.bias = bb.float().contiguous()
        self.w_ref = w.clamp(-65504.0, 65504.0).to(torch.float16)

    def tile_log_tw(self, N, H, W, margin=6):
        tp = 128                                   # 4 waves x 32 pixels (dcn_bf16.hip); window of <= 765 slots: TW <= 32
        best = None
        for ltw in ((5, 4, 3) if margin == 6 else (4, 3)):
            tw = 1 << ltw
            tr = tp // tw
            work = (-(-W // tw) * tw) * (-(-(N * H) // tr) * tr)
            if best is None or work < best[1]:
                best = (ltw, work)
        return best[0]

    def flops(self, M):
        return 2.0 * M * self.Cout * self.Cin * 9


def dcn_v2(x, om, pc, out, act=ACT_RELU):
    """x: ViewH bf16; om: ViewH float32 (>= 27 channels: the raw offset / mask conv output); out: ViewH bf16."""
    assert x.base.dtype in _SUFFIX and om.base.dtype == torch.float32 and out.base.dtype == x.base.dtype
    assert x.C == pc.Cin and out.C == pc.Cout and om.C >= 27 and x.blocked and out.blocked and not om.blocked
    assert (om.N, om.H, om.W) == (x.N, x.H, x.W) == (out.N, out.H, out.W)
    if DEBUG_OM_SCALE is not None:                 # tools only: scale the sampling offsets (how much of the time is the far-sample path?)
        om.base[..., :18] *= DEBUG_OM_SCALE
    meta = None
    if _lib.PROFILE is not None:
        M = x.N * x.H * x.W
        meta = {"kernel": "dcn_bf16_kernel<%d>" % pc.na, "flops": pc.flops(M), "exec_flops": pc.flops(M),
                "bytes": 2.0 * M * (pc.Cin + pc.Cout) + 4.0 * 27 * M + 2.0 * pc.Cout * pc.Cin * 9}
    # deep levels (small maps) sample far from the output pixel: wide window + patch slots (dcn_bf16.hip); speed only
    margin = DCN_MARGIN if DCN_MARGIN is not None else (10 if x.H * x.W <= DCN_WIDE_MAX_HW else 6)
    _lib.call("sgta_dcnv2_fwd_" + _sfx(x.base), x.ptr, x.bs, om.ptr, om.ld, pc.wt.data_ptr(), pc.wt.numel() * 2,
              pc.bias.data_ptr(), out.ptr, out.bs, x.N, x.H, x.W, pc.Cin, pc.Cout, pc.na,
              pc.tile_log_tw(x.N, x.H, x.W, margin), margin, act, _stream(), meta=meta)
    return out


def stem(img, hm, pc16, out):
    """relu(bn(conv7x7(img))) + relu(bn(conv7x7(hm))) -> out ViewH bf16 (16 channels).  img / hm: f32 NCHW tensors;
    pc16: ops.PackedC16 of the two stems (fp32 weights, f32 MFMA: the inputs are fp32 at the boundary)."""
    assert img.dtype == torch.float32 and hm.dtype == torch.float32 and img.is_contiguous() and hm.is_contiguous()
    B, _, H, W = img.shape
    assert out.base.dtype in _SUFFIX and out.blocked and (out.N, out.H, out.W, out.C) == (B, H, W, 16) and out.base.shape[1] == 1
    meta = None
    if _lib.PROFILE is not None:
        M = B * H * W
        meta = {"kernel": "stem_c16_kernel<bf16>", "flops": pc16.flops(M), "exec_flops": pc16.flops(M) * 200.0 / 196.0,
                "bytes": M * (16.0 + 32.0)}
    _lib.call("sgta_stem_" + _sfx(out.base), img.data_ptr(), 3 * H * W, hm.data_ptr(), H * W, pc16.wt.data_ptr(),
              pc16.bias.data_ptr(), out.ptr, 16, B, H, W, _stream(), meta=meta)
    return out


_ROW_PERM = (0, 2, 1, 3)      # MFMA row m = 4 j + i carries output channel 4 * _ROW_PERM[j] + i (stem_l0_bf16.hip)


class PackedStemL0BF16:
    """bf16 A fragments of sgta_stem_level0_bf16: the two 7x7 stems (image 3 -> 16, prior map 1 -> 16) and level0
    (3x3, 16 -> 16), BatchNorm folded.  ``w_img_ref`` / ``w_hm_ref`` / ``w_l0_ref`` are the rounded weights (tests)."""

    def __init__(self, w_img, bn_img, w_hm, bn_hm, w_l0, bn_l0, dtype=None):
        self.dtype = dtype = _PACK_DTYPE[-1] if dtype is None else dtype
        wi, bi = fold_bn(w_img.detach().float(), None, bn_img)
        wh, bh = fold_bn(w_hm.detach().float(), None, bn_hm)
        wl, bl = fold_bn(w_l0.detach().float(), None, bn_l0)
        assert tuple(wi.shape) == (16, 3, 7, 7) and tuple(wh.shape) == (16, 1, 7, 7) and tuple(wl.shape) == (16, 16, 3, 3)
        dev = wi.device
        lane = torch.arange(64, device=dev)
        m, g = lane & 15, lane >> 4
        cout = 4 * torch.tensor(_ROW_PERM, device=dev)[m >> 2] + (m & 3)                   # [64]
        # stem A fragments [ky][image / prior stem][lane][j]: tap kx = 2 g + (j >> 2), channel j & 3 (3 = the prior map)
        ws = torch.zeros(7, 2, 64, 8, device=dev)
        wi8 = torch.cat([wi, wi.new_zeros(16, 3, 7, 1)], 3)                                # kx 0..7 (the 8th tap is zero)
        wh8 = torch.cat([wh, wh.new_zeros(16, 1, 7, 1)], 3)
        for ky in range(7):
            for j in range(8):
                kx, ch = 2 * g + (j >> 2), j & 3
                if ch < 3:
                    ws[ky, 0, :, j] = wi8[cout, ch, ky, kx]
                else:
                    ws[ky, 1, :, j] = wh8[cout, 0, ky, kx]
        wl9 = torch.cat([wl.reshape(16, 16, 9), wl.new_zeros(16, 16, 1)], 2)              # [co, ci, tap 0..9]
        wf = torch.zeros(5, 64, 8, device=dev)
        for s in range(5):
            t = 2 * s + (g >> 1)
            for j in range(8):
                wf[s, :, j] = wl9[cout, 8 * (g & 1) + j, t]
        self.frag_stem_f32, self.frag_l0_f32 = ws, wf                                      # fp32 fragments (split-operand pack)
        self.w_stem = _to16(ws, dtype).contiguous()
        self.w_l0 = _to16(wf, dtype).contiguous()
        self.b_stem = torch.cat([bi, bh]).float().contiguous()
        self.b_l0 = bl.float().contiguous()
        self.w_img_ref, self.w_hm_ref, self.w_l0_ref = _to16(wi, dtype), _to16(wh, dtype), _to16(wl, dtype)
        self.b_img_ref, self.b_hm_ref = bi, bh

    @staticmethod
    def flops(M):
        return 2.0 * M * 16 * (4 * 49 + 16 * 9)


def stem_level0(img, hm, pk, out):
    """out (ViewH bf16, 16 channels) = level0(relu(bn(conv7x7(img))) + relu(bn(conv7x7(hm)))) in one kernel; img / hm: f32
    NCHW tensors at the API boundary."""
    assert img.dtype == torch.float32 and hm.dtype == torch.float32 and img.is_contiguous() and hm.is_contiguous()
    B, _, H, W = img.shape
    assert out.base.dtype == pk.dtype and out.blocked and (out.N, out.H, out.W, out.C) == (B, H, W, 16) and out.base.shape[1] == 1
    meta = None
    if _lib.PROFILE is not None:
        M = B * H * W
        meta = {"kernel": "stem_l0_bf16_kernel", "flops": pk.flops(M), "exec_flops": 2.0 * M * 16 * (14 * 32 + 5 * 32),
                "bytes": M * (16.0 + 32.0)}
    _lib.call("sgta_stem_level0_" + _SUFFIX[pk.dtype], img.data_ptr(), 3 * H * W, hm.data_ptr(), H * W, pk.w_stem.data_ptr(),
              pk.b_stem.data_ptr(), pk.w_l0.data_ptr(), pk.b_l0.data_ptr(), out.ptr, out.bs, B, H, W, _stream(), meta=meta)
    return out


def maxpool2(x, out):
    assert x.base.dtype in _SUFFIX and out.base.dtype == x.base.dtype and out.C == x.C
    _lib.call("sgta_maxpool2_" + _sfx(x.base), x.ptr, x.bs, out.ptr, out.bs, x.N, x.C, x.H, x.W, _stream())
    return out


def pack_up_weight(w):
    """ConvTranspose2d weight [C,1,k,k] -> f32 [k][k][C] for sgta_upsample_add_bf16."""
    return w.detach().float()[:, 0].permute(1, 2, 0).contiguous()


def upsample_add(x, w_kkc, skip, out, f):
    sp, lds = (skip.ptr, skip.bs) if skip is not None else (None, 0)
    assert out.base.dtype == x.base.dtype and (skip is None or skip.base.dtype == x.base.dtype)
    _lib.call("sgta_upsample_add_" + _sfx(x.base), x.ptr, x.bs, w_kkc.data_ptr(), sp, lds, out.ptr, out.bs, x.N, x.C, x.H, x.W, f,
              _stream())
    return out


def window_gather(feats, fid, tok, col0=0):
    B, n = fid.shape
    assert tok.dtype == torch.float32 and tok.stride(-1) == 1
    _lib.call("sgta_window_gather_" + _sfx(feats.base), feats.ptr, feats.bs, fid.data_ptr(), tok.data_ptr(), B, feats.C, feats.H,
              feats.W, n, tok.stride(0), col0, _stream())
    return tok


def window_scatter(tok, fid, win, feats):
    B, n = fid.shape
    assert tok.dtype == torch.float32 and tok.stride(-1) == 1 and tok.shape[-1] == feats.C
    _lib.call("sgta_window_scatter_" + _sfx(feats.base), tok.data_ptr(), tok.stride(0), fid.data_ptr(), win.data_ptr(), feats.ptr,
              feats.bs, B, feats.C, feats.H, feats.W, n, _stream())
    return feats

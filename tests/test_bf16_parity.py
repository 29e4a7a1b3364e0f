"""GPU (-m gpu): the bf16 data path (BASELINE configs[2]) through the C ABI.

Operator level: each HIP kernel against the same operator evaluated by torch in fp32 ON THE bf16-ROUNDED inputs and
weights -- so the only differences are accumulation order and the final rounding to bf16 (relative 2^-8): asserted as
|got - ref| <= 8e-3 |ref| + 2e-3 max|ref|.  Network level: the bf16 engine against the fp32 oracle with the stated
bf16 tolerance (see test_bf16_network_matches_fp32_oracle).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from util import ARGS, golden_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def bops():
    from sgtapose_amd import _lib, ops_bf16
    _lib.load()
    assert _lib.call("sgta_check_device") == 0
    return ops_bf16


def nhwc(t_nchw):
    """f32 NCHW (CPU) -> bf16 blocked channels-last [N, C/16, H, W, 16] on the device."""
    N, C, H, W = t_nchw.shape
    return t_nchw.reshape(N, C // 16, 16, H, W).permute(0, 1, 3, 4, 2).contiguous().to(BF).to(DEV)


def nchw(t_blk):
    """bf16 blocked [N, C/16, H, W, 16] -> f32 NCHW on the CPU."""
    N, Cb, H, W, _ = t_blk.shape
    return t_blk.float().cpu().permute(0, 1, 4, 2, 3).reshape(N, Cb * 16, H, W).contiguous()


def blocked(N, C, H, W, fill=None):
    t = torch.empty(N, C // 16, H, W, 16, dtype=BF, device=DEV)
    return t.zero_() if fill is None else t.fill_(fill)


def close(got, ref, rtol=8e-3, atol_rel=2e-3):
    tol = rtol * ref.abs() + atol_rel * float(ref.abs().max())
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), "max err %.4g (ref max %.4g), %d bad" % (
        float((got - ref).abs().max()), float(ref.abs().max()), int(bad.sum()))


@pytest.mark.parametrize("N,cin,cout,H,W,ks,stride,act,res", [
    (2, 32, 64, 9, 13, 1, 1, 0, False), (1, 64, 128, 16, 16, 1, 1, 1, False), (3, 448, 128, 6, 10, 1, 1, 1, False),
    (2, 16, 16, 20, 36, 3, 1, 1, False), (1, 64, 64, 17, 23, 3, 1, 1, True), (3, 128, 256, 8, 12, 3, 1, 1, True),
    (2, 16, 32, 16, 24, 3, 2, 1, False), (1, 64, 128, 30, 40, 3, 2, 1, False), (5, 256, 512, 6, 8, 3, 2, 1, False),
    (4, 512, 512, 15, 20, 3, 1, 1, True), (1, 64, 768, 24, 40, 3, 1, 1, False), (2, 16, 16, 64, 96, 3, 1, 1, False),
    (2, 64, 32, 12, 70, 3, 1, 0, False), (1, 1280, 512, 15, 20, 1, 1, 1, False),
])
def test_conv_bf16(bops, N, cin, cout, H, W, ks, stride, act, res):
    g = torch.Generator().manual_seed(N * 1000 + cin + cout + H)
    x = torch.randn(N, cin, H, W, generator=g).to(BF).float()
    w = (torch.randn(cout, cin, ks, ks, generator=g) * (1.5 / (cin * ks * ks) ** 0.5))
    b = torch.randn(cout, generator=g) * 0.2
    pc = bops.PackedConvBF16(w.to(DEV), b.to(DEV), None, stride=stride)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(N, cout, Ho, Wo, generator=g).to(BF).float() if res else None
    ref = F.conv2d(x, pc.w_ref.float().cpu(), b, stride=stride, padding=ks // 2)
    if res:
        ref = ref + r
    if act:
        ref = F.relu(ref)
    out = blocked(N, cout, Ho, Wo, 7.0)
    bops.conv(bops.ViewH(nhwc(x)), pc, bops.ViewH(out), act, res=bops.ViewH(nhwc(r)) if res else None)
    close(nchw(out), ref)


@pytest.mark.parametrize("N,cin,cout,H,W,ks", [(3, 128, 256, 30, 40, 3), (2, 64, 64, 17, 23, 3), (2, 448, 128, 15, 20, 1), (1, 16, 16, 40, 72, 3)])
def test_conv_bf16_half_size_tiles(bops, N, cin, cout, H, W, ks):
    """The 256-position tiles (npf = 1) the dispatcher picks for small maps give the same result as the 512-position ones."""
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(N, cin, H, W, generator=g).to(BF).float()
    w = torch.randn(cout, cin, ks, ks, generator=g) * (1.5 / (cin * ks * ks) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.2
    pc = bops.PackedConvBF16(w.to(DEV), b.to(DEV))
    ref = F.relu(F.conv2d(x, pc.w_ref.float().cpu(), b, padding=ks // 2))
    outs = []
    for npf in (1, 2):
        bops.FORCE_NPF = npf
        try:
            out = blocked(N, cout, H, W, 7.0)
            bops.conv(bops.ViewH(nhwc(x)), pc, bops.ViewH(out), 1)
        finally:
            bops.FORCE_NPF = None
        close(nchw(out), ref)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


def test_conv_bf16_channel_slices_and_fp32_outputs(bops):
    """Input / output / residual as channel slices of wider buffers (the Root concat buffers), fp32 NHWC output (DCN
    offsets) and fp32 NCHW output with sigmoid on the first channels (heads)."""
    g = torch.Generator().manual_seed(5)
    N, H, W = 2, 14, 22
    big = torch.randn(N, 96, H, W, generator=g).to(BF).float()
    w = torch.randn(64, 32, 3, 3, generator=g) * 0.08
    b = torch.randn(64, generator=g) * 0.1
    pc = bops.PackedConvBF16(w.to(DEV), b.to(DEV))
    xin = bops.ViewH(nhwc(big)).slice(32, 32)
    outbuf = blocked(N, 160, H, W)
    bops.conv(xin, pc, bops.ViewH(outbuf).slice(64, 64), 1)
    ref = F.relu(F.conv2d(big[:, 32:64], pc.w_ref.float().cpu(), b, padding=1))
    close(nchw(outbuf[:, 4:8]), ref)
    assert float(outbuf[:, :4].abs().max()) == 0 and float(outbuf[:, 8:].abs().max()) == 0
    # fp32 NHWC (27 -> 32 channel offset conv)
    w27 = torch.randn(27, 32, 3, 3, generator=g) * 0.1
    b27 = torch.randn(27, generator=g) * 0.3
    pco = bops.PackedConvBF16(w27.to(DEV), b27.to(DEV), cout_pad_to=32)
    om = torch.full((N, H, W, 32), 3.0, dtype=torch.float32, device=DEV)
    bops.conv(xin, pco, bops.ViewH(om), 0, out_mode=bops.OUT_F32_NHWC)
    refo = F.conv2d(big[:, 32:64], pco.w_ref.float().cpu()[:27], b27, padding=1)
    got = om.cpu().permute(0, 3, 1, 2)
    assert torch.allclose(got[:, :27], refo, atol=2e-3, rtol=1e-3) and float(got[:, 27:].abs().max()) == 0
    # fp32 NCHW + sigmoid on the first 7 of 11 channels (1x1, 64 -> 11)
    w11 = torch.randn(11, 64, 1, 1, generator=g) * 0.2
    b11 = torch.randn(11, generator=g)
    pch = bops.PackedConvBF16(w11.to(DEV), b11.to(DEV))
    x64 = torch.randn(N, 64, H, W, generator=g).to(BF).float()
    o = torch.empty(N, 11, H, W, device=DEV)
    bops.conv(bops.ViewH(nhwc(x64)), pch, o, 0, out_mode=bops.OUT_F32_NCHW, nsig=7)
    refh = F.conv2d(x64, pch.w_ref.float().cpu(), b11)
    refh[:, :7] = torch.sigmoid(refh[:, :7])
    assert torch.allclose(o.cpu(), refh, atol=2e-3, rtol=1e-3)


@pytest.mark.parametrize("N,cin,cout,H,W,off_scale", [(2, 64, 64, 17, 23, 1.5), (1, 128, 128, 15, 20, 4.0),
                                                      (3, 64, 256, 9, 12, 0.5), (1, 512, 256, 15, 20, 2.0),
                                                      (2, 256, 64, 30, 40, 8.0), (4, 128, 128, 60, 80, 3.5),
                                                      (2, 64, 128, 30, 40, 5.0), (1, 64, 64, 120, 160, 3.0)])
def test_dcn_bf16_matches_oracle(bops, N, cin, cout, H, W, off_scale):
    """sgta_dcnv2_fwd_bf16 against oracle.dcn_v2_im2col + GEMM on the bf16-rounded x / weights with the SAME fp32
    offsets and mask logits (border, far-out-of-image and fractional samples)."""
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(N, cin, H, W, generator=g).to(BF).float()
    w = torch.randn(cout, cin, 3, 3, generator=g) * (1.5 / (cin * 9) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.2
    om = torch.randn(N, 27, H, W, generator=g) * off_scale
    pc = bops.PackedDCNBF16(w.to(DEV), b.to(DEV))
    col = oracle.dcn_v2_im2col(x, om[:, :18], torch.sigmoid(om[:, 18:27]))
    ref = torch.matmul(pc.w_ref.float().cpu().reshape(cout, cin * 9), col.reshape(N, cin * 9, H * W)).view(N, cout, H, W)
    ref = F.relu(ref + b.view(1, -1, 1, 1))
    om32 = torch.zeros(N, H, W, 32, device=DEV)
    om32[..., :27] = om.permute(0, 2, 3, 1).to(DEV)
    # both window margins: +-6 pixels (far samples send the wave through the global-memory path) and +-10 pixels with patch
    # slots (off_scale 0.5 .. 8: no far samples, a few patched ones, more than the patch blocks hold)
    for margin in (6, 10):
        out = blocked(N, cout, H, W, 5.0)
        bops.DCN_MARGIN = margin
        try:
            bops.dcn_v2(bops.ViewH(nhwc(x)), bops.ViewH(om32), pc, bops.ViewH(out), 1)
        finally:
            bops.DCN_MARGIN = None
        # the blended sample is rounded to fp16 before the product (the fp32 oracle keeps it in fp32)
        close(nchw(out), ref, rtol=1e-2, atol_rel=4e-3)


def test_elementwise_bf16(bops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 32, 10, 14, generator=g).to(BF).float()
    out = blocked(3, 64, 5, 7)
    bops.maxpool2(bops.ViewH(nhwc(x)), bops.ViewH(out).slice(16, 32))
    assert torch.equal(nchw(out[:, 1:3]), F.max_pool2d(x, 2))
    assert float(out[:, :1].abs().max()) == 0 and float(out[:, 3:].abs().max()) == 0
    for f in (2, 4):
        w = torch.randn(32, 1, 2 * f, 2 * f, generator=g)
        skip = torch.randn(3, 32, 10 * f, 14 * f, generator=g).to(BF).float()
        ref = F.conv_transpose2d(x, w, stride=f, padding=f // 2, groups=32) + skip
        o = blocked(3, 32, 10 * f, 14 * f)
        bops.upsample_add(bops.ViewH(nhwc(x)), bops.pack_up_weight(w.to(DEV)), bops.ViewH(nhwc(skip)), bops.ViewH(o), f)
        close(nchw(o), ref, rtol=8e-3, atol_rel=2e-3)
    # window gather / ordered scatter
    B, C, H, W, n = 3, 32, 10, 14, 20
    fid = torch.randint(0, H * W, (B, n), generator=g)
    fid[0, 5] = fid[0, 2]                                  # duplicate: the LAST token wins
    win = torch.ones(B, n, dtype=torch.uint8)              # winners = tokens with no later duplicate (sgta_window_index)
    for b in range(B):
        for t_ in range(n):
            win[b, t_] = int(not bool((fid[b, t_ + 1:] == fid[b, t_]).any()))
    assert win[0, 2] == 0
    feats = bops.ViewH(nhwc(x))
    tok = torch.zeros(B * n, 40, device=DEV)
    bops.window_gather(feats, fid.to(DEV), tok, col0=8)
    ref = torch.gather(x.reshape(B, C, -1), 2, fid.view(B, 1, n).expand(B, C, n)).permute(0, 2, 1).reshape(B * n, C)
    assert torch.equal(tok[:, 8:].cpu(), ref) and float(tok[:, :8].abs().max()) == 0
    new = torch.randn(B * n, C, generator=g)
    bops.window_scatter(new.to(DEV), fid.to(DEV), win.to(DEV), feats)
    want = x.reshape(B, C, -1).clone()
    for b in range(B):
        for t in range(n):
            want[b, :, fid[b, t]] = new[b * n + t].to(BF).float()
    assert torch.equal(nchw(feats.base).reshape(B, C, -1), want)


def test_stem_bf16(bops):
    from sgtapose_amd.ops import PackedC16
    g = torch.Generator().manual_seed(8)
    B, H, W = 2, 24, 40
    img = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    hm = torch.rand(B, 1, H, W, generator=g)
    wi, wh = torch.randn(16, 3, 7, 7, generator=g) * 0.1, torch.randn(16, 1, 7, 7, generator=g) * 0.15
    bn = lambda: (torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1,      # noqa: E731
                  torch.randn(16, generator=g) * 0.1, torch.rand(16, generator=g) + 0.5)
    bi, bh = bn(), bn()
    pc = PackedC16(wi.to(DEV), tuple(t.to(DEV) for t in bi), wh.to(DEV), tuple(t.to(DEV) for t in bh))
    out = blocked(B, 16, H, W)
    bops.stem(img.to(DEV), hm.to(DEV), pc, bops.ViewH(out))
    ref = (F.relu(F.batch_norm(F.conv2d(img, wi, padding=3), bi[2], bi[3], bi[0], bi[1], False, 0.0, 1e-5)) +
           F.relu(F.batch_norm(F.conv2d(hm, wh, padding=3), bh[2], bh[3], bh[0], bh[1], False, 0.0, 1e-5)))
    close(nchw(out), ref, rtol=5e-3, atol_rel=1e-3)


@pytest.mark.parametrize("B,H,W", [(2, 24, 40), (1, 48, 136), (3, 16, 64), (1, 37, 68)])
def test_stem_level0_fused_bf16(bops, B, H, W):
    """sgta_stem_level0_bf16 (both 7x7 stems + sum + level0 in one kernel, bf16 MFMA) against fp32 torch on the SAME
    bf16-rounded image / weights with the stem map rounded to bf16 where the kernel rounds it (tiles with borders on
    every side, maps that are not multiples of the 16x64 tile)."""
    g = torch.Generator().manual_seed(H + W)
    img = (torch.rand(B, 3, H, W, generator=g) * 4 - 2)
    hm = torch.rand(B, 1, H, W, generator=g)
    wi, wh = torch.randn(16, 3, 7, 7, generator=g) * 0.1, torch.randn(16, 1, 7, 7, generator=g) * 0.15
    wl = torch.randn(16, 16, 3, 3, generator=g) * 0.12
    bn = lambda: (torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1,      # noqa: E731
                  torch.randn(16, generator=g) * 0.1, torch.rand(16, generator=g) + 0.5)
    bi, bh, bl = bn(), bn(), bn()
    dv = lambda t: tuple(x.to(DEV) for x in t)                                               # noqa: E731
    pk = bops.PackedStemL0BF16(wi.to(DEV), dv(bi), wh.to(DEV), dv(bh), wl.to(DEV), dv(bl))
    out = blocked(B, 16, H, W, 9.0)
    bops.stem_level0(img.to(DEV), hm.to(DEV), pk, bops.ViewH(out))
    r = lambda t: t.to(BF).float()                                                           # noqa: E731
    s = (F.relu(F.conv2d(r(img), pk.w_img_ref.float().cpu(), pk.b_img_ref.cpu(), padding=3)) +
         F.relu(F.conv2d(r(hm), pk.w_hm_ref.float().cpu(), pk.b_hm_ref.cpu(), padding=3)))
    ref = F.relu(F.conv2d(r(s), pk.w_l0_ref.float().cpu(), pk.b_l0.cpu(), padding=1))
    # the stem map is rounded to bf16 in both; a last-bit flip of a stem value (fp32 summation order) moves an output by
    # 2^-9 |s| |w|: tolerance as for the other bf16 convs
    close(nchw(out), ref, rtol=8e-3, atol_rel=3e-3)


@pytest.mark.parametrize("N,H,W,npf", [(2, 14, 22, 2), (1, 30, 40, 1), (3, 8, 64, 2)])
def test_heads_fused_bf16(bops, N, H, W, npf):
    """sgta_heads_bf16 (3x3 -> ReLU -> 1x1 of the three heads, hidden tensor in registers) against fp32 torch on the same
    bf16-rounded weights with the hidden activations rounded to bf16; sigmoid on the hm rows."""
    g = torch.Generator().manual_seed(H)
    x = torch.randn(N, 64, H, W, generator=g).to(BF).float()
    heads, ref = [], []
    for n in (7, 2, 2):
        w3 = torch.randn(256, 64, 3, 3, generator=g) * 0.06
        b3 = torch.randn(256, generator=g) * 0.1
        w1 = torch.randn(n, 256, 1, 1, generator=g) * 0.1
        b1 = torch.randn(n, generator=g)
        heads.append(tuple(t.to(DEV) for t in (w3, b3, w1, b1)))
        hid = F.relu(F.conv2d(x, w3.to(BF).float(), b3, padding=1)).to(BF).float()
        ref.append(F.conv2d(hid, w1.to(BF).float(), b1))
    ref[0] = torch.sigmoid(ref[0])
    pk = bops.PackedHeadsBF16(heads)
    outs = [torch.full((N, n, H, W), 5.0, device=DEV) for n in (7, 2, 2)]
    bops.FORCE_NPF = npf
    try:
        bops.heads_fused(bops.ViewH(nhwc(x)), pk, outs, nsig=7)
    finally:
        bops.FORCE_NPF = None
    for o, r in zip(outs, ref):
        assert torch.allclose(o.cpu(), r, atol=4e-3, rtol=2e-3), float((o.cpu() - r).abs().max())


# ---- network level ---------------------------------------------------------------------------------------------------
BF16_LOGIT_TOL = 6e-2      # max |logit_bf16 - logit_fp32 oracle| asserted for hm / reg / tracking (measured ~2e-2)
BF16_LOGIT_MEAN_TOL = 8e-3


def _bf16_model(peaked=False):
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
    m = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(), default_opt())
    m.load_state_dict(golden_state_dict(0, peaked=peaked), strict=True)
    m = m.to(DEV).eval()
    m.engine_dtype = "bf16"
    return m


@pytest.mark.parametrize("B,H,W,seed", [(2, 128, 160, 11), (1, 480, 640, 321), (3, 96, 224, 13)])
def test_bf16_network_matches_fp32_oracle(B, H, W, seed):
    """The whole bf16 engine against the fp32 CPU oracle: bf16 feature maps / weights (8 significant bits) through ~60
    layers.  Stated tolerance: max |logit error| <= 6e-2 and mean <= 8e-3 on all three heads (fp32 path: 1e-3)."""
    from sgtapose_amd.engine_bf16 import EngineBF16
    from sgtapose_amd.synthetic import make_inputs
    m = _bf16_model()
    assert isinstance(m.engine(), EngineBF16)
    inp = make_inputs(B, H, W, seed=seed)
    out = m(*[inp[k].to(DEV) for k in ARGS])[-1]
    ref = oracle.plan_a_window_forward(golden_state_dict(0), *[inp[k] for k in ARGS])
    for k in ("hm", "reg", "tracking"):
        assert out[k].dtype == torch.float32 and out[k].shape == ref[k].shape
        d = (out[k].cpu() - ref[k]).abs()
        print("bf16 vs fp32 oracle %-8s max %.4f mean %.5f (ref std %.3f)" % (k, float(d.max()), float(d.mean()), float(ref[k].std())))
        assert float(d.max()) <= BF16_LOGIT_TOL and float(d.mean()) <= BF16_LOGIT_MEAN_TOL, (k, float(d.max()), float(d.mean()))


@pytest.mark.parametrize("T,B,H,W", [(4, 2, 128, 160), (3, 1, 96, 224)])
def test_bf16_temporal_window_matches_fp32_oracle(T, B, H, W):
    """The T-frame model (BASELINE configs[4]) on the bf16 data path (EngineSeqBF16) against the fp32 oracle: the same stated
    tolerance as the two-frame model on all four heads."""
    from sgtapose_amd.engine_seq_bf16 import EngineSeqBF16
    from sgtapose_amd.model import create_model, default_head_conv, default_opt
    from sgtapose_amd.synthetic import SEQ_ARGS, fill_state_dict_, make_inputs_seq
    heads = {"hm": 7, "reg": 2, "wh": 2, "tracking": 2}
    m = create_model("dlapawdT_34", heads, default_head_conv(heads), default_opt(seq_len=T))
    sd = fill_state_dict_({k: v.clone() for k, v in m.state_dict().items()}, 3)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    m.engine_dtype = "bf16"
    assert isinstance(m.engine(), EngineSeqBF16)
    inp = make_inputs_seq(B, H, W, T=T, seed=77)
    mv = lambda v: [u.to(DEV) for u in v] if isinstance(v, list) else v.to(DEV)      # noqa: E731
    out = m(*[mv(inp[k]) for k in SEQ_ARGS])[-1]
    ref = oracle.plan_a_window_seq_forward(sd, *[inp[k] for k in SEQ_ARGS])
    assert sorted(out) == ["hm", "reg", "tracking", "wh"]
    for k in out:
        d = (out[k].cpu() - ref[k]).abs()
        print("bf16 T=%d vs fp32 oracle %-8s max %.4f mean %.5f" % (T, k, float(d.max()), float(d.mean())))
        assert out[k].shape == ref[k].shape
        assert float(d.max()) <= BF16_LOGIT_TOL and float(d.mean()) <= BF16_LOGIT_MEAN_TOL, (k, float(d.max()), float(d.mean()))


@pytest.mark.parametrize("shared", [False, True])
def test_bf16_three_frame_model_matches_fp32_oracle(shared):
    """dlapawd3_34 (the reference's DLA_PlanAWindow_Three, separate and shared cross-attention weights) on the bf16 data path
    against the fp32 oracle: the stated bf16 tolerance on all four heads."""
    from sgtapose_amd.engine_three_bf16 import EngineThreeBF16
    from sgtapose_amd.model import create_model, default_head_conv, default_opt
    from sgtapose_amd.synthetic import THREE_ARGS, fill_state_dict_, make_inputs_three
    heads = {"hm": 7, "reg": 2, "wh": 2, "tracking": 2}
    m = create_model("dlapawd3_34", heads, default_head_conv(heads), default_opt(shared_ca=shared))
    sd = fill_state_dict_({k: v.clone() for k, v in m.state_dict().items()}, 4)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    m.engine_dtype = "bf16"
    assert isinstance(m.engine(), EngineThreeBF16)
    inp = make_inputs_three(2, 128, 160, seed=21)
    out = m(*[inp[k].to(DEV) for k in THREE_ARGS])[-1]
    ref = oracle.plan_a_window_three_forward(sd, *[inp[k] for k in THREE_ARGS], shared_ca=shared)
    for k in ("hm", "reg", "wh", "tracking"):
        d = (out[k].cpu() - ref[k]).abs()
        print("bf16 three-frame (shared=%s) %-8s max %.4f mean %.5f" % (shared, k, float(d.max()), float(d.mean())))
        assert float(d.max()) <= BF16_LOGIT_TOL and float(d.mean()) <= BF16_LOGIT_MEAN_TOL, (k, float(d.max()), float(d.mean()))


def test_bf16_step_is_deterministic_across_streams():
    """The bf16 step runs on several streams (attention beside the backbone, dead levels beside the neck, two batch slices per
    backbone level, DCN projections beside the nodes): twelve consecutive steps on the same inputs must give bit-identical
    outputs -- a missing stream dependency shows up here as a flicker."""
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.synthetic import make_inputs
    m = _bf16_model()
    det = BatchedDetector(m)
    inp = make_inputs(6, 128, 160, seed=5)
    dev_in = [inp[k].to(DEV) for k in ARGS]
    first = None
    for i in range(12):
        out, dets = det.process_device(*dev_in, static=True)
        cur = {k: v.clone() for k, v in out.items()}
        cur["kp"] = dets["kp"].clone()
        if first is None:
            first = cur
        else:
            for k in first:
                assert torch.equal(first[k], cur[k]), (i, k)
    torch.cuda.synchronize()


def test_bf16_detector_keypoints_on_peaked_weights():
    """network -> sigmoid -> decode in bf16 with the peaked weight set.  That set multiplies the 'hm' logits by 30, so the
    bf16 logit noise (~3e-2) becomes ~1 logit and the random-init peak structure is NOT stable under it: the test
    reports the heat-map error and asserts only what bf16 can promise there -- saturated reference key-points (score >
    0.9) reappear within one pixel in at least half of the cases, and whatever both paths find at the same integer pixel
    agrees to 0.1 px after the offset head.  (Integer-peak identity is asserted where it is meaningful: on peaked MAPS fed
    to the decode, test_hip_parity.py, and for the fp32 path end to end.)"""
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.synthetic import make_inputs
    m = _bf16_model(peaked=True)
    inp = make_inputs(4, 128, 160, seed=11)
    out, dets = BatchedDetector(m).process_device(*[inp[k].to(DEV) for k in ARGS])
    ref_out, ref_dets = oracle.process_reference(golden_state_dict(0, peaked=True), *[inp[k] for k in ARGS])
    hm_err = float((out["hm"].cpu() - ref_out["hm"]).abs().max())
    clear = ref_dets["scores"] > 0.9
    xs, ys = dets["xs"].cpu().numpy(), dets["ys"].cpu().numpy()
    near = (np.abs(xs - ref_dets["xs"]) <= 1) & (np.abs(ys - ref_dets["ys"]) <= 1) & (dets["scores"].cpu().numpy() > 0)
    same = (xs == ref_dets["xs"]) & (ys == ref_dets["ys"]) & (ref_dets["scores"] > 0) & (dets["scores"].cpu().numpy() > 0)
    print("bf16 peaked: hm max abs %.4f; saturated reference key-points %d, within 1 px %d; identical integer peaks %d" % (
        hm_err, int(clear.sum()), int((near & clear).sum()), int(same.sum())))
    assert int(clear.sum()) >= 2 and int((near & clear).sum()) * 2 >= int(clear.sum())
    if same.any():
        l2 = np.sqrt(((dets["cts_wreg"].cpu().numpy() - ref_dets["cts_wreg"])[same] ** 2).sum(1)).max()
        assert l2 < 0.1, l2


def test_bf16_benched_configuration_matches_fp32_oracle():
    """BASELINE configs[2]'s per-GPU shard AT ITS BENCHED SIZE (B = 32, 480x640): first / middle / last samples of the
    bf16 batch against the fp32 oracle on those samples, with the stated bf16 tolerance."""
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.synthetic import make_inputs
    m = _bf16_model()
    inp = make_inputs(32, 480, 640, seed=317)
    out, dets = BatchedDetector(m).process_device(*[inp[k].to(DEV) for k in ARGS], static=True)
    sd = golden_state_dict(0)
    for i in (0, 15, 31):
        ref = oracle.plan_a_window_forward(sd, *[inp[k][i:i + 1] for k in ARGS])
        d = (torch.logit(out["hm"][i:i + 1].cpu().clamp(1e-7, 1 - 1e-7)) - ref["hm"]).abs()
        assert float(d.max()) <= BF16_LOGIT_TOL and float(d.mean()) <= BF16_LOGIT_MEAN_TOL, (i, float(d.max()), float(d.mean()))
        for k in ("reg", "tracking"):
            d = (out[k][i:i + 1].cpu() - ref[k]).abs()
            assert float(d.max()) <= BF16_LOGIT_TOL and float(d.mean()) <= BF16_LOGIT_MEAN_TOL, (i, k, float(d.max()))
    assert dets["kp"].shape == (32, 7, 3)
    m.engine().bufs.clear()
    torch.cuda.empty_cache()


def test_bench_cli_bf16_and_temporal_window_lines():
    """bench.py --dtype bf16 and --arch dlapawdT_34 --frames 4 (BASELINE configs[2] / configs[4] shapes, small): one JSON
    line each, dtype / unit / roofline fields as the contract asks."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4",
            "--height", "128", "--width", "160", "--no-cpu-baseline"]
    for extra, dtype, unit in ((["--dtype", "bf16", "--secondary", "f16,f32x3"], "bf16", "frame-pairs/s"),
                               (["--arch", "dlapawdT_34", "--frames", "4"], "f32", "frame-windows/s")):
        p = subprocess.run(base + extra, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        r = json.loads(lines[0])
        assert r["dtype"] == dtype and r["unit"] == unit and r["value"] > 0
        assert 0 < r["roofline"]["frac"] <= 1.0
        if dtype == "bf16":                      # further dtypes of the same workload ride in the same JSON line
            assert sorted(r["secondary"]) == ["f16", "f32x3"]
            assert all(v["value"] > 0 and 0 < v["roofline"]["frac"] <= 1.0 for v in r["secondary"].values())

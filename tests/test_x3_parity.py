"""GPU (-m gpu): fp32 emulated on the 16-bit matrix pipe by operand splitting (C ABI ``sgta_*_x3``): every activation is two
IEEE-half planes (value = hi + lo, 22 significant bits), weights are split the same way, each product is three 16-bit MFMA
products accumulated in fp32.  Bar: agreement with torch's fp32 operator on the UNROUNDED fp32 inputs to 2e-5 of the
output's largest magnitude (the fp32 kernels themselves sit at ~1e-6; bf16 at 1e-2, f16 at 1e-3)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def xo():
    from sgtapose_amd import _lib, ops_x3
    _lib.load()
    return ops_x3


def close(got, ref, rel=2e-5):
    err = float((got - ref).abs().max())
    assert err <= rel * float(ref.abs().max()) + 1e-7, "max err %.3e (ref max %.3e)" % (err, float(ref.abs().max()))
    return err


def test_split_is_exact_to_22_bits(xo):
    g = torch.Generator().manual_seed(0)
    t = torch.randn(4, 32, 6, 10, generator=g) * 3
    v = xo.x3_from_nchw(t, DEV)
    back = v.nchw().cpu()
    assert float((back - t).abs().max()) <= 3e-7 * 3 * 4          # 2^-22 relative of values up to ~12


@pytest.mark.parametrize("N,cin,cout,H,W,ks,stride,act,res", [
    (2, 16, 16, 20, 36, 3, 1, 1, False), (1, 64, 64, 17, 23, 3, 1, 1, True), (3, 128, 256, 8, 12, 3, 1, 1, True),
    (2, 32, 64, 16, 24, 3, 2, 1, False), (5, 256, 512, 6, 8, 3, 2, 1, False), (4, 512, 512, 15, 20, 3, 1, 1, True),
    (2, 128, 64, 9, 13, 1, 1, 1, False), (3, 448, 128, 6, 10, 1, 1, 1, False), (1, 1280, 512, 15, 20, 1, 1, 1, False),
    (2, 64, 32, 12, 70, 3, 1, 0, False), (2, 64, 128, 30, 40, 3, 1, 1, False),
])
def test_conv_x3_matches_fp32(xo, N, cin, cout, H, W, ks, stride, act, res):
    g = torch.Generator().manual_seed(N * 1000 + cin + cout + H)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) * (1.5 / (cin * ks * ks) ** 0.5)
    b = torch.randn(cout, generator=g) * 0.2
    pc = xo.PackedConvX3(w.to(DEV), b.to(DEV), None, stride=stride)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(N, cout, Ho, Wo, generator=g) if res else None
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=ks // 2)
    if res:
        ref = ref + r.double()
    if act:
        ref = F.relu(ref)
    out = xo.ViewX.alloc(N, cout, Ho, Wo, DEV)
    out.base.fill_(7.0)
    xo.conv(xo.x3_from_nchw(x, DEV), pc, out, act, res=xo.x3_from_nchw(r, DEV) if res else None)
    close(out.nchw().cpu().double(), ref)


def test_conv_x3_channel_slices_fp32_outputs_and_batch_views(xo):
    from sgtapose_amd.ops_bf16 import ViewH
    g = torch.Generator().manual_seed(5)
    N, H, W = 4, 14, 22
    big = torch.randn(N, 96, H, W, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) * 0.08
    b = torch.randn(64, generator=g) * 0.1
    pc = xo.PackedConvX3(w.to(DEV), b.to(DEV))
    xin = xo.x3_from_nchw(big, DEV)
    outbuf = xo.ViewX.alloc(N, 160, H, W, DEV)
    outbuf.base.zero_()
    outbuf.lo.base.zero_()
    # a channel slice of the input into a channel slice of the output, for the second half of the batch only
    xo.conv(xin.slice(32, 32).images(2, 4), pc, outbuf.slice(64, 64).images(2, 4), 1)
    ref = F.relu(F.conv2d(big[2:4, 32:64].double(), w.double(), b.double(), padding=1))
    got = outbuf.nchw().cpu().double()
    close(got[2:4, 64:128], ref)
    assert float(got[:2].abs().max()) == 0 and float(got[2:, :64].abs().max()) == 0 and float(got[2:, 128:].abs().max()) == 0
    # fp32 NHWC (the 27 -> 32 channel offset conv) and fp32 NCHW with sigmoid
    w27 = torch.randn(27, 32, 3, 3, generator=g) * 0.1
    b27 = torch.randn(27, generator=g) * 0.3
    pco = xo.PackedConvX3(w27.to(DEV), b27.to(DEV), cout_pad_to=32)
    om = torch.full((N, H, W, 32), 3.0, dtype=torch.float32, device=DEV)
    xo.conv(xin.slice(32, 32), pco, ViewH(om), 0, out_mode=xo.OUT_F32_NHWC)
    refo = F.conv2d(big[:, 32:64].double(), w27.double(), b27.double(), padding=1)
    gm = om.cpu().permute(0, 3, 1, 2).double()
    close(gm[:, :27], refo)
    assert float(gm[:, 27:].abs().max()) == 0
    w11 = torch.randn(11, 64, 1, 1, generator=g) * 0.2
    b11 = torch.randn(11, generator=g)
    pch = xo.PackedConvX3(w11.to(DEV), b11.to(DEV))
    o = torch.empty(N, 11, H, W, device=DEV)
    xo.conv(xin.slice(0, 64), pch, o, 0, out_mode=xo.OUT_F32_NCHW, nsig=7)
    refh = F.conv2d(big[:, :64].double(), w11.double(), b11.double())
    refh[:, :7] = torch.sigmoid(refh[:, :7])
    close(o.cpu().double(), refh, rel=5e-5)


@pytest.mark.parametrize("N,H,W", [(2, 14, 22), (1, 30, 40)])
def test_heads_x3_matches_fp32(xo, N, H, W):
    g = torch.Generator().manual_seed(H)
    x = torch.randn(N, 64, H, W, generator=g)
    heads, ref = [], []
    for n in (7, 2, 2):
        w3 = torch.randn(256, 64, 3, 3, generator=g) * 0.06
        b3 = torch.randn(256, generator=g) * 0.1
        w1 = torch.randn(n, 256, 1, 1, generator=g) * 0.1
        b1 = torch.randn(n, generator=g)
        heads.append(tuple(t.to(DEV) for t in (w3, b3, w1, b1)))
        hid = F.relu(F.conv2d(x.double(), w3.double(), b3.double(), padding=1))
        ref.append(F.conv2d(hid, w1.double(), b1.double()))
    ref[0] = torch.sigmoid(ref[0])
    pk = xo.PackedHeadsX3(heads)
    outs = [torch.full((N, n, H, W), 5.0, device=DEV) for n in (7, 2, 2)]
    xo.heads_fused(xo.x3_from_nchw(x, DEV), pk, outs, nsig=7)
    for o, r in zip(outs, ref):
        close(o.cpu().double(), r, rel=5e-5)


@pytest.mark.parametrize("B,cin,cout,H,W,off_scale,mt", [
    (2, 64, 64, 24, 40, 1.5, 2), (2, 128, 64, 12, 20, 3.0, 2), (1, 512, 256, 6, 12, 1.0, 4), (2, 256, 128, 9, 16, 8.0, 4),
    (3, 16, 40, 8, 8, 0.0, 2), (2, 64, 128, 30, 44, 0.5, 4), (1, 8, 100, 33, 36, 4.0, 2), (2, 64, 64, 17, 160, 2.0, 2)])
def test_dcnv2_x3_matches_oracle_and_the_fp32_kernel(xo, B, cin, cout, H, W, off_scale, mt):
    """sgta_dcnv2_x3 (fp32 blend, split-operand product) against the oracle's DCNv2 restatement at the fp32 kernel's own
    tolerance, and against the fp32 kernel to 2e-5 of the output's largest magnitude: partial tiles, channel counts that do not
    fill a work-group, both work-group widths, offsets up to +-8 pixels (in-window and global-fallback samples)."""
    import oracle
    from sgtapose_amd import ops
    g = torch.Generator().manual_seed(cin + cout + H + W)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    om = torch.randn(B, 27, H, W, generator=g) * off_scale
    col = oracle.dcn_v2_im2col(x, om[:, :18], torch.sigmoid(om[:, 18:27]))          # [B,Ci,9,H,W]
    ref = F.relu(torch.matmul(w.reshape(cout, cin * 9).double(), col.reshape(B, cin * 9, H * W).double()).view(B, cout, H, W)
                 + bias.double().view(1, cout, 1, 1))
    pc = ops.PackedConv(w.to(DEV), bias.to(DEV), tap_major=True)
    pk = xo.PackedDCNX3(pc)
    xd, omd = x.to(DEV), om.to(DEV)
    out = torch.full(ref.shape, float("nan"), device=DEV)
    assert pk.usable(ops.as_view(xd))
    xo.dcn_v2(xd, omd, pk, out, ops.ACT_RELU, mt=mt)
    f32 = torch.full(ref.shape, float("nan"), device=DEV)
    ops.dcn_v2(xd, omd, pc, f32, ops.ACT_RELU)
    err = float((out.cpu().double() - ref).abs().max())
    assert err == err and torch.allclose(out.cpu().double(), ref, atol=5e-5, rtol=1e-4), err
    close(out.double(), f32.double())


def test_dcnv2_x3_on_channel_slices_and_batch_views(xo):
    """Input and output as channel slices of wider buffers, no activation, negative outputs kept."""
    from sgtapose_amd import ops
    g = torch.Generator().manual_seed(3)
    B, H, W = 3, 14, 24
    big = torch.randn(B, 48, H, W, generator=g)
    w = torch.randn(24, 32, 3, 3, generator=g) / 17
    bias = torch.randn(24, generator=g) * 0.1
    om = torch.randn(B, 27, H, W, generator=g) * 2.0
    pc = ops.PackedConv(w.to(DEV), bias.to(DEV), tap_major=True)
    pk = xo.PackedDCNX3(pc)
    bigd = big.to(DEV)
    out = torch.zeros(B, 40, H, W, device=DEV)
    xo.dcn_v2(ops.View(bigd, 8, 32), om.to(DEV), pk, ops.View(out, 8, 24))
    f32 = torch.zeros(B, 24, H, W, device=DEV)
    ops.dcn_v2(ops.View(bigd, 8, 32), om.to(DEV), pc, f32)
    close(out[:, 8:32].double(), f32.double())
    assert float(f32.min()) < 0 and float(out[:, :8].abs().sum()) == 0 and float(out[:, 32:].abs().sum()) == 0


@pytest.mark.parametrize("B,H,W", [(2, 32, 128), (1, 19, 68), (3, 8, 64), (1, 41, 200)])
def test_stem_level0_x3_matches_fp64(xo, B, H, W):
    """sgta_stem_level0_x3 (both 7x7 stems with their own ReLUs, their sum, level0: one kernel) against torch in fp64 on the
    unrounded operands: both output forms (two half planes / fp32 NCHW), partial tiles in both directions, folded BatchNorm."""
    from sgtapose_amd import ops
    g = torch.Generator().manual_seed(H * W)
    rnd = lambda *s: torch.randn(*s, generator=g)                                  # noqa: E731
    bn = lambda c: (rnd(c).abs() + 0.5, rnd(c) * 0.2, rnd(c) * 0.1, rnd(c).abs() + 0.5)   # noqa: E731
    w_img, w_hm, w_l0 = rnd(16, 3, 7, 7) / 12, rnd(16, 1, 7, 7) / 7, rnd(16, 16, 3, 3) / 12
    bns = [bn(16) for _ in range(3)]
    img, hm = rnd(B, 3, H, W) * 1.5, rnd(B, 1, H, W).abs().clamp(max=1.0)
    fold = lambda w, p: ops.fold_bn(w.double(), None, tuple(t.double() for t in p))    # noqa: E731
    (wi, bi), (wh, bh), (wl, bl) = fold(w_img, bns[0]), fold(w_hm, bns[1]), fold(w_l0, bns[2])
    stem = F.relu(F.conv2d(img.double(), wi, bi, padding=3)) + F.relu(F.conv2d(hm.double(), wh, bh, padding=3))
    ref = F.relu(F.conv2d(stem, wl, bl, padding=1))
    d = lambda t: tuple(u.to(DEV) for u in t)                                       # noqa: E731
    pk = xo.PackedStemL0X3(w_img.to(DEV), d(bns[0]), w_hm.to(DEV), d(bns[1]), w_l0.to(DEV), d(bns[2]))
    out = xo.ViewX.alloc(B, 16, H, W, DEV)
    out.base.fill_(7.0)
    xo.stem_level0(img.to(DEV), hm.to(DEV), pk, out)
    close(out.nchw().cpu().double(), ref)
    o32 = torch.full((B, 16, H, W), float("nan"), device=DEV)
    xo.stem_level0(img.to(DEV), hm.to(DEV), pk, ops.View(o32))
    close(o32.cpu().double(), ref)
    assert float((o32 - out.nchw()).abs().max()) <= 4e-7 * float(ref.abs().max())     # the planes hold the fp32 result to 22 bits


def test_elementwise_x3(xo):
    from sgtapose_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 32, 10, 14, generator=g)
    xv = xo.x3_from_nchw(x, DEV)
    x22 = xv.nchw().cpu()                                   # the 22-bit values the planes hold
    out = xo.ViewX.alloc(3, 64, 5, 7, DEV)
    out.base.zero_()
    out.lo.base.zero_()
    xo.maxpool2(xv, out.slice(16, 32))
    got = out.nchw().cpu()
    assert torch.equal(got[:, 16:48], F.max_pool2d(x22, 2))
    assert float(got[:, :16].abs().max()) == 0 and float(got[:, 48:].abs().max()) == 0
    B, C, H, W, n = 3, 32, 10, 14, 20
    fid = torch.randint(0, H * W, (B, n), generator=g)
    fid[0, 5] = fid[0, 2]
    win = torch.ones(B, n, dtype=torch.uint8)
    for b in range(B):
        for t_ in range(n):
            win[b, t_] = int(not bool((fid[b, t_ + 1:] == fid[b, t_]).any()))
    tok = torch.zeros(B * n, 40, device=DEV)
    xo.window_gather(xv, fid.to(DEV), tok, col0=8)
    ref = torch.gather(x22.reshape(B, C, -1), 2, fid.view(B, 1, n).expand(B, C, n)).permute(0, 2, 1).reshape(B * n, C)
    assert torch.equal(tok[:, 8:].cpu(), ref) and float(tok[:, :8].abs().max()) == 0
    new = torch.randn(B * n, C, generator=g)
    xo.window_scatter(new.to(DEV), fid.to(DEV), win.to(DEV), xv)
    want = x22.reshape(B, C, -1).clone()
    for b in range(B):
        for t in range(n):
            want[b, :, fid[b, t]] = new[b * n + t]
    assert float((xv.nchw().cpu().reshape(B, C, -1) - want).abs().max()) <= 2e-6
    # fp32 NCHW <-> x3 (a channel slice of a wider fp32 buffer in, exact on the way back)
    wide = torch.randn(3, 48, 10, 14, generator=g).to(DEV)
    xv2 = xo.ViewX.alloc(3, 32, 10, 14, DEV)
    xo.from_nchw(ops.View(wide, 16, 32), xv2)
    assert float((xv2.nchw() - wide[:, 16:48]).abs().max()) <= 2e-6
    back = torch.zeros(3, 48, 10, 14, device=DEV)
    xo.to_nchw(xv2, ops.View(back, 16, 32))
    assert torch.equal(back[:, 16:48], xv2.nchw()) and float(back[:, :16].abs().max()) == 0


# ---- network level: the hybrid engine must pass the fp32 bars unchanged ------------------------------------------------------
def _x3_model(peaked=False):
    from util import golden_state_dict
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
    m = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(), default_opt())
    m.load_state_dict(golden_state_dict(0, peaked=peaked), strict=True)
    m = m.to(DEV).eval()
    m.engine_dtype = "f32x3"
    return m


@pytest.mark.parametrize("B,H,W,seed", [(2, 128, 160, 11), (1, 480, 640, 321), (3, 96, 224, 13), (1, 64, 96, 12), (1, 256, 64, 14),
                                        (5, 224, 288, 16)])
def test_x3_network_matches_fp32_oracle_at_the_fp32_bar(B, H, W, seed):
    import oracle
    from util import ARGS, golden_state_dict
    from sgtapose_amd.engine_x3 import EngineX3
    from sgtapose_amd.synthetic import make_inputs
    m = _x3_model()
    assert isinstance(m.engine(), EngineX3)
    inp = make_inputs(B, H, W, seed=seed, first_frame=(seed == 14))           # seed 14: empty priors (first frame)
    out = m(*[inp[k].to(DEV) for k in ARGS])[-1]
    ref = oracle.plan_a_window_forward(golden_state_dict(0), *[inp[k] for k in ARGS])
    for k in ("hm", "reg", "tracking"):
        d = (out[k].cpu() - ref[k]).abs()
        print("f32x3 vs fp32 oracle %-8s max %.2e mean %.2e" % (k, float(d.max()), float(d.mean())))
        assert float(d.max()) <= 1e-3, (k, float(d.max()))
        assert float(d.max()) <= 5e-5, (k, float(d.max()))       # what the emulation actually delivers


@pytest.mark.parametrize("tag", ["128", "128_first", "96x160", "480", "480x640"])
def test_x3_network_matches_the_references_own_outputs(tag):
    """The committed golden vectors (outputs of the reference itself on seeded inputs) at the bar the fp32 engine is held to:
    heads and the neck's output within 1e-3, the level-0 window ids identical."""
    import numpy as np
    from util import ARGS, GOLDEN
    from sgtapose_amd.synthetic import make_inputs
    import os
    g = np.load(os.path.join(GOLDEN, "fullnet_%s.npz" % tag))
    B, H, W, seed, first = [int(v) for v in g["meta"]]
    inp = make_inputs(B, H, W, seed=seed, first_frame=bool(first))
    m = _x3_model()
    out = m(*[inp[k].to(DEV) for k in ARGS])[-1]
    assert sorted(out) == ["hm", "reg", "tracking"]
    if "feat" in g.files:
        feat = m.engine().last_feat.tensor().cpu()
        assert float((feat - torch.from_numpy(g["feat"])).abs().max()) <= 1e-3
        assert torch.equal(m.engine().last_fid0[0].cpu(), torch.from_numpy(g["pre_fid0"]))
        assert torch.equal(m.engine().last_fid0[1].cpu(), torch.from_numpy(g["cur_fid0"]))
    for k, ref in (("hm", "hm_logits"), ("reg", "reg"), ("tracking", "tracking")):
        err = float((out[k].cpu() - torch.from_numpy(g[ref])).abs().max())
        print("f32x3 vs reference golden %s %-8s max %.2e" % (tag, k, err))
        assert out[k].shape == g[ref].shape and out[k].dtype == torch.float32 and err < 1e-3, (k, err)


def test_x3_temporal_window_and_three_frame_models_match_fp32_oracle_at_the_fp32_bar():
    """The T-frame window (T = 4) and the three-frame model on the split-operand kernels against the fp32 oracle: all four heads
    within the fp32 bar (1e-3), as the fp32 engines are held to."""
    import oracle
    from sgtapose_amd.engine_x3 import EngineSeqX3, EngineThreeX3
    from sgtapose_amd.model import create_model, default_head_conv, default_opt
    from sgtapose_amd.synthetic import SEQ_ARGS, THREE_ARGS, fill_state_dict_, make_inputs_seq, make_inputs_three
    heads = {"hm": 7, "reg": 2, "wh": 2, "tracking": 2}
    m = create_model("dlapawdT_34", heads, default_head_conv(heads), default_opt(seq_len=4))
    sd = fill_state_dict_({k: v.clone() for k, v in m.state_dict().items()}, 3)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    m.engine_dtype = "f32x3"
    assert isinstance(m.engine(), EngineSeqX3)
    inp = make_inputs_seq(2, 128, 160, T=4, seed=77)
    mv = lambda v: [u.to(DEV) for u in v] if isinstance(v, list) else v.to(DEV)      # noqa: E731
    out = m(*[mv(inp[k]) for k in SEQ_ARGS])[-1]
    ref = oracle.plan_a_window_seq_forward(sd, *[inp[k] for k in SEQ_ARGS])
    assert sorted(out) == ["hm", "reg", "tracking", "wh"]
    for k in out:
        d = (out[k].cpu() - ref[k]).abs()
        print("f32x3 T=4 vs fp32 oracle %-8s max %.2e mean %.2e" % (k, float(d.max()), float(d.mean())))
        assert out[k].shape == ref[k].shape and float(d.max()) <= 1e-3, (k, float(d.max()))
    for shared in (True, False):
        m3 = create_model("dlapawd3_34", heads, default_head_conv(heads), default_opt(shared_ca=shared))
        sd3 = fill_state_dict_({k: v.clone() for k, v in m3.state_dict().items()}, 4)
        m3.load_state_dict(sd3, strict=True)
        m3 = m3.to(DEV).eval()
        m3.engine_dtype = "f32x3"
        assert isinstance(m3.engine(), EngineThreeX3)
        inp3 = make_inputs_three(2, 128, 160, seed=21)
        out3 = m3(*[inp3[k].to(DEV) for k in THREE_ARGS])[-1]
        ref3 = oracle.plan_a_window_three_forward(sd3, *[inp3[k] for k in THREE_ARGS], shared_ca=shared)
        for k in ("hm", "reg", "wh", "tracking"):
            d = (out3[k].cpu() - ref3[k]).abs()
            print("f32x3 three-frame (shared_ca=%s) %-8s max %.2e mean %.2e" % (shared, k, float(d.max()), float(d.mean())))
            assert float(d.max()) <= 1e-3, (k, float(d.max()))


def test_x3_engine_names_its_batch_limit():
    """A batch whose level-0 map passes the 2 GiB the kernels can address is refused with the maximum named, before any launch."""
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
    m = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(), default_opt()).to(DEV).eval()
    m.engine_dtype = "f32x3"
    eng = m.engine()
    frames = ((torch.empty(70, 3, 480, 640, device="meta"), None),) * 2
    with pytest.raises(ValueError, match="max 109 frames"):
        eng.backbone_frames(frames)


def test_x3_keypoints_on_peaked_weights_are_the_references():
    """network -> sigmoid -> decode on the hybrid engine with the peaked weight set: integer peaks and the found / missing table
    identical to the fp32 oracle's, sub-pixel positions to 1e-3 px -- the fp32 path's own bar."""
    import numpy as np
    import oracle
    from util import ARGS, golden_state_dict
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.synthetic import make_inputs
    m = _x3_model(peaked=True)
    inp = make_inputs(6, 128, 160, seed=23)
    out, dets = BatchedDetector(m).process_device(*[inp[k].to(DEV) for k in ARGS])
    _, rd = oracle.process_reference(golden_state_dict(0, peaked=True), *[inp[k] for k in ARGS])
    assert int((rd["scores"] >= 0).sum()) >= 3
    assert np.array_equal(dets["xs"].cpu().numpy(), rd["xs"]) and np.array_equal(dets["ys"].cpu().numpy(), rd["ys"])
    f = rd["scores"] >= 0
    assert np.array_equal(dets["scores"].cpu().numpy() >= 0, f)
    assert float(np.abs(dets["cts_wreg"].cpu().numpy() - rd["cts_wreg"])[f].max()) <= 1e-3


def test_x3_step_is_deterministic_across_streams():
    from util import ARGS
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.synthetic import make_inputs
    det = BatchedDetector(_x3_model())
    inp = make_inputs(6, 128, 160, seed=5)
    dev_in = [inp[k].to(DEV) for k in ARGS]
    first = None
    for i in range(8):
        out, dets = det.process_device(*dev_in, static=True)
        cur = {k: v.clone() for k, v in out.items()}
        cur["kp"] = dets["kp"].clone()
        if first is None:
            first = cur
        else:
            for k in first:
                assert torch.equal(first[k], cur[k]), (i, k)
    torch.cuda.synchronize()

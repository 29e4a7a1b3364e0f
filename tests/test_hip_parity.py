"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the golden vectors
generated from the reference.  Integer / index outputs must be bit-exact; floating point within the
north-star tolerance of 1e-3 abs (most kernels are asserted much tighter)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
import oracle
from oracle import temporal as T
from util import ARGS, golden, golden_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def ops():
    from sgtapose_amd import _lib, ops as _ops
    _lib.load()
    assert _lib.call("sgta_check_device") == 0
    return _ops


@pytest.fixture(scope="module")
def model():
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
    m = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(), default_opt())
    m.load_state_dict(golden_state_dict(0), strict=True)
    return m.to(DEV).eval()


@pytest.fixture(scope="module")
def model_peaked():
    """Second weight set (the 'hm' head emits isolated peaks, so the decode finds key-points."""
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
    m = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(), default_opt())
    m.load_state_dict(golden_state_dict(0, peaked=True), strict=True)
    return m.to(DEV).eval()


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,ks,stride,tile", [
    (3, 16, 7, 1, 0), (1, 16, 7, 1, 1), (16, 16, 3, 1, 0), (16, 32, 3, 2, 0), (32, 64, 1, 1, 2),
    (64, 64, 3, 1, 2), (64, 64, 3, 1, 1), (64, 128, 3, 2, 3), (128, 128, 3, 1, 3), (128, 128, 3, 1, 2),
    (448, 128, 1, 1, 0), (64, 27, 3, 1, 0), (64, 768, 3, 1, 0), (256, 7, 1, 1, 0)])
def test_conv2d_bn_relu_residual(ops, cin, cout, ks, stride, tile):
    g = torch.Generator().manual_seed(cin * 131 + cout)
    B, H, W = 3, 24, 40
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1,
          torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    ref = F.batch_norm(F.conv2d(x, w, None, stride, ks // 2), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    res = torch.randn(ref.shape, generator=g)
    pc = ops.PackedConv(w.to(DEV), None, tuple(v.to(DEV) for v in bn), stride=stride)
    out = torch.empty(ref.shape, device=DEV)
    ops.conv2d(x.to(DEV), pc, out, ops.ACT_RELU, res=res.to(DEV), res_mode=1, tile=tile)
    assert torch.allclose(out.cpu(), F.relu(ref + res), atol=2e-5, rtol=1e-5)
    ops.conv2d(x.to(DEV), pc, out, ops.ACT_RELU, res=res.to(DEV), res_mode=2, tile=tile)
    assert torch.allclose(out.cpu(), F.relu(ref) + res, atol=2e-5, rtol=1e-5)
    ops.conv2d(x.to(DEV), pc, out, ops.ACT_SIGMOID, tile=tile)
    assert torch.allclose(out.cpu(), torch.sigmoid(ref), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("cin,cout,H,W", [(64, 64, 24, 40), (128, 27, 15, 20), (16, 96, 17, 23), (64, 768, 8, 12), (512, 512, 15, 20),
                                          (32, 64, 5, 130), (24, 40, 64, 6), (8, 16, 64, 16), (24, 48, 30, 132), (64, 64, 34, 36),
                                          (16, 16, 64, 256), (16, 16, 40, 36), (16, 27, 24, 132)])
def test_conv3x3_implementations_agree(ops, cin, cout, H, W):
    """Winograd F(2x2,3x3), the LDS-range direct kernel and the generic implicit GEMM against torch,
    incl. odd map sizes (partial 2x2 tiles), residual + ReLU, and a non-multiple-of-32 Cout."""
    g = torch.Generator().manual_seed(cin + cout + H)
    B = 2
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, cout, H, W, generator=g)
    ref = F.relu(F.conv2d(x, w, bias, padding=1) + res)
    pc = ops.PackedConv(w.to(DEV), bias.to(DEV))
    assert pc.u_wino is not None and pc.wt_patch is not None
    out = torch.empty(ref.shape, device=DEV)
    flags = (ops.USE_WINOGRAD, ops.USE_PATCH_KERNEL, ops.USE_WINO4, ops.WINO4_MIN_FILL)
    try:
        ops.WINO4_MIN_FILL = 0.0        # F(4x4,3x3) whenever its alignment contract holds (W % 4 == 0)
        # F(4x4,3x3) (or F(2x2,3x3) when W % 4 != 0), F(2x2,3x3), LDS-range direct, generic implicit GEMM
        for wino, w4, patch in ((True, True, True), (True, False, True), (False, False, True), (False, False, False)):
            ops.USE_WINOGRAD, ops.USE_PATCH_KERNEL, ops.USE_WINO4 = wino, patch, w4
            out.zero_()
            ops.conv2d(x.to(DEV), pc, out, ops.ACT_RELU, res=res.to(DEV), res_mode=1)
            err = float((out.cpu() - ref).abs().max())
            assert err < (2e-4 if w4 else (1e-4 if wino else 3e-5)), (wino, w4, patch, err)
    finally:
        ops.USE_WINOGRAD, ops.USE_PATCH_KERNEL, ops.USE_WINO4, ops.WINO4_MIN_FILL = flags


@pytest.mark.parametrize("B,cin,cout,H,W", [(3, 16, 32, 15, 20), (5, 8, 40, 7, 12), (2, 24, 27, 3, 5), (1, 8, 32, 15, 20),
                                            (64, 32, 32, 15, 20), (7, 16, 64, 5, 9)])
def test_winograd_stacks_small_maps(ops, B, cin, cout, H, W):
    """F(2x2,3x3) puts 2 or 4 small images (H <= 2*rows - 1) into one 64-tile block; the zero row between them is shared.
    Odd batch sizes leave empty image slots in the last block; residual, split-K (27 channels, no activation) and
    every tile code must agree with torch."""
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(B, cout, H, W, generator=g)
    ref = F.conv2d(x, w, bias, padding=1)
    pc = ops.PackedConv(w.to(DEV), bias.to(DEV))
    old = ops.USE_WINO4
    ops.USE_WINO4 = False
    try:
        for code in (None, 0, 1, 2, 3):              # automatic, then every block shape
            ops.WINO_TILE_OVERRIDE = code
            out = torch.full(ref.shape, 7.0, device=DEV)
            ops.conv2d(x.to(DEV), pc, out, ops.ACT_RELU, res=res.to(DEV), res_mode=1)
            err = float((out.cpu() - F.relu(ref + res)).abs().max())
            assert err < 1e-4, (code, err)
        ops.WINO_TILE_OVERRIDE = None
        out = torch.full(ref.shape, 7.0, device=DEV)
        ops.conv2d(x.to(DEV), pc, out, ops.ACT_NONE)
        assert float((out.cpu() - ref).abs().max()) < 1e-4
    finally:
        ops.USE_WINO4 = old
        ops.WINO_TILE_OVERRIDE = None


@pytest.mark.parametrize("cin,H,W", [(512, 15, 20), (256, 30, 40), (64, 9, 7)])
def test_winograd_split_k_offset_conv(ops, cin, H, W):
    """The 27-channel DCN offset convs of the small levels are split over slices of the input channels (few work
    items otherwise); the fixed-order reduction must reproduce the plain convolution and be run-to-run identical."""
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(2, cin, H, W, generator=g)
    w = torch.randn(27, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(27, generator=g) * 0.1
    ref = F.conv2d(x, w, bias, padding=1)
    pc = ops.PackedConv(w.to(DEV), bias.to(DEV))
    old = ops.USE_WINO4
    ops.USE_WINO4 = False
    try:
        a = torch.zeros(ref.shape, device=DEV)
        b = torch.zeros(ref.shape, device=DEV)
        ops.conv2d(x.to(DEV), pc, a, ops.ACT_NONE)
        ops.conv2d(x.to(DEV), pc, b, ops.ACT_NONE)
    finally:
        ops.USE_WINO4 = old
    assert torch.equal(a, b)
    assert float((a.cpu() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("cin,cout,H,W", [(16, 32, 48, 64), (64, 128, 24, 40), (256, 512, 30, 40), (32, 64, 17, 23)])
def test_conv3x3_stride2_kernel(ops, cin, cout, H, W):
    g = torch.Generator().manual_seed(cin + H)
    x = torch.randn(2, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, w, bias, stride=2, padding=1))
    pc = ops.PackedConv(w.to(DEV), bias.to(DEV), stride=2)
    out = torch.empty(ref.shape, device=DEV)
    for flag in (True, False):
        old, ops.USE_S2_KERNEL = ops.USE_S2_KERNEL, flag
        try:
            out.zero_()
            ops.conv2d(x.to(DEV), pc, out, ops.ACT_RELU)
        finally:
            ops.USE_S2_KERNEL = old
        assert torch.allclose(out.cpu(), ref, atol=3e-5, rtol=1e-5), flag


@pytest.mark.parametrize("B,cin,cout,H,W", [(2, 16, 32, 48, 128), (3, 8, 64, 34, 70), (1, 16, 40, 16, 66)])
def test_stride2_kernel_with_fused_maxpool(ops, B, cin, cout, H, W):
    """level1's conv also emits MaxPool2d(2,2) of its output (the next Tree's ``bottom``, dla.py:218-223): both outputs
    against torch, incl. odd output sizes (the pool floors) and partial tiles."""
    g = torch.Generator().manual_seed(B + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, w, bias, stride=2, padding=1))
    pc = ops.PackedConv(w.to(DEV), bias.to(DEV), stride=2)
    keep, ops.S2_MIN_FILL = ops.S2_MIN_FILL, 0.0
    try:
        out = torch.full(ref.shape, 7.0, device=DEV)
        pool = torch.full((B, cout, ref.shape[2] // 2, ref.shape[3] // 2), 7.0, device=DEV)
        fused = ops.conv3x3s2_pool(x.to(DEV), pc, out, pool)
    finally:
        ops.S2_MIN_FILL = keep
    if not fused:
        pytest.skip("the 8x32 tile is not the best fit for this map")
    assert torch.allclose(out.cpu(), ref, atol=3e-5, rtol=1e-5)
    assert torch.equal(pool, F.max_pool2d(out, 2, 2))


@pytest.mark.parametrize("B,cin,cout,H,W", [(3, 16, 64, 30, 40), (5, 8, 32, 13, 22), (2, 24, 64, 7, 9), (64, 32, 64, 30, 40),
                                            (1, 8, 64, 30, 40), (7, 12, 40, 15, 31)])
def test_stride2_kernel_stacks_small_maps(ops, B, cin, cout, H, W):
    """The stride-2 kernel puts 2 or 4 small images (H <= 2*rows - 1) into one 256-pixel tile (shared zero row); odd
    batch sizes leave empty image slots in the last tile."""
    g = torch.Generator().manual_seed(B * 10 + H)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(x, w, bias, stride=2, padding=1))
    pc = ops.PackedConv(w.to(DEV), bias.to(DEV), stride=2)
    keep, ops.S2_MIN_FILL = ops.S2_MIN_FILL, 0.0      # the stride-2 kernel whatever the fill
    try:
        assert pc.s2_tile(H, W) != 0
        out = torch.full(ref.shape, 7.0, device=DEV)
        ops.conv2d(x.to(DEV), pc, out, ops.ACT_RELU)
    finally:
        ops.S2_MIN_FILL = keep
    assert torch.allclose(out.cpu(), ref, atol=3e-5, rtol=1e-5), float((out.cpu() - ref).abs().max())


@pytest.mark.parametrize("cin,cout,H,W,act", [(256, 7, 24, 40, 2), (256, 2, 24, 40, 0), (64, 5, 10, 14, 1), (100, 8, 6, 6, 0),
                                              (128, 64, 24, 40, 1), (192, 96, 12, 20, 0), (256, 7, 9, 11, 2)])
def test_conv1x1_gemm_and_skinny_kernels(ops, cin, cout, H, W, act):
    g = torch.Generator().manual_seed(cin * cout + H)
    x = torch.randn(3, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x, w, bias)
    ref = {0: ref, 1: F.relu(ref), 2: torch.sigmoid(ref)}[act]
    pc = ops.PackedConv(w.to(DEV), bias.to(DEV))
    out = torch.empty(ref.shape, device=DEV)
    for flag in (True, False):          # dedicated 1x1 kernels, then the generic implicit GEMM
        old, ops.USE_CONV1X1 = ops.USE_CONV1X1, flag
        try:
            out.zero_()
            ops.conv2d(x.to(DEV), pc, out, act)
        finally:
            ops.USE_CONV1X1 = old
        assert torch.allclose(out.cpu(), ref, atol=3e-5, rtol=1e-5), flag


def test_conv_c16_level0_and_fused_stems(ops):
    g = torch.Generator().manual_seed(21)
    bn = lambda: (torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.1,
                  torch.randn(16, generator=g) * 0.1, torch.rand(16, generator=g) + 0.5)
    bnf = lambda y, p: F.batch_norm(y, p[2], p[3], p[0], p[1], False, 0.0, 1e-5)
    for (B, H, W) in ((2, 16, 64), (3, 20, 45)):
        x = torch.randn(B, 16, H, W, generator=g)
        w = torch.randn(16, 16, 3, 3, generator=g) / 12
        p0 = bn()
        pc = ops.PackedC16(w.to(DEV), tuple(v.to(DEV) for v in p0))
        out = torch.empty(B, 16, H, W, device=DEV)
        ops.conv_c16(x.to(DEV), None, pc, out)
        assert torch.allclose(out.cpu(), F.relu(bnf(F.conv2d(x, w, None, 1, 1), p0)), atol=2e-5, rtol=1e-5)
        img, hm = torch.randn(B, 3, H, W, generator=g), torch.rand(B, 1, H, W, generator=g)
        wi, wh = torch.randn(16, 3, 7, 7, generator=g) / 12, torch.randn(16, 1, 7, 7, generator=g) / 7
        p1, p2 = bn(), bn()
        ps = ops.PackedC16(wi.to(DEV), tuple(v.to(DEV) for v in p1), wh.to(DEV), tuple(v.to(DEV) for v in p2))
        ops.conv_c16(img.to(DEV), hm.to(DEV), ps, out)
        ref = F.relu(bnf(F.conv2d(img, wi, None, 1, 3), p1)) + F.relu(bnf(F.conv2d(hm, wh, None, 1, 3), p2))
        assert torch.allclose(out.cpu(), ref, atol=2e-5, rtol=1e-5)


def test_conv2d_channel_slice_views(ops):
    """input / output / residual as channel slices of wider (concat) buffers, batch > 1."""
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 16, 24
    big_in = torch.randn(B, 96, H, W, generator=g)
    w = torch.randn(32, 64, 3, 3, generator=g) / 24
    pc = ops.PackedConv(w.to(DEV), torch.randn(32, generator=g).to(DEV))
    big_out = torch.zeros(B, 80, H, W, device=DEV)
    vin = ops.View(big_in.to(DEV), 16, 64)
    vout = ops.View(big_out, 40, 32)
    ops.conv2d(vin, pc, vout, ops.ACT_NONE)
    ref = F.conv2d(big_in[:, 16:80], w, pc.bias[:32].cpu(), padding=1)
    assert torch.allclose(big_out[:, 40:72].cpu(), ref, atol=2e-5, rtol=1e-5)
    assert float(big_out[:, :40].abs().sum()) == 0 and float(big_out[:, 72:].abs().sum()) == 0


@pytest.mark.parametrize("cin,cout,H,W,tile,off_scale", [
    (64, 64, 24, 40, 2, 1.5), (64, 64, 24, 40, 1, 1.5), (128, 64, 12, 20, 0, 3.0), (512, 256, 6, 10, 3, 1.0),
    (256, 128, 9, 14, 2, 8.0), (16, 16, 8, 8, 0, 0.0)])
def test_dcnv2_matches_oracle(ops, cin, cout, H, W, tile, off_scale):
    g = torch.Generator().manual_seed(cin + cout + H)
    B = 2
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    omw = torch.randn(27, cin, 3, 3, generator=g) * (off_scale / (cin * 9) ** 0.5)
    omb = torch.randn(27, generator=g) * 0.2
    ref = oracle.dcn_v2_forward(x, w, b, omw, omb)
    pc_om = ops.PackedConv(omw.to(DEV), omb.to(DEV))
    pc = ops.PackedConv(w.to(DEV), b.to(DEV), tap_major=True)
    om = torch.empty(B, 27, H, W, device=DEV)
    out = torch.empty(B, cout, H, W, device=DEV)
    xd = x.to(DEV)
    ops.conv2d(xd, pc_om, om)
    assert torch.allclose(om.cpu(), F.conv2d(x, omw, omb, padding=1), atol=2e-5, rtol=1e-5)
    ops.dcn_v2(xd, om, pc, out, tile=tile)
    assert torch.allclose(out.cpu(), ref, atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize("cin,cout,H,W,off_scale", [(32, 128, 17, 23, 1.0), (16, 256, 9, 40, 6.0), (64, 128, 24, 33, 0.3)])
def test_dcnv2_128_channel_work_groups(ops, cin, cout, H, W, off_scale):
    """The 128-output-channel form of the LDS-window kernel (every gathered sample feeds four MFMAs) against the
    oracle and, bit for bit, against the 64-channel form (same per-channel arithmetic)."""
    g = torch.Generator().manual_seed(cin + cout + W)
    B = 2
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    omw = torch.randn(27, cin, 3, 3, generator=g) * (off_scale / (cin * 9) ** 0.5)
    omb = torch.randn(27, generator=g) * 0.2
    ref = oracle.dcn_v2_forward(x, w, b, omw, omb)
    pc = ops.PackedConv(w.to(DEV), b.to(DEV), tap_major=True)
    om = F.conv2d(x, omw, omb, padding=1).to(DEV)
    outs = []
    keep = (ops.DCN_MT, ops.DCN_MT4_MIN_WGS)
    try:
        for mt in (4, 2):
            ops.DCN_MT, ops.DCN_MT4_MIN_WGS = mt, 0
            assert pc.dcn_tile(H, W, B)[0] // 10 == mt
            out = torch.full((B, cout, H, W), 7.0, device=DEV)
            ops.dcn_v2(x.to(DEV), om, pc, out)
            outs.append(out)
    finally:
        ops.DCN_MT, ops.DCN_MT4_MIN_WGS = keep
    assert torch.equal(outs[0], outs[1])
    assert torch.allclose(outs[0].cpu(), ref, atol=5e-5, rtol=1e-4)


def test_dcn_module_zero_init_is_half_conv(ops):
    """Known answer: zero offset/mask conv -> 0.5 * conv3x3(x) + bias."""
    from sgtapose_amd.dcn_v2 import DCN
    torch.manual_seed(0)
    d = DCN(32, 48, (3, 3), 1, 1).to(DEV)
    x = torch.randn(2, 32, 10, 12, device=DEV)
    with torch.no_grad():
        d.bias.uniform_(-1, 1)
        y = d(x)
        ref = 0.5 * F.conv2d(x.cpu(), d.weight.cpu(), None, padding=1) + d.bias.cpu().view(1, -1, 1, 1)
    assert torch.allclose(y.cpu(), ref, atol=2e-5, rtol=1e-5)


def test_maxpool_and_upsample_add(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 24, 12, 20, generator=g)
    out = torch.empty(2, 24, 6, 10, device=DEV)
    ops.maxpool2(x.to(DEV), out)
    assert torch.equal(out.cpu(), F.max_pool2d(x, 2, 2))
    for f in (2, 4):
        for (h, w_) in ((12, 20), (7, 5), (3, 3), (1, 2)):          # 4-pixel fast path and the generic one
            xs = torch.randn(2, 24, h, w_, generator=g)
            w = torch.rand(24, 1, 2 * f, 2 * f, generator=g)
            skip = torch.randn(2, 24, h * f, w_ * f, generator=g)
            ref = F.conv_transpose2d(xs, w, None, stride=f, padding=f // 2, groups=24) + skip
            o = torch.empty(ref.shape, device=DEV)
            ops.upsample_add(xs.to(DEV), w.to(DEV), skip.to(DEV), o, f)
            assert torch.allclose(o.cpu(), ref, atol=1e-5, rtol=1e-5), (f, h, w_)


# ---------------------------------------------------------------------------------------------
def test_window_index_gather_scatter_match_reference(ops):
    g = golden("window_ops.npz")
    case = cases.window_case()
    idx = ops.prior_argmax(case["pre_cls"].to(DEV), case["cur_cls"].to(DEV))
    assert torch.equal(idx[0].cpu(), t(g["pre_idx"])) and torch.equal(idx[1].cpu(), t(g["cur_idx"]))
    sd = golden_state_dict()
    for i, (pre_f, cur_f, tok) in enumerate(case["levels"]):
        B, C, H, W = cur_f.shape
        fid, win = ops.window_index(idx, float(cases.LEVEL_SCALE[i]), cases.LEVEL_KS[i], H, W)
        assert torch.equal(fid[0].cpu(), t(g["pre_fid%d" % i])), "pre fid level %d" % i
        assert torch.equal(fid[1].cpu(), t(g["cur_fid%d" % i])), "cur fid level %d" % i
        n = fid.shape[2]
        key = torch.empty(B * n, C, device=DEV)
        ops.window_gather(pre_f.to(DEV), fid[0], key)
        assert torch.equal(key.view(B, n, C).cpu(), t(g["key%d" % i]))
        # cat-MLP + ordered scatter, exactly as the engine does it
        cat_tok = torch.empty(B * n, 2 * C, device=DEV)
        cur_d = cur_f.to(DEV).clone()
        cat_tok[:, :C] = tok.to(DEV).view(B * n, C)
        ops.window_gather(cur_d, fid[1], cat_tok, col0=C)
        w0, b0 = sd["cat_layer.%d.0.weight" % i].to(DEV), sd["cat_layer.%d.0.bias" % i].to(DEV)
        w2, b2 = sd["cat_layer.%d.2.weight" % i].to(DEV), sd["cat_layer.%d.2.bias" % i].to(DEV)
        h1 = ops.linear(cat_tok, w0, b0, torch.empty(B * n, 4 * C, device=DEV), relu=True)
        z = ops.linear(h1, w2, b2, torch.empty(B * n, C, device=DEV))
        ops.window_scatter(z, fid[1], win, cur_d)
        got = T.window_gather(cur_d.cpu(), fid[1].cpu())
        assert torch.allclose(got, t(g["sub_tok%d" % i]), atol=1e-4, rtol=1e-4), "substitute level %d" % i
        untouched = torch.ones(B, H * W, dtype=torch.bool).scatter_(1, fid[1].cpu(), False)
        same = (cur_d.cpu() == cur_f).reshape(B, C, -1).all(1)
        assert bool(same[untouched].all())
    # non-square: both axes clamp to H-1
    idn = ops.prior_argmax(case["ns_cls"].to(DEV), case["ns_cls"].to(DEV))
    f1, _ = ops.window_index(idn, 1.0, 3, 24, 40)
    f2, _ = ops.window_index(idn, 0.5, 1, 12, 20)
    assert torch.equal(f1[0].cpu(), t(g["ns_fid"])) and torch.equal(f2[0].cpu(), t(g["ns_fid2"]))


def test_linear_with_relu_residual_layernorm(ops):
    g = torch.Generator().manual_seed(9)
    for (M, N, K, ln) in ((300, 64, 128, True), (77, 16, 1024, True), (1183, 1024, 16, False), (224, 2048, 1024, False),
                          (130, 32, 64, True)):
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g)
        ref = F.linear(x, w, b)
        out = torch.empty(M, N, device=DEV)
        ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), out, relu=True)
        assert torch.allclose(out.cpu(), F.relu(ref), atol=3e-5, rtol=1e-5)
        if ln:
            gm, bt = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
            ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), out, res=r.to(DEV), ln=(gm.to(DEV), bt.to(DEV)))
            assert torch.allclose(out.cpu(), F.layer_norm(ref + r, (N,), gm, bt, 1e-5), atol=5e-5, rtol=1e-5)


@pytest.mark.parametrize("n,d", [(63, 16), (343, 8), (1183, 4), (17, 4), (3, 4), (16, 4), (100, 4)])
def test_attention_tiled_bias_equals_raw_bias(ops, n, d):
    """The structure-prior bias re-packed into 16 x 16 tiles (one 16-byte load per lane and key tile, padding mask
    baked in) gives the attention outputs of the [heads, n, n] form (bit-identical where the kernel is the same)."""
    g = torch.Generator().manual_seed(n)
    B, hid = 2, 8 * d
    q, k, v = (torch.randn(B * n, hid, generator=g).to(DEV) for _ in range(3))
    pos = (torch.randn(8, n, n, generator=g) * 0.5).to(DEV)
    a = ops.mhca(q, k, v, pos, torch.empty(B * n, hid, device=DEV), B, n, 8, d)
    b = ops.mhca(q, k, v, ops.PosTiles(pos), torch.empty(B * n, hid, device=DEV), B, n, 8, d)
    if d == 4:       # head_dim 4 + tiled bias runs the per-lane-state kernel (VALU PV): same maths, other summation order
        assert torch.allclose(a, b, atol=3e-6, rtol=1e-5), float((a - b).abs().max())
    else:
        assert torch.equal(a, b)
    s = (q.view(B, n, 8, d).permute(0, 2, 1, 3) @ k.view(B, n, 8, d).permute(0, 2, 3, 1)) / d ** 0.5 + pos
    ref = (torch.softmax(s, -1) @ v.view(B, n, 8, d).permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B * n, hid)
    assert torch.allclose(a, ref, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("M,C", [(37, 16), (1183, 16), (4100, 16), (343, 32), (8300, 32), (63, 64), (2016, 64), (33000, 16)])
def test_fused_feed_forward_block(ops, M, C):
    """LayerNorm(x + W2 relu(W1 x + b1) + b2) in one kernel (hidden activations in registers, dla.py:721-738) against
    torch and against the two-launch form; strided output view (the last encoder iteration writes into cat_tok)."""
    g = torch.Generator().manual_seed(M + C)
    Fh = 1024
    x = torch.randn(M, C, generator=g)
    w1 = torch.randn(Fh, C, generator=g) / C ** 0.5
    b1 = torch.randn(Fh, generator=g) * 0.1
    w2 = torch.randn(C, Fh, generator=g) / Fh ** 0.5
    b2 = torch.randn(C, generator=g) * 0.1
    gm, bt = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ref = F.layer_norm(x + F.linear(F.relu(F.linear(x, w1, b1)), w2, b2), (C,), gm, bt, 1e-5)
    d = [t_.to(DEV) for t_ in (x, w1, b1, w2, b2, gm, bt)]
    wide = torch.full((M, 2 * C), 7.0, device=DEV)
    out = wide[:, :C]
    assert ops.ffn_ln(d[0], d[1], d[2], d[3], d[4], (d[5], d[6]), out) is not None
    assert torch.allclose(out.cpu(), ref, atol=5e-5, rtol=1e-5), float((out.cpu() - ref).abs().max())
    assert float(wide[:, C:].min()) == 7.0                      # nothing written beside the view
    hdn = torch.empty(M, Fh, device=DEV)
    two = torch.empty(M, C, device=DEV)
    ops.linear(d[0], d[1], d[2], hdn, relu=True)
    ops.linear(hdn, d[3], d[4], two, res=d[0], ln=(d[5], d[6]))
    assert torch.allclose(out, two, atol=5e-5, rtol=1e-5)


def test_linear_split_k_matches_direct_and_is_deterministic(ops):
    """The few-hundred-token products of cat_layer (dla.py:1537-1543) split K over work-groups: same results as the
    un-split kernel within fp32 summation-order noise, bit-identical from run to run, residual / ReLU after the sum."""
    g = torch.Generator().manual_seed(19)
    for (M, N, K) in ((224, 2048, 1024), (224, 512, 2048), (7, 128, 512), (231, 1024, 512), (224, 256, 1024)):
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g)
        r = torch.randn(M, N, generator=g)
        ref = F.relu(F.linear(x, w, b)) + r
        xd, wd, bd, rd = x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
        out = [torch.empty(M, N, device=DEV) for _ in range(3)]
        ops.linear(xd, wd, bd, out[0], relu=True, res=rd)
        ops.linear(xd, wd, bd, out[1], relu=True, res=rd)
        keep, ops.LINEAR_SPLITK_MAX_ROWS = ops.LINEAR_SPLITK_MAX_ROWS, 0
        try:
            ops.linear(xd, wd, bd, out[2], relu=True, res=rd)
        finally:
            ops.LINEAR_SPLITK_MAX_ROWS = keep
        assert torch.equal(out[0], out[1])
        assert torch.allclose(out[0].cpu(), ref, atol=3e-5, rtol=1e-5)
        assert torch.allclose(out[2].cpu(), ref, atol=3e-5, rtol=1e-5)


@pytest.mark.parametrize("i", [2, 1, 0])
def test_attention_and_encoder_match_reference(ops, model, i):
    g = golden("encoder_ops.npz")
    q, k = cases.encoder_case()[i]
    B, n, C = q.shape
    eng = model.engine()
    Tf = eng.tf[i]
    hid = Tf["hid"]
    M = B * n
    qd, kd = q.to(DEV).view(M, C), k.to(DEV).view(M, C)
    kv = ops.linear(kd, Tf["wkv"], None, torch.empty(M, 2 * hid, device=DEV))
    qq = ops.linear(qd, Tf["wq"], None, torch.empty(M, hid, device=DEV))
    att = ops.mhca(qq, kv[:, :hid], kv[:, hid:], Tf["pos"], torch.empty(M, hid, device=DEV), B, n, 8, hid // 8)
    a = ops.linear(att, Tf["fc_w"], Tf["fc_b"], torch.empty(M, C, device=DEV))
    assert torch.allclose(a.view(B, n, C).cpu(), t(g["mhca%d" % i]), atol=1e-4, rtol=1e-4)
    cur = qd
    for it in range(3):
        ops.linear(cur, Tf["wq"], None, qq)
        ops.mhca(qq, kv[:, :hid], kv[:, hid:], Tf["pos"], att, B, n, 8, hid // 8)
        y1 = ops.linear(att, Tf["fc_w"], Tf["fc_b"], torch.empty(M, C, device=DEV), res=cur, ln=Tf["ln1"])
        hdn = ops.linear(y1, Tf["w1"], Tf["b1"], torch.empty(M, 1024, device=DEV), relu=True)
        cur = ops.linear(hdn, Tf["w2"], Tf["b2"], torch.empty(M, C, device=DEV), res=y1, ln=Tf["ln3"])
        if it == 0:
            assert torch.allclose(cur.view(B, n, C).cpu(), t(g["layer%d" % i]), atol=2e-4, rtol=1e-4)
    assert torch.allclose(cur.view(B, n, C).cpu(), t(g["enc%d" % i]), atol=5e-4, rtol=1e-4)


# ---------------------------------------------------------------------------------------------
def test_decode_matches_reference_bit_exact(ops):
    from sgtapose_amd.decode import dream_generic_decode
    g = golden("decode_ops.npz")
    hm, reg, trk = cases.decode_case()
    dets = dream_generic_decode({"hm": hm.to(DEV), "reg": reg.to(DEV), "tracking": trk.to(DEV)}, K=7)
    for b in range(2):
        assert list(dets["npeaks"][b].cpu().numpy()) == list(g["peaks_count_%d" % b])
        for k in ("scores", "clses", "xs", "ys", "cts", "cts_wreg", "regs", "tracking"):
            ref = g["det%d_%s" % (b, k)]
            got = dets[k][b:b + 1].cpu().numpy()
            assert got.shape == ref.shape and got.dtype == ref.dtype, k
            assert np.array_equal(got, ref), k
    sa = ops.soft_argmax(hm.to(DEV))
    assert torch.allclose(sa.cpu(), t(g["softargmax"]), atol=1e-4)      # same bar as the oracle's own check


def test_gaussian_filter_bit_exact_vs_scipy(ops):
    from scipy.ndimage import gaussian_filter
    rng = np.random.RandomState(1)
    for shape in ((1, 2, 120, 160), (2, 1, 15, 9), (1, 1, 30, 7)):
        m = torch.from_numpy(rng.rand(*shape).astype(np.float32))
        ws = (torch.empty(shape, device=DEV), torch.empty(shape, device=DEV))
        ops.peaks_decode(m.to(DEV), ws=ws)
        got = ws[1].cpu().numpy()
        for b in range(shape[0]):
            for k in range(shape[1]):
                assert np.array_equal(got[b, k], gaussian_filter(m[b, k].numpy(), sigma=3))


def test_fused_decode_equals_three_pass(ops):
    """One-kernel Gaussian + local maxima + per-tile candidates (24-row tiles merged by a second kernel) against the
    three-kernel form: identical filtered maps and identical dictionaries, on map heights that are not tile multiples,
    peaks on tile seams, several peaks per map, empty maps and plateaus."""
    rng = np.random.RandomState(5)
    for (B, K, H, W) in ((2, 7, 120, 160), (1, 3, 49, 37), (2, 2, 24, 40), (1, 2, 25, 13), (1, 1, 15, 9)):
        hm = np.zeros((B, K, H, W), np.float32)
        for b in range(B):
            for k in range(K):
                for _ in range(rng.randint(0, 4)):
                    cy, cx = rng.randint(0, H), rng.randint(0, W)
                    if rng.rand() < 0.4:
                        cy = min(H - 1, 24 * rng.randint(0, (H + 23) // 24) + rng.randint(-1, 1) % H)   # tile seams
                    yy, xx = np.mgrid[0:H, 0:W]
                    hm[b, k] = np.maximum(hm[b, k], rng.uniform(0.3, 1.0) *
                                          np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * rng.uniform(1.0, 3.0) ** 2)))
        hm[0, 0, : min(H, 6), : min(W, 6)] = 0.5                      # a plateau
        reg = rng.rand(B, 2, H, W).astype(np.float32)
        trk = rng.randn(B, 2, H, W).astype(np.float32)
        h, r, t_ = (torch.from_numpy(a).to(DEV) for a in (hm, reg, trk))
        ws_a = (torch.empty_like(h), torch.empty_like(h))
        ws_b = (torch.empty_like(h), torch.empty_like(h))
        a = ops.peaks_decode(h, r, t_, ws=ws_a)
        b_ = ops.peaks_decode(h, r, t_, ws=ws_b, three_pass=True)
        assert torch.equal(ws_a[1], ws_b[1])
        for k in b_:
            assert torch.equal(a[k], b_[k]), k
        assert int(a["npeaks"].sum()) > 0


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["128", "128_first", "96x160", "480", "480x640"])
def test_fullnet_matches_reference(model, tag):
    from sgtapose_amd.synthetic import make_inputs
    g = golden("fullnet_%s.npz" % tag)
    B, H, W, seed, first = [int(v) for v in g["meta"]]
    inp = make_inputs(B, H, W, seed=seed, first_frame=bool(first))
    out = model(*[inp[k].to(DEV) for k in ARGS])[-1]
    assert sorted(out) == ["hm", "reg", "tracking"]
    if "feat" in g.files:
        feat = model.engine().last_feat.tensor().cpu()
        assert torch.allclose(feat, t(g["feat"]), atol=1e-3), float((feat - t(g["feat"])).abs().max())
        assert torch.equal(model.engine().last_fid0[0].cpu(), t(g["pre_fid0"]))
        assert torch.equal(model.engine().last_fid0[1].cpu(), t(g["cur_fid0"]))
    for k, ref in (("hm", "hm_logits"), ("reg", "reg"), ("tracking", "tracking")):
        err = float((out[k].cpu() - t(g[ref])).abs().max())
        assert out[k].shape == g[ref].shape and out[k].dtype == torch.float32
        assert err < 1e-3, (k, err)


def test_side_stream_overlap_is_bit_identical(model):
    from sgtapose_amd.synthetic import make_inputs
    inp = make_inputs(2, 128, 160, seed=9)
    dev = [inp[k].to(DEV) for k in ARGS]
    eng = model.engine()
    a = {k: v.clone() for k, v in model(*dev)[-1].items()}
    fid_a = eng.last_fid0.clone()
    l0_a = eng.bufs[("l0", (4, 16, 128, 160), torch.float32)].clone()
    eng.use_side_stream = False
    try:
        b = model(*dev)[-1]
        torch.cuda.synchronize()
        assert torch.equal(fid_a, eng.last_fid0)
        assert torch.equal(l0_a, eng.bufs[("l0", (4, 16, 128, 160), torch.float32)])   # level-0 substitution result
    finally:
        eng.use_side_stream = True
    for k in a:
        assert torch.equal(a[k], b[k])


def test_skipping_dead_levels_changes_nothing(model):
    from sgtapose_amd.synthetic import make_inputs
    inp = make_inputs(2, 128, 128, seed=317)
    dev = [inp[k].to(DEV) for k in ARGS]
    eng = model.engine()
    a = {k: v.clone() for k, v in model(*dev)[-1].items()}
    eng.compute_dead_levels = False
    try:
        b = model(*dev)[-1]
    finally:
        eng.compute_dead_levels = True
    for k in a:
        assert torch.equal(a[k], b[k])


def test_process_contract_and_keypoints_on_peaked_maps(model):
    """SGTADetector.process contract for B>1 + decode parity on peaked maps (random-init logits are flat)."""
    from sgtapose_amd.decode import dream_generic_decode
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.synthetic import make_inputs, peaked_heatmaps
    inp = make_inputs(3, 128, 160, seed=5)
    dev = [inp[k].to(DEV) for k in ARGS]
    det = BatchedDetector(model)
    output, dets = det.process(*dev[:4], pre_hms_cls=dev[4], repro_hms_cls=dev[5])
    sd = golden_state_dict()
    ref_out, ref_dets = oracle.process_reference(sd, *[inp[k] for k in ARGS])
    assert torch.allclose(output["hm"].cpu(), ref_out["hm"], atol=1e-3)
    assert dets["scores"].shape == (3, 7) and dets["cts_wreg"].shape == (3, 7, 2, 1)
    # keypoint parity on peaked maps fed to both decoders
    hm, _ = peaked_heatmaps(3, 32, 40, seed=2)
    o = {"hm": hm.to(DEV), "reg": output["reg"], "tracking": output["tracking"]}
    got = dream_generic_decode(o, K=7)
    ref = oracle.dream_decode({"hm": hm, "reg": output["reg"].cpu(), "tracking": output["tracking"].cpu()})
    assert torch.equal(got["xs"].cpu(), ref["xs"]) and torch.equal(got["ys"].cpu(), ref["ys"])
    assert torch.equal(got["cts_wreg"].cpu(), ref["cts_wreg"])
    assert int((got["scores"] > 0).sum()) == 21


def test_large_nonsquare_input_matches_oracle(model):
    """BASELINE configs[3] geometry (720p padded to 736x1280, feature maps 184x320): exercises the kernels'
    fallbacks for wide maps against the CPU oracle (no reference vector exists at this size)."""
    from sgtapose_amd.synthetic import make_inputs
    inp = make_inputs(1, 736, 1280, seed=41)
    out = model(*[inp[k].to(DEV) for k in ARGS])[-1]
    ref = oracle.plan_a_window_forward(golden_state_dict(), *[inp[k] for k in ARGS])
    for k in ("hm", "reg", "tracking"):
        assert out[k].shape == ref[k].shape
        err = float((out[k].cpu() - ref[k]).abs().max())
        assert err < 1e-3, (k, err)


# ---- the steps either side of the network ---------------------------------------
def _prepost_metas(cs, g):
    metas = []
    for b, (rw, rh) in enumerate(cs["raws"]):
        metas.append(dict(c=cs["c"][b], s=cs["s"][b], width=rw, height=rh, inp_width=cs["inp_w"],
                          inp_height=cs["inp_h"], out_width=cs["out_w"], out_height=cs["out_h"],
                          trans_input=t(g["trans_input"][b]).numpy(), trans_output=t(g["trans_output"][b]).numpy()))
    return metas


def test_prior_rendering_matches_reference_bit_exact(ops):
    """sgta_render_priors against the reference's get_prev_hm_wo_noise(_cls) outputs (golden): integer centres and
    float32 Gaussian values, bit-exact; the host affine matrices equal the fixtures'."""
    from sgtapose_amd import prepost
    g = golden("prepost_ops.npz")
    cs = cases.prepost_case()
    B = len(cs["raws"])
    t_in = np.stack([prepost.get_affine_transform(cs["c"][b], cs["s"][b], 0, [cs["inp_w"], cs["inp_h"]]) for b in range(B)])
    t_out = np.stack([prepost.get_affine_transform(cs["c"][b], cs["s"][b], 0, [cs["out_w"], cs["out_h"]]) for b in range(B)])
    assert np.array_equal(t_in, g["trans_input"]) and np.array_equal(t_out, g["trans_output"])
    wh = np.array(cs["raws"], np.int32)
    for q, kps in enumerate((cs["prev"], cs["nxt"])):
        hm = prepost.get_prev_hm_wo_noise(kps, t_in, cs["inp_w"], cs["inp_h"], wh, cs["has_prior"], device=DEV)
        hc = prepost.get_prev_hm_wo_noise_cls(kps, t_out, cs["out_w"], cs["out_h"], wh, cs["has_prior"], device=DEV)
        assert hm.shape == (B, 1, cs["inp_h"], cs["inp_w"]) and hc.shape == (B, cs["K"], cs["out_h"], cs["out_w"])
        assert np.array_equal(hm[:, 0].cpu().numpy(), g["hm"][:, q]), q
        assert np.array_equal(hc.cpu().numpy(), g["hm_cls"][:, q]), q
    with pytest.raises(ValueError):
        prepost.get_prev_hm_wo_noise(np.full((1, 7, 2), np.nan), t_in[:1], 8, 8, wh[:1], device=DEV)


def test_post_process_and_final_kps_match_reference(ops, model):
    """sgta_post_process against dream_generic_post_process + _get_final_kps of the reference (golden): validity
    and the missing marker exact, fp32 affine coordinates within 1 ulp of a 1e3-pixel value (2.5e-4)."""
    from sgtapose_amd import prepost
    from sgtapose_amd.detector import BatchedDetector
    g = golden("prepost_ops.npz")
    cs = cases.prepost_case()
    dets = {k: torch.from_numpy(v).to(DEV) for k, v in cs["dets"].items()}
    pp = prepost.post_process_device(dets, g["trans_inv"], cs["out_thresh"])
    assert np.array_equal(pp["valid"].cpu().numpy(), g["pp_valid"])
    valid = g["pp_valid"].astype(bool)
    for k in ("ct", "ct_wreg", "reg", "tracking"):
        got = pp[k].cpu().numpy()
        assert np.allclose(got[valid], g["pp_" + k][valid], atol=2.5e-4, rtol=0), k
    fk = pp["final_kps"].cpu().numpy()
    assert fk.dtype == np.float64
    assert np.array_equal(fk[~valid], g["final_kps"][~valid])                 # -999.999 * 4, exactly
    assert np.allclose(fk[valid], g["final_kps"][valid], atol=2.5e-4, rtol=0)
    # detector-level mirror: the reference's list-of-dicts structure and the rendered inputs through the metas
    det = BatchedDetector(model, model.opt)
    old = det.opt.out_thresh
    det.opt.out_thresh = cs["out_thresh"]
    try:
        items, final = det.post_process(dets, _prepost_metas(cs, g))
    finally:
        det.opt.out_thresh = old
    assert [len(x) for x in items] == g["pp_valid"].sum(1).tolist()
    assert items[0][0]["class"] == 1 and set(items[0][0]) >= {"score", "ct", "ct_wreg", "reg", "tracking", "ct_output"}
    assert np.allclose(final[valid], g["final_kps"][valid], atol=2.5e-4, rtol=0)
    maps = det.render_inputs(cs["prev"], cs["nxt"], _prepost_metas(cs, g), has_prior=cs["has_prior"])
    assert np.array_equal(maps[1][:, 0].cpu().numpy(), g["hm"][:, 1])
    assert np.array_equal(maps[2].cpu().numpy(), g["hm_cls"][:, 0])


# ---- seeded random sweeps: every dispatch path of ops.conv2d / ops.dcn_v2 on shapes nobody hand-picked --------------
def test_conv2d_random_shape_sweep(ops):
    """60 seeded random layers (3x3 stride 1/2, 1x1; odd / tiny / wide maps; any channel counts; all epilogues) against
    torch: exercises F(4x4,3x3) in both work-group forms, F(2x2,3x3) incl. split-K, the stride-2, 1x1, skinny and
    generic kernels and their fall-backs through the same dispatcher the engine uses."""
    rng = np.random.RandomState(20231003)
    g = torch.Generator().manual_seed(99)
    worst = 0.0
    for case in range(60):
        ks, stride = [(3, 1), (3, 1), (3, 1), (3, 2), (1, 1)][rng.randint(5)]
        cin = int(rng.choice([3, 4, 8, 12, 16, 24, 32, 40, 64, 96, 128, 256]))
        cout = int(rng.choice([2, 7, 16, 27, 32, 33, 48, 64, 96, 128, 200]))
        H = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 30, 33, 60]))
        W = int(rng.choice([2, 3, 4, 6, 8, 12, 20, 23, 40, 64, 132]))
        if ks == 3 and stride == 2 and (H < 2 or W < 2):
            H, W = max(H, 2), max(W, 2)
        B = int(rng.choice([1, 2, 3]))
        act = int(rng.choice([0, 1, 1, 2]))
        use_res = ks == 3 and stride == 1 and act == 1 and rng.rand() < 0.5
        x = torch.randn(B, cin, H, W, generator=g)
        w = torch.randn(cout, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        ref = F.conv2d(x, w, bias, stride=stride, padding=ks // 2)
        res = torch.randn(ref.shape, generator=g) if use_res else None
        if use_res:
            ref = ref + res
        ref = {0: ref, 1: F.relu(ref), 2: torch.sigmoid(ref)}[act]
        pc = ops.PackedConv(w.to(DEV), bias.to(DEV), stride=stride)
        out = torch.full(ref.shape, float("nan"), device=DEV)
        ops.conv2d(x.to(DEV), pc, out, act, res=None if res is None else res.to(DEV), res_mode=1 if use_res else 0)
        err = float((out.cpu() - ref).abs().max())
        assert err == err and err < 2e-4, (case, ks, stride, cin, cout, H, W, B, act, use_res, err)
        worst = max(worst, err)
    assert worst < 2e-4


def test_dcn_random_shape_sweep(ops):
    """12 seeded random deformable layers (odd maps, offsets up to +-8 pixels: in-window and global-fallback samples)
    against the oracle's DCNv2 restatement."""
    rng = np.random.RandomState(7)
    g = torch.Generator().manual_seed(5)
    for case in range(12):
        cin = int(rng.choice([4, 8, 32, 64, 128]))
        cout = int(rng.choice([8, 64, 100, 128]))
        H, W = int(rng.choice([3, 8, 15, 24, 31])), int(rng.choice([2, 9, 20, 33, 48]))
        scale = float(rng.choice([0.3, 1.0, 4.0]))
        x = torch.randn(2, cin, H, W, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        bias = torch.randn(cout, generator=g) * 0.1
        om = torch.randn(2, 27, H, W, generator=g) * scale
        col = oracle.dcn_v2_im2col(x, om[:, :18], torch.sigmoid(om[:, 18:27]))          # [B,Ci,9,H,W]
        ref = F.relu(torch.matmul(w.reshape(cout, cin * 9), col.reshape(2, cin * 9, H * W)).view(2, cout, H, W)
                     + bias.view(1, cout, 1, 1))
        pc = ops.PackedConv(w.to(DEV), bias.to(DEV), tap_major=True)
        out = torch.full(ref.shape, float("nan"), device=DEV)
        ops.dcn_v2(x.to(DEV), om.to(DEV), pc, out, ops.ACT_RELU)
        err = float((out.cpu() - ref).abs().max())
        assert err == err and err < 2e-4, (case, cin, cout, H, W, scale, err)


@pytest.mark.parametrize("B,H,W,seed", [(1, 64, 96, 11), (2, 160, 128, 12), (3, 96, 224, 13), (1, 256, 64, 14)])
def test_fullnet_random_sizes_match_oracle(model, B, H, W, seed):
    """Whole network at sizes with no golden vector (different Winograd block shapes, split-K, tile fall-backs per
    level) against the CPU oracle, incl. a first-frame (empty prior) sample."""
    from sgtapose_amd.synthetic import make_inputs
    inp = make_inputs(B, H, W, seed=seed, first_frame=(seed == 14))
    out = model(*[inp[k].to(DEV) for k in ARGS])[-1]
    ref = oracle.plan_a_window_forward(golden_state_dict(), *[inp[k] for k in ARGS])
    for k in ("hm", "reg", "tracking"):
        err = float((out[k].cpu() - ref[k]).abs().max())
        assert out[k].shape == ref[k].shape and err < 1e-3, (k, err)


# ---- the three-frame variant ----------------------------------------------------------
@pytest.mark.parametrize("tag", ["128", "96x160", "shared_128"])
def test_three_frame_model_matches_reference(tag):
    """dlapawd3_34 on the HIP path against DLA_PlanAWindow_Three of the reference (golden head logits, incl. 'wh')."""
    from sgtapose_amd.model import create_model, default_head_conv, default_opt
    from sgtapose_amd.synthetic import THREE_ARGS, make_inputs_three
    g = golden("fullnet_three_%s.npz" % tag)
    B, H, W, seed, shared = [int(v) for v in g["meta"]]
    heads = {"hm": 7, "reg": 2, "wh": 2, "tracking": 2}
    m = create_model("dlapawd3_34", heads, default_head_conv(heads), default_opt(shared_ca=bool(shared)))
    m.load_state_dict(golden_state_dict(keys="model_keys_three_shared.json" if shared else "model_keys_three.json"),
                      strict=True)
    m = m.to(DEV).eval()
    inp = make_inputs_three(B, H, W, seed=seed)
    out = m(*[inp[k].to(DEV) for k in THREE_ARGS])[-1]
    assert sorted(out) == ["hm", "reg", "tracking", "wh"]
    for k in out:
        err = float((out[k].cpu() - t(g[k])).abs().max())
        assert out[k].shape == g[k].shape and err < 1e-3, (k, err)


# ---- end to end with real key-points: network -> sigmoid -> decode -> post-process -> final_kps ---------------------
@pytest.mark.parametrize("tag", ["peaked_128x160", "peaked_480x640"])
def test_peaked_weights_keypoints_end_to_end(model_peaked, tag):
    """The REFERENCE network with the peaked weight set produced these decode dictionaries (fixture): integer peaks
    bit-exact, sub-pixel centres / offsets / tracking within 1e-3, then both dictionaries through post-process ->
    final_kps (the key-points the PnP solver consumes)."""
    from oracle import prepost as op
    from sgtapose_amd import prepost
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.synthetic import make_inputs
    g = golden("fullnet_%s.npz" % tag)
    B, H, W, seed, first = [int(v) for v in g["meta"]]
    inp = make_inputs(B, H, W, seed=seed, first_frame=bool(first))
    det = BatchedDetector(model_peaked)
    out, dets = det.process_device(*[inp[k].to(DEV) for k in ARGS])
    found = g["det_scores"] > 0
    assert int(found.sum()) >= 2
    if "hm_logits" in g.files:
        assert float((out["hm"].cpu() - torch.sigmoid(t(g["hm_logits"]))).abs().max()) < 1e-3
    else:
        pix = t(g["probe_pix"])
        got = out["hm"].cpu().reshape(B, 7, -1)[:, :, pix]
        assert float((got - torch.sigmoid(t(g["hm_probe"]))).abs().max()) < 1e-3
    for k in ("xs", "ys", "cts"):
        assert np.array_equal(dets[k].cpu().numpy(), g["det_" + k]), k
    assert np.array_equal(dets["scores"].cpu().numpy() > 0, found)
    for k in ("scores", "cts_wreg", "regs", "tracking"):
        assert np.allclose(dets[k].cpu().numpy(), g["det_" + k], atol=1e-3, rtol=0), k
    kp = det.keypoints(dets).cpu().numpy()
    assert np.allclose(kp[..., :2], g["det_cts_wreg"][..., 0], atol=1e-3) and np.allclose(kp[..., 2], g["det_scores"], atol=1e-3)
    # post-process + final key-points: HIP chain on the HIP dets vs the oracle chain on the REFERENCE's dets
    c, s_ = np.array([W / 2.0, H / 2.0], np.float32), float(max(W, H))
    t_inv = np.stack([op.get_affine_transform(c, s_, 0, (W // 4, H // 4), inv=1)] * B)
    pp = prepost.post_process_device(dets, t_inv.astype(np.float32), 0.01)
    ref_dets = {k[4:]: g[k] for k in g.files if k.startswith("det_")}
    ref_items = op.dream_generic_post_process(ref_dets, t_inv, 0.01)
    fk = pp["final_kps"].cpu().numpy()
    for b in range(B):
        ref_fk = op.get_final_kps(ref_items[b], 7)
        miss = ref_fk[:, 0] < -3000
        assert np.array_equal(fk[b][miss], ref_fk[miss])
        assert np.allclose(fk[b][~miss], ref_fk[~miss], atol=4e-3, rtol=0)      # 1e-3 map pixels x 4 (output stride)
    assert int((fk[..., 0] > -3000).sum()) == int((g["det_scores"] >= 0.01).sum())


@pytest.mark.parametrize("B,H,W,picks", [(32, 480, 640, (0, 15, 31)), (16, 736, 1280, (0, 15))])
def test_benched_configurations_match_oracle(model, model_peaked, B, H, W, picks):
    """BASELINE configs[1] (B=32, 480x640) and configs[3] (B=16, 736x1280) AT THEIR BENCHED BATCH SIZE -- the
    dispatcher takes other kernels there (128-channel DCN work-groups, stacked images, persistent grids) than at the
    B<=2 of the fixtures.  First / middle / last samples of the HIP batch against the oracle run on those samples
    alone (frame pairs are independent), with both weight sets; key-points of the peaked set compared too."""
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.synthetic import make_inputs
    inp = make_inputs(B, H, W, seed=317)
    dev_in = [inp[k].to(DEV) for k in ARGS]
    for m, peaked in ((model, False), (model_peaked, True)):
        det = BatchedDetector(m)
        out, dets = det.process_device(*dev_in, static=True)
        sd = golden_state_dict(0, peaked=peaked)
        for i in picks:
            ref_out, ref_dets = oracle.process_reference(sd, *[inp[k][i:i + 1] for k in ARGS])
            for k in ("hm", "reg", "tracking"):
                err = float((out[k][i:i + 1].cpu() - ref_out[k]).abs().max())
                assert err < 1e-3, (peaked, i, k, err)
            assert np.array_equal(dets["xs"][i].cpu().numpy(), ref_dets["xs"][0]), (peaked, i)
            assert np.array_equal(dets["ys"][i].cpu().numpy(), ref_dets["ys"][0]), (peaked, i)
            assert np.allclose(dets["cts_wreg"][i].cpu().numpy(), ref_dets["cts_wreg"][0], atol=1e-3, rtol=0)
            assert np.allclose(dets["scores"][i].cpu().numpy(), ref_dets["scores"][0], atol=1e-3, rtol=0)
        del det
    m.engine().bufs.clear()
    model_peaked.engine().bufs.clear()
    model.engine().bufs.clear()
    torch.cuda.empty_cache()


def test_dcnv2_hip_matches_torchvision_deform_conv2d(ops):
    """Third-party pin of the HIP DCN where torchvision is installed (not in this image)."""
    tvo = pytest.importorskip("torchvision.ops")
    from sgtapose_amd.ops import PackedConv, View
    g = torch.Generator().manual_seed(12)
    for (B, Ci, Co, H, W, s) in ((2, 16, 64, 17, 23, 1.5), (1, 64, 64, 30, 40, 4.0)):
        x = torch.randn(B, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.2
        b = torch.randn(Co, generator=g)
        om = torch.randn(B, 27, H, W, generator=g) * s
        o1, o2, m = torch.chunk(om, 3, dim=1)
        ref = tvo.deform_conv2d(x, torch.cat([o1, o2], 1), w, b, padding=1, mask=torch.sigmoid(m))
        pc = PackedConv(w.to(DEV), b.to(DEV), None, tap_major=True)
        out = torch.empty(B, Co, H, W, device=DEV)
        ops.dcn_v2(View(x.to(DEV)), View(om.to(DEV)), pc, View(out))
        assert torch.allclose(out.cpu(), ref, atol=2e-4), float((out.cpu() - ref).abs().max())


def test_rccl_all_gather_of_keypoints_world_of_one():
    """The path's only collective over RCCL (backend "nccl") on device tensors: a world of ONE still runs
    all_gather_into_tensor through RCCL (more ranks need more GPUs; the gloo tests cover world 2)."""
    import os
    import socket
    import torch.distributed as dist
    from sgtapose_amd.distributed import gather_keypoints
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        kp = torch.randn(5, 7, 3, device=DEV)
        got = gather_keypoints(kp, 5)
        torch.cuda.synchronize()
        assert got.data_ptr() != kp.data_ptr() and torch.equal(got, kp)
        assert dist.get_backend() == "nccl"
    finally:
        dist.destroy_process_group()


def test_bench_cli_emits_one_json_line_with_roofline_and_parity():
    """bench.py end to end on a small workload: one JSON line, roofline.frac <= 1, parity of the benched batch with
    real key-points from the peaked weight set."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--batch", "8", "--height", "128", "--width", "160"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["dtype"] == "f32"
    assert 0 < r["roofline"]["frac"] <= 1.0 and r["roofline"]["bound"] in ("mfma", "hbm")
    par = r["cpu_baseline"]["parity"]
    assert par["hm_max_abs"] < 1e-3 and par["reg_max_abs"] < 1e-3 and par["tracking_max_abs"] < 1e-3
    pk = par["peaked_weights"]
    assert pk["integer_peaks_identical"] and pk["keypoint_l2_max_px_on_shared_peaks"] < 1e-3
    assert pk["keypoints_same_integer_peak"] == pk["keypoints_found_ref"]
    assert par["within_bar"] and r["parity_ok"] is True and r["sustained_value"] > 0
    assert r["cpu_baseline"]["cores"] >= 1 and "b8" in r["cpu_baseline"]


# ---- T-frame temporal window (BASELINE configs[4]: seq_len 4) ---------------------------------------------------------
def _seq_model(T, legacy_slip=False, sd=None, seed=3):
    from sgtapose_amd.model import create_model, default_head_conv, default_opt
    from sgtapose_amd.synthetic import fill_state_dict_
    heads = {"hm": 7, "reg": 2, "wh": 2, "tracking": 2}
    m = create_model("dlapawdT_34", heads, default_head_conv(heads), default_opt(seq_len=T, legacy_slip=legacy_slip))
    if sd is None:
        sd = fill_state_dict_({k: v.clone() for k, v in m.state_dict().items()}, seed)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval(), sd


@pytest.mark.parametrize("tag", ["128", "96x160"])
def test_seq_model_at_three_frames_matches_reference(tag):
    """dlapawdT_34 with seq_len 3 + legacy_slip = the reference's DLA_PlanAWindow_Three (golden head logits)."""
    from sgtapose_amd.synthetic import SEQ_ARGS, make_inputs_seq
    g = golden("fullnet_three_%s.npz" % tag)
    B, H, W, seed, shared = [int(v) for v in g["meta"]]
    m, _ = _seq_model(3, True, oracle.seq_keys_from_three(golden_state_dict(keys="model_keys_three.json")))
    inp = make_inputs_seq(B, H, W, T=3, seed=seed)
    mv = lambda v: [u.to(DEV) for u in v] if isinstance(v, list) else v.to(DEV)      # noqa: E731
    out = m(*[mv(inp[k]) for k in SEQ_ARGS])[-1]
    for k in ("hm", "reg", "wh", "tracking"):
        err = float((out[k].cpu() - t(g[k])).abs().max())
        assert err < 1e-3, (k, err)


@pytest.mark.parametrize("T,B,H,W", [(4, 2, 128, 160), (4, 1, 96, 224), (5, 1, 128, 128)])
def test_seq_model_four_frames_matches_oracle(T, B, H, W):
    """seq_len 4 (and 5): no reference exists beyond T = 3, so the HIP path is checked against the oracle's
    generalisation (itself pinned at T = 3)."""
    from sgtapose_amd.synthetic import SEQ_ARGS, make_inputs_seq
    m, sd = _seq_model(T)
    inp = make_inputs_seq(B, H, W, T=T, seed=77)
    mv = lambda v: [u.to(DEV) for u in v] if isinstance(v, list) else v.to(DEV)      # noqa: E731
    out = m(*[mv(inp[k]) for k in SEQ_ARGS])[-1]
    ref = oracle.plan_a_window_seq_forward(sd, *[inp[k] for k in SEQ_ARGS])
    assert sorted(out) == ["hm", "reg", "tracking", "wh"]
    for k in out:
        err = float((out[k].cpu() - ref[k]).abs().max())
        assert out[k].shape == ref[k].shape and err < 1e-3, (k, err)
    # the detector step works on it too (decode of every sample)
    from sgtapose_amd.detector import BatchedDetector
    _, dets = BatchedDetector(m).process_device(*[mv(inp[k]) for k in SEQ_ARGS])
    assert dets["scores"].shape == (B, 7)


def test_sequence_mode_stem_caching_is_bit_identical(model):
    """A 4-frame video through SequenceDetector (image stem of frame t cached and reused
    at step t+1) equals the stateless detector on the same frame pairs, bit for bit (same kernel arithmetic, the two
    stems are only summed)."""
    from sgtapose_amd.detector import BatchedDetector, SequenceDetector
    from sgtapose_amd.synthetic import make_inputs
    B, H, W = 2, 128, 160
    frames = [make_inputs(B, H, W, seed=50 + i) for i in range(4)]
    seq, ref = SequenceDetector(model), BatchedDetector(model)
    prev_img = frames[0]["pre_img"].to(DEV)
    for i, f in enumerate(frames):
        img = f["x"].to(DEV)
        pri = [f[k].to(DEV) for k in ("pre_hm", "repro_hm", "pre_hm_cls", "repro_hm_cls")]
        out_s, det_s = seq.step_device(img, *pri, pre_image=prev_img if i == 0 else None)
        out_s = {k: v.clone() for k, v in out_s.items()}
        det_s = {k: v.clone() for k, v in det_s.items()}
        out_r, det_r = ref.process_device(img, prev_img, *pri)
        for k in out_r:
            assert torch.equal(out_s[k], out_r[k]), (i, k)
        for k in det_r:
            assert torch.equal(det_s[k], det_r[k]), (i, k)
        prev_img = img
    with pytest.raises(ValueError):
        SequenceDetector(model).step_device(img, *pri)          # first step without pre_image


@pytest.mark.parametrize("Hr,Wr,inp_h,inp_w", [(480, 640, 480, 640), (720, 1280, 736, 1280), (257, 333, 128, 160), (96, 64, 128, 128)])
def test_device_pre_process_matches_oracle_bit_exact(ops, Hr, Wr, inp_h, inp_w):
    """sgta_pre_process (affine warp + normalise, SGTADetector.pre_process) against oracle.prepost.pre_process: every
    float bit-identical (integer fixed-point warp, then the same two float32 roundings torch performs)."""
    from types import SimpleNamespace
    from oracle import prepost as op
    from sgtapose_amd import prepost
    rng = np.random.RandomState(Hr + Wr)
    imgs = rng.randint(0, 256, (3, Hr, Wr, 3)).astype(np.uint8)
    opt = SimpleNamespace(fix_short=0, fix_res=True, input_h=inp_h, input_w=inp_w, pad=31)
    out, metas = prepost.pre_process(imgs, opt)
    assert tuple(out.shape) == (3, 3, inp_h, inp_w) and len(metas) == 3
    assert metas[0]["out_height"] == inp_h // 4 and metas[0]["trans_output"].shape == (2, 3)
    for b in range(3):
        ref = op.pre_process(imgs[b], metas[b]["trans_input"], inp_w, inp_h)
        assert torch.equal(out[b:b + 1].cpu(), ref), b


# ---- pose step downstream of the key-points ---------------------------------------------------
def test_batched_pnp_and_gauss_newton_match_oracle(ops, model):
    """sgta_pnp_batched / sgta_gn_refine against oracle.pnp (solve_pnp / is_pnp unpinned vs OpenCV, GN pinned vs the
    reference's python twin through tests/golden/pnp_ops.npz): poses to 1e-7, projections to 1e-5 px, the generating
    pose recovered; missing key-points given by the detector's -999.999*4 marker."""
    from oracle import pnp as op
    from sgtapose_amd import pnp
    from test_oracle_golden import _gn_converged, _pnp_problems
    op, K, x3d, probs = _pnp_problems()
    B = len(probs)
    x2d = np.stack([p["x2d"] for p in probs])
    for b, p in enumerate(probs):
        x2d[b][~p["v"]] = -999.999 * 4                              # as SGTADetector._get_final_kps marks them
    X = np.stack([x3d] * B)
    r = pnp.solve_pnp(X, x2d, K, next_points=X)
    assert r["retval"].cpu().numpy().tolist() == [1] * B
    for b, p in enumerate(probs):
        v = p["v"]
        ok, t, q = op.solve_pnp(x3d[v], p["x2d"][v], K)
        tq = r["quaternion"][b].cpu().numpy()
        assert np.abs(r["translation"][b].cpu().numpy() - t).max() < 1e-7, b
        assert min(np.abs(tq - q).max(), np.abs(tq + q).max()) < 1e-7, b
        _, proj = op.is_pnp(x3d[v], p["x2d"][v], x3d, p["x2d"], K)
        assert np.abs(r["next_kp_projs_est"][b].cpu().numpy() - proj).max() < 1e-5, b
        if np.abs(p["c"]["noise"]).max() == 0:
            assert np.abs(r["translation"][b].cpu().numpy() - p["c"]["t"]).max() < 1e-8
    # fewer than four usable key-points: retval 0 and the detections stand in for the projections (is_pnp's else branch)
    bad = x2d[:2].copy()
    bad[:, 3:] = -999.999 * 4
    rb = pnp.solve_pnp(X[:2], bad, K, next_points=X[:2])
    assert rb["retval"].cpu().numpy().tolist() == [0, 0] and np.array_equal(rb["next_kp_projs_est"].cpu().numpy(), bad)
    # Gauss-Newton refinement: where the reference converged, the same end value; f / J through one step from value0
    v0 = np.stack([p["value0"] for p in probs])
    valid = np.stack([p["v"] for p in probs]).astype(np.uint8)
    q, T, it = pnp.register_GN_C(np.stack([p["x2d"] for p in probs]), X, v0[:, :4], v0[:, 4:], None, K, valid=valid)
    got = torch.cat([q, T], 1).cpu().numpy()
    n_conv = 0
    for b, p in enumerate(probs):
        if _gn_converged(op, K, x3d, p):
            n_conv += 1
            # the loop stops when sum|delta| <= 1e-4: end values agree to that order, not to rounding
            assert np.abs(got[b] - p["gn"]).max() < 5e-4, b
    assert n_conv >= 3
    q1, T1, it1 = pnp.register_GN_C(np.stack([p["x2d"] for p in probs]), X, v0[:, :4], v0[:, 4:], None, K, valid=valid, max_iter=1)
    for b, p in enumerate(probs):
        v = p["v"]
        w = [[1.0, 1.0]] * int(v.sum())
        with np.errstate(all="ignore"):
            one = op.gn_refine(p["value0"], p["x2d"][v], x3d[v], w, K, max_iter=1)
        assert np.allclose(torch.cat([q1, T1], 1)[b].cpu().numpy(), one, rtol=1e-6, atol=1e-9), b
    assert it1.cpu().numpy().tolist() == [1] * B
    # the loop between two frames on the device: final key-points -> PnP -> projections -> the four prior maps
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd import prepost
    det = BatchedDetector(model)
    c, s = np.array([320.0, 240.0], np.float32), 640.0
    meta = dict(c=c, s=s, width=640, height=480, inp_width=640, inp_height=480, out_width=160, out_height=120,
                trans_input=prepost.get_affine_transform(c, s, 0, [640, 480]),
                trans_output=prepost.get_affine_transform(c, s, 0, [160, 120]))
    maps = det.next_frame_priors(torch.from_numpy(x2d).to(DEV), X, X, K, [meta] * B)
    assert tuple(maps[0].shape) == (B, 1, 480, 640) and tuple(maps[3].shape) == (B, 7, 120, 160)
    ref_maps = det.render_inputs(x2d, r["next_kp_projs_est"], [meta] * B)
    for a, b_ in zip(maps[:4], ref_maps):
        assert torch.equal(a, b_)
    assert float(maps[1].max()) == 1.0


def test_fp32_step_is_deterministic_across_streams(model):
    """The step runs on three streams (attention beside the backbone, dead levels beside the neck, DCN projections beside the
    nodes): consecutive steps on the same inputs give bit-identical outputs -- a missing dependency would flicker."""
    from sgtapose_amd.detector import BatchedDetector
    from sgtapose_amd.synthetic import make_inputs
    det = BatchedDetector(model)
    inp = make_inputs(6, 128, 160, seed=5)
    dev_in = [inp[k].to(DEV) for k in ARGS]
    first = None
    for i in range(10):
        out, dets = det.process_device(*dev_in, static=True)
        cur = {k: v.clone() for k, v in out.items()}
        cur["kp"] = dets["kp"].clone()
        if first is None:
            first = cur
        else:
            for k in first:
                assert torch.equal(first[k], cur[k]), (i, k)

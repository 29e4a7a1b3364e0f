"""Python faces of the C-ABI entry points: torch tensors in, device pointers out to libsgta_hip.so.

torch is plumbing only (device memory + the current HIP stream); every function below enqueues
hand-written HIP kernels and raises if the library is missing (sgtapose_amd/_lib.py).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, dtype=torch.float32):
    if not (t.is_cuda and t.dtype == dtype):
        raise _lib.SgtaHipError("expected a %s tensor on the GPU, got %s on %s" % (dtype, t.dtype, t.device))
    return t


class View:
    """A [B, C, H, W] float32 activation that may be a channel slice of a wider buffer."""
    __slots__ = ("base", "c0", "C")

    def __init__(self, base, c0=0, C=None):
        self.base = _req(base)
        assert base.is_contiguous() and base.dim() == 4
        self.c0 = c0
        self.C = base.shape[1] - c0 if C is None else C
        assert 0 <= c0 and c0 + self.C <= base.shape[1]

    @property
    def B(self):
        return self.base.shape[0]

    @property
    def H(self):
        return self.base.shape[2]

    @property
    def W(self):
        return self.base.shape[3]

    @property
    def ptr(self):
        return self.base.data_ptr() + 4 * self.c0 * self.H * self.W

    @property
    def bs(self):
        return self.base.shape[1] * self.H * self.W

    def tensor(self):
        return self.base[:, self.c0:self.c0 + self.C]

    def slice(self, c0, C):
        return View(self.base, self.c0 + c0, C)


def as_view(t):
    return t if isinstance(t, View) else View(t.contiguous())


# ---------------------------------------------------------------------------------------------
# weight packing (host side, once per model): BN folding + the layouts the kernels read
# ---------------------------------------------------------------------------------------------
def _round_up(v, m):
    return (v + m - 1) // m * m


def fold_bn(weight, bias, bn, eps=1e-5):
    """Return (weight * s[co], shift) for conv -> BatchNorm(eval);  bn = (gamma, beta, mean, var) or None."""
    co = weight.shape[0]
    b = bias if bias is not None else torch.zeros(co, dtype=weight.dtype, device=weight.device)
    if bn is None:
        return weight, b
    gamma, beta, mean, var = bn
    s = gamma / torch.sqrt(var + eps)
    return weight * s.view(-1, *([1] * (weight.dim() - 1))), (b - mean) * s + beta


class PackedConv:
    """Packed weights of one conv / DCN main GEMM: wt [Kpad][Cout_pad], bias [Cout_pad]."""

    def __init__(self, weight, bias=None, bn=None, stride=1, pad=None, tap_major=False):
        w, b = fold_bn(weight.detach().float(), None if bias is None else bias.detach().float(), bn)
        co, ci, kh, kw = w.shape
        assert kh == kw
        self._w_folded = w
        self.Cout, self.Cin, self.ks, self.stride = co, ci, kh, stride
        self.pad = kh // 2 if pad is None else pad
        K = ci * kh * kw
        self.Kpad, self.Cout_pad = _round_up(K, 16), _round_up(co, 32)
        if tap_major:                           # DCN: k = tap*Cin + c
            flat = w.permute(2, 3, 1, 0).reshape(K, co)
        else:                                   # conv: k = c*ks*ks + tap
            flat = w.reshape(co, K).t()
        wt = torch.zeros(self.Kpad, self.Cout_pad, dtype=torch.float32, device=w.device)
        wt[:K, :co] = flat
        # kernel layout: [K chunk of 16][Cout_pad][16 k] -- one float4 = 4 consecutive k of one cout
        self.wt = wt.view(self.Kpad // 16, 16, self.Cout_pad).permute(0, 2, 1).contiguous()
        bb = torch.zeros(self.Cout_pad, dtype=torch.float32, device=w.device)
        bb[:co] = b
        self.bias = bb.contiguous()
        self.tap_major = tap_major
        self._ktab = {}
        # 3x3 / stride 1 / pad 1: second layout for the LDS-patch kernel, [Cin/4][Cout_pad][36]
        self.wt_patch = None
        if kh == 3 and stride in (1, 2) and self.pad == 1 and ci % 4 == 0 and not tap_major:
            wp = torch.zeros(ci // 4, self.Cout_pad, 36, dtype=torch.float32, device=w.device)
            wp[:, :co] = w.reshape(co, ci // 4, 36).permute(1, 0, 2)
            self.wt_patch = wp.contiguous()

        # 1x1: GEMM layout of sgta_conv1x1_f32, [Cin_pad/16][Cout_pad][20]
        self.wt_1x1 = None
        if kh == 1 and stride == 1 and not tap_major:
            w1 = torch.zeros(self.Kpad // 16, self.Cout_pad, 20, dtype=torch.float32, device=w.device)
            wf = torch.zeros(self.Cout_pad, self.Kpad, dtype=torch.float32, device=w.device)
            wf[:co, :ci] = w.reshape(co, ci)
            w1[:, :, :16] = wf.view(self.Cout_pad, self.Kpad // 16, 16).permute(1, 0, 2)
            self.wt_1x1 = w1.contiguous()
        # 1x1 with <= 8 outputs (head output convs): plain [Cout][Cin] rows for sgta_conv1x1_skinny_f32
        self.w_skinny = None
        if kh == 1 and stride == 1 and co <= 8 and not tap_major:
            self.w_skinny = w.reshape(co, ci).contiguous()
            self.b_skinny = self.bias[:co].contiguous()
        # Winograd F(2x2,3x3): U = G g G^T, packed [Cin/8][8][Cout_pad][16 positions + 4 pad]
        self.u_wino = None
        if kh == 3 and stride == 1 and self.pad == 1 and ci % 8 == 0 and not tap_major:
            G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64, device=w.device)
            U = torch.einsum("xi,ocij,yj->ocxy", G, w.double(), G).float()          # [co, ci, 4, 4]
            ur = torch.zeros(ci // 8, 8, self.Cout_pad, 20, dtype=torch.float32, device=w.device)
            ur[:, :, :co, :16] = U.reshape(co, ci // 8, 8, 16).permute(1, 2, 0, 3)
            self.u_wino = ur.contiguous()
        # Winograd F(4x4,3x3): U = G g G^T (6x6), packed [Cin/4][4][Cout_pad][36]
        self.u_wino4 = None
        if kh == 3 and stride == 1 and self.pad == 1 and ci % 4 == 0 and not tap_major:
            G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                               [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                              dtype=torch.float64, device=w.device)
            U4 = torch.einsum("xi,ocij,yj->ocxy", G4, w.double(), G4).float()       # [co, ci, 6, 6]
            u4 = torch.zeros(ci // 4, 4, self.Cout_pad, 36, dtype=torch.float32, device=w.device)
            u4[:, :, :co] = U4.reshape(co, ci // 4, 4, 36).permute(1, 2, 0, 3)
            self.u_wino4 = u4.contiguous()
        # 3x3 / stride 2: k-major rows [Cin/4][36][Cout_pad] (k_local = c_local*9 + tap) for sgta_conv3x3s2_f32
        self.wt_s2 = None
        if kh == 3 and stride == 2 and self.pad == 1 and ci % 4 == 0 and not tap_major:
            ws = torch.zeros(ci // 4, 36, self.Cout_pad, dtype=torch.float32, device=w.device)
            ws[:, :, :co] = w.reshape(co, ci // 4, 36).permute(1, 2, 0)
            self.wt_s2 = ws.contiguous()
        # DCN main GEMM: layout of the LDS-window kernel, [Cin/4][36][Cout_pad] (k_local = c_local*9 + tap)
        self.wt_dcn8 = None
        if tap_major and kh == 3 and ci % 4 == 0 and self.Cout_pad % 64 == 0:
            wd = torch.zeros(ci // 4, 36, self.Cout_pad, dtype=torch.float32, device=w.device)
            wd[:, :, :co] = w.reshape(co, ci // 4, 36).permute(1, 2, 0)
            self.wt_dcn8 = wd.contiguous()

    def dcn_tile(self, H, W, B=1):
        """(tile code, utilisation) of the best 128-pixel 2-D tile for sgta_dcnv2_lds_f32.  128 output channels per
        work-group (every gathered sample feeds four MFMAs instead of two: -15 % at 128 -> 128 on 60x80 maps) where
        that still leaves two rounds of work-groups; the small maps keep 64 (measured equal or slower with 128)."""
        best = None
        for code, tw in ((0, 32), (1, 16), (2, 8)):
            th = 128 // tw
            tiles = (-(-W // tw)) * (-(-H // th))
            util = (H * W) / float(tiles * 128)
            if best is None or util > best[1] + 1e-9:
                best = (code, util, tiles)
        mt = 2
        if DCN_MT == 4 and self.Cout_pad % 128 == 0 and B * best[2] * (self.Cout_pad // 128) >= DCN_MT4_MIN_WGS:
            mt = 4
        return (mt * 10 + best[0], best[1])

    def s2_tile(self, H, W):
        """Tile code for sgta_conv3x3s2_f32: best-filling 256-pixel output tile.  Inputs of at most 2*th/n - 1 rows are
        stacked n images to a tile by the kernel (same rule as its launcher)."""
        mt = 2 if self.Cout_pad % 64 == 0 else 1
        ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        best = None
        for code, tw in ((0, 32), (1, 16), (2, 8)):
            th = 256 // tw
            rows = -(-ho // th) * th
            for n in (4, 2):
                if th % n == 0 and H <= 2 * (th // n) - 1:
                    rows = th // n
                    break
            util = (ho * wo) / float((-(-wo // tw) * tw) * rows)
            if best is None or util > best[1] + 1e-9:
                best = (mt * 10 + code, util)
        return best[0] if best[1] >= S2_MIN_FILL else 0  # poorly filled tiles: generic implicit GEMM wins

    def wino4_tile(self, H, W):
        """(tile code, utilisation) of the best-filling 64-tile block shape for sgta_conv3x3_wino4_f32 (4x4 tiles)."""
        ty, tx = -(-H // 4), -(-W // 4)
        best = None
        for code, tc in ((0, 32), (1, 16), (2, 8), (3, 4)):
            tr = 64 // tc
            util = (H * W) / float((-(-ty // tr) * tr) * (-(-tx // tc) * tc) * 16)
            if best is None or util > best[1] + 1e-9:
                best = (code, util)
        return best

    def wino_tile(self, H, W):
        """Tile code for sgta_conv3x3_wino_f32: best-filling 64-tile block shape.  Maps of at most 2*tr/n - 1 rows
        are stacked n images to a block by the kernel (same rule as launch_wino in conv3x3_wino.hip)."""
        ty, tx = -(-H // 2), -(-W // 2)
        best = None
        for code, tc in ((0, 32), (1, 16), (2, 8), (3, 4)):
            tr = 64 // tc
            rows = -(-ty // tr) * tr
            for n in (4, 2):
                if tr % n == 0 and H <= 2 * (tr // n) - 1:
                    rows = tr // n
                    break
            util = (ty * tx) / float(rows * (-(-tx // tc) * tc))
            if best is None or util > best[1] + 1e-9:
                best = (code, util)
        return best[0]

    def patch_tile(self, W, tile=0, HW=None):
        """Tile id for sgta_conv3x3s1_f32, or 0 when that kernel does not cover (W, tile)."""
        if self.wt_patch is None or self.stride != 1:
            return 0
        if tile == 0:
            # measured (tools/conv_bench.py): 64x256 is the fastest tile whenever it applies; small
            # maps (level 5: 15x20 per image) waste fewer pixel slots with the 128-pixel tile
            tile = 2 if self.Cout_pad % 64 == 0 else 1
            if self.Cout_pad % 128 == 0 and (HW is not None and HW < 1024 or W > 255):
                tile = 3
        if tile == 3 and self.Cout_pad % 128:
            tile = 2
        if tile == 2 and self.Cout_pad % 64:
            tile = 1
        limit = {1: 767, 2: 255, 3: 191}[tile]
        return tile if W <= limit else 0

    def unpacked(self):
        """[Kpad, Cout_pad] view of the packed weights (tests)."""
        return self.wt.permute(0, 2, 1).reshape(self.Kpad, self.Cout_pad)

    def ktab(self, H, W):
        """int32 [Kpad, 2] = (c*H*W + ky*W + kx, tap) per GEMM k; tap 63 marks the zero padding of K."""
        key = (H, W)
        t = self._ktab.get(key)
        if t is None:
            K, kk = self.Cin * self.ks * self.ks, self.ks * self.ks
            k = torch.arange(self.Kpad, dtype=torch.int64)
            c, tap = k // kk, k % kk
            off = c * (H * W) + (tap // self.ks) * W + tap % self.ks
            tab = torch.stack([torch.where(k < K, off, torch.zeros_like(off)),
                               torch.where(k < K, tap, torch.full_like(tap, 63))], 1)
            t = tab.to(torch.int32).contiguous().to(self.wt.device)
            self._ktab[key] = t
        return t

    def flops(self, M):
        return 2.0 * M * self.Cout * self.Cin * self.ks * self.ks


def conv2d(x, pc, out, act=ACT_NONE, res=None, res_mode=0, tile=0):
    """out = act(conv(x) + bias [+ res]) ; x/out/res are Views (or contiguous tensors)."""
    x, out = as_view(x), as_view(out)
    assert x.C == pc.Cin and out.C == pc.Cout and out.B == x.B
    rp, rbs = None, 0
    if res is not None:
        res = as_view(res)
        assert res_mode in (1, 2) and res.C == pc.Cout and res.H == out.H and res.W == out.W
        rp, rbs = res.ptr, res.bs
    if (USE_CONV1X1 and pc.w_skinny is not None and tile == 0 and res is None and (x.H * x.W) % 4 == 0
            and x.bs % 4 == 0 and out.bs % 4 == 0 and x.ptr % 16 == 0 and out.ptr % 16 == 0):
        meta = None
        if _lib.PROFILE is not None:
            meta = _conv_meta(pc, out, 2, False)
            meta["kernel"] = "conv1x1_skinny_kernel<%d>" % pc.Cout
        _lib.call("sgta_conv1x1_skinny_f32", x.ptr, x.bs, pc.w_skinny.data_ptr(), pc.b_skinny.data_ptr(), out.ptr,
                  out.bs, x.B, pc.Cin, x.H, x.W, pc.Cout, act, _stream(), meta=meta)
        return out
    if (USE_CONV1X1 and pc.wt_1x1 is not None and tile == 0 and (x.H * x.W) % 4 == 0 and x.bs % 4 == 0
            and x.ptr % 16 == 0):
        meta = None
        if _lib.PROFILE is not None:
            meta = _conv_meta(pc, out, 2, False)
            meta["kernel"] = "conv1x1_kernel<%d>" % (2 if pc.Cout_pad % 64 == 0 else 1)
        _lib.call("sgta_conv1x1_f32", x.ptr, x.bs, pc.wt_1x1.data_ptr(), pc.bias.data_ptr(), rp, rbs, out.ptr,
                  out.bs, x.B, pc.Cin, x.H, x.W, pc.Cout, pc.Cout_pad, pc.Kpad, act,
                  res_mode if res is not None else 0, _stream(), meta=meta)
        return out
    if (USE_S2_KERNEL and pc.ks == 3 and pc.stride == 2 and pc.wt_s2 is not None and tile == 0
            and act == ACT_RELU and res is None):
        code = pc.s2_tile(x.H, x.W)
    else:
        code = 0
    if code:
        meta = None
        if _lib.PROFILE is not None:
            meta = _conv_meta(pc, out, 2, False)
            meta["kernel"] = "conv3x3_s2_kernel<%d,%d>" % (code // 10, (32, 16, 8)[code % 10])
        _lib.call("sgta_conv3x3s2_f32", x.ptr, x.bs, pc.wt_s2.data_ptr(), pc.bias.data_ptr(), out.ptr, out.bs,
                  x.B, pc.Cin, x.H, x.W, pc.Cout, pc.Cout_pad, code, _stream(), meta=meta)
        return out
    # epilogues the two Winograd entry points instantiate: (act, res_mode) in _WINO_EPILOGUES; others fall through
    wino_ok = (act, res_mode if res is not None else 0) in _WINO_EPILOGUES
    if (wino_ok and USE_WINOGRAD and USE_WINO4 and pc.u_wino4 is not None and tile == 0 and x.W % 4 == 0 and x.bs % 4 == 0
            and out.bs % 4 == 0 and x.ptr % 16 == 0 and out.ptr % 16 == 0
            and (res is None or (rbs % 4 == 0 and rp % 16 == 0))):
        code, util = pc.wino4_tile(x.H, x.W)
        if util >= WINO4_MIN_FILL:
            meta = None
            if _lib.PROFILE is not None:
                meta = _conv_meta(pc, out, 2, False)
                meta["kernel"] = "conv3x3_wino4_kernel<%d>" % (32, 16, 8, 4)[code]
                meta["exec_flops"] = meta["flops"] * 36.0 / 144.0     # F(4x4,3x3): 36 of 144 multiplies
            _lib.call("sgta_conv3x3_wino4_f32", x.ptr, x.bs, pc.u_wino4.data_ptr(), pc.bias.data_ptr(), rp, rbs,
                      out.ptr, out.bs, x.B, pc.Cin, x.H, x.W, pc.Cout, pc.Cout_pad, act,
                      res_mode if res is not None else 0, code, _stream(), meta=meta)
            return out
    if wino_ok and USE_WINOGRAD and pc.u_wino is not None and tile == 0:
        code = pc.wino_tile(x.H, x.W) if WINO_TILE_OVERRIDE is None else WINO_TILE_OVERRIDE
        meta = None
        if _lib.PROFILE is not None:
            meta = _conv_meta(pc, out, 2, False)
            meta["kernel"] = "conv3x3_wino_kernel<%d>" % (32, 16, 8, 4)[code]
            meta["exec_flops"] = meta["flops"] * 16.0 / 36.0      # F(2x2,3x3): 16 of 36 multiplies
        ws, nws = None, 0
        if act == ACT_NONE and res is None and pc.Cout_pad == 32:      # split-K candidates: the DCN offset convs
            nws = 8 * x.B * pc.Cout * x.H * x.W
            ws = _splitk_ws(out.base.device, nws)
        _lib.call("sgta_conv3x3_wino_f32", x.ptr, x.bs, pc.u_wino.data_ptr(), pc.bias.data_ptr(), rp, rbs,
                  out.ptr, out.bs, x.B, pc.Cin, x.H, x.W, pc.Cout, pc.Cout_pad, act,
                  res_mode if res is not None else 0, code, None if ws is None else ws.data_ptr(), nws, _stream(),
                  meta=meta)
        return out
    pt = pc.patch_tile(x.W, tile, x.H * x.W) if USE_PATCH_KERNEL else 0
    if pt:
        _lib.call("sgta_conv3x3s1_f32", x.ptr, x.bs, pc.wt_patch.data_ptr(), pc.bias.data_ptr(), rp, rbs,
                  out.ptr, out.bs, x.B, pc.Cin, x.H, x.W, pc.Cout, pc.Cout_pad, act,
                  res_mode if res is not None else 0, pt, _stream(),
                  meta=_conv_meta(pc, out, pt, False, patch=True) if _lib.PROFILE is not None else None)
        return out
    _lib.call("sgta_conv2d_f32", x.ptr, x.bs, pc.wt.data_ptr(), pc.bias.data_ptr(),
              pc.ktab(x.H, x.W).data_ptr(), rp, rbs, out.ptr,
              out.bs, x.B, pc.Cin, x.H, x.W, pc.Cout, pc.Cout_pad, pc.Kpad, pc.ks, pc.stride, pc.pad,
              act, res_mode if res is not None else 0, tile, _stream(),
              meta=_conv_meta(pc, out, tile, False) if _lib.PROFILE is not None else None)
    return out


DEBUG_OM_SCALE = None          # tools: multiply the DCN sampling offsets (never set in the product)
_WINO_EPILOGUES = {(ACT_NONE, 0), (ACT_RELU, 0), (ACT_RELU, 1), (ACT_RELU, 2), (ACT_SIGMOID, 0)}
_SPLITK_WS = {}


def _splitk_ws(dev, nfloats):
    """Workspace of the split-K Winograd path: one buffer per (device, stream) -- layers on different streams run
    concurrently -- grown on demand (<= 8*B*27*H*W floats for the level-4/5 offset convs)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if dev is None else dev
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    t = _SPLITK_WS.get(key)
    if t is None or t.numel() < nfloats:
        t = _SPLITK_WS[key] = torch.empty(nfloats, dtype=torch.float32, device=dev)
    return t


LINEAR_SPLITK_MAX_ROWS = 1024   # token GEMMs with at most this many rows may split K (0 disables)
DCN_MT4_MIN_WGS = 1024         # work-groups a 128-channel DCN launch must still have (two rounds at 2 per CU)
DCN_MT = 4                     # 4: 128 output channels per DCN work-group where it pays (dcn_tile), 2: always 64
S2_MIN_FILL = 0.7              # minimum useful fraction of the stride-2 kernel's 256-pixel tiles
WINO_TILE_OVERRIDE = None      # tests / tools: force the F(2x2,3x3) block shape (tile code 0..3)
USE_WINO4 = True               # F(4x4,3x3) where the map fills its 64-tile blocks (else F(2x2,3x3))
WINO4_MIN_FILL = 0.7           # minimum useful fraction of the 4x4-tile blocks
USE_WINOGRAD = True            # 3x3 s1 convs with Cin % 8 == 0 go through Winograd F(2x2,3x3) (fp32)
USE_S2_KERNEL = True           # LDS-window kernel for 3x3 stride-2 convs (False: generic implicit GEMM)
USE_CONV1X1 = True             # LDS-DMA GEMM kernel for 1x1 convs (False: generic implicit GEMM)
USE_DCN_LDS_KERNEL = True      # False: gather-from-global DCN (conv_igemm.hip)
USE_PATCH_KERNEL = True        # tools/tests flip this to time / check the generic implicit-GEMM path


def _conv_meta(pc, out, tile, dcn, patch=False):
    """Algorithmic work of one launch: real (un-padded) MACs and compulsory bytes."""
    M = out.B * out.H * out.W
    if tile == 0:
        tile = 3 if pc.Cout_pad % 128 == 0 else (2 if pc.Cout_pad % 64 == 0 else 1)
    mt_nt = {1: "1,2", 2: "2,2", 3: "4,1"}[tile]
    kern = "conv_igemm_kernel<%d,%s,%s>" % (pc.ks, mt_nt, "true" if dcn else "false")
    if patch:
        kern = "conv3x3_patch_kernel<%s>" % mt_nt
    hw_in = M * pc.stride * pc.stride
    byts = 4.0 * (pc.Cin * hw_in + pc.Cout * M + (27 * M if dcn else 0)) + 4.0 * pc.Cout * pc.Cin * pc.ks ** 2
    return {"kernel": kern, "flops": pc.flops(M), "bytes": byts, "exec_flops": pc.flops(M)}


class PackedC16:
    """Weights of the 16-cout full-resolution layers (sgta_conv_c16_f32): level0 or the fused stems."""

    def __init__(self, w_main, bn_main, w_hm=None, bn_hm=None):
        w, b = fold_bn(w_main.detach().float(), None, bn_main)
        co, ci, ks, _ = w.shape
        assert co == 16
        if w_hm is None:                        # level0: [144][16]
            assert ci == 16 and ks == 3
            self.mode, self.ks, self.cin = 0, 3, 16
            self.wt = w.reshape(16, ci * ks * ks).t().contiguous()
            self.bias = b.contiguous()
        else:                                   # stems: channel group {r, g, b, hm}
            assert ci == 3 and ks == 7
            w2, b2 = fold_bn(w_hm.detach().float(), None, bn_hm)
            kk = ks * ks
            # flattened k: image conv k = ch*49 + tap (147 -> 148 rows), heat-map conv k = tap (49 -> 52 rows)
            a_img = torch.zeros(148, 16, dtype=torch.float32, device=w.device)
            a_hm = torch.zeros(52, 16, dtype=torch.float32, device=w.device)
            a_img[:3 * kk] = w.reshape(16, 3 * kk).t()
            a_hm[:kk] = w2.reshape(16, kk).t()
            self.mode, self.ks, self.cin = 1, 7, 4
            self.wt = torch.cat([a_img, a_hm], 0).contiguous()
            self.bias = torch.cat([b, b2]).contiguous()

    def flops(self, M):
        return 2.0 * M * 16 * self.cin * self.ks * self.ks


def conv_c16(x, x2, pc, out, mode=None):
    """mode None: pc.mode (0 level0 / 1 fused stems); 2: image stem only; 3: cached image stem ``x`` + heat-map stem of
    ``x2`` (sequence mode, sgta_conv_c16_f32)."""
    x, out = as_view(x), as_view(out)
    x2p, x2bs = None, 0
    if x2 is not None:
        x2 = as_view(x2)
        x2p, x2bs = x2.ptr, x2.bs
    meta = None
    if _lib.PROFILE is not None:
        M = out.B * out.H * out.W
        meta = {"kernel": "conv_c16_kernel<%d>" % pc.ks, "flops": pc.flops(M), "bytes": 4.0 * M * (pc.cin + 16),
                "exec_flops": pc.flops(M) * (200.0 / 196.0 if pc.mode == 1 else 1.0)}     # 148 + 52 padded k
    _lib.call("sgta_conv_c16_f32", x.ptr, x.bs, x2p, x2bs, pc.wt.data_ptr(), pc.bias.data_ptr(), out.ptr,
              out.bs, x.B, x.H, x.W, pc.mode if mode is None else mode, _stream(), meta=meta)
    return out


def conv3x3s2_pool(x, pc, out, pool):
    """relu(bn(conv3x3 stride 2)) -> out AND MaxPool2d(2,2)(out) -> pool in one launch when the stride-2 kernel takes the
    layer with its 8x32 tile; returns False (nothing done) otherwise."""
    x, out, pool = as_view(x), as_view(out), as_view(pool)
    if not (USE_S2_KERNEL and pc.ks == 3 and pc.stride == 2 and pc.wt_s2 is not None and x.H >= 16):
        return False
    code = pc.s2_tile(x.H, x.W)
    if code == 0 or code % 10 != 0:
        return False
    assert pool.B == out.B and pool.C == out.C and pool.H == out.H // 2 and pool.W == out.W // 2
    meta = None
    if _lib.PROFILE is not None:
        meta = _conv_meta(pc, out, 2, False)
        meta["kernel"] = "conv3x3_s2_kernel<%d,%d>" % (code // 10, 32)
    _lib.call("sgta_conv3x3s2_pool_f32", x.ptr, x.bs, pc.wt_s2.data_ptr(), pc.bias.data_ptr(), out.ptr, out.bs,
              pool.ptr, pool.bs, x.B, pc.Cin, x.H, x.W, pc.Cout, pc.Cout_pad, code, _stream(), meta=meta)
    return True


def dcn_v2(x, om, pc, out, act=ACT_NONE, tile=0):
    """Fused modulated deformable conv: ``om`` is the raw 27-channel offset/mask conv output."""
    x, om, out = as_view(x), as_view(om), as_view(out)
    assert om.C == 27 and x.C == pc.Cin and out.C == pc.Cout
    if DEBUG_OM_SCALE is not None:                 # tools only: scale the sampling offsets (cost of the far-sample path)
        om.tensor()[:, :18] *= DEBUG_OM_SCALE
    if USE_DCN_LDS_KERNEL and pc.wt_dcn8 is not None and tile == 0:
        code, _ = pc.dcn_tile(x.H, x.W, x.B)
        meta = None
        if _lib.PROFILE is not None:
            meta = _conv_meta(pc, out, 2, True)
            meta["kernel"] = "dcn_patch_kernel<%d,%d>" % (code // 10, (32, 16, 8)[code % 10])
        _lib.call("sgta_dcnv2_lds_f32", x.ptr, x.bs, om.ptr, om.bs, pc.wt_dcn8.data_ptr(), pc.bias.data_ptr(),
                  out.ptr, out.bs, x.B, pc.Cin, x.H, x.W, pc.Cout, pc.Cout_pad, act, code, _stream(), meta=meta)
        return out
    _lib.call("sgta_dcnv2_fwd_f32", x.ptr, x.bs, om.ptr, om.bs, pc.wt.data_ptr(), pc.bias.data_ptr(),
              out.ptr, out.bs, x.B, pc.Cin, x.H, x.W, pc.Cout, pc.Cout_pad, act, tile, _stream(),
              meta=_conv_meta(pc, out, tile, True) if _lib.PROFILE is not None else None)
    return out


def maxpool2(x, out):
    x, out = as_view(x), as_view(out)
    _lib.call("sgta_maxpool2_f32", x.ptr, x.bs, out.ptr, out.bs, x.B, x.C, x.H, x.W, _stream())
    return out


def upsample_add(x, w, skip, out, f):
    x, out = as_view(x), as_view(out)
    sp, sbs = None, 0
    if skip is not None:
        skip = as_view(skip)
        sp, sbs = skip.ptr, skip.bs
    _lib.call("sgta_upsample_add_f32", x.ptr, x.bs, _req(w).data_ptr(), sp, sbs, out.ptr, out.bs, x.B,
              x.C, x.H, x.W, f, _stream())
    return out


def sigmoid_(t):
    _req(t)
    assert t.is_contiguous()
    _lib.call("sgta_sigmoid_f32", t.data_ptr(), t.data_ptr(), t.numel(), _stream())
    return t


def prior_argmax(pre_cls, cur_cls, out=None):
    B, K, h, w = pre_cls.shape
    out = torch.empty(2, B, K, 2, dtype=torch.float32, device=pre_cls.device) if out is None else out
    _lib.call("sgta_prior_argmax", _req(pre_cls.contiguous()).data_ptr(), _req(cur_cls.contiguous()).data_ptr(),
              out.data_ptr(), B, K, h, w, _stream())
    return out


def window_index(idx_xy, scale, ks, H, W, fid=None, win=None):
    _, B, K, _ = idx_xy.shape
    n = K * (1 + 2 * (ks // 2)) ** 2
    fid = torch.empty(2, B, n, dtype=torch.int64, device=idx_xy.device) if fid is None else fid
    win = torch.empty(B, n, dtype=torch.uint8, device=idx_xy.device) if win is None else win
    _lib.call("sgta_window_index", idx_xy.data_ptr(), fid.data_ptr(), win.data_ptr(), B, K,
              ctypes.c_float(scale), ks, H, W, _stream())
    return fid, win


def window_gather(feats, fid, tok, col0=0):
    feats = as_view(feats)
    B, n = fid.shape
    ld = tok.shape[-1]
    _lib.call("sgta_window_gather", feats.ptr, feats.bs, fid.data_ptr(), tok.data_ptr(), B, feats.C,
              feats.H, feats.W, n, ld, col0, _stream())
    return tok


def window_scatter(tok, fid, win, feats):
    feats = as_view(feats)
    B, n = fid.shape
    assert tok.is_contiguous() and tok.shape[-1] == feats.C
    _lib.call("sgta_window_scatter", tok.data_ptr(), fid.data_ptr(), win.data_ptr(), feats.ptr, feats.bs,
              B, feats.C, feats.H, feats.W, n, _stream())
    return feats


def linear(x, w, bias, out, relu=False, res=None, ln=None, K=None):
    """out[M,N] = LN?(relu?(x[:, :K] @ w.T + bias) + res); x/out/res are 2-D row-major (last dim = ld)."""
    M = x.shape[0]
    N, Kw = w.shape
    K = Kw if K is None else K
    assert x.stride(-1) == 1 and out.stride(-1) == 1 and w.is_contiguous()
    g, bt = (ln[0].data_ptr(), ln[1].data_ptr()) if ln is not None else (None, None)
    rp, ldr = (res.data_ptr(), res.stride(0)) if res is not None else (None, 0)
    ws, nws = None, 0
    if ln is None and M <= LINEAR_SPLITK_MAX_ROWS and K >= 256:
        # few-hundred-token products (cat_layer of levels 3-5): split-K over work-groups, see sgta_linear_ws_f32
        nws = min(32 * M * N, 4 << 20)
        ws = _splitk_ws(x.device, nws)
    meta = None
    if _lib.PROFILE is not None:           # token kernels always run on the f32 matrix pipe, whatever the engine's dtype
        meta = {"kernel": "linear_kernel (token GEMMs, f32)", "flops": 2.0 * M * N * K, "exec_flops": 2.0 * M * N * K,
                "bytes": 4.0 * (M * K + N * K + M * N), "peak_tflops": F32_MFMA_PEAK_TFLOPS}
    _lib.call("sgta_linear_ws_f32", x.data_ptr(), x.stride(0), w.data_ptr(),
              None if bias is None else bias.data_ptr(), rp, ldr, g, bt, out.data_ptr(), out.stride(0),
              M, N, K, int(relu), None if ws is None else ws.data_ptr(), nws, _stream(), meta=meta)
    return out


F32_MFMA_PEAK_TFLOPS = 157.3    # f32-input MFMA peak of one MI355X (bench.py's roofline denominators)
USE_FUSED_FFN = True            # encoder feed-forward block in one kernel (False: two linear launches, tests)


def ffn_ln(x, w1, b1, w2, b2, ln, out):
    """out = LayerNorm(x + relu(x @ w1.T + b1) @ w2.T + b2): fused feed-forward block (sgta_ffn_ln_f32) when the
    shape / alignment contract holds, else None (the caller falls back to two ``linear`` calls)."""
    M, C = x.shape
    F = w1.shape[0]
    ok = (USE_FUSED_FFN and C in (16, 32, 64) and F % 16 == 0 and x.stride(-1) == 1 and out.stride(-1) == 1
          and x.stride(0) % 4 == 0 and out.stride(0) % 4 == 0 and w1.is_contiguous() and w2.is_contiguous()
          and all(t.data_ptr() % 16 == 0 for t in (x, out, w1, b1, w2, b2, ln[0], ln[1])))
    if not ok:
        return None
    meta = None
    if _lib.PROFILE is not None:
        meta = {"kernel": "ffn_ln_kernel (encoder FFN, f32)", "flops": 4.0 * M * C * F, "exec_flops": 4.0 * M * C * F,
                "bytes": 4.0 * (2 * M * C + 2 * C * F), "peak_tflops": F32_MFMA_PEAK_TFLOPS}
    _lib.call("sgta_ffn_ln_f32", x.data_ptr(), x.stride(0), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
              ln[0].data_ptr(), ln[1].data_ptr(), out.data_ptr(), out.stride(0), M, C, F, _stream(), meta=meta)
    return out


def mhca(q, k, v, pos, out, B, n, n_heads, head_dim):
    """q/k/v/out: 2-D [B*n, ld] row-major views (column slices allowed).  ``pos``: [n_heads, n, n] bias, a ``PosTiles``
    (the same bias re-packed by ``pack_pos_tiles``), or None."""
    meta = None
    if _lib.PROFILE is not None:
        # algorithmic: QK^T and PV, 2 n^2 d each per (sample, head).  Issued to the matrix pipe: QK^T over 16-padded token
        # tiles (k = head_dim, fully used) and, for head_dim >= 8, PV as 16-row MFMAs of which head_dim rows are used;
        # head_dim 4 does PV on the VALU.  bytes: the bias stream (4 B per (head, query, key)) + q, k, v, out.
        n16 = -(-n // 16) * 16
        meta = {"kernel": "mhca_kernel<head_dim %d> (windowed attention, f32)" % head_dim,
                "flops": 4.0 * B * n_heads * n * n * head_dim,
                "exec_flops": 2.0 * B * n_heads * n16 * n16 * (head_dim + (16 if head_dim >= 8 else 0)),
                "bytes": 4.0 * B * n_heads * (n * n + 4 * n * head_dim), "peak_tflops": F32_MFMA_PEAK_TFLOPS}
    if isinstance(pos, PosTiles):
        assert pos.n == n and pos.tiles.shape[0] == n_heads
        _lib.call("sgta_mhca_tiled_fwd_f32", q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(),
                  v.stride(0), pos.tiles.data_ptr(), out.data_ptr(), out.stride(0), B, n, n_heads, head_dim, _stream(),
                  meta=meta)
        return out
    _lib.call("sgta_mhca_fwd_f32", q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(),
              v.stride(0), None if pos is None else pos.data_ptr(), out.data_ptr(), out.stride(0), B, n,
              n_heads, head_dim, _stream())
    return out


class PosTiles:
    """The structure-prior bias [heads, n, n] (MHCA_ein.pos_embed, dla.py:866-868) in the 16 x 16-tile layout of
    sgta_mhca_tiled_fwd_f32: tiles[h, it, jt, li, c] = pos[h, 16 it + li, 16 jt + c]; key columns beyond n are -inf
    (the padding mask the kernel would otherwise apply), query rows beyond n are 0."""

    def __init__(self, pos):
        heads, n, _ = pos.shape
        T = -(-n // 16)
        full = torch.zeros(heads, T * 16, T * 16, dtype=torch.float32, device=pos.device)
        full[:, :, n:] = float("-inf")
        full[:, :n, :n] = pos
        self.n = n
        self.tiles = full.view(heads, T, 16, T, 16).permute(0, 1, 3, 2, 4).contiguous()


_GAUSS = {}


def gauss_taps(device):
    """The 13 float64 taps (far .. centre) of scipy's sigma=3, truncate=4 kernel, computed on the
    host exactly as scipy does (exp in float64, normalised by the float64 sum)."""
    key = str(device)
    if key not in _GAUSS:
        x = np.arange(-12, 13)
        phi = np.exp(-0.5 / 9.0 * x ** 2)
        phi = phi / phi.sum()
        _GAUSS[key] = torch.from_numpy(phi[:13].copy()).to(device)
    return _GAUSS[key]


def decode_buffers(B, K, dev, with_reg=True, alloc=None):
    """The output tensors of ``peaks_decode`` (the engine passes its workspace allocator so that a step allocates
    nothing)."""
    alloc = alloc or (lambda name, shape, dtype: torch.empty(shape, dtype=dtype, device=dev))
    out = dict(scores=alloc("dec_scores", (B, K), torch.float32), xs=alloc("dec_xs", (B, K), torch.int64),
               ys=alloc("dec_ys", (B, K), torch.int64), regs=alloc("dec_regs", (B, K, 2), torch.float32),
               trks=alloc("dec_trks", (B, K, 2), torch.float32), npeaks=alloc("dec_npeaks", (B, K), torch.int32),
               cts=alloc("dec_cts", (B, K, 2), torch.int64), kp=alloc("dec_kp", (B, K, 3), torch.float32))
    if with_reg:
        out["cts_wreg"] = alloc("dec_cts_wreg", (B, K, 2), torch.float32)
        out["regs_rel"] = alloc("dec_regs_rel", (B, K, 2), torch.float32)
    return out


def peaks_decode(hm, reg=None, tracking=None, ws=None, three_pass=False, out=None):
    """hm [B,K,H,W] post-sigmoid.  Returns dict(scores, xs, ys, regs, trks, npeaks, cts, kp[, cts_wreg, regs_rel]) of
    device tensors (the last two only with ``reg``; ``kp`` [B,K,3] = x + reg_x, y + reg_y, score).  ``three_pass``
    forces the un-fused kernels (cross-check); ``out`` = pre-allocated ``decode_buffers``."""
    _req(hm)
    B, K, H, W = hm.shape
    dev = hm.device
    hm = hm.contiguous()
    if ws is None:
        ws = (torch.empty_like(hm), torch.empty_like(hm))
    if out is None:
        out = decode_buffers(B, K, dev, reg is not None)
    _lib.call("sgta_peaks_decode_kp", hm.data_ptr(), None if reg is None else _req(reg.contiguous()).data_ptr(),
              None if tracking is None else _req(tracking.contiguous()).data_ptr(),
              gauss_taps(dev).data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), out["scores"].data_ptr(),
              out["xs"].data_ptr(), out["ys"].data_ptr(), out["regs"].data_ptr(), out["trks"].data_ptr(),
              out["npeaks"].data_ptr(), out["cts"].data_ptr(),
              out["cts_wreg"].data_ptr() if reg is not None else None,
              out["regs_rel"].data_ptr() if reg is not None else None, out["kp"].data_ptr(), B, K, H, W,
              int(three_pass), _stream())
    return out


def peaks_decode_static(hm, reg, tracking, out, three_pass=False):
    """``peaks_decode`` for callers that may not allocate: the two workspaces live in a device array owned by the library
    (sgta_peaks_decode_kp_static), every output is the caller's (``out`` = ``decode_buffers``).  B*K*H*W is limited to
    ``sgta_peaks_decode_static_max_elems()``; calls that overlap on one GPU race on the shared array."""
    _req(hm)
    B, K, H, W = hm.shape
    if not hm.is_contiguous() or (reg is not None and not reg.is_contiguous()) or \
            (tracking is not None and not tracking.is_contiguous()):
        raise _lib.SgtaHipError("peaks_decode_static takes contiguous tensors (it makes no copies)")
    lim = _lib.load().sgta_peaks_decode_static_max_elems()
    if B * K * H * W > lim:
        raise _lib.SgtaHipError("peaks_decode_static: B*K*H*W = %d exceeds the library's static scratch (%d elements); "
                                "use peaks_decode with caller workspaces" % (B * K * H * W, lim))
    _lib.call("sgta_peaks_decode_kp_static", hm.data_ptr(), None if reg is None else _req(reg).data_ptr(),
              None if tracking is None else _req(tracking).data_ptr(), gauss_taps(hm.device).data_ptr(),
              out["scores"].data_ptr(), out["xs"].data_ptr(), out["ys"].data_ptr(), out["regs"].data_ptr(),
              out["trks"].data_ptr(), out["npeaks"].data_ptr(), out["cts"].data_ptr(),
              out["cts_wreg"].data_ptr() if reg is not None else None,
              out["regs_rel"].data_ptr() if reg is not None else None, out["kp"].data_ptr(), B, K, H, W,
              int(three_pass), _stream())
    return out


def soft_argmax(hm, beta=25.0, size_mult=1.0):
    _req(hm)
    B, K, H, W = hm.shape
    out = torch.empty(B, K, 2, dtype=torch.float32, device=hm.device)
    _lib.call("sgta_softargmax", hm.contiguous().data_ptr(), out.data_ptr(), B, K, H, W,
              ctypes.c_float(beta), ctypes.c_float(size_mult), _stream())
    return out


def head_scale(head_dim):
    return 1.0 / math.sqrt(head_dim)

"""bf16 data path of the three-frame variant ``DLA_PlanAWindow_Three`` (reference dla.py:1400-1456): engine_three.EngineThree's
structure (two encoder passes per level 0-2, the folded ``cat[pre, pre, cur]`` 1x1 convolutions at levels 3-5, 'wh' head) on
engine_bf16.EngineBF16's kernels.  Token kernels stay fp32; the level 3-5 cat-MLP runs as bf16 1x1 convolutions."""
import torch

from . import _lib, ops
from .engine_bf16 import EngineBF16
from .engine_three import EngineThree
from .ops import ACT_NONE, ACT_RELU
from .ops_bf16 import PackedConvBF16


class EngineThreeBF16(EngineBF16):
    _encoder = EngineThree._encoder                     # fp32 token kernels

    def _pack(self, m):
        super()._pack(m)
        self._pack_three(m, PackedConvBF16, self.P)

    def _pack_three(self, m, pack_conv, dest):
        self.tf = [self._pack_encoder(m.prev_transformer[i]) for i in range(3)]
        self.tf_pp = self.tf if m.shared_ca else [self._pack_encoder(m.pprev_transformer[i]) for i in range(3)]
        for i in (3, 4, 5):                              # full-map cat-MLP as 1x1 convs (the two ``pre`` slices folded)
            s = m.cat_layer[i]
            C = s[2].out_features
            w0 = s[0].weight.detach().float()
            dest["cat%d.pre" % i] = pack_conv((w0[:, :C] + w0[:, C:2 * C]).reshape(-1, C, 1, 1).contiguous(), None)
            dest["cat%d.cur" % i] = pack_conv(w0[:, 2 * C:].reshape(-1, C, 1, 1).contiguous(), s[0].bias)
            dest["cat%d.out" % i] = pack_conv(s[2].weight.detach().float().reshape(C, -1, 1, 1).contiguous(), s[2].bias)

    def temporal_three(self, lv, idx_pp, idx_p, B, levels=range(6)):
        K = self.num_classes
        CP = self.conv_packs
        for i in levels:
            feats = lv[i]
            C, H, W = feats.C, feats.H, feats.W
            pp_f, pre_f, cur_f = feats.images(0, B), feats.images(B, 2 * B), feats.images(2 * B, 3 * B)
            if i <= 2:
                n = K * (1 + 2 * (self.ks[i] // 2)) ** 2
                M = B * n
                fid_pp, _ = ops.window_index(idx_pp, float(self.scales[i]), self.ks[i], H, W,
                                             self.buf("fidpp%d" % i, (2, B, n), torch.int64),
                                             self.buf("winpp%d" % i, (B, n), torch.uint8))
                fid, win = ops.window_index(idx_p, float(self.scales[i]), self.ks[i], H, W,
                                            self.buf("fid%d" % i, (2, B, n), torch.int64),
                                            self.buf("win%d" % i, (B, n), torch.uint8))
                cat_tok = self.buf("cat_tok%d" % i, (M, 3 * C))
                key_pp, key_p, qry = (self.buf("%s%d" % (t, i), (M, C)) for t in ("keypp", "keyp", "qry"))
                self.kops.window_gather(pp_f, fid_pp[0], key_pp)
                self.kops.window_gather(pre_f, fid[0], key_p)
                self.kops.window_gather(cur_f, fid[1], qry)
                self.kops.window_gather(cur_f, fid[1], cat_tok, col0=2 * C)
                self._encoder(self.tf_pp[i], "pp", i, key_pp, qry, cat_tok[:, :C], B, n)
                self._encoder(self.tf[i], "p", i, key_p, qry, cat_tok[:, C:2 * C], B, n)
                Cc = self.cat[i]
                h1 = ops.linear(cat_tok, Cc["w0"], Cc["b0"], self.buf("cat_h%d" % i, (M, 6 * C)), relu=True)
                z = ops.linear(h1, Cc["w2"], Cc["b2"], self.buf("cat_z%d" % i, (M, C)))
                self.kops.window_scatter(z, fid[1], win, cur_f)
            else:
                t = self.mbuf("cat_t%d" % i, B, H, W, 6 * C)
                h = self.mbuf("cat_hm%d" % i, B, H, W, 6 * C)
                self.kops.conv(cur_f, CP["cat%d.cur" % i], t, ACT_NONE)
                self.kops.conv(pre_f, CP["cat%d.pre" % i], h, ACT_RELU, res=t)
                self.kops.conv(h, CP["cat%d.out" % i], cur_f, ACT_NONE)

    def forward(self, x, ppre_img, pre_img, ppre_hm, pre_hm, repro_hm, ppre_hm_cls, pre_hm_cls, repro_hm_cls,
                hm_act=ACT_NONE, static=False):
        args = [t.contiguous().float() for t in (x, ppre_img, pre_img, ppre_hm, pre_hm, repro_hm, ppre_hm_cls,
                                                 pre_hm_cls, repro_hm_cls)]
        x, ppre_img, pre_img, ppre_hm, pre_hm, repro_hm, ppre_hm_cls, pre_hm_cls, repro_hm_cls = args
        for t in args:
            if t.device != self.dev:
                raise _lib.SgtaHipError("input on %s, model on %s" % (t.device, self.dev))
        B, _, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError("network input must be a multiple of 32 in both dimensions, got %dx%d" % (H, W))
        for t in (ppre_hm_cls, pre_hm_cls, repro_hm_cls):
            if tuple(t.shape) != (B, self.num_classes, H // 4, W // 4):
                raise ValueError("the class priors must be [B,%d,H/4,W/4]" % self.num_classes)
        with torch.no_grad():
            K = self.num_classes
            idx_pp = ops.prior_argmax(ppre_hm_cls, repro_hm_cls, self.buf("idxpp", (2, B, K, 2)))
            idx_p = ops.prior_argmax(pre_hm_cls, repro_hm_cls, self.buf("idx", (2, B, K, 2)))
            lv = self.backbone_frames(((ppre_img, ppre_hm), (pre_img, pre_hm), (x, repro_hm)))
            main = torch.cuda.current_stream()
            side = self._side_stream() if (self.use_side_stream and _lib.PROFILE is None and self.compute_dead_levels) else None
            if side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self.temporal_three(lv, idx_pp, idx_p, B, (0, 1))
                self.temporal_three(lv, idx_pp, idx_p, B, (2, 3, 4, 5))
            else:
                self.temporal_three(lv, idx_pp, idx_p, B, range(6) if self.compute_dead_levels else (2, 3, 4, 5))
            cur = [v.images(2 * B, 3 * B) for v in lv]
            out = self.neck_heads(cur, hm_act, static)
            if side is not None:
                main.wait_stream(side)
            return out


class EngineThreeF16(EngineThreeBF16):
    """The same engine on IEEE-half feature maps and weights (engine_bf16.EngineF16)."""
    dtype = "f16"
    act_dtype = torch.float16

"""HIP pipeline of the three-frame variant ``DLA_PlanAWindow_Three`` (reference dla.py:1400-1456): the same kernels as engine.Engine, three frames per sample.

* backbone: ONE batch of 3B images (pre-previous, previous, current);
* levels 0-2: the current-frame window tokens attend to the windows of both older frames (two encoder passes, the
  keys / values of each projected once), ``cat[ppre_out, pre_out, cur]`` (3C) -> cat-MLP -> ordered scatter;
* levels 3-5: cat_layer on EVERY pixel of ``cat[pre, pre, cur]`` (the reference assigns ``ppre_f = pre_feats``,
  dla.py:1435 -- kept): three 1x1 convolutions on the NCHW maps, the two identical ``pre`` slices folded into one
  weight block (W[:, :C] + W[:, C:2C]; one rounding away from the reference's single GEMM);
* neck and heads as in the two-frame model ('wh' head included, base_model.py:213-262).
"""
import torch

from . import _lib, ops
from .engine import D_FFN, N_HEADS, Engine
from .ops import ACT_NONE, ACT_RELU, PackedConv, View


class EngineThree(Engine):
    def _pack(self, m):
        super()._pack(m)
        self.tf = [self._pack_encoder(m.prev_transformer[i]) for i in range(3)]
        self.tf_pp = self.tf if m.shared_ca else [self._pack_encoder(m.pprev_transformer[i]) for i in range(3)]
        for i in (3, 4, 5):                                  # full-map cat-MLP as 1x1 convs
            s = m.cat_layer[i]
            C = s[2].out_features
            w0 = s[0].weight.detach().float()
            self.P["cat%d.pre" % i] = PackedConv((w0[:, :C] + w0[:, C:2 * C]).reshape(-1, C, 1, 1).contiguous(), None)
            self.P["cat%d.cur" % i] = PackedConv(w0[:, 2 * C:].reshape(-1, C, 1, 1).contiguous(), s[0].bias)
            self.P["cat%d.out" % i] = PackedConv(s[2].weight.detach().float().reshape(C, -1, 1, 1).contiguous(), s[2].bias)

    # ------------------------------------------------------------------------------------------
    def _encoder(self, T, tag, i, key, qry, out, B, n):
        """Three iterations of ONE shared layer (dla.py:788-803) with keys / values of one older frame."""
        M, C, hid = B * n, T["C"], T["hid"]
        kv = ops.linear(key, T["wkv"], None, self.buf("kv%s%d" % (tag, i), (M, 2 * hid)))
        q = self.buf("q%s%d" % (tag, i), (M, hid))
        att = self.buf("att%s%d" % (tag, i), (M, hid))
        y1 = self.buf("y1%s%d" % (tag, i), (M, C))
        nxt = [self.buf("qa%s%d" % (tag, i), (M, C)), self.buf("qb%s%d" % (tag, i), (M, C))]
        cur = qry
        for it in range(3):
            ops.linear(cur, T["wq"], None, q)
            ops.mhca(q, kv[:, :hid], kv[:, hid:], T["pos"], att, B, n, N_HEADS, hid // N_HEADS)
            ops.linear(att, T["fc_w"], T["fc_b"], y1, res=cur, ln=T["ln1"])
            dst = out if it == 2 else nxt[it]
            if ops.ffn_ln(y1, T["w1"], T["b1"], T["w2"], T["b2"], T["ln3"], dst) is None:
                hdn = self.buf("ffn%s%d" % (tag, i), (M, D_FFN))   # two-launch fall-back only
                ops.linear(y1, T["w1"], T["b1"], hdn, relu=True)
                ops.linear(hdn, T["w2"], T["b2"], dst, res=y1, ln=T["ln3"])
            cur = dst

    def temporal_three(self, lv, idx_pp, idx_p, B, levels=range(6)):
        K = self.num_classes
        for i in levels:
            feats = lv[i]
            C, H, W = feats.C, feats.H, feats.W
            pp_f, pre_f, cur_f = View(feats.base[:B]), View(feats.base[B:2 * B]), View(feats.base[2 * B:])
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
                ops.window_gather(pp_f, fid_pp[0], key_pp)
                ops.window_gather(pre_f, fid[0], key_p)
                ops.window_gather(cur_f, fid[1], qry)
                ops.window_gather(cur_f, fid[1], cat_tok, col0=2 * C)
                self._encoder(self.tf_pp[i], "pp", i, key_pp, qry, cat_tok[:, :C], B, n)
                self._encoder(self.tf[i], "p", i, key_p, qry, cat_tok[:, C:2 * C], B, n)
                Cc = self.cat[i]
                h1 = ops.linear(cat_tok, Cc["w0"], Cc["b0"], self.buf("cat_h%d" % i, (M, 6 * C)), relu=True)
                z = ops.linear(h1, Cc["w2"], Cc["b2"], self.buf("cat_z%d" % i, (M, C)))
                ops.window_scatter(z, fid[1], win, cur_f)
            else:
                t = self.vbuf("cat_t%d" % i, B, 6 * C, H, W)
                h = self.vbuf("cat_hm%d" % i, B, 6 * C, H, W)
                ops.conv2d(cur_f, self.P["cat%d.cur" % i], t, ACT_NONE)
                ops.conv2d(pre_f, self.P["cat%d.pre" % i], h, ACT_RELU, res=t, res_mode=1)
                ops.conv2d(h, self.P["cat%d.out" % i], cur_f, ACT_NONE)

    # ------------------------------------------------------------------------------------------
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
            # levels 0 / 1 never reach the heads (dla_up starts at level 2) and their stage is a chain of small
            # latency-bound kernels: it runs on a side stream beside the matrix-bound rest of the step (engine.py)
            main = torch.cuda.current_stream()
            side = self._side_stream() if (self.use_side_stream and _lib.PROFILE is None and self.compute_dead_levels) else None
            if side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self.temporal_three(lv, idx_pp, idx_p, B, (0, 1))
                self.temporal_three(lv, idx_pp, idx_p, B, (2, 3, 4, 5))
            else:
                self.temporal_three(lv, idx_pp, idx_p, B, range(6) if self.compute_dead_levels else (2, 3, 4, 5))
            cur = [View(v.base[2 * B:]) for v in lv]
            out = self.neck_heads(cur, hm_act, static)
            if side is not None:
                main.wait_stream(side)             # the step is complete only when the side work is
            return out

"""bf16 data path of the T-frame temporal window (``DLAPlanAWindowSeq``; BASELINE configs[4]): engine_seq.EngineSeq's
structure on engine_bf16.EngineBF16's kernels.

* backbone (fused stems + level0, pipelined bf16 convs): ONE batch of T*B images, oldest frame first;
* levels 0-2: fp32 token kernels exactly as in the fp32 engine (one encoder stream per previous frame); the gather converts
  the bf16 blocked maps to fp32 tokens, the ordered scatter converts back;
* levels 3-5: cat_layer on every pixel as a chain of bf16 1x1 convolutions that accumulate the first linear layer frame by
  frame (the running sum is a bf16 map: one extra rounding per previous frame, inside the bf16 tolerance);
* neck (DCN) and the fused head kernel ('wh' included: 13 output rows).
"""
import torch

from . import _lib, ops
from .engine_bf16 import EngineBF16
from .engine_three import EngineThree
from .ops import ACT_NONE, ACT_RELU
from .ops_bf16 import PackedConvBF16


class EngineSeqBF16(EngineBF16):
    _encoder = EngineThree._encoder                     # fp32 token kernels: three iterations of one shared layer per frame

    def _pack(self, m):
        super()._pack(m)
        self._pack_seq(m, PackedConvBF16, self.P)

    def _pack_seq(self, m, pack_conv, dest):
        """Token-encoder weights per previous frame; the full-map cat-MLP of levels 3-5 as 1x1 convolutions (``pack_conv``
        objects stored in ``dest``), one weight block per frame."""
        self.T = int(m.n_frames)
        self.legacy_slip = bool(m.legacy_slip)
        n_prev = self.T - 1
        self.tf_k = [[self._pack_encoder(m.frame_transformer[k][i]) for i in range(3)] for k in range(n_prev)]
        self.tf = self.tf_k[-1]
        for i in (3, 4, 5):
            s = m.cat_layer[i]
            C = s[2].out_features
            w0 = s[0].weight.detach().float()
            blk = [w0[:, k * C:(k + 1) * C] for k in range(self.T)]
            if self.legacy_slip:                         # cat[pre, pre, cur]: both blocks act on the newest previous frame
                blk = [None, blk[0] + blk[1], blk[2]]
            for k in range(n_prev):
                if blk[k] is not None:
                    dest["cat%d.f%d" % (i, k)] = pack_conv(blk[k].reshape(-1, C, 1, 1).contiguous(), None)
            dest["cat%d.cur" % i] = pack_conv(blk[-1].reshape(-1, C, 1, 1).contiguous(), s[0].bias)
            dest["cat%d.out" % i] = pack_conv(s[2].weight.detach().float().reshape(C, -1, 1, 1).contiguous(), s[2].bias)

    # ------------------------------------------------------------------------------------------
    def temporal_seq(self, lv, idxs, B, levels=range(6)):
        K, T = self.num_classes, self.T
        CP = self.conv_packs
        n_prev = T - 1
        for i in levels:
            feats = lv[i]
            C, H, W = feats.C, feats.H, feats.W
            frame = [feats.images(k * B, (k + 1) * B) for k in range(T)]
            cur_f = frame[-1]
            if i <= 2:
                n = K * (1 + 2 * (self.ks[i] // 2)) ** 2
                M = B * n
                cat_tok = self.buf("cat_tok%d" % i, (M, T * C))
                qry = self.buf("qry%d" % i, (M, C))
                fid = win = None
                for k in range(n_prev):
                    fid, win = ops.window_index(idxs[k], float(self.scales[i]), self.ks[i], H, W,
                                                self.buf("fid%d_%d" % (k, i), (2, B, n), torch.int64),
                                                self.buf("win%d_%d" % (k, i), (B, n), torch.uint8))
                    if k == 0:
                        self.kops.window_gather(cur_f, fid[1], qry)
                        self.kops.window_gather(cur_f, fid[1], cat_tok, col0=n_prev * C)
                    key = self.buf("key%d_%d" % (k, i), (M, C))
                    self.kops.window_gather(frame[k], fid[0], key)
                    self._encoder(self.tf_k[k][i], "f%d_" % k, i, key, qry, cat_tok[:, k * C:(k + 1) * C], B, n)
                Cc = self.cat[i]
                h1 = ops.linear(cat_tok, Cc["w0"], Cc["b0"], self.buf("cat_h%d" % i, (M, 2 * T * C)), relu=True)
                z = ops.linear(h1, Cc["w2"], Cc["b2"], self.buf("cat_z%d" % i, (M, C)))
                self.kops.window_scatter(z, fid[1], win, cur_f)
            else:
                ta = self.mbuf("cat_ta%d" % i, B, H, W, 2 * T * C)
                tb = self.mbuf("cat_tb%d" % i, B, H, W, 2 * T * C)
                self.kops.conv(cur_f, CP["cat%d.cur" % i], ta, ACT_NONE)
                ks_ = [k for k in range(n_prev) if ("cat%d.f%d" % (i, k)) in CP]
                for j, k in enumerate(ks_):
                    last = j == len(ks_) - 1
                    self.kops.conv(frame[k], CP["cat%d.f%d" % (i, k)], tb, ACT_RELU if last else ACT_NONE, res=ta)
                    ta, tb = tb, ta
                self.kops.conv(ta, CP["cat%d.out" % i], cur_f, ACT_NONE)

    # ------------------------------------------------------------------------------------------
    def forward(self, x, pre_imgs, pre_hms, repro_hm, pre_hm_clss, repro_hm_cls, hm_act=ACT_NONE, static=False):
        f = lambda t: t.contiguous().float()                 # noqa: E731
        x, repro_hm, repro_hm_cls = f(x), f(repro_hm), f(repro_hm_cls)
        pre_imgs, pre_hms, pre_hm_clss = [f(t) for t in pre_imgs], [f(t) for t in pre_hms], [f(t) for t in pre_hm_clss]
        n_prev = self.T - 1
        if not (len(pre_imgs) == len(pre_hms) == len(pre_hm_clss) == n_prev):
            raise ValueError("expected %d previous frames" % n_prev)
        for t in [x, repro_hm, repro_hm_cls] + pre_imgs + pre_hms + pre_hm_clss:
            if t.device != self.dev:
                raise _lib.SgtaHipError("input on %s, model on %s" % (t.device, self.dev))
        B, _, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError("network input must be a multiple of 32 in both dimensions, got %dx%d" % (H, W))
        for t in pre_hm_clss + [repro_hm_cls]:
            if tuple(t.shape) != (B, self.num_classes, H // 4, W // 4):
                raise ValueError("the class priors must be [B,%d,H/4,W/4]" % self.num_classes)
        with torch.no_grad():
            K = self.num_classes
            idxs = [ops.prior_argmax(pre_hm_clss[k], repro_hm_cls, self.buf("idx%d" % k, (2, B, K, 2))) for k in range(n_prev)]
            frames = tuple((pre_imgs[k], pre_hms[k]) for k in range(n_prev)) + ((x, repro_hm),)
            lv = self.backbone_frames(frames)
            main = torch.cuda.current_stream()
            side = self._side_stream() if (self.use_side_stream and _lib.PROFILE is None and self.compute_dead_levels) else None
            if side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self.temporal_seq(lv, idxs, B, (0, 1))
                self.temporal_seq(lv, idxs, B, (2, 3, 4, 5))
            else:
                self.temporal_seq(lv, idxs, B, range(6) if self.compute_dead_levels else (2, 3, 4, 5))
            cur = [v.images(n_prev * B, self.T * B) for v in lv]
            out = self.neck_heads(cur, hm_act, static)
            if side is not None:
                main.wait_stream(side)
            return out


class EngineSeqF16(EngineSeqBF16):
    """The same engine on IEEE-half feature maps and weights (engine_bf16.EngineF16)."""
    dtype = "f16"
    act_dtype = torch.float16

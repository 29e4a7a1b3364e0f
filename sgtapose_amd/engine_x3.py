"""``DLAPlanAWindowL3New.forward`` at fp32 ACCURACY with the FLOP-carrying convolutions on the 16-bit matrix pipe
(``model.engine_dtype = "f32x3"``; same call stack and reference code as engine.py).

fp32 MFMA peaks at 157 TFLOP/s; the 16-bit pipe at 2500.  An fp32 product emulated by operand splitting -- x = x_hi + x_lo,
w = w_hi + w_lo in IEEE half, x_hi w_hi + x_hi w_lo + x_lo w_hi accumulated in fp32 -- costs three 16-bit products and keeps
22 significant bits per operand: results agree with the fp32 operators to ~1e-6 relative (tests/test_x3_parity.py), far
inside the 1e-3 bar the fp32 fixtures assert.  The step on this engine:

  stems + level0             sgta_stem_level0_x3: ONE kernel, the 16-channel stem map never leaves LDS, two-plane output
  level1                     sgta_conv_x3 (stride 2) on the two-plane level0 map
  levels 2-5 of the backbone sgta_conv_x3 / sgta_maxpool2_x3 on two-plane blocked maps (the 16-bit conv kernel over 3x the chunks)
  temporal stage             tokens are fp32 as everywhere (linear / attention / FFN kernels of engine.py): sgta_window_gather_x3 /
                             scatter_x3 between the two-plane maps and the token matrices
  current-frame maps -> fp32 sgta_x3_to_nchw (exact), then engine.py's neck on fp32 NCHW maps: offset convs and up-sampling on the fp32
                             kernels, the deformable convs through sgta_dcnv2_x3 (fp32 blend, split-operand product)
  neck output -> x3          sgta_nchw_to_x3, then the three head branches in ONE split-operand kernel (sgta_heads_x3), fp32 out

It is reported under its own dtype label, never as plain f32: the arithmetic is emulated, the accuracy is fp32's.
"""
import torch

from . import _lib, ops, ops_x3 as xops
from .engine import Engine, _bn_tuple
from .ops import ACT_NONE, ACT_RELU, View
from .ops_x3 import PackedConvX3, ViewX

F16 = torch.float16


class EngineX3(Engine):
    dtype = "f32x3"
    split_levels = (2, 3, 4, 5)        # levels run as two batch slices on two streams (as engine_bf16.EngineBF16)
    split_parts = 2
    fuse_stem = True                   # sgta_stem_level0_x3 + level1 as sgta_conv_x3 instead of the fp32 stem / level0 / level1 kernels

    def _pack(self, m):
        super()._pack(m)               # fp32 packs: stems, level0 / level1, the DCN neck, token weights (levels 2-5 unused)
        X = {}
        b = m.base

        def pack_tree(node, name):
            if node.levels == 1:
                for t in ("tree1", "tree2"):
                    blk = getattr(node, t)
                    X["%s.%s.conv1" % (name, t)] = PackedConvX3(blk.conv1.weight, None, _bn_tuple(blk.bn1), stride=blk.stride)
                    X["%s.%s.conv2" % (name, t)] = PackedConvX3(blk.conv2.weight, None, _bn_tuple(blk.bn2))
                X[name + ".root"] = PackedConvX3(node.root.conv.weight, None, _bn_tuple(node.root.bn))
                if hasattr(node, "project"):
                    X[name + ".project"] = PackedConvX3(node.project[0].weight, None, _bn_tuple(node.project[1]))
            else:
                pack_tree(node.tree1, name + ".tree1")
                pack_tree(node.tree2, name + ".tree2")

        for lv in range(2, 6):
            pack_tree(getattr(b, "level%d" % lv), "level%d" % lv)
        for k in [k for k in self.P if k.startswith(("level2", "level3", "level4", "level5"))]:
            del self.P[k]              # their fp32 packs (Winograd transforms) are not used here
        try:
            X["heads.fused"] = xops.PackedHeadsX3([(getattr(m, h)[0].weight, getattr(m, h)[0].bias, getattr(m, h)[2].weight,
                                                    getattr(m, h)[2].bias) for h in self.head_names])
        except _lib.SgtaHipError:
            X["heads.fused"] = None    # unusual head configuration: the fp32 head kernels
        X["stem_l0"] = xops.PackedStemL0X3(b.pre_img_layer[0].weight, _bn_tuple(b.pre_img_layer[1]), b.pre_hm_layer[0].weight,
                                           _bn_tuple(b.pre_hm_layer[1]), b.level0[0].weight, _bn_tuple(b.level0[1]))
        X["level1"] = PackedConvX3(b.level1[0].weight, None, _bn_tuple(b.level1[1]), stride=2)
        for k, pc in self.P.items():   # the deformable convs of the neck: fp32 tensors in and out, split-operand product
            if getattr(pc, "tap_major", False):
                try:
                    X[k] = xops.PackedDCNX3(pc)
                except _lib.SgtaHipError:
                    pass               # Cin % 8 != 0: the fp32 kernel
        self.X = X

    def xbuf(self, name, n, h, w, c):
        return ViewX.of(self.buf(name, (2, n, c // 16, h, w, 16), F16))

    kops = xops                        # the hooks the multi-frame engines are written against (engine_bf16.EngineBF16)
    conv_packs = property(lambda self: self.X)
    mbuf = xbuf

    # ---- levels 2-5 on two-plane maps (the structure of Engine._tree / _tree_l1) ------------------------------------------
    def _xtree_l1(self, node, name, x, out, cat, bottom=None):
        X, c, cin, N = self.X, node.cout, node.cin, x.N
        h, w = x.H // node.stride, x.W // node.stride
        if node.stride > 1:
            if bottom is None:
                bottom = cat.slice(2 * c, cin) if node.level_root else self.xbuf("x.pool", N, h, w, cin)
                xops.maxpool2(x, bottom)
        else:
            bottom = x
        if cin != c:
            residual = self.xbuf("x.proj", N, h, w, c)
            xops.conv(bottom, X[name + ".project"], residual, ACT_NONE)
        else:
            residual = bottom
        a = self.xbuf("x.blk_a", N, h, w, c)
        x1, x2 = cat.slice(c, c), cat.slice(0, c)
        xops.conv(x, X[name + ".tree1.conv1"], a, ACT_RELU)
        xops.conv(a, X[name + ".tree1.conv2"], x1, ACT_RELU, res=residual)
        xops.conv(x1, X[name + ".tree2.conv1"], a, ACT_RELU)
        xops.conv(a, X[name + ".tree2.conv2"], x2, ACT_RELU, res=x1)
        xops.conv(cat, X[name + ".root"], out, ACT_RELU)
        return out

    def _xtree(self, node, name, x, out):
        N, c, cin = x.N, node.cout, node.cin
        h, w = x.H // node.stride, x.W // node.stride
        if node.levels == 1:
            cat = self.xbuf("x." + name + ".cat", N, h, w, node.root_dim)
            return self._xtree_l1(node, name, x, out, cat)
        inner = node.tree2
        cat2 = self.xbuf("x." + name + ".cat2", N, h, w, inner.root_dim)
        bottom = cat2.slice(2 * c, cin)
        xops.maxpool2(x, bottom)
        t1 = cat2.slice(2 * c + cin, c)
        cat1 = self.xbuf("x." + name + ".cat1", N, h, w, node.tree1.root_dim)
        self._xtree_l1(node.tree1, name + ".tree1", x, t1, cat1, bottom=bottom)
        return self._xtree_l1(inner, name + ".tree2", t1, out, cat2)

    def backbone_frames(self, frames, after_level=None):
        B, _, H, W = frames[-1][0].shape
        N = len(frames) * B
        P = self.P
        # the kernels address a two-plane map with 31-bit byte offsets: the largest one is level 0 (N x 16 channels x 2 planes)
        if N * H * W * 16 * 2 * 2 >= 2 ** 31:
            raise ValueError("f32x3 engine: %d frames of %dx%d per step exceed the 2 GiB per-tensor addressing of its kernels "
                             "(max %d frames at this size): split the batch" % (N, H, W, (2 ** 31 - 1) // (64 * H * W)))
        if self.fuse_stem and W % 4 == 0 and all(t.data_ptr() % 16 == 0 for f in frames for t in f):
            # both stems, their sum and level0 in one split-operand kernel; level1 as a split-operand stride-2 conv
            lv = [self.xbuf("x.l0", N, H, W, 16)]
            for half, (img, hm) in enumerate(frames):
                xops.stem_level0(img.contiguous(), hm.contiguous(), self.X["stem_l0"], lv[0].images(half * B, (half + 1) * B))
            if after_level:
                after_level(0, lv[0])
            lv.append(self.xbuf("x.l1", N, H // 2, W // 2, 32))
            xops.conv(lv[0], self.X["level1"], lv[1], ACT_RELU)
            if after_level:
                after_level(1, lv[1])
            prev = lv[1]
        else:
            # rows that are not 16-byte aligned: stems, level 0 and level 1 on the fp32 kernels, handed over as two-plane maps
            s = self.buf("stem", (N, 16, H, W))
            for half, (img, hm) in enumerate(frames):
                ops.conv_c16(img, hm, P["stem.c16"], View(s[half * B:(half + 1) * B]))
            l0 = self.vbuf("l0", N, 16, H, W)
            ops.conv2d(View(s), P["level0"], l0, ACT_RELU)
            l1 = self.vbuf("l1", N, 32, H // 2, W // 2)
            ops.conv2d(l0, P["level1"], l1, ACT_RELU)
            lv = [xops.from_nchw(l0, self.xbuf("x.l0", N, H, W, 16)), xops.from_nchw(l1, self.xbuf("x.l1", N, H // 2, W // 2, 32))]
            if after_level:
                after_level(0, lv[0])
                after_level(1, lv[1])
            prev = lv[1]
        ch = self.model.base.channels
        main = torch.cuda.current_stream()
        parts = self.split_parts if (self.use_side_stream and _lib.PROFILE is None and self.split_levels and N % self.split_parts == 0) else 1
        for i in range(2, 6):
            out = self.xbuf("x.l%d" % i, N, H >> i, W >> i, ch[i])
            node, name = getattr(self.model.base, "level%d" % i), "level%d" % i
            if parts > 1 and i in self.split_levels:
                hN = N // parts
                streams = self._part_streams(parts - 1)
                for k, st in enumerate(streams):
                    st.wait_stream(main)
                    with torch.cuda.stream(st):
                        self.buf_tag = ".p%d" % (k + 1)
                        try:
                            self._xtree(node, name, prev.images((k + 1) * hN, (k + 2) * hN), out.images((k + 1) * hN, (k + 2) * hN))
                        finally:
                            self.buf_tag = ""
                self._xtree(node, name, prev.images(0, hN), out.images(0, hN))
                for st in streams:
                    main.wait_stream(st)
            else:
                self._xtree(node, name, prev, out)
            lv.append(out)
            prev = out
            if after_level:
                after_level(i, out)
        return lv

    # ---- temporal stage: fp32 tokens; the maps of levels 2-5 are two-plane ---------------------------------------------------
    def temporal_tokens(self, feats, i, idx, B):
        if not isinstance(feats, ViewX):
            return super().temporal_tokens(feats, i, idx, B)
        K = self.num_classes
        C, H, W = feats.C, feats.H, feats.W
        pre_f, cur_f = feats.images(0, B), feats.images(B, 2 * B)
        n = K * (1 + 2 * (self.ks[i] // 2)) ** 2
        fid, win = ops.window_index(idx, float(self.scales[i]), self.ks[i], H, W,
                                    self.buf("fid%d" % i, (2, B, n), torch.int64),
                                    self.buf("win%d" % i, (B, n), torch.uint8))
        M = B * n
        cat_tok = self.buf("cat_tok%d" % i, (M, 2 * C))
        xops.window_gather(cur_f, fid[1], cat_tok, col0=C)
        if i <= 2:
            T = self.tf[i]
            hid = T["hid"]
            key = self.buf("key%d" % i, (M, C))
            cur = self.buf("qry%d" % i, (M, C))
            xops.window_gather(pre_f, fid[0], key)
            xops.window_gather(cur_f, fid[1], cur)
            kv = ops.linear(key, T["wkv"], None, self.buf("kv%d" % i, (M, 2 * hid)))
            q = self.buf("q%d" % i, (M, hid))
            att = self.buf("att%d" % i, (M, hid))
            y1 = self.buf("y1_%d" % i, (M, C))
            nxt = self.buf("qry_b%d" % i, (M, C))
            for it in range(3):
                ops.linear(cur, T["wq"], None, q)
                ops.mhca(q, kv[:, :hid], kv[:, hid:], T["pos"], att, B, n, 8, hid // 8)
                ops.linear(att, T["fc_w"], T["fc_b"], y1, res=cur, ln=T["ln1"])
                dst = cat_tok[:, :C] if it == 2 else nxt
                if ops.ffn_ln(y1, T["w1"], T["b1"], T["w2"], T["b2"], T["ln3"], dst) is None:
                    hdn = self.buf("ffn%d" % i, (M, 1024))
                    ops.linear(y1, T["w1"], T["b1"], hdn, relu=True)
                    ops.linear(hdn, T["w2"], T["b2"], dst, res=y1, ln=T["ln3"])
                cur, nxt = dst, cur
        else:
            xops.window_gather(pre_f, fid[0], cat_tok, col0=0)
        Cc = self.cat[i]
        h1 = ops.linear(cat_tok, Cc["w0"], Cc["b0"], self.buf("cat_h%d" % i, (M, 4 * C)), relu=True)
        z = ops.linear(h1, Cc["w2"], Cc["b2"], self.buf("cat_z%d" % i, (M, C)))
        return z, fid, win

    def temporal_scatter(self, feats, B, tok):
        if not isinstance(feats, ViewX):
            return super().temporal_scatter(feats, B, tok)
        z, fid, win = tok
        xops.window_scatter(z, fid[1], win, feats.images(B, 2 * B))

    @staticmethod
    def _frame(v, k, B):
        return v.images(k * B, (k + 1) * B) if isinstance(v, ViewX) else View(v.base[k * B:(k + 1) * B])

    def _deform(self, name, x, out, om_name="om"):
        pk = self.X.get(name)
        if pk is None or not pk.usable(x):
            return super()._deform(name, x, out, om_name)
        om = self.vbuf(om_name, x.B, 27, x.H, x.W)
        ops.conv2d(x, self.P[name + ".om"], om, ACT_NONE)
        return xops.dcn_v2(x, om, pk, out, ACT_RELU)

    # ---- neck on fp32 NCHW maps (offset convs fp32, deformable convs split-operand), heads on the split-operand kernel ---------------------------------------------------------
    neck_parts = 1                     # > 1: neck + heads as that many batch slices on as many streams

    def neck_heads(self, cur, hm_act=ACT_NONE, static=False):
        first = self.model.dla_up.startp                                               # the neck reads levels first .. 5 only
        pk = self.X.get("heads.fused")

        def neck_slice(lo, hi):
            cur32 = []
            for i, v in enumerate(cur):
                if i >= first:
                    v = self._images(v, lo, hi)
                    if isinstance(v, ViewX):
                        v = xops.to_nchw(v, self.vbuf("l%d.f32" % i, v.N, v.C, v.H, v.W))   # exact
                cur32.append(v)
            return self.neck(cur32)

        B = cur[-1].N if isinstance(cur[-1], ViewX) else cur[-1].B
        parts = self.neck_parts if (self.use_side_stream and _lib.PROFILE is None and pk is not None) else 1
        if parts <= 1 or B % parts:
            feat = neck_slice(0, B)
            self.last_feat = feat
            if pk is None:
                return self.heads(feat, hm_act, static)
        h, w = cur[first].H, cur[first].W
        outs = [self.buf("out_" + n, (B, r, h, w)) if static else torch.empty((B, r, h, w), dtype=torch.float32, device=self.dev)
                for n, r in zip(self.head_names, pk.rows)]
        nsig = pk.rows[0] if (self.head_names[0] == "hm" and hm_act == ops.ACT_SIGMOID) else 0

        def heads_slice(feat, lo, hi):
            fx = xops.from_nchw(feat, self.xbuf("x.feat", feat.B, h, w, feat.C))
            xops.heads_fused(fx, pk, [o[lo:hi] for o in outs], nsig)

        if parts <= 1 or B % parts:
            heads_slice(feat, 0, B)
            return dict(zip(self.head_names, outs))
        main, hb = torch.cuda.current_stream(), B // parts
        streams = self._part_streams(parts - 1)
        saved, self.use_side_stream = self.use_side_stream, False       # a slice's neck runs serially on its own stream
        try:
            for k, st in enumerate(streams):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    self.buf_tag = ".n%d" % (k + 1)
                    try:
                        heads_slice(neck_slice((k + 1) * hb, (k + 2) * hb), (k + 1) * hb, (k + 2) * hb)
                    finally:
                        self.buf_tag = ""
            self.last_feat = neck_slice(0, hb)
            heads_slice(self.last_feat, 0, hb)
        finally:
            self.use_side_stream = saved
        for st in streams:
            main.wait_stream(st)
        return dict(zip(self.head_names, outs))

    @staticmethod
    def _images(v, lo, hi):
        return v.images(lo, hi) if isinstance(v, ViewX) else View(v.base[lo:hi], v.c0, v.C)


class EngineThreeX3(EngineX3):
    """``DLA_PlanAWindow_Three`` (reference dla.py:1400-1456) on the split-operand kernels: engine_three_bf16.EngineThreeBF16's
    structure (fp32 token kernels, the level 3-5 cat-MLP as 1x1 convolutions, 'wh' head) with two-plane maps."""
    from .engine_three import EngineThree as _E3
    from .engine_three_bf16 import EngineThreeBF16 as _B3
    _encoder = _E3._encoder
    temporal_three = _B3.temporal_three
    forward = _B3.forward

    def _pack(self, m):
        super()._pack(m)
        self._B3._pack_three(self, m, PackedConvX3, self.X)


class EngineSeqX3(EngineX3):
    """The T-frame temporal window (``DLAPlanAWindowSeq``; BASELINE configs[4]) on the split-operand kernels:
    engine_seq_bf16.EngineSeqBF16's structure with two-plane maps (the running sum of the cat-MLP's first layer keeps 22 bits)."""
    from .engine_three import EngineThree as _E3
    from .engine_seq_bf16 import EngineSeqBF16 as _BS
    _encoder = _E3._encoder
    temporal_seq = _BS.temporal_seq
    forward = _BS.forward

    def _pack(self, m):
        super()._pack(m)
        self._BS._pack_seq(self, m, PackedConvX3, self.X)

"""16-bit data path of ``DLAPlanAWindowL3New.forward`` (BASELINE configs[2]; same call stack as engine.py): bfloat16
(``EngineBF16``) or IEEE half (``EngineF16``) feature maps and weights.

Feature maps are 16-bit BLOCKED CHANNELS-LAST [N,C/16,H,W,16] in HBM (half the bytes of the fp32 path; the MFMA operand of a
pixel and a deformable-sampling corner are contiguous 16-byte runs, and a 16-channel chunk of a window row is one
contiguous run of memory for the LDS-DMA); weights are 16-bit; every product runs on v_mfma_f32_32x32x16_{bf16,f16} with
fp32 accumulation (direct-form convs: at this matrix rate the path is memory-bound, so Winograd is neither needed nor
numerically wise).  What is rounded where (the numerics contract of this path):

* the boundary inputs (fp32 NCHW images and prior maps) are rounded ONCE to the 16-bit format inside the fused stem kernel
  (``fuse_stem``, the default: both 7x7 stems, their sum and level0 on the 16-bit MFMA, the stem map rounded in LDS); with
  ``fuse_stem = False`` the stems run on the f32 matrix pipe from the fp32 inputs and only their sum is rounded;
* every feature map between kernels is 16-bit; residual adds, bias, ReLU happen in fp32 on the accumulator before the store;
* the DCN offset / mask maps and the sampling coordinates / weights are fp32; the bilinear blend itself runs in PACKED
  fp16 (the staged window is converted to fp16 -- exact for bf16 values in range --, corner weights rounded to fp16, three
  fp16 fma roundings) and the main product on the fp16 MFMA with fp16 weights, in both engines;
* the whole windowed attention stage (a few hundred token rows: QKV / softmax / LayerNorm / FFN / cat-MLP) is fp32: tokens are
  converted at the gather / scatter; the head outputs and the decode are fp32.
The API boundary (inputs, output dict, decode dictionary) is unchanged.
"""
import torch

from . import _lib, ops, ops_bf16 as bops
from .engine import Engine, _bn_tuple
from .ops import ACT_NONE, ACT_RELU
from .ops_bf16 import BF16, PackedConvBF16, PackedDCNBF16, ViewH


class EngineBF16(Engine):
    dtype = "bf16"
    act_dtype = BF16          # element format of the feature maps and weights (EngineF16: IEEE half)

    def __init__(self, model):
        with bops.pack_dtype(self.act_dtype):
            super().__init__(model)
    fuse_heads = True         # sgta_heads_bf16: the 3x3 -> ReLU -> 1x1 branches in one kernel, no hidden tensor in memory
    kops = bops               # kernel faces of this data path (the multi-frame engines are written against these three hooks)
    conv_packs = property(lambda self: self.P)

    def mbuf(self, name, n, h, w, c):
        return self.hbuf(name, n, h, w, c)

    fuse_stem = True          # sgta_stem_level0_bf16 (bf16 MFMA, stem map in LDS) instead of the f32 stem kernel + level0 conv

    # ------------------------------------------------------------------------------------------
    def _pack(self, m):
        P = {}
        b = m.base
        P["stem.c16"] = ops.PackedC16(b.pre_img_layer[0].weight, _bn_tuple(b.pre_img_layer[1]),
                                       b.pre_hm_layer[0].weight, _bn_tuple(b.pre_hm_layer[1]))
        P["level0"] = PackedConvBF16(b.level0[0].weight, None, _bn_tuple(b.level0[1]))
        P["stem_l0"] = bops.PackedStemL0BF16(b.pre_img_layer[0].weight, _bn_tuple(b.pre_img_layer[1]),
                                             b.pre_hm_layer[0].weight, _bn_tuple(b.pre_hm_layer[1]),
                                             b.level0[0].weight, _bn_tuple(b.level0[1]))
        P["level1"] = PackedConvBF16(b.level1[0].weight, None, _bn_tuple(b.level1[1]), stride=2)

        def pack_tree(node, name):
            if node.levels == 1:
                for t in ("tree1", "tree2"):
                    blk = getattr(node, t)
                    P["%s.%s.conv1" % (name, t)] = PackedConvBF16(blk.conv1.weight, None, _bn_tuple(blk.bn1), stride=blk.stride)
                    P["%s.%s.conv2" % (name, t)] = PackedConvBF16(blk.conv2.weight, None, _bn_tuple(blk.bn2))
                P[name + ".root"] = PackedConvBF16(node.root.conv.weight, None, _bn_tuple(node.root.bn))
                if hasattr(node, "project"):
                    P[name + ".project"] = PackedConvBF16(node.project[0].weight, None, _bn_tuple(node.project[1]))
            else:
                pack_tree(node.tree1, name + ".tree1")
                pack_tree(node.tree2, name + ".tree2")

        for lv in range(2, 6):
            pack_tree(getattr(b, "level%d" % lv), "level%d" % lv)

        def pack_ida(ida, name):
            for j in range(1, len(ida.channels)):
                for kind in ("proj", "node"):
                    dc = getattr(ida, "%s_%d" % (kind, j))
                    com = dc.conv.conv_offset_mask
                    P["%s.%s_%d.om" % (name, kind, j)] = PackedConvBF16(com.weight, com.bias, cout_pad_to=32)
                    P["%s.%s_%d" % (name, kind, j)] = PackedDCNBF16(dc.conv.weight, dc.conv.bias, _bn_tuple(dc.actf[0]))
                P["%s.up_%d" % (name, j)] = bops.pack_up_weight(getattr(ida, "up_%d" % j).weight)

        for i in range(3):
            pack_ida(getattr(m.dla_up, "ida_%d" % i), "dla_up.ida_%d" % i)
        pack_ida(m.ida_up, "ida_up")

        f = lambda t: t.detach().float().contiguous()          # noqa: E731
        self.tf = [self._pack_encoder(m.transformer[i]) for i in range(3)] if hasattr(m, "transformer") else []
        self.cat = [dict(w0=f(s[0].weight), b0=f(s[0].bias), w2=f(s[2].weight), b2=f(s[2].bias)) for s in m.cat_layer]

        self.head_names = [h for h in m.heads if hasattr(m, h)]
        hw = torch.cat([getattr(m, h)[0].weight.detach().float() for h in self.head_names], 0)
        hb = torch.cat([getattr(m, h)[0].bias.detach().float() for h in self.head_names], 0)
        P["heads.0"] = PackedConvBF16(hw, hb)
        self.head_hidden = [getattr(m, h)[0].out_channels for h in self.head_names]
        for h in self.head_names:
            P["heads.2." + h] = PackedConvBF16(getattr(m, h)[2].weight, getattr(m, h)[2].bias)
        try:
            P["heads.fused"] = bops.PackedHeadsBF16([(getattr(m, h)[0].weight, getattr(m, h)[0].bias, getattr(m, h)[2].weight,
                                                      getattr(m, h)[2].bias) for h in self.head_names])
        except _lib.SgtaHipError:
            P["heads.fused"] = None               # unusual head configuration: 3x3 conv + three 1x1 kernels
        self.P = P

    split_levels = (2, 3, 4, 5)   # backbone levels run as ``split_parts`` slices of the batch on as many streams: a level is a
    split_parts = 2               # chain of launches with 1.2-round tails and latency-bound prologues; independent slices
    #                               side by side fill those gaps (measured: -3 % on the step)

    def forward(self, x, *args, **kw):
        # the kernels address a tensor with 31-bit byte offsets: the largest map (level0, 2B images x 16 channels) bounds
        # the batch of one launch
        B, _, H, W = x.shape
        if 2 * B * H * W * 16 * 2 >= 2 ** 31:
            raise ValueError("16-bit engine: batch %d at %dx%d exceeds the 2 GiB per-tensor addressing of its kernels "
                             "(max batch %d at this size): split the batch" % (B, H, W, (2 ** 31 - 1) // (64 * H * W)))
        return super().forward(x, *args, **kw)

    def hbuf(self, name, n, h, w, c, dtype=None):
        """16-bit maps: blocked [n, c/16, h, w, 16]; fp32 maps (DCN offsets): plain channels-last [n, h, w, c]."""
        if dtype is None:
            return ViewH(self.buf(name, (n, c // 16, h, w, 16), self.act_dtype))
        return ViewH(self.buf(name, (n, h, w, c), dtype))

    @staticmethod
    def _frame(v, k, B):
        return v.images(k * B, (k + 1) * B)

    # ------------------------------------------------------------------------------------------
    def _tree_l1(self, node, name, x, out, cat, bottom=None):
        P, c, cin, N = self.P, node.cout, node.cin, x.N
        h, w = x.H // node.stride, x.W // node.stride
        if node.stride > 1:
            if bottom is None:
                bottom = cat.slice(2 * c, cin) if node.level_root else self.hbuf("pool", N, h, w, cin)
                bops.maxpool2(x, bottom)
        else:
            bottom = x
        if cin != c:
            residual = self.hbuf("proj", N, h, w, c)
            bops.conv(bottom, P[name + ".project"], residual, ACT_NONE)
        else:
            residual = bottom
        a = self.hbuf("blk_a", N, h, w, c)
        x1, x2 = cat.slice(c, c), cat.slice(0, c)
        bops.conv(x, P[name + ".tree1.conv1"], a, ACT_RELU)
        bops.conv(a, P[name + ".tree1.conv2"], x1, ACT_RELU, res=residual)
        bops.conv(x1, P[name + ".tree2.conv1"], a, ACT_RELU)
        bops.conv(a, P[name + ".tree2.conv2"], x2, ACT_RELU, res=x1)
        bops.conv(cat, P[name + ".root"], out, ACT_RELU)
        return out

    def _tree(self, node, name, x, out, bottom=None):
        N, c, cin = x.N, node.cout, node.cin
        h, w = x.H // node.stride, x.W // node.stride
        if node.levels == 1:
            cat = self.hbuf(name + ".cat", N, h, w, node.root_dim)
            if bottom is not None and node.level_root:
                bottom = None
            return self._tree_l1(node, name, x, out, cat, bottom=bottom)
        inner = node.tree2
        cat2 = self.hbuf(name + ".cat2", N, h, w, inner.root_dim)
        bottom = cat2.slice(2 * c, cin)
        bops.maxpool2(x, bottom)
        t1 = cat2.slice(2 * c + cin, c)
        cat1 = self.hbuf(name + ".cat1", N, h, w, node.tree1.root_dim)
        self._tree_l1(node.tree1, name + ".tree1", x, t1, cat1, bottom=bottom)
        return self._tree_l1(inner, name + ".tree2", t1, out, cat2)

    def backbone_frames(self, frames, after_level=None):
        B, _, H, W = frames[-1][0].shape
        N = len(frames) * B
        P = self.P
        lv = [self.hbuf("l0", N, H, W, 16)]
        if self.fuse_stem:
            # stems + level0 in one kernel: the 16-channel stem map never leaves the chip
            for k, (img, hm) in enumerate(frames):
                bops.stem_level0(img, hm, P["stem_l0"], lv[0].images(k * B, (k + 1) * B))
        else:
            s = self.hbuf("stem", N, H, W, 16)
            for k, (img, hm) in enumerate(frames):
                bops.stem(img, hm, P["stem.c16"], s.images(k * B, (k + 1) * B))
            bops.conv(s, P["level0"], lv[0], ACT_RELU)
        if after_level:
            after_level(0, lv[0])
        lv.append(self.hbuf("l1", N, H // 2, W // 2, 32))
        bops.conv(lv[0], P["level1"], lv[1], ACT_RELU)
        if after_level:
            after_level(1, lv[1])
        ch = self.model.base.channels
        main = torch.cuda.current_stream()
        parts = self.split_parts if (self.use_side_stream and _lib.PROFILE is None and self.split_levels) else 1
        if parts > 1 and N % parts:
            parts = 2 if N % 2 == 0 else 1
        for i in range(2, 6):
            out = self.hbuf("l%d" % i, N, H >> i, W >> i, ch[i])
            node, name = getattr(self.model.base, "level%d" % i), "level%d" % i
            if parts > 1 and i in self.split_levels:
                # the batch in ``parts`` slices, slice 0 on this stream and the others on their own streams (a slice depends only
                # on its own slice of the previous level); they join after every level: the attention hooks and the next level
                # see the whole map
                h = N // parts
                streams = self._part_streams(parts - 1)
                for k, st in enumerate(streams):
                    st.wait_stream(main)
                    with torch.cuda.stream(st):
                        self.buf_tag = ".p%d" % (k + 1)
                        try:
                            self._tree(node, name, lv[i - 1].images((k + 1) * h, (k + 2) * h), out.images((k + 1) * h, (k + 2) * h))
                        finally:
                            self.buf_tag = ""
                self._tree(node, name, lv[i - 1].images(0, h), out.images(0, h))
                for st in streams:
                    main.wait_stream(st)
            else:
                self._tree(node, name, lv[i - 1], out)
            lv.append(out)
            if after_level:
                after_level(i, out)
        return lv

    # ------------------------------------------------------------------------------------------
    def temporal_tokens(self, feats, i, idx, B):
        """As Engine.temporal_tokens; tokens are fp32 (converted from the bf16 map by the gather)."""
        K = self.num_classes
        C, H, W = feats.C, feats.H, feats.W
        pre_f, cur_f = feats.images(0, B), feats.images(B, 2 * B)
        n = K * (1 + 2 * (self.ks[i] // 2)) ** 2
        fid, win = ops.window_index(idx, float(self.scales[i]), self.ks[i], H, W,
                                    self.buf("fid%d" % i, (2, B, n), torch.int64),
                                    self.buf("win%d" % i, (B, n), torch.uint8))
        M = B * n
        cat_tok = self.buf("cat_tok%d" % i, (M, 2 * C))
        bops.window_gather(cur_f, fid[1], cat_tok, col0=C)
        if i <= 2:
            T = self.tf[i]
            hid = T["hid"]
            key = self.buf("key%d" % i, (M, C))
            cur = self.buf("qry%d" % i, (M, C))
            bops.window_gather(pre_f, fid[0], key)
            bops.window_gather(cur_f, fid[1], cur)
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
            bops.window_gather(pre_f, fid[0], cat_tok, col0=0)
        Cc = self.cat[i]
        h1 = ops.linear(cat_tok, Cc["w0"], Cc["b0"], self.buf("cat_h%d" % i, (M, 4 * C)), relu=True)
        z = ops.linear(h1, Cc["w2"], Cc["b2"], self.buf("cat_z%d" % i, (M, C)))
        return z, fid, win

    def temporal_scatter(self, feats, B, tok):
        z, fid, win = tok
        bops.window_scatter(z, fid[1], win, feats.images(B, 2 * B))

    # ------------------------------------------------------------------------------------------
    def _deform(self, name, x, out, om_name="om"):
        om = self.hbuf(om_name, x.N, x.H, x.W, 32, torch.float32)
        bops.conv(x, self.P[name + ".om"], om, ACT_NONE, out_mode=bops.OUT_F32_NHWC)
        return bops.dcn_v2(x, om, self.P[name], out, ACT_RELU)

    def _ida(self, ida, name, layers, startp, endp):
        main = torch.cuda.current_stream()
        side = self._neck_stream() if (self.use_side_stream and _lib.PROFILE is None and endp - startp > 2) else None
        early = {}
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for i in range(startp + 2, endp):
                    j = i - startp
                    x = layers[i]
                    p = self._deform("%s.proj_%d" % (name, j), x, self.hbuf("%s.p%d" % (name, j), x.N, x.H, x.W, ida.o),
                                     om_name="%s.om%d" % (name, j))
                    ev = torch.cuda.Event()
                    ev.record(side)
                    early[i] = (p, ev)
        for i in range(startp + 1, endp):
            j = i - startp
            x = layers[i]
            o = ida.o
            f = ida.up_f[j]
            if i in early:
                p, ev = early[i]
                main.wait_event(ev)
            else:
                p = self._deform("%s.proj_%d" % (name, j), x, self.hbuf("ida_p", x.N, x.H, x.W, o))
            u = bops.upsample_add(p, self.P["%s.up_%d" % (name, j)], layers[i - 1],
                                  self.hbuf("ida_u", x.N, x.H * f, x.W * f, o), f)
            layers[i] = self._deform("%s.node_%d" % (name, j), u, self.hbuf("%s.n%d" % (name, j), x.N, u.H, u.W, o))

    neck_parts = 1            # > 1: neck + heads as that many batch slices on as many streams (measured: -0.6 % at two slices, the host then enqueues 4.9 ms per step)

    def neck_heads(self, cur, hm_act=ACT_NONE, static=False):
        B = cur[0].N
        pk = self.P.get("heads.fused") if self.fuse_heads else None
        parts = self.neck_parts if (self.use_side_stream and _lib.PROFILE is None and pk is not None) else 1
        if parts <= 1 or B % parts:
            return super().neck_heads(cur, hm_act, static)
        h, w = cur[2].H, cur[2].W
        outs = [self.buf("out_" + n, (B, r, h, w)) if static else torch.empty((B, r, h, w), dtype=torch.float32, device=self.dev)
                for n, r in zip(self.head_names, pk.rows)]
        main, hb = torch.cuda.current_stream(), B // parts
        streams = self._part_streams(parts - 1)
        saved, self.use_side_stream = self.use_side_stream, False       # a slice's neck runs serially on its own stream

        def run(k):
            feat = self.neck([v.images(k * hb, (k + 1) * hb) for v in cur])
            self.heads(feat, hm_act, static, outs=[o[k * hb:(k + 1) * hb] for o in outs])
            return feat
        try:
            for k, st in enumerate(streams):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    self.buf_tag = ".n%d" % (k + 1)
                    try:
                        run(k + 1)
                    finally:
                        self.buf_tag = ""
            self.last_feat = run(0)
        finally:
            self.use_side_stream = saved
        for st in streams:
            main.wait_stream(st)
        return dict(zip(self.head_names, outs))

    def heads(self, feat, hm_act=ACT_NONE, static=False, outs=None):
        B, h, w = feat.N, feat.H, feat.W
        pk = self.P.get("heads.fused") if self.fuse_heads else None
        if pk is not None:
            if outs is None:
                outs = [self.buf("out_" + n, (B, r, h, w)) if static else torch.empty((B, r, h, w), dtype=torch.float32, device=self.dev)
                        for n, r in zip(self.head_names, pk.rows)]
            nsig = pk.rows[0] if (self.head_names[0] == "hm" and hm_act == ops.ACT_SIGMOID) else 0
            bops.heads_fused(feat, pk, outs, nsig)
            return dict(zip(self.head_names, outs))
        hidden = self.hbuf("head_hidden", B, h, w, sum(self.head_hidden))
        bops.conv(feat, self.P["heads.0"], hidden, ACT_RELU)
        out, off = {}, 0
        for name, hc in zip(self.head_names, self.head_hidden):
            pc = self.P["heads.2." + name]
            o = self.buf("out_" + name, (B, pc.Cout, h, w)) if static else \
                torch.empty((B, pc.Cout, h, w), dtype=torch.float32, device=self.dev)
            nsig = pc.Cout if (name == "hm" and hm_act == ops.ACT_SIGMOID) else 0
            bops.conv(hidden.slice(off, hc), pc, o, ACT_NONE, out_mode=bops.OUT_F32_NCHW, nsig=nsig)
            out[name] = o
            off += hc
        return out


class EngineF16(EngineBF16):
    """The same engine on IEEE-half feature maps and weights (C ABI ``sgta_*_f16``): 11 significant bits instead of bf16's 8
    at the same bytes and matrix rate -- head outputs within a few 1e-3 of the fp32 path instead of a few 1e-2 -- for a
    range of +-65504 (clamped on store).  ``model.engine_dtype = "f16"``."""
    dtype = "f16"
    act_dtype = bops.F16

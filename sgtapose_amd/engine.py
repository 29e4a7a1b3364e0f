"""The HIP pipeline behind ``DLAPlanAWindowL3New.forward`` (reference call stack: DLA_PlanAWindow_l3new.forward -> imgpre2feats, dla.py:1505-1554).

Host side only plans: it packs weights once (BatchNorm folded, GEMM layouts), owns the activation
workspace (allocated once per input shape, reused every step; sized for HBM residency, nothing is
re-allocated inside a step) and enqueues the hand-written kernels of libsgta_hip.so on the current
HIP stream.  Nothing here computes on the host or in PyTorch.

Data layout in HBM: float32 NCHW everywhere (the reference's layout at the API boundary).  The two
frames of a pair run through the shared-weight backbone as ONE batch of 2B images (previous frames
first), so every backbone conv is launched once.  Tensors that the reference concatenates for a Root
conv (dla.py:169) are never copied: their producers write straight into channel slices of the
concat buffer (kernels take a batch stride).  The temporal stage edits the current-frame feature
maps in place (the reference builds a permuted copy of the whole map per level, dla.py:1009-1016).
"""
import torch

from . import _lib, ops
from .ops import ACT_NONE, ACT_RELU, PackedConv, View

N_HEADS = 8
D_FFN = 1024


def _bn_tuple(bn):
    return (bn.weight.detach().float(), bn.bias.detach().float(), bn.running_mean.detach().float(),
            bn.running_var.detach().float())


class CachedStem:
    """Stand-in for a frame's image in sequence mode: ``stem`` = relu(bn(conv7x7(image))) [B,16,H,W], computed when the
    frame was the current one (exact because the two stems are only summed, dla.py:325-328)."""

    def __init__(self, stem):
        self.stem = stem
        self.shape = (stem.shape[0], 3, stem.shape[2], stem.shape[3])

    def contiguous(self):
        return self

    def float(self):
        return self

    @property
    def device(self):
        return self.stem.device



# Side streams are shared by every engine of a process (one set per device): HIP maps streams onto a handful of hardware
# queues, and an engine that took fresh streams of its own could land two streams that are meant to overlap on one queue
# (measured: the second and third engine built in one process ran 3-6 % slower than the first).
_STREAMS = {}


def shared_stream(dev, name):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), name)
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(device=dev)        # every stream at the default priority
    return st


class Engine:
    def __init__(self, model):
        p = next(model.parameters())
        if not p.is_cuda:
            raise _lib.SgtaHipError("model parameters must live on the GPU (model.to('cuda')): the HIP path "
                                    "has no CPU fallback")
        _lib.load()
        self.dev = p.device
        self.model = model
        self.use_pos = bool(getattr(model.opt, "pos_embed", True))
        self.ks = list(model.kernel_list)
        self.scales = list(model.scale_list)
        self.num_classes = int(model.opt.num_classes)
        self.use_side_stream = True          # side stream: early_levels beside the backbone, levels 0/1 beside the neck
        self.early_levels = (2, 3)           # levels whose attention (minus write-back) overlaps the deeper backbone
        self.use_c16 = True                  # 16-cout MFMA kernel for the fused 7x7 stems (False: generic conv)
        self.level0_c16 = False              # level0 (3x3, 16->16) through the 16-cout direct kernel instead of Winograd F(4x4,3x3)
        self.compute_dead_levels = True      # levels 0/1 never reach the heads (dla.py:600-606); the
        #                                      reference still computes them, and so do we by default
        self.bufs = {}
        with torch.no_grad():
            self._pack(model)

    # ------------------------------------------------------------------------------------------
    def _pack(self, m):
        P = {}
        b = m.base
        for stem in ("pre_img_layer", "pre_hm_layer"):
            seq = getattr(b, stem)
            P[stem] = PackedConv(seq[0].weight, None, _bn_tuple(seq[1]), stride=1)
        P["level0"] = PackedConv(b.level0[0].weight, None, _bn_tuple(b.level0[1]), stride=1)
        P["level0.c16"] = ops.PackedC16(b.level0[0].weight, _bn_tuple(b.level0[1]))
        P["stem.c16"] = ops.PackedC16(b.pre_img_layer[0].weight, _bn_tuple(b.pre_img_layer[1]),
                                       b.pre_hm_layer[0].weight, _bn_tuple(b.pre_hm_layer[1]))
        P["level1"] = PackedConv(b.level1[0].weight, None, _bn_tuple(b.level1[1]), stride=2)

        def pack_tree(node, name):
            if node.levels == 1:
                for t in ("tree1", "tree2"):
                    blk = getattr(node, t)
                    P["%s.%s.conv1" % (name, t)] = PackedConv(blk.conv1.weight, None, _bn_tuple(blk.bn1), stride=blk.stride)
                    P["%s.%s.conv2" % (name, t)] = PackedConv(blk.conv2.weight, None, _bn_tuple(blk.bn2), stride=1)
                P[name + ".root"] = PackedConv(node.root.conv.weight, None, _bn_tuple(node.root.bn), stride=1)
                if hasattr(node, "project"):
                    P[name + ".project"] = PackedConv(node.project[0].weight, None, _bn_tuple(node.project[1]), stride=1)
            else:
                pack_tree(node.tree1, name + ".tree1")
                pack_tree(node.tree2, name + ".tree2")
                # a 2-level node's own ``project`` output is never consumed (dla.py:221-224): not packed

        for lv in range(2, 6):
            pack_tree(getattr(b, "level%d" % lv), "level%d" % lv)

        def pack_ida(ida, name):
            for j in range(1, len(ida.channels)):
                for kind in ("proj", "node"):
                    dc = getattr(ida, "%s_%d" % (kind, j))
                    com = dc.conv.conv_offset_mask
                    P["%s.%s_%d.om" % (name, kind, j)] = PackedConv(com.weight, com.bias)
                    P["%s.%s_%d" % (name, kind, j)] = PackedConv(dc.conv.weight, dc.conv.bias,
                                                                  _bn_tuple(dc.actf[0]), tap_major=True)
                P["%s.up_%d" % (name, j)] = getattr(ida, "up_%d" % j).weight.detach().float().contiguous()

        for i in range(3):
            pack_ida(getattr(m.dla_up, "ida_%d" % i), "dla_up.ida_%d" % i)
        pack_ida(m.ida_up, "ida_up")

        f = lambda t: t.detach().float().contiguous()
        self.tf = [self._pack_encoder(m.transformer[i]) for i in range(3)] if hasattr(m, "transformer") else []
        self.cat = [dict(w0=f(s[0].weight), b0=f(s[0].bias), w2=f(s[2].weight), b2=f(s[2].bias)) for s in m.cat_layer]

        self.head_names = [h for h in m.heads if hasattr(m, h)]      # l3new has no 'wh' module (base_model.py:114-115)
        hw = torch.cat([getattr(m, h)[0].weight.detach().float() for h in self.head_names], 0)
        hb = torch.cat([getattr(m, h)[0].bias.detach().float() for h in self.head_names], 0)
        P["heads.0"] = PackedConv(hw, hb)
        self.head_hidden = [getattr(m, h)[0].out_channels for h in self.head_names]
        for h in self.head_names:
            P["heads.2." + h] = PackedConv(getattr(m, h)[2].weight, getattr(m, h)[2].bias)
        self.P = P

    def _pack_encoder(self, enc):
        """One TransformerEncoder (its three layers alias ONE module, dla.py:788-789) -> flat weight dict."""
        f = lambda t: t.detach().float().contiguous()          # noqa: E731
        lay = enc.layers[0]
        ca = lay.cross_attn
        return dict(
            wq=f(ca.w_q.weight), wkv=torch.cat([f(ca.w_k.weight), f(ca.w_v.weight)], 0).contiguous(),
            fc_w=f(ca.fc.weight), fc_b=f(ca.fc.bias),
            pos=ops.PosTiles(f(ca.pos_embed)) if self.use_pos else None,     # re-packed once: it is a weight
            ln1=(f(lay.norm1.weight), f(lay.norm1.bias)), ln3=(f(lay.norm3.weight), f(lay.norm3.bias)),
            w1=f(lay.linear1.weight), b1=f(lay.linear1.bias), w2=f(lay.linear2.weight), b2=f(lay.linear2.bias),
            n=ca.n, C=ca.inp_dim, hid=ca.hid_dim)

    # ------------------------------------------------------------------------------------------
    buf_tag = ""                  # suffix of workspace names while a slice of the batch runs on its own stream (split_levels)
    split_levels = ()             # backbone levels run as ``split_parts`` slices of the batch on as many streams
    split_parts = 2

    def _part_streams(self, n):
        return [shared_stream(self.dev, "part%d" % k) for k in range(n)]

    def buf(self, name, shape, dtype=torch.float32):
        key = (name + self.buf_tag, tuple(shape), dtype)
        t = self.bufs.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.dev)
            self.bufs[key] = t
        return t

    def vbuf(self, name, n, c, h, w):
        return View(self.buf(name, (n, c, h, w)))

    def workspace_bytes(self):
        return sum(t.numel() * t.element_size() for t in self.bufs.values())

    # ------------------------------------------------------------------------------------------
    def _tree_l1(self, node, name, x, out, cat, bottom=None):
        """1-level aggregation node (two BasicBlocks + Root).  ``cat`` is the Root's concat buffer; its
        child slots (channels >= 2*cout) are filled by the caller, except this node's own ``bottom``."""
        P, c, cin, N = self.P, node.cout, node.cin, x.B
        h, w = x.H // node.stride, x.W // node.stride
        if node.stride > 1:
            if bottom is None:
                bottom = cat.slice(2 * c, cin) if node.level_root else self.vbuf("pool", N, cin, h, w)
                ops.maxpool2(x, bottom)
        else:
            bottom = x
        if cin != c:
            residual = self.vbuf("proj", N, c, h, w)
            ops.conv2d(bottom, P[name + ".project"], residual, ACT_NONE)
        else:
            residual = bottom
        a = self.vbuf("blk_a", N, c, h, w)
        x1, x2 = cat.slice(c, c), cat.slice(0, c)
        ops.conv2d(x, P[name + ".tree1.conv1"], a, ACT_RELU)
        ops.conv2d(a, P[name + ".tree1.conv2"], x1, ACT_RELU, res=residual, res_mode=1)
        ops.conv2d(x1, P[name + ".tree2.conv1"], a, ACT_RELU)
        ops.conv2d(a, P[name + ".tree2.conv2"], x2, ACT_RELU, res=x1, res_mode=1)
        ops.conv2d(cat, P[name + ".root"], out, ACT_RELU)
        return out

    def _tree(self, node, name, x, out, bottom=None):
        N, c, cin = x.B, node.cout, node.cin
        h, w = x.H // node.stride, x.W // node.stride
        if node.levels == 1:
            cat = self.vbuf(name + ".cat", N, node.root_dim, h, w)
            if bottom is not None and node.level_root:       # a pre-pooled input must sit in the Root's concat slot
                bottom = None
            return self._tree_l1(node, name, x, out, cat, bottom=bottom)
        # 2-level, level-root node (levels 3, 4): inner Root sees [x2 | x1 | bottom | tree1-out]
        inner = node.tree2
        cat2 = self.vbuf(name + ".cat2", N, inner.root_dim, h, w)
        bottom = cat2.slice(2 * c, cin)
        ops.maxpool2(x, bottom)
        t1 = cat2.slice(2 * c + cin, c)
        cat1 = self.vbuf(name + ".cat1", N, node.tree1.root_dim, h, w)
        self._tree_l1(node.tree1, name + ".tree1", x, t1, cat1, bottom=bottom)
        return self._tree_l1(inner, name + ".tree2", t1, out, cat2)

    def image_stem(self, img, out=None):
        """relu(bn(conv7x7(img))) alone -> [B,16,H,W] (sequence mode: cached for the next step)."""
        B, _, H, W = img.shape
        out = torch.empty((B, 16, H, W), dtype=torch.float32, device=self.dev) if out is None else out
        ops.conv_c16(img.contiguous().float(), None, self.P["stem.c16"], out, mode=2)
        return out

    def backbone(self, x, pre_img, pre_hm, repro_hm, after_level=None):
        """Both frames as one batch of 2B (previous first).  Returns the six level Views [2B, C_i, ...].
        ``after_level(i, view)`` is called as soon as level i's kernels are enqueued."""
        return self.backbone_frames(((pre_img, pre_hm), (x, repro_hm)), after_level)

    def backbone_frames(self, frames, after_level=None):
        """``frames`` = ((image, heat-map), ...) oldest first, current last: ONE batch of len(frames) * B images
        through the shared-weight backbone."""
        B, _, H, W = frames[-1][0].shape
        N = len(frames) * B
        P = self.P
        s = self.buf("stem", (N, 16, H, W))
        for half, (img, hm) in enumerate(frames):
            sl = slice(half * B, (half + 1) * B)
            if isinstance(img, CachedStem):        # sequence mode: the image stem of this frame already exists
                ops.conv_c16(img.stem, hm, P["stem.c16"], View(s[sl]), mode=3)
            elif self.use_c16:                     # both 7x7 stems + BN + ReLU + sum in one kernel
                ops.conv_c16(img, hm, P["stem.c16"], View(s[sl]))
            else:
                t = self.buf("stem_hm", (N, 16, H, W))
                ops.conv2d(hm, P["pre_hm_layer"], View(t[sl]), ACT_RELU)
                ops.conv2d(img, P["pre_img_layer"], View(s[sl]), ACT_RELU, res=View(t[sl]), res_mode=2)
        lv = [self.vbuf("l0", N, 16, H, W)]
        if self.use_c16 and self.level0_c16:
            ops.conv_c16(View(s), None, P["level0.c16"], lv[0])
        else:
            # F(4x4,3x3) (16 -> 16 at full resolution: 0.93 ms against 1.15 ms for the direct 16-cout kernel at B = 32)
            ops.conv2d(View(s), P["level0"], lv[0], ACT_RELU)
        if after_level:
            after_level(0, lv[0])
        lv.append(self.vbuf("l1", N, 32, H // 2, W // 2))
        # level2 = Tree(levels = 1): its ``bottom`` = MaxPool2d(2)(level1 output) rides in the stride-2 kernel's epilogue
        pooled = self.vbuf("pool", N, 32, H // 4, W // 4)
        if not ops.conv3x3s2_pool(lv[0], P["level1"], lv[1], pooled):
            ops.conv2d(lv[0], P["level1"], lv[1], ACT_RELU)
            pooled = None
        if after_level:
            after_level(1, lv[1])
        ch = self.model.base.channels
        main = torch.cuda.current_stream()
        parts = self.split_parts if (self.use_side_stream and _lib.PROFILE is None and self.split_levels) else 1
        if parts > 1 and N % parts:
            parts = 2 if N % 2 == 0 else 1
        for i in range(2, 6):
            out = self.vbuf("l%d" % i, N, ch[i], H >> i, W >> i)
            node, name = getattr(self.model.base, "level%d" % i), "level%d" % i
            bottom = pooled if i == 2 else None
            if parts > 1 and i in self.split_levels:
                # slices of the batch side by side on their own streams (engine_bf16.py); joined after every level
                h = N // parts
                sl = lambda v, k: None if v is None else View(v.base[k * h:(k + 1) * h], v.c0, v.C)     # noqa: E731
                streams = self._part_streams(parts - 1)
                for k, st in enumerate(streams):
                    st.wait_stream(main)
                    with torch.cuda.stream(st):
                        self.buf_tag = ".p%d" % (k + 1)
                        try:
                            self._tree(node, name, sl(lv[i - 1], k + 1), sl(out, k + 1), bottom=sl(bottom, k + 1))
                        finally:
                            self.buf_tag = ""
                self._tree(node, name, sl(lv[i - 1], 0), sl(out, 0), bottom=sl(bottom, 0))
                for st in streams:
                    main.wait_stream(st)
            else:
                self._tree(node, name, lv[i - 1], out, bottom=bottom)
            lv.append(out)
            if after_level:
                after_level(i, out)
        return lv

    # ------------------------------------------------------------------------------------------
    def temporal_tokens(self, feats, i, idx, B):
        """Temporal stage of pyramid level ``i`` up to (not including) the write-back: reads both frame halves of
        ``feats`` only, so it may run beside the deeper backbone levels.  -> (z, fid, win) for temporal_scatter."""
        K = self.num_classes
        C, H, W = feats.C, feats.H, feats.W
        pre_f, cur_f = View(feats.base[:B]), View(feats.base[B:])
        n = K * (1 + 2 * (self.ks[i] // 2)) ** 2
        fid, win = ops.window_index(idx, float(self.scales[i]), self.ks[i], H, W,
                                    self.buf("fid%d" % i, (2, B, n), torch.int64),
                                    self.buf("win%d" % i, (B, n), torch.uint8))
        M = B * n
        cat_tok = self.buf("cat_tok%d" % i, (M, 2 * C))
        ops.window_gather(cur_f, fid[1], cat_tok, col0=C)
        if i <= 2:
            T = self.tf[i]
            hid = T["hid"]
            key = self.buf("key%d" % i, (M, C))
            cur = self.buf("qry%d" % i, (M, C))
            ops.window_gather(pre_f, fid[0], key)
            ops.window_gather(cur_f, fid[1], cur)
            kv = ops.linear(key, T["wkv"], None, self.buf("kv%d" % i, (M, 2 * hid)))
            q = self.buf("q%d" % i, (M, hid))
            att = self.buf("att%d" % i, (M, hid))
            y1 = self.buf("y1_%d" % i, (M, C))
            nxt = self.buf("qry_b%d" % i, (M, C))
            for it in range(3):                    # ONE shared layer, three times (dla.py:788-803)
                ops.linear(cur, T["wq"], None, q)
                ops.mhca(q, kv[:, :hid], kv[:, hid:], T["pos"], att, B, n, N_HEADS, hid // N_HEADS)
                ops.linear(att, T["fc_w"], T["fc_b"], y1, res=cur, ln=T["ln1"])
                dst = cat_tok[:, :C] if it == 2 else nxt
                if ops.ffn_ln(y1, T["w1"], T["b1"], T["w2"], T["b2"], T["ln3"], dst) is None:
                    hdn = self.buf("ffn%d" % i, (M, D_FFN))        # two-launch fall-back only: 155 MB at level 0
                    ops.linear(y1, T["w1"], T["b1"], hdn, relu=True)
                    ops.linear(hdn, T["w2"], T["b2"], dst, res=y1, ln=T["ln3"])
                cur, nxt = dst, cur
        else:
            ops.window_gather(pre_f, fid[0], cat_tok, col0=0)
        Cc = self.cat[i]
        h1 = ops.linear(cat_tok, Cc["w0"], Cc["b0"], self.buf("cat_h%d" % i, (M, 4 * C)), relu=True)
        z = ops.linear(h1, Cc["w2"], Cc["b2"], self.buf("cat_z%d" % i, (M, C)))
        return z, fid, win

    def temporal_scatter(self, feats, B, tok):
        """Write the substituted tokens into the current-frame half of ``feats`` (in place, last-wins)."""
        z, fid, win = tok
        ops.window_scatter(z, fid[1], win, View(feats.base[B:]))

    def temporal(self, lv, idx, B, levels):
        """In-place temporal stage on the current-frame halves of ``lv`` for the given pyramid levels;
        ``idx`` = prior arg-max coordinates.  Returns the level-0 window ids (or None)."""
        fid0 = None
        for i in levels:
            tok = self.temporal_tokens(lv[i], i, idx, B)
            self.temporal_scatter(lv[i], B, tok)
            if i == 0:
                fid0 = tok[1]
        return fid0

    # ------------------------------------------------------------------------------------------
    def _deform(self, name, x, out, om_name="om"):
        om = self.vbuf(om_name, x.B, 27, x.H, x.W)
        ops.conv2d(x, self.P[name + ".om"], om, ACT_NONE)
        return ops.dcn_v2(x, om, self.P[name], out, ACT_RELU)

    def _ida(self, ida, name, layers, startp, endp):
        """IDAUp.forward (dla.py:571-577) on a list of Views; new buffers for every node output.
        ``proj_j(layers[i])`` only reads the level as it was when the call started, so the projections of the later
        iterations (small maps that fill a fraction of the chip) run on a second stream beside the node chain."""
        main = torch.cuda.current_stream()
        side = self._neck_stream() if (self.use_side_stream and _lib.PROFILE is None and endp - startp > 2) else None
        early = {}
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for i in range(startp + 2, endp):
                    j = i - startp
                    x = layers[i]
                    p = self._deform("%s.proj_%d" % (name, j), x, self.vbuf("%s.p%d" % (name, j), x.B, ida.o, x.H, x.W),
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
                p = self._deform("%s.proj_%d" % (name, j), x, self.vbuf("ida_p", x.B, o, x.H, x.W))
            u = ops.upsample_add(p, self.P["%s.up_%d" % (name, j)], layers[i - 1],
                                 self.vbuf("ida_u", x.B, o, x.H * f, x.W * f), f)
            layers[i] = self._deform("%s.node_%d" % (name, j), u, self.vbuf("%s.n%d" % (name, j), x.B, o, u.H, u.W))

    def neck(self, layers):
        m = self.model
        layers = list(layers)
        out = [layers[-1]]
        for i in range(len(layers) - m.dla_up.startp - 1):
            self._ida(getattr(m.dla_up, "ida_%d" % i), "dla_up.ida_%d" % i, layers, len(layers) - i - 2, len(layers))
            out.insert(0, layers[-1])
        y = out[:m.last_level - m.first_level]
        self._ida(m.ida_up, "ida_up", y, 0, len(y))
        return y[-1]

    def neck_heads(self, cur, hm_act=ACT_NONE, static=False):
        """DCN neck + head branches on the current-frame maps of the six levels."""
        feat = self.neck(cur)
        self.last_feat = feat
        return self.heads(feat, hm_act, static)

    def heads(self, feat, hm_act=ACT_NONE, static=False):
        B, h, w = feat.B, feat.H, feat.W
        hidden = self.vbuf("head_hidden", B, sum(self.head_hidden), h, w)
        ops.conv2d(feat, self.P["heads.0"], hidden, ACT_RELU)
        out, off = {}, 0
        for name, hc in zip(self.head_names, self.head_hidden):
            pc = self.P["heads.2." + name]
            o = self.buf("out_" + name, (B, pc.Cout, h, w)) if static else \
                torch.empty((B, pc.Cout, h, w), dtype=torch.float32, device=self.dev)
            ops.conv2d(hidden.slice(off, hc), pc, o, hm_act if name == "hm" else ACT_NONE)
            out[name] = o
            off += hc
        return out

    # ------------------------------------------------------------------------------------------
    def forward(self, x, pre_img, pre_hm, repro_hm, pre_hm_cls, repro_hm_cls, hm_act=ACT_NONE, static=False):
        """-> {'hm': logits [B,7,H/4,W/4], 'reg': [B,2,..], 'tracking': [B,2,..]} (hm_act=ACT_SIGMOID fuses
        SGTADetector._sigmoid_output into the head epilogue)."""
        args = [t.contiguous().float() for t in (x, pre_img, pre_hm, repro_hm, pre_hm_cls, repro_hm_cls)]
        x, pre_img, pre_hm, repro_hm, pre_hm_cls, repro_hm_cls = args
        for t in args:
            if t.device != self.dev:
                raise _lib.SgtaHipError("input on %s, model on %s" % (t.device, self.dev))
        B, _, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError("network input must be a multiple of 32 in both dimensions, got %dx%d" % (H, W))
        if tuple(pre_hm_cls.shape) != (B, self.num_classes, H // 4, W // 4):
            raise ValueError("pre_hm_cls must be [B,%d,H/4,W/4]" % self.num_classes)
        with torch.no_grad():
            idx = ops.prior_argmax(pre_hm_cls, repro_hm_cls, self.buf("idx", (2, B, self.num_classes, 2)))
            # The attention of a level only READS that level's features until its final write-back, and it is a
            # chain of small latency-bound kernels.  ``early_levels`` run on a side stream as soon as their
            # features exist, beside the MFMA-bound deeper backbone levels; their write-backs follow on the main
            # stream once the backbone no longer reads the features.  Levels 0/1 never reach the heads (dla_up
            # starts at level 2, dla.py:600-606; the reference still computes them, and so do we unless
            # compute_dead_levels is off): by default their whole stage runs beside the neck / heads instead.
            main = torch.cuda.current_stream()
            side = self._side_stream() if (self.use_side_stream and _lib.PROFILE is None) else None
            early = self.early_levels if side is not None else ()
            tok = {}

            def after_level(i, feats):
                if i in early:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        tok[i] = self.temporal_tokens(feats, i, idx, B)

            lv = self.backbone(x, pre_img, pre_hm, repro_hm, after_level)
            if early:
                main.wait_stream(side)
            self.last_fid0 = None
            dead_on_side = False
            if self.compute_dead_levels:
                if side is not None and 0 not in early and 1 not in early:
                    # levels 0/1 are never read again: their whole stage runs beside the neck / heads
                    dead_on_side = True
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        self.last_fid0 = self.temporal(lv, idx, B, (0, 1))
                else:
                    for i in (0, 1):
                        t01 = tok.get(i) or self.temporal_tokens(lv[i], i, idx, B)
                        self.temporal_scatter(lv[i], B, t01)
                        if i == 0:
                            self.last_fid0 = t01[1]
            for i in (2, 3, 4, 5):
                self.temporal_scatter(lv[i], B, tok.get(i) or self.temporal_tokens(lv[i], i, idx, B))
            cur = [self._frame(v, 1, B) for v in lv]
            out = self.neck_heads(cur, hm_act, static)
            if dead_on_side:
                main.wait_stream(side)             # the step is complete only when the side work is
            return out

    @staticmethod
    def _frame(v, k, B):
        """Frame ``k`` (0 = oldest) of a level whose batch holds the frames back to back."""
        return View(v.base[k * B:(k + 1) * B])

    def _neck_stream(self):
        return shared_stream(self.dev, "neck")

    def _side_stream(self):
        return shared_stream(self.dev, "side")

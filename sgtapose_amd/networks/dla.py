"""``dlapawdl3new_34`` as a parameter tree whose ``state_dict()`` is key- and shape-compatible with the
reference's ``DLA_PlanAWindow_l3new`` (sgtapose/lib/model/networks/dla.py:1458-1554 on top of
BaseModelPlanA, sgtapose/lib/model/networks/base_model.py:102-200) -- 560 entries, including the
aliased ``transformer.i.layers.{0,1,2}.*`` keys that the reference's ``_get_clones`` produces by
registering ONE layer object three times (dla.py:788-789).  Reference checkpoints therefore load
with ``strict=True``.

The modules below only HOLD parameters (their ``forward`` is never used); the computation is the HIP
pipeline in ``sgtapose_amd/engine.py``, reached through ``DLAPlanAWindowL3New.forward`` which keeps
the reference's signature and return type (a list with one dict of head outputs).
"""
import math

import torch
from torch import nn

from ..dcn_v2 import DCN

BN_MOMENTUM = 0.1
DLA34_LEVELS = (1, 1, 1, 2, 2, 1)
DLA34_CHANNELS = (16, 32, 64, 128, 256, 512)


def _bn(c):
    return nn.BatchNorm2d(c, momentum=BN_MOMENTUM)


def _conv_bn_relu(cin, cout, k, stride=1):
    """Sequential(conv, bn, relu): keys '<name>.0.weight', '<name>.1.*'."""
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False), _bn(cout),
                         nn.ReLU(inplace=True))


class Block(nn.Module):
    """BasicBlock parameters: conv1/bn1/conv2/bn2 (3x3)."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = _bn(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, stride=1, padding=1, bias=False)
        self.bn2 = _bn(cout)
        self.stride = stride


class RootNode(nn.Module):
    """1x1 aggregation conv over the channel concat + BN (+ReLU)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, bias=False)
        self.bn = _bn(cout)


class TreeNode(nn.Module):
    """Hierarchical aggregation node.  ``root_dim`` bookkeeping follows the DLA paper / reference:
    a level-root node also feeds its (pooled) input into the innermost Root."""

    def __init__(self, levels, cin, cout, stride=1, level_root=False, root_dim=0):
        super().__init__()
        if root_dim == 0:
            root_dim = 2 * cout
        if level_root:
            root_dim += cin
        if levels == 1:
            self.tree1 = Block(cin, cout, stride)
            self.tree2 = Block(cout, cout, 1)
            self.root = RootNode(root_dim, cout)
        else:
            self.tree1 = TreeNode(levels - 1, cin, cout, stride)
            self.tree2 = TreeNode(levels - 1, cout, cout, 1, root_dim=root_dim + cout)
        if cin != cout:
            self.project = nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), _bn(cout))
        self.levels, self.cin, self.cout = levels, cin, cout
        self.stride, self.level_root, self.root_dim = stride, level_root, root_dim


class DLA34(nn.Module):
    """Backbone parameters incl. the two live stems and the dead ``base_layer`` (kept for key parity)."""

    def __init__(self):
        super().__init__()
        ch, lv = DLA34_CHANNELS, DLA34_LEVELS
        self.channels = list(ch)
        self.base_layer = _conv_bn_relu(3, ch[0], 7)
        self.level0 = _conv_bn_relu(ch[0], ch[0], 3)
        self.level1 = _conv_bn_relu(ch[0], ch[1], 3, stride=2)
        self.level2 = TreeNode(lv[2], ch[1], ch[2], 2, level_root=False)
        self.level3 = TreeNode(lv[3], ch[2], ch[3], 2, level_root=True)
        self.level4 = TreeNode(lv[4], ch[3], ch[4], 2, level_root=True)
        self.level5 = TreeNode(lv[5], ch[4], ch[5], 2, level_root=True)
        self.pre_img_layer = _conv_bn_relu(3, ch[0], 7)
        self.pre_hm_layer = _conv_bn_relu(1, ch[0], 7)


class DeformConv(nn.Module):
    """DCN + BN + ReLU parameters: keys 'actf.0.*', 'conv.{weight,bias,conv_offset_mask.*}'."""

    def __init__(self, cin, cout):
        super().__init__()
        self.actf = nn.Sequential(_bn(cout), nn.ReLU(inplace=True))
        self.conv = DCN(cin, cout, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)


def _bilinear_up_init(up):
    w = up.weight.data
    k = w.shape[2]
    f = math.ceil(k / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    a = 1 - (torch.arange(k, dtype=torch.float32) / f - c).abs()
    w.copy_((a[:, None] * a[None, :]).expand_as(w))


class IDAUp(nn.Module):
    def __init__(self, o, channels, up_f):
        super().__init__()
        self.o, self.channels, self.up_f = o, list(channels), [int(f) for f in up_f]
        for i in range(1, len(channels)):
            f = int(up_f[i])
            setattr(self, "proj_%d" % i, DeformConv(channels[i], o))
            up = nn.ConvTranspose2d(o, o, f * 2, stride=f, padding=f // 2, output_padding=0, groups=o, bias=False)
            _bilinear_up_init(up)
            setattr(self, "up_%d" % i, up)
            setattr(self, "node_%d" % i, DeformConv(o, o))


class DLAUp(nn.Module):
    def __init__(self, startp, channels, scales):
        super().__init__()
        self.startp = startp
        channels = list(channels)
        in_channels = list(channels)
        scales = [int(s) for s in scales]
        for i in range(len(channels) - 1):
            j = -i - 2
            setattr(self, "ida_%d" % i, IDAUp(channels[j], in_channels[j:], [s // scales[j] for s in scales[j:]]))
            scales[j + 1:] = [scales[j]] * len(scales[j + 1:])
            in_channels[j + 1:] = [channels[j]] * len(in_channels[j + 1:])


class MHCA(nn.Module):
    def __init__(self, n_heads, inp_dim, hid_dim, n):
        super().__init__()
        self.n_heads, self.inp_dim, self.hid_dim, self.n = n_heads, inp_dim, hid_dim, n
        self.w_q = nn.Linear(inp_dim, hid_dim, bias=False)
        self.w_k = nn.Linear(inp_dim, hid_dim, bias=False)
        self.w_v = nn.Linear(inp_dim, hid_dim, bias=False)
        self.fc = nn.Linear(hid_dim, inp_dim)
        self.pos_embed = nn.Parameter(torch.zeros(n_heads, n, n))


class EncoderLayer(nn.Module):
    def __init__(self, d_inp, d_model, n_k, d_ffn=1024, n_heads=8):
        super().__init__()
        self.cross_attn = MHCA(n_heads, d_inp, d_model * n_heads, n_k)
        self.dropout1 = nn.Dropout(0.1)
        self.norm1 = nn.LayerNorm(d_inp)
        self.linear1 = nn.Linear(d_inp, d_ffn)
        self.activation = nn.ReLU()
        self.dropout3 = nn.Dropout(0.1)
        self.linear2 = nn.Linear(d_ffn, d_inp)
        self.dropout4 = nn.Dropout(0.1)
        self.norm3 = nn.LayerNorm(d_inp)


class Encoder(nn.Module):
    """``layers`` registers the SAME layer object num_layers times (weight sharing + aliased keys)."""

    def __init__(self, layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([layer] * num_layers)
        self.num_layers = num_layers


def _head(cin, hidden, classes, kernel):
    return nn.Sequential(nn.Conv2d(cin, hidden, kernel, padding=kernel // 2, bias=True), nn.ReLU(inplace=True),
                         nn.Conv2d(hidden, classes, 1, bias=True))


class DLAPlanAWindowL3New(nn.Module):
    """Constructor signature of the reference's network classes: (num_layers, heads, head_convs, opt)."""

    scale_list = (4, 2, 1, 1 / 2, 1 / 4, 1 / 8)

    def __init__(self, num_layers, heads, head_convs, opt):
        super().__init__()
        if num_layers != 34:
            raise ValueError("only dlapawdl3new_34 is built (the live model of the reference)")
        if getattr(opt, "dla_node", "dcn") != "dcn":
            raise ValueError("only --dla_node dcn is supported")
        self.opt = opt
        self.heads = dict(heads)
        self.num_stacks = 1
        head_kernel = getattr(opt, "head_kernel", 3)
        for name, classes in self.heads.items():
            if "wh" in name:                      # skipped by the reference too (base_model.py:114-115)
                continue
            hc = head_convs[name]
            if len(hc) != 1:
                raise ValueError("head_conv must have exactly one hidden width (reference default [256])")
            fc = _head(64, hc[0], classes, head_kernel)
            if "hm" in name:
                fc[-1].bias.data.fill_(opt.prior_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.constant_(m.bias, 0)
            setattr(self, name, fc)
        self.first_level, self.last_level = 2, 5
        self.base = DLA34()
        ch = self.base.channels
        self.dla_up = DLAUp(self.first_level, ch[self.first_level:], [2 ** i for i in range(len(ch) - self.first_level)])
        self.ida_up = IDAUp(ch[self.first_level], ch[self.first_level:self.last_level],
                            [2 ** i for i in range(self.last_level - self.first_level)])
        self.K_list = [int(getattr(opt, "k_list_%d" % i)) for i in range(1, 7)]
        self.kernel_list = [int(getattr(opt, "ks%d" % i)) for i in range(1, 7)]
        if any(k != 1 for k in self.K_list):
            raise ValueError("k_list_* must be 1 (the reference's inference setting)")
        ncls = opt.num_classes
        self.transformer = nn.ModuleList([
            Encoder(EncoderLayer(16 * 2 ** i, 4 * 2 ** i,
                                 ncls * self.K_list[i] * (1 + 2 * (self.kernel_list[i] // 2)) ** 2), 3)
            for i in range(3)])
        self.cat_layer = nn.ModuleList([
            nn.Sequential(nn.Linear(16 * 2 ** (i + 1), 32 * 2 ** (i + 1)), nn.ReLU(), nn.Linear(32 * 2 ** (i + 1), 16 * 2 ** i))
            for i in range(6)])
        self._engine = None

    # -- the reference's forward contract: list (num_stacks = 1) of {'hm','reg','tracking'} logits ------
    def forward(self, x, pre_img=None, pre_hm=None, repro_hm=None, pre_hm_cls=None, repro_hm_cls=None):
        if any(t is None for t in (pre_img, pre_hm, repro_hm, pre_hm_cls, repro_hm_cls)):
            raise ValueError("phase PlanA_win needs all six inputs (sgta_detector.py:899-900)")
        if self.training:
            raise RuntimeError("the HIP path is inference-only: call model.eval() (sgta_detector.py:54)")
        return [self.engine().forward(x, pre_img, pre_hm, repro_hm, pre_hm_cls, repro_hm_cls)]

    # 'f32' (the reference's arithmetic, BASELINE configs[1]), 'bf16' (configs[2]: bf16 feature maps / weights on bf16
    # MFMA, fp32 accumulate / softmax / LayerNorm / decode) or 'f16' (the same data path on IEEE half: an eighth of bf16's
    # rounding error at the same speed, range +-65504) or 'f32x3' (fp32 ACCURACY with the stems, every convolution, the deformable
    # convs and the heads emulated on the 16-bit matrix pipe by operand splitting, engine_x3.py); also read from opt.compute_dtype
    @property
    def engine_dtype(self):
        return getattr(self, "_engine_dtype", getattr(self.opt, "compute_dtype", "f32"))

    @engine_dtype.setter
    def engine_dtype(self, v):
        if v not in ("f32", "bf16", "f16", "f32x3"):
            raise ValueError("engine_dtype must be 'f32', 'bf16', 'f16' or 'f32x3'")
        object.__setattr__(self, "_engine_dtype", v)
        self._engine = None

    def engine(self):
        if self._engine is None:
            if self.engine_dtype == "f32x3":          # fp32 accuracy, convolutions emulated on the 16-bit matrix pipe
                from ..engine_x3 import EngineX3
                self._engine = EngineX3(self)
            elif self.engine_dtype in ("bf16", "f16"):
                from ..engine_bf16 import EngineBF16, EngineF16
                self._engine = (EngineF16 if self.engine_dtype == "f16" else EngineBF16)(self)
            else:
                from ..engine import Engine
                self._engine = Engine(self)
        return self._engine

    def invalidate(self):
        """Drop packed weights (call after changing parameters in place)."""
        self._engine = None

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)


class DLAPlanAWindowThree(nn.Module):
    """``DLA_PlanAWindow_Three`` ('dlapawd3', reference dla.py:1349-1456 over BaseModelPlanA_Three,
    base_model.py:203-299): the current frame attends to the windows of TWO previous frames at levels 0-2 (one encoder
    per frame unless ``opt.shared_ca``), levels 3-5 run cat_layer on every pixel.  Same constructor signature and
    state_dict (690 / 564 keys) as the reference."""

    scale_list = (4, 2, 1)

    def __init__(self, num_layers, heads, head_convs, opt):
        super().__init__()
        if num_layers != 34:
            raise ValueError("only dlapawd3_34 is built")
        if getattr(opt, "dla_node", "dcn") != "dcn":
            raise ValueError("only --dla_node dcn is supported")
        self.opt = opt
        self.heads = dict(heads)
        self.num_stacks = 1
        self.n_frames = 3
        head_kernel = getattr(opt, "head_kernel", 3)
        for name, classes in self.heads.items():          # BaseModelPlanA_Three keeps every head, 'wh' included
            hc = head_convs[name]
            if len(hc) != 1:
                raise ValueError("head_conv must have exactly one hidden width (reference default [256])")
            fc = _head(64, hc[0], classes, head_kernel)
            if "hm" in name:
                fc[-1].bias.data.fill_(opt.prior_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.constant_(m.bias, 0)
            setattr(self, name, fc)
        self.first_level, self.last_level = 2, 5
        self.base = DLA34()
        ch = self.base.channels
        self.dla_up = DLAUp(self.first_level, ch[self.first_level:], [2 ** i for i in range(len(ch) - self.first_level)])
        self.ida_up = IDAUp(ch[self.first_level], ch[self.first_level:self.last_level],
                            [2 ** i for i in range(self.last_level - self.first_level)])
        self.K_list = [int(getattr(opt, "k_list_%d" % i)) for i in range(1, 4)]
        self.kernel_list = [int(getattr(opt, "ks%d" % i)) for i in range(1, 4)]
        if any(k != 1 for k in self.K_list):
            raise ValueError("k_list_* must be 1 (the reference's inference setting)")
        self.shared_ca = bool(getattr(opt, "shared_ca", False))
        ncls = opt.num_classes

        def encoders():
            return nn.ModuleList([
                Encoder(EncoderLayer(16 * 2 ** i, 4 * 2 ** i,
                                     ncls * self.K_list[i] * (1 + 2 * (self.kernel_list[i] // 2)) ** 2), 3)
                for i in range(3)])
        if not self.shared_ca:
            self.pprev_transformer = encoders()
        self.prev_transformer = encoders()
        self.cat_layer = nn.ModuleList([
            nn.Sequential(nn.Linear(16 * 2 ** i * 3, 32 * 2 ** i * 3), nn.ReLU(), nn.Linear(32 * 2 ** i * 3, 16 * 2 ** i))
            for i in range(6)])
        self._engine = None

    def forward(self, x, ppre_img=None, pre_img=None, ppre_hm=None, pre_hm=None, repro_hm=None,
                ppre_hm_cls=None, pre_hm_cls=None, repro_hm_cls=None):
        """base_model.py:267-299: list (num_stacks = 1) of {'hm','reg','wh','tracking'} logits."""
        args = (ppre_img, pre_img, ppre_hm, pre_hm, repro_hm, ppre_hm_cls, pre_hm_cls, repro_hm_cls)
        if any(t is None for t in args):
            raise ValueError("the three-frame model needs all nine inputs")
        if self.training:
            raise RuntimeError("the HIP path is inference-only: call model.eval()")
        return [self.engine().forward(x, *args)]

    # 'f32' (default) or 'bf16' (engine_three_bf16.EngineThreeBF16); also read from opt.compute_dtype
    @property
    def engine_dtype(self):
        return getattr(self, "_engine_dtype", getattr(self.opt, "compute_dtype", "f32"))

    @engine_dtype.setter
    def engine_dtype(self, v):
        if v not in ("f32", "bf16", "f16", "f32x3"):
            raise ValueError("engine_dtype must be 'f32', 'bf16', 'f16' or 'f32x3'")
        object.__setattr__(self, "_engine_dtype", v)
        self._engine = None

    def engine(self):
        if self._engine is None:
            if self.engine_dtype == "f32x3":
                from ..engine_x3 import EngineThreeX3
                self._engine = EngineThreeX3(self)
            elif self.engine_dtype in ("bf16", "f16"):
                from ..engine_three_bf16 import EngineThreeBF16, EngineThreeF16
                self._engine = (EngineThreeF16 if self.engine_dtype == "f16" else EngineThreeBF16)(self)
            else:
                from ..engine_three import EngineThree
                self._engine = EngineThree(self)
        return self._engine

    def invalidate(self):
        self._engine = None

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)


class DLAPlanAWindowSeq(DLAPlanAWindowThree):
    """``DLA_PlanAWindow_Three`` generalised to a temporal window of T = ``opt.seq_len`` frames (T - 1 previous frames +
    the current one; BASELINE configs[4] asks for T = 4, the reference stops at T = 3).
    One attention stream per previous frame against the same current-frame queries at levels 0-2
    (``frame_transformer.k.i``, k = 0 the OLDEST frame), their outputs concatenated with the current tokens before
    ``cat_layer`` of width T*C -> 2T*C -> C; at levels 3-5 cat_layer runs on every pixel of cat[frame_0 .. frame_{T-2}, cur].
    ``opt.legacy_slip`` reproduces the reference's T = 3 behaviour (``ppre_f = pre_feats`` at levels 3-5, dla.py:1435) so
    that the T = 3 instance is checked against the reference's own outputs (keys: pprev_transformer -> frame_transformer.0,
    prev_transformer -> frame_transformer.1)."""

    def __init__(self, num_layers, heads, head_convs, opt):
        T = int(getattr(opt, "seq_len", 4))
        if T < 3:
            raise ValueError("seq_len must be >= 3 (the two-frame model is dlapawdl3new_34)")
        super().__init__(num_layers, heads, head_convs, opt)
        self.n_frames = T
        self.legacy_slip = bool(getattr(opt, "legacy_slip", False))
        if self.legacy_slip and T != 3:
            raise ValueError("legacy_slip only exists for seq_len 3")
        ncls = opt.num_classes
        for name in ("pprev_transformer", "prev_transformer"):
            if hasattr(self, name):
                delattr(self, name)
        self.shared_ca = False
        self.frame_transformer = nn.ModuleList([
            nn.ModuleList([Encoder(EncoderLayer(16 * 2 ** i, 4 * 2 ** i,
                                                ncls * self.K_list[i] * (1 + 2 * (self.kernel_list[i] // 2)) ** 2), 3)
                           for i in range(3)]) for _ in range(T - 1)])
        self.cat_layer = nn.ModuleList([
            nn.Sequential(nn.Linear(16 * 2 ** i * T, 32 * 2 ** i * T), nn.ReLU(), nn.Linear(32 * 2 ** i * T, 16 * 2 ** i))
            for i in range(6)])
        # levels 3-5 have no window size of their own in the three-frame class: same defaults as the two-frame model
        self.kernel_list = self.kernel_list + [1, 1, 1]
        self.scale_list = (4, 2, 1, 1 / 2, 1 / 4, 1 / 8)

    def forward(self, x, pre_imgs=None, pre_hms=None, repro_hm=None, pre_hm_clss=None, repro_hm_cls=None):
        """x [B,3,H,W]; pre_imgs / pre_hms / pre_hm_clss: sequences of T-1 tensors, OLDEST first; -> [{'hm','reg','wh','tracking'}]."""
        n = self.n_frames - 1
        if any(t is None for t in (pre_imgs, pre_hms, repro_hm, pre_hm_clss, repro_hm_cls)) or \
                not (len(pre_imgs) == len(pre_hms) == len(pre_hm_clss) == n):
            raise ValueError("the %d-frame model needs %d previous images / prior maps / class priors" % (n + 1, n))
        if self.training:
            raise RuntimeError("the HIP path is inference-only: call model.eval()")
        return [self.engine().forward(x, list(pre_imgs), list(pre_hms), repro_hm, list(pre_hm_clss), repro_hm_cls)]

    # 'f32' (default) or 'bf16': engine_seq_bf16.EngineSeqBF16 (same interface; also read from opt.compute_dtype)
    @property
    def engine_dtype(self):
        return getattr(self, "_engine_dtype", getattr(self.opt, "compute_dtype", "f32"))

    @engine_dtype.setter
    def engine_dtype(self, v):
        if v not in ("f32", "bf16", "f16", "f32x3"):
            raise ValueError("engine_dtype must be 'f32', 'bf16', 'f16' or 'f32x3'")
        object.__setattr__(self, "_engine_dtype", v)
        self._engine = None

    def engine(self):
        if self._engine is None:
            if self.engine_dtype == "f32x3":
                from ..engine_x3 import EngineSeqX3
                self._engine = EngineSeqX3(self)
            elif self.engine_dtype in ("bf16", "f16"):
                from ..engine_seq_bf16 import EngineSeqBF16, EngineSeqF16
                self._engine = (EngineSeqF16 if self.engine_dtype == "f16" else EngineSeqBF16)(self)
            else:
                from ..engine_seq import EngineSeq
                self._engine = EngineSeq(self)
        return self._engine

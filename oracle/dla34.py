"""Oracle: DLA-34 two-stem backbone and the DCN up-sampling neck, functional over a state_dict.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Follows (restates, does not copy) the reference's
  * stems + levels      sgtapose/lib/model/networks/dla.py:234-337 (DLA), :354-363 (dla34)
  * BasicBlock / Root / Tree                                dla.py:41-69, :157-175, :178-231
  * DeformConv (= DCN + BN + ReLU)                          dla.py:538-550
  * IDAUp / DLAUp                                           dla.py:552-577, :581-606
All BatchNorms are evaluated in inference mode (the detector calls model.eval(),
sgtapose/lib/sgta_detector.py:54), eps 1e-5.  ``sd`` maps the reference's parameter names
(e.g. ``base.level2.tree1.conv1.weight``) to tensors; ``p`` is the key prefix.
"""
import torch
import torch.nn.functional as F

from .dcnv2 import dcn_v2_forward

DLA34_LEVELS = (1, 1, 1, 2, 2, 1)
DLA34_CHANNELS = (16, 32, 64, 128, 256, 512)


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _conv(x, sd, p, stride=1, pad=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=pad)


def _stem(x, sd, p):
    """conv7x7 + BN + ReLU  (dla.py:260-270)."""
    return F.relu(_bn(_conv(x, sd, p + ".0", 1, 3), sd, p + ".1"))


def _conv_level(x, sd, p, stride):
    """_make_conv_level with one conv (dla.py:302-312): conv3x3 + BN + ReLU."""
    return F.relu(_bn(_conv(x, sd, p + ".0", stride, 1), sd, p + ".1"))


def _basic_block(x, residual, sd, p, stride):
    """dla.py:55-69: conv-bn-relu-conv-bn, add residual, relu."""
    out = F.relu(_bn(_conv(x, sd, p + ".conv1", stride, 1), sd, p + ".bn1"))
    out = _bn(_conv(out, sd, p + ".conv2", 1, 1), sd, p + ".bn2")
    return F.relu(out + (x if residual is None else residual))


def _root(xs, sd, p):
    """dla.py:167-175 with residual=False (dla34): 1x1 conv over the channel concat, BN, ReLU."""
    return F.relu(_bn(_conv(torch.cat(xs, 1), sd, p + ".conv"), sd, p + ".bn"))


def _tree(x, sd, p, levels, cin, cout, stride, level_root, children=None):
    """dla.py:218-231.  ``children`` accumulates the extra inputs of the innermost Root."""
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
    if levels == 1:
        # only a 1-level tree hands ``residual`` to a BasicBlock; a deeper tree passes it to a
        # sub-tree that immediately overwrites it (dla.py:221-224), so it is not evaluated here.
        residual = _bn(_conv(bottom, sd, p + ".project.0"), sd, p + ".project.1") \
            if cin != cout else bottom
    if level_root:
        children.append(bottom)
    if levels == 1:
        x1 = _basic_block(x, residual, sd, p + ".tree1", stride)
        x2 = _basic_block(x1, None, sd, p + ".tree2", 1)
        return _root([x2, x1] + children, sd, p + ".root")
    x1 = _tree(x, sd, p + ".tree1", levels - 1, cin, cout, stride, False)
    children.append(x1)
    return _tree(x1, sd, p + ".tree2", levels - 1, cout, cout, 1, False, children)


def dla34_backbone(img, hm, sd, p="base"):
    """DLA.forward with x=None (dla.py:324-335): both frames use pre_img_layer + pre_hm_layer.
    img [B,3,H,W], hm [B,1,H,W] -> list of 6 maps (16@H .. 512@H/32)."""
    x = _stem(img, sd, p + ".pre_img_layer") + _stem(hm, sd, p + ".pre_hm_layer")
    ys = []
    x = _conv_level(x, sd, p + ".level0", 1)
    ys.append(x)
    x = _conv_level(x, sd, p + ".level1", 2)
    ys.append(x)
    ch = DLA34_CHANNELS
    for lv in range(2, 6):
        x = _tree(x, sd, "%s.level%d" % (p, lv), DLA34_LEVELS[lv], ch[lv - 1], ch[lv], 2,
                  level_root=(lv > 2))
        ys.append(x)
    return ys


def _deform_conv(x, sd, p):
    """DeformConv (dla.py:538-550): DCN then BN then ReLU."""
    y = dcn_v2_forward(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"],
                       sd[p + ".conv.conv_offset_mask.weight"], sd[p + ".conv.conv_offset_mask.bias"])
    return F.relu(_bn(y, sd, p + ".actf.0"))


def _up(x, sd, p):
    """depthwise ConvTranspose2d(o,o,2f,stride f,pad f//2,groups=o,no bias) (dla.py:561-563)."""
    w = sd[p + ".weight"]
    f = w.shape[2] // 2
    return F.conv_transpose2d(x, w, None, stride=f, padding=f // 2, groups=w.shape[0])


def ida_up_forward(layers, startp, endp, sd, p):
    """IDAUp.forward (dla.py:571-577); mutates and returns ``layers``."""
    for i in range(startp + 1, endp):
        j = i - startp
        layers[i] = _up(_deform_conv(layers[i], sd, "%s.proj_%d" % (p, j)), sd, "%s.up_%d" % (p, j))
        layers[i] = _deform_conv(layers[i] + layers[i - 1], sd, "%s.node_%d" % (p, j))
    return layers


def dla_up_forward(layers, sd, p="dla_up", startp=2):
    """DLAUp.forward (dla.py:600-606) on the 6-level list; returns [64@/4, 128@/8, 256@/16, 512@/32]."""
    layers = list(layers)
    out = [layers[-1]]
    for i in range(len(layers) - startp - 1):
        ida_up_forward(layers, len(layers) - i - 2, len(layers), sd, "%s.ida_%d" % (p, i))
        out.insert(0, layers[-1])
    return out

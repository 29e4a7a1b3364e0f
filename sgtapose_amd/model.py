"""Model factory + checkpoint I/O with the reference's interface (drop-in boundary #1).

Mirrors ``create_model / load_model / save_model`` of sgtapose/lib/model/model.py:24-29, :43-103,
:105-114: same arguments, same checkpoint dictionary (``{'epoch', 'state_dict'[, 'optimizer']}``),
same tolerant loading rules (``module.`` prefix stripped, shape-mismatched or missing entries keep
the model's own values, ``strict=False``).  Only the live architecture is built:
``dlapawdl3new_34`` (sgtapose/scripts/inference_scripts.sh:12-13).
"""
from types import SimpleNamespace

import torch

from .networks.dla import DLAPlanAWindowL3New, DLAPlanAWindowSeq, DLAPlanAWindowThree

_network_factory = {"dlapawdl3new": DLAPlanAWindowL3New,          # the live model (sgtapose/lib/model/model.py:19)
                    "dlapawd3": DLAPlanAWindowThree,              # three-frame variant (model.py:18)
                    "dlapawdT": DLAPlanAWindowSeq}                # this build's T-frame generalisation (opt.seq_len, default 4)

DEFAULT_HEADS = {"hm": 7, "reg": 2, "wh": 2, "tracking": 2}        # opts_parallel.py:392-395


def default_opt(**over):
    """The ``opt`` fields the path reads at the inference script's values."""
    o = dict(head_kernel=3, prior_bias=-4.6, dla_node="dcn", load_model="x", pre_img=True, pre_hm=True,
             ct_modify=False, num_classes=7, pos_embed=True, model_output_list=False,
             device=torch.device("cuda"), phase="PlanA_win", zero_tracking=False, flip_test=False,
             out_thresh=0.001, shared_ca=False)       # max(track_thresh 0.001, out_thresh -1): opts_parallel.py:212, 263, 333
    for i, ks in enumerate((12, 6, 3, 1, 1, 1), 1):
        o["k_list_%d" % i] = 1
        o["ks%d" % i] = ks
    o.update(over)
    return SimpleNamespace(**o)


def default_head_conv(heads=None, width=256):
    return {h: [width] for h in (heads or DEFAULT_HEADS)}


def create_model(arch, head, head_conv, opt=None):
    num_layers = int(arch[arch.find("_") + 1:]) if "_" in arch else 0
    name = arch[:arch.find("_")] if "_" in arch else arch
    if name not in _network_factory:
        raise KeyError("architecture %r is outside the hot path; supported: dlapawdl3new_34, dlapawd3_34, dlapawdT_34" % arch)
    return _network_factory[name](num_layers, heads=head, head_convs=head_conv, opt=opt or default_opt())


def load_model(model, model_path, opt, optimizer=None):
    checkpoint = torch.load(model_path, map_location=lambda storage, loc: storage)
    print("loaded {}, epoch {}".format(model_path, checkpoint.get("epoch", "?")))
    src = checkpoint["state_dict"]
    state_dict = {}
    for k, v in src.items():
        state_dict[k[7:] if k.startswith("module") and not k.startswith("module_list") else k] = v
    own = model.state_dict()
    for k in list(state_dict):
        if k in own:
            if state_dict[k].shape != own[k].shape:
                print("Skip loading parameter {}, required shape {}, loaded shape {}.".format(
                    k, tuple(own[k].shape), tuple(state_dict[k].shape)))
                state_dict[k] = own[k]
        else:
            print("Drop parameter {}.".format(k))
    for k in own:
        if k not in state_dict:
            print("No param {}.".format(k))
            state_dict[k] = own[k]
    model.load_state_dict(state_dict, strict=False)
    return (model, optimizer, checkpoint.get("epoch", 0)) if optimizer is not None else model


def save_model(path, epoch, model, optimizer=None):
    sd = model.module.state_dict() if hasattr(model, "module") else model.state_dict()
    data = {"epoch": epoch, "state_dict": sd}
    if optimizer is not None:
        data["optimizer"] = optimizer.state_dict()
    torch.save(data, path)

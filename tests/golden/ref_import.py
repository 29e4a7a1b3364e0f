"""Load the reference's hot-path Python files in THIS container (never on the GPU box).

Used only by tests/golden/make_golden.py.  The reference package cannot be imported as a whole
(its __init__ pulls cv2 / ruamel / pyrr, rf_tools/LM.py loads a hard-coded .so path), so the
individual files are loaded by path under a scratch package with import stubs -- the recipe of
SURVEY.md Appendix C.  The un-vendored DCNv2 extension is replaced by the oracle's restatement.
"""
import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import torch

REF = os.environ.get("SGTA_REFERENCE", "/root/reference/sgtapose")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    if "reflib.model.networks.dla" in sys.modules:
        return sys.modules["reflib"].handles
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle.dcnv2 import OracleDCN

    for name in ("torchvision", "torchvision.models", "torchvision.transforms",
                 "torchvision.transforms.functional", "cv2", "webcolors"):
        _stub(name)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sg = _stub("sgtapose")
    root = _stub("reflib")
    _stub("reflib.model")
    _stub("reflib.model.networks")
    _stub("reflib.model.networks.DCNv2")
    _stub("reflib.model.networks.DCNv2.dcn_v2", DCN=OracleDCN)
    # no GPU here: the reference hard-codes .cuda() in _peaks_info / SoftArgmaxPavlo
    torch.Tensor.cuda = lambda self, *a, **k: self

    image_proc = _load("sgtapose.image_proc", os.path.join(REF, "image_proc.py"))
    sg.image_proc = image_proc
    base_model = _load("reflib.model.networks.base_model",
                       os.path.join(REF, "lib/model/networks/base_model.py"))
    dla = _load("reflib.model.networks.dla", os.path.join(REF, "lib/model/networks/dla.py"))
    mutils = _load("reflib.model.utils", os.path.join(REF, "lib/model/utils.py"))
    decode = _load("reflib.model.decode", os.path.join(REF, "lib/model/decode.py"))
    softmax = _load("reflib.spatial_softmax", os.path.join(REF, "spatial_softmax.py"))
    root.handles = SimpleNamespace(dla=dla, base_model=base_model, utils=mutils, decode=decode,
                                   image_proc=image_proc, spatial_softmax=softmax)
    return root.handles


def reference_opt(**over):
    """The opt fields the path reads, at the inference script's values."""
    o = dict(head_kernel=3, prior_bias=-4.6, dla_node="dcn", load_model="x", pre_img=True,
             pre_hm=True, ct_modify=False, num_classes=7, pos_embed=True, model_output_list=False,
             device=torch.device("cpu"), phase="PlanA_win", zero_tracking=False)
    for i, ks in enumerate((12, 6, 3, 1, 1, 1), 1):
        o["k_list_%d" % i] = 1
        o["ks%d" % i] = ks
    o.update(over)
    return SimpleNamespace(**o)


HEADS = {"hm": 7, "reg": 2, "wh": 2, "tracking": 2}
HEAD_CONV = {h: [256] for h in HEADS}


def build_reference_model(opt=None):
    ref = load_reference()
    opt = opt or reference_opt()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):      # the constructors print banners
        model = ref.dla.DLA_PlanAWindow_l3new(34, dict(HEADS), dict(HEAD_CONV), opt)
    return model.eval(), opt


def build_reference_model_three(shared_ca=False):
    """DLA_PlanAWindow_Three ('dlapawd3', sgtapose/lib/model/model.py:18): two previous frames."""
    ref = load_reference()
    opt = reference_opt(shared_ca=shared_ca)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.dla.DLA_PlanAWindow_Three(34, dict(HEADS), dict(HEAD_CONV), opt)
    return model.eval(), opt

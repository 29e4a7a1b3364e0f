"""CPU: the C-ABI library builds, loads, and exports every symbol include/sgta_hip.h declares;
host-side packing / model-factory logic (no kernel is launched here)."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from sgtapose_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib.load()


def header_symbols():
    src = open(os.path.join(REPO, "include", "sgta_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|long)\s+(sgta_\w+)\s*\(", src, flags=re.M)))


def test_header_symbols_exported_and_bound(lib):
    from sgtapose_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), "libsgta_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "no ctypes signature for %s" % s
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.sgta_version() >= 100


def test_header_argument_counts_match_ctypes_signatures():
    from sgtapose_amd import _lib
    src = open(os.path.join(REPO, "include", "sgta_hip.h")).read()
    for name, args in re.findall(r"^(?:int|long)\s+(sgta_\w+)\s*\(([^;]*?)\);", src, flags=re.M | re.S):
        n = 0 if args.strip() == "void" else len([a for a in args.split(",") if a.strip()])
        assert n == len(_lib.SIGNATURES[name]), name


def test_missing_library_fails_loudly(monkeypatch):
    from sgtapose_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsgta_hip.so")
    with pytest.raises(_lib.SgtaHipError):
        _lib.load()


def test_cpu_model_refuses_to_run_without_gpu():
    from sgtapose_amd._lib import SgtaHipError
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
    m = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(), default_opt()).eval()
    z = torch.zeros(1, 3, 32, 32)
    with pytest.raises(SgtaHipError):
        m(z, z, z[:, :1], z[:, :1], torch.zeros(1, 7, 8, 8), torch.zeros(1, 7, 8, 8))


def test_state_dict_keys_shapes_and_aliasing_match_reference():
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt
    from util import model_keys
    m = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(), default_opt())
    sd = m.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == model_keys()
    for i in range(3):
        p = "transformer.%d.layers." % i
        assert sd[p + "0.linear1.weight"].data_ptr() == sd[p + "2.linear1.weight"].data_ptr()
    assert float(sd["hm.2.bias"][0]) == pytest.approx(-4.6)


def test_checkpoint_roundtrip_and_tolerant_load(tmp_path):
    from sgtapose_amd.model import DEFAULT_HEADS, create_model, default_head_conv, default_opt, load_model, save_model
    from util import golden_state_dict
    m = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(), default_opt())
    m.load_state_dict(golden_state_dict(), strict=True)
    path = str(tmp_path / "model_1.pth")
    save_model(path, 1, m)
    ck = torch.load(path)
    assert set(ck) == {"epoch", "state_dict"}
    # reference-style checkpoint: DataParallel prefix, one wrong-shaped and one unknown entry
    sd = {"module." + k: v for k, v in ck["state_dict"].items()}
    sd["module.hm.2.bias"] = torch.zeros(3)
    sd["module.not_a_param"] = torch.zeros(1)
    torch.save({"epoch": 7, "state_dict": sd}, path)
    m2 = create_model("dlapawdl3new_34", DEFAULT_HEADS, default_head_conv(), default_opt())
    load_model(m2, path, default_opt())
    a, b = m.state_dict(), m2.state_dict()
    assert torch.equal(a["base.level2.tree1.conv1.weight"], b["base.level2.tree1.conv1.weight"])
    assert float(b["hm.2.bias"][0]) == pytest.approx(-4.6)          # wrong-shaped entry skipped


def test_packed_conv_layouts():
    from sgtapose_amd.ops import PackedConv
    w = torch.randn(5, 3, 3, 3)
    bn = (torch.rand(5) + 0.5, torch.randn(5), torch.randn(5), torch.rand(5) + 0.5)
    pc = PackedConv(w, None, bn, stride=2)
    assert pc.wt.shape == (2, 32, 16) and pc.Kpad == 32 and pc.Cout_pad == 32
    s = bn[0] / torch.sqrt(bn[3] + 1e-5)
    flat = pc.unpacked()
    assert torch.allclose(flat[:27, :5], (w * s.view(-1, 1, 1, 1)).reshape(5, 27).t())
    assert torch.allclose(pc.bias[:5], bn[1] - bn[2] * s)
    assert float(flat[27:].abs().sum()) == 0 and float(flat[:, 5:].abs().sum()) == 0
    tab = pc.ktab(10, 12)
    assert tab.shape == (32, 2) and tab.dtype == torch.int32
    assert tab[13].tolist() == [1 * 120 + 1 * 12 + 1, 4] and tab[27].tolist() == [0, 63]
    pd = PackedConv(torch.randn(4, 16, 3, 3), torch.randn(4), tap_major=True)
    assert pd.wt.shape == (9, 32, 16)
    wd = pd.unpacked()
    assert float(wd[2 * 16 + 5, 3]) == float(pd_w(pd, 3, 5, 0, 2))


def pd_w(pd, co, c, ky, kx):
    return pd._w_folded[co, c, ky, kx]


def test_dcn_signature_and_guards():
    from sgtapose_amd.dcn_v2 import DCN
    d = DCN(64, 32, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    assert sorted(k for k, _ in d.named_parameters()) == ["bias", "conv_offset_mask.bias", "conv_offset_mask.weight", "weight"]
    assert float(d.conv_offset_mask.weight.abs().sum()) == 0
    with pytest.raises(ValueError):
        DCN(64, 32, kernel_size=3, stride=2, padding=1)


def test_three_frame_state_dict_matches_reference():
    """dlapawd3_34 mirror: key order, shapes and layer aliasing of DLA_PlanAWindow_Three (both shared_ca settings)."""
    import torch
    from sgtapose_amd.model import create_model, default_head_conv, default_opt
    from util import model_keys
    heads = {"hm": 7, "reg": 2, "wh": 2, "tracking": 2}
    for shared, kf in ((False, "model_keys_three.json"), (True, "model_keys_three_shared.json")):
        m = create_model("dlapawd3_34", heads, default_head_conv(heads), default_opt(shared_ca=shared, device=torch.device("cpu")))
        sd = m.state_dict()
        assert [[k, list(v.shape)] for k, v in sd.items()] == model_keys(kf)
        a, b = sd["prev_transformer.0.layers.0.linear1.weight"], sd["prev_transformer.0.layers.2.linear1.weight"]
        assert a.data_ptr() == b.data_ptr()
        assert ("pprev_transformer.0.layers.0.linear1.weight" in sd) == (not shared)

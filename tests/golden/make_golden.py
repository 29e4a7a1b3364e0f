#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python files.

Run in the build container only (needs /root/reference):

    python -O tests/golden/make_golden.py

``-O`` disables the reference's ``assert H == W`` (dla.py:931) so the non-square (480x640) vectors
can be produced; nothing else depends on asserts.  torch runs with ONE thread: torch's CPU
``index_put_`` is only deterministic for duplicate indices when serial (last write wins), and the
reference's scatter-back (dla.py:1012-1014) hits duplicates on every frame (overlapping / clamped
windows).  That serial behaviour is the canonical semantics of this build.

Fixtures are DATA ONLY: seeds + the reference's outputs (inputs and weights are regenerated from the
seeds by sgtapose_amd/synthetic.py on whatever machine runs the tests).  The un-vendored DCNv2
extension is replaced by oracle/dcnv2.py (parity unpinned for that operator, see oracle/__init__.py).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

from ref_import import build_reference_model, build_reference_model_three, load_reference  # noqa: E402
from sgtapose_amd.synthetic import THREE_ARGS, fill_state_dict_, make_inputs, make_inputs_three  # noqa: E402
import cases  # noqa: E402

torch.set_num_threads(1)
ARGS = ("x", "pre_img", "pre_hm", "repro_hm", "pre_hm_cls", "repro_hm_cls")


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("wrote %-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


def fullnet(model, tag, B, H, W, seed, first_frame=False, extra=False, probe=False):
    ref = load_reference()
    inp = make_inputs(B, H, W, seed=seed, first_frame=first_frame)
    opt = type("O", (), {"zero_tracking": False})()
    with torch.no_grad():
        out = model(*[inp[k] for k in ARGS])[-1]
        hm_logits = out["hm"].clone()
        out["hm"] = out["hm"].sigmoid_()
        dets = []
        for b in range(B):                       # the reference decodes batch element 0 only
            ob = {k: v[b:b + 1].clone() for k, v in out.items()}
            ob["pre_inds"] = None
            dets.append(ref.decode.dream_generic_decode(ob, K=7, opt=opt))
    arrs = dict(meta=np.array([B, H, W, seed, int(first_frame)]), hm_logits=hm_logits,
                reg=out["reg"], tracking=out["tracking"])
    if probe:
        # large batches: the heads at 4096 seeded probe pixels per sample + per-map statistics instead of full maps
        g = torch.Generator().manual_seed(seed)
        pix = torch.randint(0, (H // 4) * (W // 4), (4096,), generator=g)
        flat = lambda t: t.reshape(B, t.shape[1], -1)             # noqa: E731
        arrs = dict(meta=arrs["meta"], probe_pix=pix, hm_probe=flat(hm_logits)[:, :, pix],
                    reg_probe=flat(out["reg"])[:, :, pix], tracking_probe=flat(out["tracking"])[:, :, pix],
                    hm_map_max=flat(hm_logits).max(2)[0], hm_map_mean=flat(hm_logits).double().mean(2))
    for k in dets[0]:
        arrs["det_" + k] = torch.cat([d[k] for d in dets], 0)
    if extra:
        # intermediate maps for kernel-level debugging (small resolutions only)
        with torch.no_grad():
            xp = model.base(pre_img=inp["pre_img"], pre_hm=inp["pre_hm"])
            xc = model.base(pre_img=inp["x"], pre_hm=inp["repro_hm"])
            feats, pf, cf = model.imgpre2feats(*[inp[k] for k in ARGS])
        for i in (2, 3, 4, 5):
            arrs["pre_l%d" % i] = xp[i]
            arrs["cur_l%d" % i] = xc[i]
        arrs["feat"] = feats[0]
        arrs["pre_fid0"] = pf
        arrs["cur_fid0"] = cf
    save("fullnet_%s.npz" % tag, **arrs)


def window_ops(model):
    """get_topk_index / get_topk_features_scale / substitute_topk_features_scale (dla.py:898-1018).
    Inputs come from cases.window_case(); only the reference's outputs are stored."""
    dla = load_reference().dla
    case = cases.window_case()
    pi, ci = dla.get_topk_index(case["pre_cls"], case["cur_cls"], 1)
    arrs = dict(pre_idx=pi, cur_idx=ci)
    for i, (pre_f, cur_f, tok) in enumerate(case["levels"]):
        sc, ks = model.scale_list[i], model.kernel_list[i]
        pk, _, pf = dla.get_topk_features_scale(pre_f, pi, scale_num=sc, kernel=ks)
        cq, cb, cf = dla.get_topk_features_scale(cur_f, ci, scale_num=sc, kernel=ks)
        with torch.no_grad():
            sub = dla.substitute_topk_features_scale(tok, cur_f, cb, cf, model.cat_layer[i])
        B, C, H, W = cur_f.shape
        touched = torch.zeros(B, H * W, dtype=torch.bool)
        touched.scatter_(1, cf, True)
        same = (sub == cur_f).reshape(B, C, -1).all(1)
        assert bool((same | touched).all()), "substitute changed an untouched pixel"
        sub_tok = torch.gather(sub.reshape(B, C, -1), 2, cf.view(B, 1, -1).expand(B, C, -1)).permute(0, 2, 1)
        arrs.update({"pre_fid%d" % i: pf, "cur_fid%d" % i: cf, "key%d" % i: pk, "sub_tok%d" % i: sub_tok})
    pin, _ = dla.get_topk_index(case["ns_cls"], case["ns_cls"], 1)
    k1, _, f1 = dla.get_topk_features_scale(case["ns_f"], pin, scale_num=1, kernel=3)
    k2, _, f2 = dla.get_topk_features_scale(case["ns_f2"], pin, scale_num=0.5, kernel=1)
    arrs.update(ns_idx=pin, ns_fid=f1, ns_key=k1, ns_fid2=f2, ns_key2=k2)
    save("window_ops.npz", **arrs)


def encoder_ops(model):
    """TransformerEncoder (3 x one shared layer) incl. MHCA_ein with non-zero pos_embed."""
    arrs = {}
    for i, (q, k) in enumerate(cases.encoder_case()):
        with torch.no_grad():
            lay = model.transformer[i].layers[0]
            arrs["mhca%d" % i] = lay.cross_attn(q, k, k)
            arrs["layer%d" % i] = lay(q, k, k)
            arrs["enc%d" % i] = model.transformer[i](q, k, k)
    save("encoder_ops.npz", **arrs)


def decode_ops():
    ref = load_reference()
    hm, reg, trk = cases.decode_case()
    arrs = {}
    opt = type("O", (), {"zero_tracking": False})()
    for b in range(2):
        peaks = ref.image_proc.peaks_from_belief_maps(hm[b], offset_due_to_upsampling=0.4395)
        arrs["peaks_count_%d" % b] = np.array([len(p) for p in peaks])
        arrs["peaks_flat_%d" % b] = np.array([[p[0], p[1], p[2]] for pl in peaks for p in pl],
                                             dtype=np.float64).reshape(-1, 3)
        out = {"hm": hm[b:b + 1].clone(), "reg": reg[b:b + 1], "tracking": trk[b:b + 1], "pre_inds": None}
        dets = ref.decode.dream_generic_decode(out, K=7, opt=opt)
        for k, v in dets.items():
            arrs["det%d_%s" % (b, k)] = v
    sa = ref.spatial_softmax.SoftArgmaxPavlo.__new__(ref.spatial_softmax.SoftArgmaxPavlo)
    torch.nn.Module.__init__(sa)                 # its __init__ hard-codes .cuda(); set fields by hand
    sa.avgpool = torch.nn.AvgPool2d(7, stride=1, padding=3)
    sa.beta = torch.ones(7) * 25.0
    arrs["softargmax"] = sa(hm)
    save("decode_ops.npz", **arrs)


def prepost_ops():
    """Prior rendering + post-processing from the reference's own utilities.py,
    lib/utils/post_process.py and lib/sgta_detector.py.  cv2 is absent, so the affine matrices come from
    oracle.prepost.get_affine_transform (unpinned) and are INPUTS of both sides; the reference's
    ``get_affine_transform`` inside dream_generic_post_process is pointed at the same matrices."""
    import ref_import
    from types import SimpleNamespace
    from oracle import prepost as op
    load_reference()
    ref_import._stub("ruamel")
    ref_import._stub("ruamel.yaml", YAML=object)
    ref_import._stub("progress")
    ref_import._stub("progress.bar", Bar=object)
    REF = ref_import.REF
    util = ref_import._load("reflib.utilities", os.path.join(REF, "utilities.py"))
    sys.modules["sgtapose"].utilities = util
    ref_import._stub("reflib.utils")
    image = ref_import._load("reflib.utils.image", os.path.join(REF, "lib/utils/image.py"))
    ref_import._load("reflib.utils.ddd_utils", os.path.join(REF, "lib/utils/ddd_utils.py"))
    pp = ref_import._load("reflib.utils.post_process", os.path.join(REF, "lib/utils/post_process.py"))
    ref_import._stub("reflib.model.model", create_model=None, load_model=None, create_dream_hourglass=None)
    ref_import._stub("reflib.utils.debugger", Debugger=object)
    ref_import._stub("reflib.utils.tracker", Tracker=object)
    det = ref_import._load("reflib.sgta_detector", os.path.join(REF, "lib/sgta_detector.py"))

    cs = cases.prepost_case()
    K, B = cs["K"], len(cs["raws"])
    arrs = {}
    t_in = np.stack([op.get_affine_transform(cs["c"][b], cs["s"][b], 0, [cs["inp_w"], cs["inp_h"]]) for b in range(B)])
    t_out = np.stack([op.get_affine_transform(cs["c"][b], cs["s"][b], 0, [cs["out_w"], cs["out_h"]]) for b in range(B)])
    t_inv = np.stack([op.get_affine_transform(cs["c"][b], cs["s"][b], 0, (cs["out_w"], cs["out_h"]), inv=1)
                      for b in range(B)])
    arrs.update(trans_input=t_in, trans_output=t_out, trans_inv=t_inv)
    hms, clss, clips = [], [], []
    for b, (rw, rh) in enumerate(cs["raws"]):
        for nm, kps in (("prev", cs["prev"][b]), ("next", cs["nxt"][b])):
            src = kps if cs["has_prior"][b] else None
            hms.append(util.get_prev_hm_wo_noise(src, t_in[b], cs["inp_w"], cs["inp_h"], rw, rh))
            clss.append(util.get_prev_hm_wo_noise_cls(src, np.zeros((K, 3)), t_out[b], cs["out_w"], cs["out_h"], rw, rh))
            clips.append(util.affine_transform_and_clip(kps, t_in[b], cs["inp_w"], cs["inp_h"], rw, rh))
    arrs["hm"] = np.stack(hms).reshape(B, 2, cs["inp_h"], cs["inp_w"])
    arrs["hm_cls"] = np.stack(clss).reshape(B, 2, K, cs["out_h"], cs["out_w"])
    arrs["clip"] = np.stack(clips).reshape(B, 2, K, 2)
    # post-processing: the reference function with its affine matrix pinned to ours
    mats = iter([t_inv[b] for b in range(B)])
    pp.get_affine_transform = lambda *a, **k: next(mats)
    opt = SimpleNamespace(out_thresh=cs["out_thresh"], num_classes=K)
    dets = cs["dets"]
    out = pp.dream_generic_post_process(opt, dets, [None] * B, [None] * B, cs["out_h"], cs["out_w"], K)
    fake = SimpleNamespace(opt=opt, is_ct=True)
    valid = np.zeros((B, K), np.uint8)
    fields = {k: np.zeros((B, K, 2), np.float32) for k in ("ct", "ct_wreg", "reg", "tracking")}
    final = np.zeros((B, K, 2), np.float64)
    for b in range(B):
        for item in out[b]:
            j = item["class"] - 1
            valid[b, j] = 1
            for k in fields:
                fields[k][b, j] = item[k]
        final[b] = det.SGTADetector._get_final_kps(fake, out[b])
    arrs.update(pp_valid=valid, final_kps=final, **{"pp_" + k: v for k, v in fields.items()})
    save("prepost_ops.npz", **arrs)


def three_frames():
    """DLA_PlanAWindow_Three: state_dict keys + head logits of the reference model."""
    for shared, tag in ((False, ""), (True, "_shared")):
        model, _ = build_reference_model_three(shared)
        sd = model.state_dict()
        keys = [[k, list(v.shape)] for k, v in sd.items()]
        with open(os.path.join(HERE, "model_keys_three%s.json" % tag), "w") as f:
            json.dump({"arch": "dlapawd3_34", "shared_ca": shared, "n_keys": len(keys), "keys": keys}, f)
        fill_state_dict_(sd, seed=0)
        for name, B, H, W, seed in (("128", 2, 128, 128, 417), ("96x160", 1, 96, 160, 418)):
            if shared and name != "128":
                continue
            inp = make_inputs_three(B, H, W, seed=seed)
            with torch.no_grad():
                out = model(*[inp[k] for k in THREE_ARGS])[-1]
            save("fullnet_three%s_%s.npz" % (tag, name), meta=np.array([B, H, W, seed, int(shared)]),
                 **{k: v for k, v in out.items()})


def pnp_ops():
    """Gauss-Newton refinement of rf_tools/LM.py (fun / dfun / GN: the python twin of the C ``LM``), run from the
    reference's own file.  LM.py loads a .so from a hard-coded absolute path at import and imports cv2: both are stubbed
    (neither is touched by fun / dfun / GN)."""
    import ctypes
    import ref_import
    from oracle import pnp as opnp
    load_reference()
    real = ctypes.cdll.LoadLibrary
    ctypes.cdll.LoadLibrary = lambda path: None
    try:
        lm = ref_import._load("reflib.rf_tools_LM", os.path.join(ref_import.REF, "rf_tools/LM.py"))
    finally:
        ctypes.cdll.LoadLibrary = real
    K, x3d, cs = cases.pnp_case()
    arrs = {}
    for i, c in enumerate(cs):
        v = c["valid"]
        R = opnp.rodrigues(c["rvec"])
        x2d = opnp.project(K, R, c["t"], x3d) + c["noise"]
        q = opnp.quat_xyzw_from_rvec(c["rvec"] + 0.02)                     # a start near the pose, as PnP provides
        value0 = np.array([q[3], q[0], q[1], q[2], *(c["t"] + 0.01)])
        w = [[1.0, 1.0]] * int(v.sum())
        args = (x2d[v].tolist(), x3d[v].tolist(), w, K)
        arrs["f_%d" % i] = lm.fun(value0, *args)
        arrs["df_%d" % i] = lm.dfun(value0, *args)
        arrs["gn_%d" % i] = lm.GN(value0, *args)
        arrs["x2d_%d" % i] = x2d
        arrs["value0_%d" % i] = value0
    save("pnp_ops.npz", **arrs)


def peaked():
    """Second weight set (peaked heat-maps): reference network -> sigmoid ->
    dream_generic_decode; the decode dictionaries hold real key-points (scores > 0)."""
    model, _ = build_reference_model()
    fill_state_dict_(model.state_dict(), seed=0, peaked=True)
    fullnet(model, "peaked_128x160", 4, 128, 160, 11)
    fullnet(model, "peaked_480x640", 4, 480, 640, 317, probe=True)


def main():
    if "--three-only" in sys.argv:
        three_frames()
        return
    if "--peaked-only" in sys.argv:
        peaked()
        return
    if "--pnp-only" in sys.argv:
        pnp_ops()
        return
    if "--prepost-only" in sys.argv:
        prepost_ops()
        return
    model, opt = build_reference_model()
    sd = model.state_dict()
    keys = [[k, list(v.shape)] for k, v in sd.items()]
    with open(os.path.join(HERE, "model_keys.json"), "w") as f:
        json.dump({"arch": "dlapawdl3new_34", "n_keys": len(keys), "keys": keys}, f)
    print("wrote model_keys.json (%d keys)" % len(keys))
    fill_state_dict_(sd, seed=0)
    window_ops(model)
    encoder_ops(model)
    decode_ops()
    prepost_ops()
    three_frames()
    fullnet(model, "128", 2, 128, 128, 317, extra=True)
    fullnet(model, "128_first", 1, 128, 128, 318, first_frame=True)
    fullnet(model, "96x160", 1, 96, 160, 319)
    fullnet(model, "480", 1, 480, 480, 320)
    fullnet(model, "480x640", 1, 480, 640, 321)
    peaked()
    pnp_ops()


if __name__ == "__main__":
    main()

"""Seeded synthetic weights and frame-pair inputs (no datasets / checkpoints are reachable).

* ``fill_state_dict_`` overwrites every tensor of a ``dlapawdl3new_34`` state_dict with values that
  depend only on (seed, parameter name, shape) -- the same recipe rebuilds identical weights in
  the build container (where the golden vectors are made from the reference) and on the GPU box.
  The deformable-conv ``conv_offset_mask`` gets a non-zero init so the bilinear gather is real
  (the reference's zero init makes DCN a plain conv), ``pos_embed`` is non-zero so the structure
  prior bias matters.
* ``make_inputs`` renders the six network inputs of one frame-pair batch.  Prior heat-maps follow the
  reference's rule (sgtapose/utilities.py:800-821, :1045-1098): 9x9 Gaussian, sigma 2, integer
  centre, max-combined, dropped if the patch touches the border.
"""
import zlib

import numpy as np
import torch


def _gen(seed, name):
    g = torch.Generator()
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return g


def _uniform(shape, lo, hi, g):
    return torch.rand(shape, generator=g) * (hi - lo) + lo


PEAKED_HM_GAIN = 30.0      # second weight set (the 'hm' output conv scaled and its bias lowered
PEAKED_HM_BIAS = -8.0      # so that the random-init heat-maps carry isolated peaks the decode can find


def fill_state_dict_(sd, seed=0, peaked=False):
    """In-place, deterministic re-initialisation of a state_dict (keys follow the reference).  ``peaked`` = the
    second weight set: identical except hm.2.weight * PEAKED_HM_GAIN and hm.2.bias =
    PEAKED_HM_BIAS (random-init logits are flat around the -4.6 prior bias = the decode's 0.01 threshold, so the
    default set yields no key-points; this one does, for network -> decode -> post-process parity)."""
    done = set()
    for name in sorted(sd.keys()):
        t = sd[name]
        if t.data_ptr() in done:                 # transformer.i.layers.{0,1,2} alias one module
            continue
        done.add(t.data_ptr())
        # aliased keys differ only in 'layers.N' -> seed on the canonical 'layers.0' name
        canon = name.replace(".layers.1.", ".layers.0.").replace(".layers.2.", ".layers.0.")
        g = _gen(seed, canon)
        leaf = canon.rsplit(".", 1)[-1]
        with torch.no_grad():
            if leaf == "num_batches_tracked":
                t.zero_()
            elif leaf == "running_mean":
                t.copy_(_uniform(t.shape, -0.1, 0.1, g))
            elif leaf == "running_var":
                t.copy_(_uniform(t.shape, 0.6, 1.4, g))
            elif leaf == "pos_embed":
                t.copy_(torch.randn(t.shape, generator=g) * 0.5)
            elif "conv_offset_mask" in canon:
                fan_in = t[0].numel() if t.dim() == 4 else 1
                if leaf == "weight":
                    t.copy_(torch.randn(t.shape, generator=g) * (1.2 / fan_in ** 0.5))
                else:
                    t.copy_(_uniform(t.shape, -0.3, 0.3, g))
            elif t.dim() == 1:                   # BN / LN affine, conv / linear biases
                if leaf == "weight":
                    t.copy_(_uniform(t.shape, 0.7, 1.3, g))
                elif canon.startswith("hm.2."):
                    t.fill_(PEAKED_HM_BIAS if peaked else -4.6)      # prior_bias (base_model.py:151-152)
                else:
                    t.copy_(_uniform(t.shape, -0.1, 0.1, g))
            elif ".up_" in canon:                # depthwise transposed conv: near-bilinear, perturbed
                k = t.shape[2]
                f = (k + 1) // 2
                c = (2 * f - 1 - f % 2) / (2.0 * f)
                a = 1 - (torch.arange(k, dtype=torch.float32) / f - c).abs()
                t.copy_((a[:, None] * a[None, :]).expand_as(t) * _uniform(t.shape, 0.9, 1.1, g))
            else:                                # conv [Co,Ci,kh,kw] / linear [N,K]: He-uniform
                fan_in = t[0].numel()
                b = (3.0 / fan_in) ** 0.5 * 1.2
                t.copy_(_uniform(t.shape, -b, b, g))
                if peaked and canon == "hm.2.weight":
                    t.mul_(PEAKED_HM_GAIN)
    return sd


def gaussian_patch(radius=4, sigma=2.0):
    ax = np.arange(-radius, radius + 1, dtype=np.float32)
    g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / (2 * sigma * sigma)).astype(np.float32)
    g[g < np.finfo(np.float32).eps * g.max()] = 0
    return g


def draw_prior(hm, cx, cy, radius=4):
    """max-combine one 9x9 Gaussian at the integer centre; dropped when it would touch the border."""
    H, W = hm.shape
    x, y = int(cx), int(cy)
    if x - radius >= 0 and x + radius + 1 < W and y - radius >= 0 and y + radius + 1 < H:
        sl = hm[y - radius:y + radius + 1, x - radius:x + radius + 1]
        np.maximum(sl, gaussian_patch(radius), out=sl)
    return hm


def make_keypoints(B, h, w, seed=317, n_kp=7):
    """previous / current keypoint centres on the h x w output grid: uniform in the central 80 %,
    prev -> cur shift ~ N(0, 2 px)."""
    rng = np.random.RandomState(seed)
    cur = np.stack([rng.uniform(0.1 * w, 0.9 * w, size=(B, n_kp)),
                    rng.uniform(0.1 * h, 0.9 * h, size=(B, n_kp))], axis=-1)
    pre = cur + rng.normal(0.0, 2.0, size=cur.shape)
    return pre.astype(np.float32), cur.astype(np.float32)


def make_inputs(B, H, W, seed=317, n_kp=7, first_frame=False):
    """Returns dict of CPU float32 tensors: x, pre_img [B,3,H,W]; pre_hm, repro_hm [B,1,H,W];
    pre_hm_cls, repro_hm_cls [B,n_kp,H/4,W/4]; plus the keypoint centres used ('kp_pre','kp_cur').
    ``first_frame=True`` gives all-zero priors (sgta_detector.py:164-166, :425-426)."""
    h, w = H // 4, W // 4
    g = torch.Generator()
    g.manual_seed(int(seed))
    x = torch.rand((B, 3, H, W), generator=g) * 2 - 1
    pre_img = torch.rand((B, 3, H, W), generator=g) * 2 - 1
    kp_pre, kp_cur = make_keypoints(B, h, w, seed, n_kp)
    pre_hm = np.zeros((B, 1, H, W), np.float32)
    repro_hm = np.zeros((B, 1, H, W), np.float32)
    pre_cls = np.zeros((B, n_kp, h, w), np.float32)
    repro_cls = np.zeros((B, n_kp, h, w), np.float32)
    if not first_frame:
        for b in range(B):
            for k in range(n_kp):
                draw_prior(pre_hm[b, 0], kp_pre[b, k, 0] * 4, kp_pre[b, k, 1] * 4)
                draw_prior(repro_hm[b, 0], kp_cur[b, k, 0] * 4, kp_cur[b, k, 1] * 4)
                draw_prior(pre_cls[b, k], kp_pre[b, k, 0], kp_pre[b, k, 1])
                draw_prior(repro_cls[b, k], kp_cur[b, k, 0], kp_cur[b, k, 1])
    return {"x": x, "pre_img": pre_img,
            "pre_hm": torch.from_numpy(pre_hm), "repro_hm": torch.from_numpy(repro_hm),
            "pre_hm_cls": torch.from_numpy(pre_cls), "repro_hm_cls": torch.from_numpy(repro_cls),
            "kp_pre": torch.from_numpy(kp_pre), "kp_cur": torch.from_numpy(kp_cur)}


THREE_ARGS = ("x", "ppre_img", "pre_img", "ppre_hm", "pre_hm", "repro_hm", "ppre_hm_cls", "pre_hm_cls", "repro_hm_cls")


def make_inputs_three(B, H, W, seed=317, n_kp=7):
    """Inputs of the three-frame variant (BaseModelPlanA_Three.forward, base_model.py:267-269) in THREE_ARGS order:
    the frame pair of ``make_inputs(seed)`` plus an older frame taken from ``make_inputs(seed + 1000)``."""
    cur = make_inputs(B, H, W, seed, n_kp)
    old = make_inputs(B, H, W, seed + 1000, n_kp)
    return {"x": cur["x"], "ppre_img": old["pre_img"], "pre_img": cur["pre_img"], "ppre_hm": old["pre_hm"],
            "pre_hm": cur["pre_hm"], "repro_hm": cur["repro_hm"], "ppre_hm_cls": old["pre_hm_cls"],
            "pre_hm_cls": cur["pre_hm_cls"], "repro_hm_cls": cur["repro_hm_cls"]}


def make_inputs_seq(B, H, W, T=4, seed=317, n_kp=7):
    """Inputs of the T-frame model (DLAPlanAWindowSeq.forward): dict(x, pre_imgs, pre_hms, repro_hm, pre_hm_clss,
    repro_hm_cls) with the T-1 previous frames OLDEST first.  The newest previous frame and the current one are the pair
    of ``make_inputs(seed)``; the j-th older frame comes from ``make_inputs(seed + 1000 * j)`` (T = 3 reproduces
    ``make_inputs_three``)."""
    cur = make_inputs(B, H, W, seed, n_kp)
    older = [make_inputs(B, H, W, seed + 1000 * j, n_kp) for j in range(T - 2, 0, -1)]
    frames = older + [cur]
    return {"x": cur["x"], "pre_imgs": [f["pre_img"] for f in frames], "pre_hms": [f["pre_hm"] for f in frames],
            "repro_hm": cur["repro_hm"], "pre_hm_clss": [f["pre_hm_cls"] for f in frames],
            "repro_hm_cls": cur["repro_hm_cls"]}


SEQ_ARGS = ("x", "pre_imgs", "pre_hms", "repro_hm", "pre_hm_clss", "repro_hm_cls")


def peaked_heatmaps(B, h, w, seed=317, n_kp=7, amp=0.9, floor=0.002):
    """Post-sigmoid-like heat-maps with one Gaussian blob (sigma 2) per class at a sub-pixel
    centre -- used for decode / keypoint parity since random-init logits are flat."""
    rng = np.random.RandomState(seed + 1)
    hm = np.full((B, n_kp, h, w), floor, np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    cx = rng.uniform(6, w - 7, size=(B, n_kp)).astype(np.float32)
    cy = rng.uniform(6, h - 7, size=(B, n_kp)).astype(np.float32)
    for b in range(B):
        for k in range(n_kp):
            hm[b, k] += amp * np.exp(-((xx - cx[b, k]) ** 2 + (yy - cy[b, k]) ** 2) / 8.0)
    return torch.from_numpy(hm), torch.from_numpy(np.stack([cx, cy], -1))

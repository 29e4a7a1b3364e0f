"""The steps either side of the network on the device: prior heat-map rendering
(``sgtapose/utilities.py:get_prev_hm_wo_noise(_cls)``) and detection post-processing
(``lib/utils/post_process.py:dream_generic_post_process``, ``SGTADetector._get_final_kps``), same names and
argument meaning as the reference, batched, with every array staying in HBM between steps.

``get_affine_transform`` is host code (a 2x3 matrix per frame): the reference ends it in
``cv2.getAffineTransform`` (lib/utils/image.py:73-76); OpenCV is not a dependency here, the 3-point solve is
done in fp64 with numpy (the method OpenCV documents).
"""
import numpy as np
import torch

from . import _lib

MISSING = -999.999 * 4           # SGTADetector._get_final_kps marker of an undetected key-point


def _get_3rd_point(a, b):
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def get_affine_transform(center, scale, rot, output_size, shift=None, inv=0):
    """lib/utils/image.py:45-78 (rot = 0, no shift: the only form the detector uses) -> 2x3 float64."""
    if rot != 0 or (shift is not None and np.any(np.asarray(shift) != 0)):
        raise NotImplementedError("the detector path only uses rot = 0 and shift = 0")
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale], dtype=np.float32)
    src_w, dst_w, dst_h = scale[0], output_size[0], output_size[1]
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = np.asarray(center, np.float32) + np.array([0, src_w * -0.5], np.float32)
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + np.array([0, dst_w * -0.5], np.float32)
    src[2:, :] = _get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = _get_3rd_point(dst[0, :], dst[1, :])
    a, b = (dst, src) if inv else (src, dst)
    m = np.zeros((6, 6), np.float64)
    r = np.zeros(6, np.float64)
    for i in range(3):
        m[i, 0:3] = (a[i, 0], a[i, 1], 1.0)
        m[i + 3, 3:6] = (a[i, 0], a[i, 1], 1.0)
        r[i], r[i + 3] = b[i, 0], b[i, 1]
    return np.linalg.solve(m, r).reshape(2, 3)


_TABLE = {}


def _gauss_table(dev):
    """float32(exp(-(dx^2+dy^2)/(2*2^2))) of the 9x9 patch, evaluated in fp64 like utilities.py:846-853."""
    t = _TABLE.get(dev)
    if t is None:
        y, x = np.ogrid[-4.0:5.0, -4.0:5.0]
        h = np.exp(-(x * x + y * y) / 8.0)
        h[h < np.finfo(h.dtype).eps * h.max()] = 0
        t = _TABLE[dev] = torch.from_numpy(h.astype(np.float32).reshape(81)).to(dev)
    return t


def _dev(a, dtype, dev):
    t = a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))
    return t.to(device=dev, dtype=dtype).contiguous()


def render_prior(kp_projs_raw, trans, map_w, map_h, raw_wh, has_prior=None, per_class=False, out=None,
                 device="cuda:0"):
    """Batched get_prev_hm_wo_noise (per_class=False -> [B,1,H,W]) / get_prev_hm_wo_noise_cls (True -> [B,K,h,w]).
    kp_projs_raw [B,K,2] raw-image coordinates (fp64, host or device), trans [B,2,3] = meta['trans_input'] /
    meta['trans_output'], raw_wh [B,2] = (meta['width'], meta['height']); has_prior[b] = False stands for the
    reference's ``kp_projs_raw is None`` (zero map)."""
    dev = torch.device(device)
    kps = _dev(kp_projs_raw, torch.float64, dev)
    B, K, _ = kps.shape
    tr = _dev(trans, torch.float64, dev)
    wh = _dev(raw_wh, torch.int32, dev)
    hp = torch.ones(B, dtype=torch.uint8, device=dev) if has_prior is None else _dev(has_prior, torch.uint8, dev)
    if tuple(tr.shape) != (B, 2, 3) or tuple(wh.shape) != (B, 2) or tuple(hp.shape) != (B,):
        raise ValueError("trans must be [B,2,3], raw_wh [B,2], has_prior [B]")
    if not bool(torch.isfinite(kps).all()):
        raise ValueError("key-point coordinates must be finite")
    C = K if per_class else 1
    if out is None:
        out = torch.empty(B, C, int(map_h), int(map_w), dtype=torch.float32, device=dev)
    elif tuple(out.shape) != (B, C, int(map_h), int(map_w)) or not out.is_contiguous() or out.dtype != torch.float32:
        raise ValueError("out must be a contiguous float32 [B,%d,%d,%d]" % (C, map_h, map_w))
    _lib.call("sgta_render_priors", kps.data_ptr(), tr.data_ptr(), wh.data_ptr(), hp.data_ptr(),
              _gauss_table(dev).data_ptr(), out.data_ptr(), B, K, int(map_h), int(map_w), int(per_class),
              torch.cuda.current_stream(dev).cuda_stream)
    return out


def get_prev_hm_wo_noise(kp_projs_raw, trans_input, input_w, input_h, raw_wh, has_prior=None, **kw):
    """utilities.py:1045-1057, batched -> [B,1,input_h,input_w] on the device."""
    return render_prior(kp_projs_raw, trans_input, input_w, input_h, raw_wh, has_prior, False, **kw)


def get_prev_hm_wo_noise_cls(kp_projs_raw, trans_output, out_w, out_h, raw_wh, has_prior=None, **kw):
    """utilities.py:1086-1098, batched -> [B,K,out_h,out_w] on the device."""
    return render_prior(kp_projs_raw, trans_output, out_w, out_h, raw_wh, has_prior, True, **kw)


def post_process_device(dets, trans_inv, out_thresh):
    """dream_generic_post_process + _get_final_kps on the device.  ``dets`` = the device dict of
    decode.dream_generic_decode; trans_inv [B,2,3] = get_affine_transform(c, s, 0, (out_w, out_h), inv=1).
    -> dict(valid u8 [B,K], ct / ct_wreg / reg / tracking f32 [B,K,2], final_kps f64 [B,K,2])."""
    scores = dets["scores"]
    dev = scores.device
    B, K = scores.shape
    tr = _dev(trans_inv, torch.float32, dev)
    if tuple(tr.shape) != (B, 2, 3):
        raise ValueError("trans_inv must be [B,2,3]")
    cts = dets["cts"].to(torch.int64).contiguous()
    cw = dets["cts_wreg"].reshape(B, K, 2).float().contiguous()
    rg = dets["regs"].reshape(B, K, 2).float().contiguous()
    trk = dets["tracking"].reshape(B, K, 2).float().contiguous() if "tracking" in dets else None
    f32 = lambda: torch.empty(B, K, 2, dtype=torch.float32, device=dev)        # noqa: E731
    out = dict(valid=torch.empty(B, K, dtype=torch.uint8, device=dev), ct=f32(), ct_wreg=f32(), reg=f32(),
               final_kps=torch.empty(B, K, 2, dtype=torch.float64, device=dev))
    if trk is not None:
        out["tracking"] = f32()
    _lib.call("sgta_post_process", scores.float().contiguous().data_ptr(), cts.data_ptr(), cw.data_ptr(),
              rg.data_ptr(), None if trk is None else trk.data_ptr(), tr.data_ptr(), float(out_thresh),
              out["valid"].data_ptr(), out["ct"].data_ptr(), out["ct_wreg"].data_ptr(), out["reg"].data_ptr(),
              None if trk is None else out["tracking"].data_ptr(), out["final_kps"].data_ptr(), B, K,
              torch.cuda.current_stream(dev).cuda_stream)
    return out


def dream_generic_post_process(dets_np, pp, num_classes):
    """The reference's return value (list per sample of lists of dicts, post_process.py:93-175) assembled from the
    host copies of ``post_process_device``'s outputs (``pp``) and of the decode dict (``dets_np``)."""
    ret = []
    for i in range(pp["valid"].shape[0]):
        preds = []
        for j in range(num_classes):
            if not pp["valid"][i, j]:
                continue
            item = {"score": dets_np["scores"][i][j], "class": int(dets_np["clses"][i][j]) + 1,
                    "ct": pp["ct"][i, j], "ct_wreg": pp["ct_wreg"][i, j], "reg": pp["reg"][i, j],
                    "ct_wreg_output": dets_np["cts_wreg"][i][j].reshape(2), "reg_output": dets_np["regs"][i][j].reshape(2),
                    "ct_output": dets_np["cts"][i][j].reshape(2)}
            if "tracking" in pp:
                item["tracking"] = pp["tracking"][i, j]
            preds.append(item)
        ret.append(preds)
    return ret


# ---- SGTADetector.pre_process on the device (sgta_detector.py:335-399) ------------------------------------------------
def invert_affine(m):
    """The inverse 2x3 map exactly as cv2.warpAffine builds it from ``M`` (imgwarp.cpp: D = 1 / det, double)."""
    m = np.asarray(m, np.float64)
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    i00, i01, i10, i11 = a11, m[0, 1] * -d, m[1, 0] * -d, a22
    b1 = -i00 * m[0, 2] - i01 * m[1, 2]
    b2 = -i10 * m[0, 2] - i11 * m[1, 2]
    return np.array([[i00, i01, b1], [i10, i11, b2]], np.float64)


def transform_scale(height, width, opt):
    """``_transform_scale`` (sgta_detector.py:335-365) at scale 1 (cv2.resize to the same size is the identity) ->
    (c, s, inp_width, inp_height)."""
    if getattr(opt, "fix_short", 0) > 0:
        if height < width:
            inp_h, inp_w = opt.fix_short, (int(width / height * opt.fix_short) + 63) // 64 * 64
        else:
            inp_h, inp_w = (int(height / width * opt.fix_short) + 63) // 64 * 64, opt.fix_short
        return np.array([width / 2, height / 2], np.float32), np.array([width, height], np.float32), inp_w, inp_h
    if getattr(opt, "fix_res", True):
        return np.array([width / 2., height / 2.], np.float32), max(height, width) * 1.0, opt.input_w, opt.input_h
    inp_h, inp_w = (height | opt.pad) + 1, (width | opt.pad) + 1
    return np.array([width // 2, height // 2], np.float32), np.array([inp_w, inp_h], np.float32), inp_w, inp_h


def pre_process(images_u8, opt, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), down_ratio=4, out=None):
    """Batched ``SGTADetector.pre_process``: images_u8 uint8 [B,Hr,Wr,3] (device or host; one raw size per call) ->
    (images f32 [B,3,inp_h,inp_w] on the device, list of the reference's ``meta`` dicts).  The affine warp + normalise
    run in sgta_pre_process; only the 2x3 matrices are host work."""
    dev = torch.device("cuda", torch.cuda.current_device())
    img = images_u8 if torch.is_tensor(images_u8) else torch.from_numpy(np.ascontiguousarray(images_u8))
    if img.dtype != torch.uint8 or img.dim() != 4 or img.shape[3] != 3:
        raise ValueError("images must be uint8 [B,H,W,3]")
    img = img.to(dev).contiguous()
    B, Hr, Wr, _ = img.shape
    c, s, inp_w, inp_h = transform_scale(Hr, Wr, opt)
    t_in = get_affine_transform(c, s, 0, [inp_w, inp_h])
    out_h, out_w = int(inp_h // down_ratio), int(inp_w // down_ratio)
    t_out = get_affine_transform(c, s, 0, [out_w, out_h])
    inv = torch.from_numpy(np.stack([invert_affine(t_in)] * B)).to(dev)
    out = torch.empty((B, 3, inp_h, inp_w), dtype=torch.float32, device=dev) if out is None else out
    _lib.call("sgta_pre_process", img.data_ptr(), inv.data_ptr(), _dev(np.asarray(mean, np.float32), torch.float32, dev).data_ptr(),
              _dev(np.asarray(std, np.float32), torch.float32, dev).data_ptr(), out.data_ptr(), B, Hr, Wr, inp_h, inp_w,
              torch.cuda.current_stream(dev).cuda_stream)
    meta = {"c": c, "s": s, "height": Hr, "width": Wr, "out_height": out_h, "out_width": out_w, "inp_height": inp_h,
            "inp_width": inp_w, "trans_input": t_in, "trans_output": t_out}
    return out, [dict(meta) for _ in range(B)]

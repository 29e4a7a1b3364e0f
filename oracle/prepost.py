"""CPU oracle (numpy) for the steps either side of the network  --  TEST INFRASTRUCTURE ONLY.

Prior heat-map rendering and detection post-processing.  Every function cites the
reference lines it restates; ``tests/golden/make_golden.py`` pins them against the reference's own files
(``sgtapose/utilities.py``, ``lib/utils/image.py``, ``lib/utils/post_process.py``, ``lib/sgta_detector.py``)
imported in the build container.

Parity status: PINNED, with one exception -- ``get_affine_transform`` ends in ``cv2.getAffineTransform``
(lib/utils/image.py:73-76) and OpenCV is not in this image (nor vendored by the reference), so the 3-point affine
solve is restated from OpenCV's published algorithm (6x6 linear system in double) and is *unpinned*; the golden
vectors feed the SAME matrices to the reference functions and to this restatement, so everything downstream of
the matrix is pinned.
"""
import numpy as np


# ---- lib/utils/image.py -----------------------------------------------------------------------------------------
def get_3rd_point(a, b):
    """image.py:88-90."""
    direct = a - b
    return b + np.array([-direct[1], direct[0]], dtype=np.float32)


def get_affine_3pt(src, dst):
    """cv2.getAffineTransform(src, dst): the 2x3 double matrix M with M [x y 1]^T = dst for the three point pairs
    (OpenCV imgproc/imgwarp.cpp: 6x6 system solved in double).  UNPINNED (OpenCV absent)."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        a[i, 0:3] = (src[i, 0], src[i, 1], 1.0)
        a[i + 3, 3:6] = (src[i, 0], src[i, 1], 1.0)
        b[i] = dst[i, 0]
        b[i + 3] = dst[i, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def get_affine_transform(center, scale, rot, output_size, inv=0):
    """image.py:45-78 with rot == 0 and shift == 0 (the only way the path calls it,
    sgta_detector.py:375, 380; post_process.py:102-103)."""
    assert rot == 0
    if not isinstance(scale, (np.ndarray, list)):
        scale = np.array([scale, scale], dtype=np.float32)
    src_w = scale[0]
    dst_w, dst_h = output_size[0], output_size[1]
    src_dir = np.array([0, src_w * -0.5], np.float32)          # get_dir with rot_rad = 0
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = np.asarray(center, np.float32) + src_dir
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2:, :] = get_3rd_point(src[0, :], src[1, :])
    dst[2:, :] = get_3rd_point(dst[0, :], dst[1, :])
    return get_affine_3pt(dst, src) if inv else get_affine_3pt(src, dst)


def transform_preds_with_trans(coords, trans):
    """image.py:20-26: fp32 homogeneous coordinates times the 2x3 matrix."""
    t = np.ones((coords.shape[0], 3), np.float32)
    t[:, :2] = coords
    return np.dot(trans, t.transpose()).transpose()[:, :2]


def transform_preds_with_trans_offset(coords, trans):
    """image.py:28-34: same with a 0 in the homogeneous slot (offsets do not translate)."""
    t = np.zeros((coords.shape[0], 3), np.float32)
    t[:, :2] = coords
    return np.dot(trans, t.transpose()).transpose()[:, :2]


# ---- sgtapose/utilities.py --------------------------------------------------------------------------------------
def gaussian2d(shape, sigma):
    """utilities.py:846-853 with res = (0, 0) (draw_umich_gaussian always passes [0, 0], :806)."""
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_umich_gaussian(heatmap, center, radius, k=1):
    """utilities.py:800-821: integer (truncated) centre, sigma 2, drawn only if the whole patch (and one more
    column / row to the right / below) lies inside the map; element-wise max."""
    diameter = 2 * radius + 1
    height, width = heatmap.shape[0:2]
    x, y = int(center[0]), int(center[1])
    if x - radius >= 0 and x + radius + 1 < width and y - radius >= 0 and y + radius + 1 < height:
        gaussian = gaussian2d((diameter, diameter), sigma=2)
        left, right = min(x, radius), min(width - x, radius + 1)
        top, bottom = min(y, radius), min(height - y, radius + 1)
        masked_heatmap = heatmap[y - top:y + bottom, x - left:x + right]
        masked_gaussian = gaussian[radius - top:radius + bottom, radius - left:radius + right]
        if min(masked_gaussian.shape) > 0 and min(masked_heatmap.shape) > 0:
            np.maximum(masked_heatmap, masked_gaussian * k, out=masked_heatmap)
    return heatmap


def affine_transform_and_clip(pts, t, width, height, raw_width, raw_height):
    """utilities.py:943-972: fp64 affine, clip to the map, points outside the raw image become (0, 0)."""
    n_kp = pts.shape[0]
    new_pts = np.concatenate((pts, np.ones(n_kp).reshape(n_kp, 1)), axis=-1)
    new_pts = np.dot(t, new_pts.T).T
    new_pts[:, 0] = np.clip(new_pts[:, 0], 0, width - 1)
    new_pts[:, 1] = np.clip(new_pts[:, 1], 0, height - 1)
    out = []
    for kp in range(n_kp):
        px, py = pts[kp][0], pts[kp][1]
        out.append(new_pts[kp].tolist() if (0.0 <= px < raw_width and 0.0 <= py < raw_height) else [0, 0])
    return np.array(out)


def get_prev_hm_wo_noise(kp_projs_raw, trans_input, input_w, input_h, raw_width, raw_height):
    """utilities.py:1045-1057: ONE map, all key-points max-combined, radius 4."""
    pre_hm = np.zeros((input_h, input_w), dtype=np.float32)
    if kp_projs_raw is not None:
        kps = affine_transform_and_clip(kp_projs_raw, trans_input, input_w, input_h, raw_width, raw_height)
        for i in range(kps.shape[0]):
            draw_umich_gaussian(pre_hm, kps[i].copy(), 4, k=1)
    return pre_hm


def get_prev_hm_wo_noise_cls(kp_projs_raw, n_kp, trans_output, out_w, out_h, raw_width, raw_height):
    """utilities.py:1086-1098: one map PER key-point (the reference reads n_kp off ``kp_gts_raw.shape``)."""
    pre_hm_cls = np.zeros((n_kp, int(out_h), int(out_w)), dtype=np.float32)
    if kp_projs_raw is not None:
        assert kp_projs_raw.shape[0] == n_kp
        kps = affine_transform_and_clip(kp_projs_raw, trans_output, out_w, out_h, raw_width, raw_height)
        for i in range(n_kp):
            draw_umich_gaussian(pre_hm_cls[i], kps[i].copy(), 4, k=1)
    return pre_hm_cls


# ---- lib/utils/post_process.py / lib/sgta_detector.py -----------------------------------------------------------
def dream_generic_post_process(dets, trans_inv, out_thresh):
    """post_process.py:93-175 for the keys the decode of this path produces (scores, clses, cts, cts_wreg, regs,
    tracking); ``trans_inv[i]`` = get_affine_transform(c, s, 0, (w, h), inv=1).astype(float32) of sample i
    (:102-103).  -> list (per sample) of lists of dicts."""
    ret = []
    for i in range(len(dets["scores"])):
        preds = []
        trans = np.asarray(trans_inv[i], np.float32)
        for j in range(len(dets["scores"][i])):
            if dets["scores"][i][j] < out_thresh:
                continue
            item = {"score": dets["scores"][i][j], "class": int(dets["clses"][i][j]) + 1}
            item["ct"] = transform_preds_with_trans(dets["cts"][i][j].reshape(1, 2), trans).reshape(2)
            item["ct_wreg"] = transform_preds_with_trans(dets["cts_wreg"][i][j].reshape(1, 2), trans).reshape(2)
            item["reg"] = transform_preds_with_trans_offset(dets["regs"][i][j].reshape(1, 2), trans).reshape(2)
            item["ct_wreg_output"] = dets["cts_wreg"][i][j].reshape(2)
            item["reg_output"] = dets["regs"][i][j].reshape(2)
            item["ct_output"] = dets["cts"][i][j].reshape(2)
            if "tracking" in dets:
                tracking = transform_preds_with_trans(
                    (dets["tracking"][i][j] + dets["cts"][i][j]).reshape(1, 2), trans).reshape(2)
                item["tracking"] = tracking - item["ct"]
            preds.append(item)
        ret.append(preds)
    return ret


def get_final_kps(dets, num_classes):
    """sgta_detector.py:608-651 (``is_ct`` path): per class the ct_wreg of the best-scoring detection (ties: the
    python list sort falls back to the coordinate), missing classes stay at -999.999 * 4."""
    detected_kps = np.full((num_classes, 2), -999.999 * 4)
    cls = {i: {"x": [], "y": []} for i in range(1, num_classes + 1)}
    for det in dets:
        score, clas, ct_wreg = det["score"], det["class"], det["ct_wreg"].tolist()
        cls[clas]["x"].append([score, ct_wreg[0]])
        cls[clas]["y"].append([score, ct_wreg[1]])
    for i in range(1, num_classes + 1):
        x_list, y_list = cls[i]["x"], cls[i]["y"]
        if not x_list:
            continue
        x_list.sort()
        y_list.sort()
        if x_list[-1][0] != y_list[-1][0]:           # the reference's ``assert`` inside try/except: leave missing
            continue
        detected_kps[i - 1] = [x_list[-1][1], y_list[-1][1]]
    return detected_kps


# ---- SGTADetector.pre_process: cv2.warpAffine + normalize_img (sgta_detector.py:368-402) ------------------------------------
def invert_affine(m):
    """cv2.warpAffine's inversion of M (OpenCV imgproc/imgwarp.cpp, double)."""
    m = np.asarray(m, np.float64)
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    i00, i01, i10, i11 = a11, m[0, 1] * -d, m[1, 0] * -d, a22
    return np.array([[i00, i01, -i00 * m[0, 2] - i01 * m[1, 2]], [i10, i11, -i10 * m[0, 2] - i11 * m[1, 2]]], np.float64)


def warp_affine_u8(img, M, dsize):
    """cv2.warpAffine(img u8 [H,W,C], M, (w, h), flags=INTER_LINEAR), constant border 0, restated from OpenCV's published
    8-bit algorithm (WarpAffineInvoker + remapBilinear, imgwarp.cpp): AB_BITS = 10, INTER_BITS = 5, weights scaled by 2^15.
    UNPINNED (OpenCV is not in the image); a float bilinear warp agrees to the 1/32-pixel coordinate quantisation."""
    w, h = dsize
    Hr, Wr, C = img.shape
    iM = invert_affine(M)
    xs = np.arange(w, dtype=np.float64)
    ys = np.arange(h, dtype=np.float64)
    adelta = np.rint(iM[0, 0] * xs * 1024.0).astype(np.int64)
    bdelta = np.rint(iM[1, 0] * xs * 1024.0).astype(np.int64)
    X0 = np.rint((iM[0, 1] * ys + iM[0, 2]) * 1024.0).astype(np.int64) + 16
    Y0 = np.rint((iM[1, 1] * ys + iM[1, 2]) * 1024.0).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    out = np.zeros((h, w, C), np.int64)
    src = img.astype(np.int64)
    for dy, dx, wgt in ((0, 0, (32 - fx) * (32 - fy) * 32), (0, 1, fx * (32 - fy) * 32),
                        (1, 0, (32 - fx) * fy * 32), (1, 1, fx * fy * 32)):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < Hr) & (xx >= 0) & (xx < Wr)
        v = src[np.clip(yy, 0, Hr - 1), np.clip(xx, 0, Wr - 1)]
        out += np.where(ok[..., None], v, 0) * wgt[..., None]
    return ((out + (1 << 14)) >> 15).astype(np.uint8)


def pre_process(image_u8, trans_input, inp_w, inp_h, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
    """sgta_detector.py:381-386: warp, ((img / 255. - mean) / std).float32 as torch evaluates it, HWC -> [1,3,h,w]."""
    import torch
    warped = torch.from_numpy(warp_affine_u8(image_u8, trans_input, (inp_w, inp_h)))
    mean_t = torch.tensor(mean, dtype=torch.float32).reshape(1, 1, 3)
    std_t = torch.tensor(std, dtype=torch.float32).reshape(1, 1, 3)
    x = ((warped / 255. - mean_t) / std_t).type(torch.float32)
    return x.permute(2, 0, 1).reshape(1, 3, inp_h, inp_w)

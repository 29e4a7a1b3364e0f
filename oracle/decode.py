"""Oracle: heat-map decode (peak search + centroid), CenterNet-style gather, spatial soft-argmax.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Restates
  * peaks_from_belief_maps   sgtapose/image_proc.py:1032-1143   (scipy gaussian_filter sigma=3 restated
                             below in numpy: axis 0 then axis 1, double accumulation, float32 storage
                             after each axis, 'reflect' borders, truncate 4.0 -> 25 taps)
  * _peaks_info              sgtapose/lib/model/utils.py:207-284 (y-sorted multi-peak rule, int() truncation)
  * dream_generic_decode     sgtapose/lib/model/decode.py:184-313 (heads hm / reg / tracking only)
  * _tranpose_and_gather_feat  lib/model/utils.py:29-33
  * SoftArgmaxPavlo.forward  sgtapose/spatial_softmax.py:24-95  (== lib/model/utils.py:107-187)
The reference decodes batch element 0 only (utils.py:214-216, :255); ``peaks_info`` here applies the
same per-sample rule to every sample of the batch.
"""
import numpy as np
import torch
import torch.nn.functional as F

SIGMA = 3.0
RADIUS = int(4.0 * SIGMA + 0.5)           # scipy: int(truncate * sd + 0.5) = 12
PEAK_THRESH = 0.01                        # image_proc.py:1043
OFFSET_DUE_TO_UPSAMPLING = 0.4395         # lib/model/utils.py:212
MISSING = -999.999


def gaussian_kernel1d(sigma=SIGMA, radius=RADIUS):
    """scipy.ndimage._filters._gaussian_kernel1d (order 0), float64."""
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def _correlate1d_sym(a32, w, axis):
    """scipy NI_Correlate1D, symmetric branch: out = c*w0; for j=-r..-1: out += (a[j]+a[-j])*w[j],
    in double on a 'reflect'-extended line; result rounded to float32."""
    r = (len(w) - 1) // 2
    a = np.moveaxis(a32.astype(np.float64), axis, -1)
    pad = [(0, 0)] * (a.ndim - 1) + [(r, r)]
    p = np.pad(a, pad, mode="symmetric")          # numpy 'symmetric' == scipy 'reflect'
    L = a.shape[-1]
    acc = p[..., r:r + L] * w[r]
    for j in range(-r, 0):
        acc = acc + (p[..., r + j:r + j + L] + p[..., r - j:r - j + L]) * w[r + j]
    return np.moveaxis(acc.astype(np.float32), -1, axis)


def gaussian_filter_f32(m32):
    """gaussian_filter(m, sigma=3) for a float32 2-D map (image_proc.py:1053)."""
    w = gaussian_kernel1d()
    return _correlate1d_sym(_correlate1d_sym(np.ascontiguousarray(m32, np.float32), w, 0), w, 1)


def peaks_from_maps(maps):
    """maps float32 [K,H,W] -> list (per map) of (x, y, score) tuples in np.nonzero (row-major) order."""
    out = []
    for j in range(maps.shape[0]):
        ori = np.asarray(maps[j], dtype=np.float32)
        g = gaussian_filter_f32(ori)
        H, W = g.shape
        up = np.zeros_like(g); up[1:, :] = g[:-1, :]
        dn = np.zeros_like(g); dn[:-1, :] = g[1:, :]
        lf = np.zeros_like(g); lf[:, 1:] = g[:, :-1]
        rt = np.zeros_like(g); rt[:, :-1] = g[:, 1:]
        is_peak = (g >= up) & (g >= dn) & (g >= lf) & (g >= rt) & (g > PEAK_THRESH)
        ys, xs = np.nonzero(is_peak)
        plist = []
        for px, py in zip(xs, ys):
            wts = np.zeros((5, 5))
            yv = np.zeros((5, 5))
            xv = np.zeros((5, 5))
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    yy, xx = py + dy, px + dx
                    if yy < 0 or yy >= H or xx < 0 or xx >= W:
                        continue
                    yv[dx + 2, dy + 2] = yy
                    xv[dx + 2, dy + 2] = xx
                    wts[dx + 2, dy + 2] = ori[yy, xx]
            try:
                cx = np.average(xv, weights=wts) + OFFSET_DUE_TO_UPSAMPLING
                cy = np.average(yv, weights=wts) + OFFSET_DUE_TO_UPSAMPLING
            except ZeroDivisionError:
                cx = px + OFFSET_DUE_TO_UPSAMPLING
                cy = py + OFFSET_DUE_TO_UPSAMPLING
            plist.append((cx, cy, ori[py, px]))
        out.append(plist)
    return out


def _select(plist):
    """lib/model/utils.py:221-246: 1 peak -> it; >1 -> stable sort by y DESC, keep the first iff its
    score exceeds the second's by >= 0.25 (float32 subtraction); else missing."""
    if len(plist) == 1:
        return plist[0][0], plist[0][1]
    if len(plist) > 1:
        s = sorted(plist, key=lambda t: t[1], reverse=True)
        if s[0][2] - s[1][2] >= 0.25:
            return s[0][0], s[0][1]
    return MISSING, MISSING


def peaks_info(heat):
    """heat: torch float32 [B,K,H,W] (post-sigmoid).  Returns (scores f32 [B,K], inds i64 [B,K],
    clses i64 [B,K], ys i64 [B,K], xs i64 [B,K]) -- lib/model/utils.py:251-284."""
    B, K, H, W = heat.shape
    hm = heat.detach().cpu().numpy()
    scores = np.zeros((B, K), np.float32)
    xs = np.zeros((B, K), np.int64)
    ys = np.zeros((B, K), np.int64)
    for b in range(B):
        for k, plist in enumerate(peaks_from_maps(hm[b])):
            x, y = _select(plist)
            if x == MISSING or y == MISSING:
                scores[b, k] = -1.0
            else:
                xi, yi = int(x), int(y)
                xs[b, k], ys[b, k] = xi, yi
                scores[b, k] = hm[b, k, yi, xi]
    xs_t, ys_t = torch.from_numpy(xs), torch.from_numpy(ys)
    clses = torch.arange(K).view(1, K).expand(B, K).contiguous()
    return torch.from_numpy(scores), ys_t * W + xs_t, clses, ys_t, xs_t


def gather_feat(feat, inds):
    """_tranpose_and_gather_feat: feat [B,C,H,W], inds [B,K] -> [B,K,C]."""
    B, C = feat.shape[:2]
    flat = feat.reshape(B, C, -1)
    return torch.gather(flat, 2, inds.view(B, 1, -1).expand(B, C, inds.shape[1])).permute(0, 2, 1)


def dream_decode(output, K=7):
    """dream_generic_decode for heads {hm (post-sigmoid), reg, tracking} (decode.py:184-313)."""
    heat = output["hm"]
    B = heat.shape[0]
    scores, inds, clses, ys0, xs0 = peaks_info(heat)
    ret = {"scores": scores.view(B, K), "clses": clses.view(B, K).float(), "xs": xs0, "ys": ys0,
           "cts": torch.cat([xs0.unsqueeze(2), ys0.unsqueeze(2)], dim=2)}
    reg = gather_feat(output["reg"], inds).reshape(B, K, 2)
    xs = xs0.view(B, K, 1) + reg[:, :, 0:1]
    ys = ys0.view(B, K, 1) + reg[:, :, 1:2]
    cts_wreg = torch.cat([xs.unsqueeze(2), ys.unsqueeze(2)], dim=2)         # [B,K,2,1]
    ret["cts_wreg"] = cts_wreg
    ret["regs"] = cts_wreg - torch.cat([xs0.view(B, K, 1).unsqueeze(2), ys0.view(B, K, 1).unsqueeze(2)], dim=2)
    ret["tracking"] = gather_feat(output["tracking"], inds).reshape(B, K, -1)
    return ret


def soft_argmax_pavlo(heatmaps, beta=25.0, size_mult=1.0):
    """SoftArgmaxPavlo.forward: AvgPool2d(7,1,3) (zero pad, divisor 49) -> minus per-map max ->
    exp(beta*x) -> / (sum + 1e-8) -> (E[col], E[row]).  heatmaps [B,K,H,W] -> [B,K,2] = (x, y)."""
    B, K, H, W = heatmaps.shape
    hm = F.avg_pool2d(heatmaps, 7, stride=1, padding=3)
    flat = hm.reshape(B, K, -1)
    flat = flat - flat.max(dim=2, keepdim=True)[0]
    e = torch.exp(beta * flat)
    nrm = (e / (e.sum(dim=2, keepdim=True) + 1e-8)).view(B, K, H, W)
    col = (torch.arange(W, dtype=torch.float32) * size_mult).view(1, 1, 1, W)
    row = (torch.arange(H, dtype=torch.float32) * size_mult).view(1, 1, H, 1)
    x = (nrm * col).reshape(B, K, -1).sum(dim=2)
    y = (nrm * row).reshape(B, K, -1).sum(dim=2)
    return torch.stack((x, y), dim=2)

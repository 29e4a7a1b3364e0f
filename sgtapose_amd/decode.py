"""Device-side ``dream_generic_decode`` (reference sgtapose/lib/model/decode.py:184-313 with
``_peaks_info``, sgtapose/lib/model/utils.py:207-284): same dictionary, same shapes and dtypes, for
every sample of the batch (the reference only decodes batch element 0)."""
import torch

from . import ops


def dream_generic_decode(output, K=7, opt=None, ws=None, out=None):
    if "hm" not in output:
        return {}
    if opt is not None and getattr(opt, "zero_tracking", False):
        output["tracking"] *= 0
    heat = output["hm"]
    B, cat, H, W = heat.shape
    if cat != K:
        raise ValueError("K must equal the number of heat-map classes")
    d = ops.peaks_decode(heat, output.get("reg"), output.get("tracking"), ws, out=out)
    xs0, ys0 = d["xs"], d["ys"]
    ret = {"scores": d["scores"], "clses": _clses(B, cat, heat.device), "xs": xs0, "ys": ys0, "cts": d["cts"]}
    if "reg" in output:
        # decode.py:296-312: int64 centre + float32 offset -> float32, and back; emitted by the decode kernel
        ret["cts_wreg"] = d["cts_wreg"].view(B, K, 2, 1)
        ret["regs"] = d["regs_rel"].view(B, K, 2, 1)
    else:
        reg = torch.full((B, K, 2), 0.5, device=heat.device)
        xs = xs0.view(B, K, 1) + reg[:, :, 0:1]
        ys = ys0.view(B, K, 1) + reg[:, :, 1:2]
        cts_wreg = torch.cat([xs.unsqueeze(2), ys.unsqueeze(2)], dim=2)          # [B,K,2,1]
        ret["cts_wreg"] = cts_wreg
        ret["regs"] = cts_wreg - torch.cat([xs0.view(B, K, 1).unsqueeze(2), ys0.view(B, K, 1).unsqueeze(2)], dim=2)
    if "tracking" in output:
        ret["tracking"] = d["trks"]
    ret["npeaks"] = d["npeaks"]
    ret["kp"] = d["kp"]                # [B,K,3] = (x + reg_x, y + reg_y, score): BatchedDetector.keypoints / all-gather
    return ret


_CLSES = {}


def _clses(B, cat, device):
    """The constant class-index tensor of the dictionary (float32 [B, cat]), built once per shape."""
    key = (B, cat, str(device))
    if key not in _CLSES:
        _CLSES[key] = torch.arange(cat, device=device).view(1, cat).expand(B, cat).float()
    return _CLSES[key]

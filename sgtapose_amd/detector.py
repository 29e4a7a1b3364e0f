"""Batched detector step with the contract of ``SGTADetector.process`` (drop-in boundary #3,
sgtapose/lib/sgta_detector.py:881-927): forward -> sigmoid on 'hm' -> dream_generic_decode ->
numpy dets, but for B >= 1 frame pairs and entirely on the device until the final copy.
"""
import torch

from . import ops, prepost
from .decode import dream_generic_decode


class BatchedDetector:
    def __init__(self, model, opt=None):
        self.model = model.eval()
        self.opt = opt or model.opt
        self.K = int(self.opt.num_classes)

    @torch.no_grad()
    def process_device(self, *inputs, static=False):
        """Returns (output with post-sigmoid hm, dets as device tensors); nothing leaves the GPU.  ``inputs`` = the
        model's forward arguments: (images, pre_images, pre_hms, repro_hms, pre_hms_cls, repro_hms_cls) for
        dlapawdl3new, the nine tensors of BaseModelPlanA_Three.forward for dlapawd3."""
        eng = self.model.engine()
        out = eng.forward(*inputs, hm_act=ops.ACT_SIGMOID, static=static)
        hm = out["hm"]
        ws = (eng.buf("dec_tmp", tuple(hm.shape)), eng.buf("dec_filt", tuple(hm.shape)))
        bufs = ops.decode_buffers(hm.shape[0], self.K, hm.device, "reg" in out, alloc=eng.buf) if static else None
        dets = dream_generic_decode(out, K=self.K, opt=self.opt, ws=ws, out=bufs)
        return out, dets

    def process(self, images, pre_images=None, pre_hms=None, repro_hms=None, pre_inds=None,
                return_time=False, pre_hms_cls=None, repro_hms_cls=None):
        import time
        torch.cuda.synchronize()
        output, dets = self.process_device(images, pre_images, pre_hms, repro_hms, pre_hms_cls, repro_hms_cls)
        output["pre_inds"] = pre_inds
        torch.cuda.synchronize()
        forward_time = time.time()
        hms = output["hm"][0]
        dets = {k: v.detach().cpu().numpy() for k, v in dets.items() if k != "kp"}   # 'kp' is this build's extra
        return (output, dets, forward_time, hms) if return_time else (output, dets)

    # ---- the steps either side of the network, on the device ------------------
    def render_inputs(self, prev_kp_projs, next_kp_projs, metas, has_prior=None):
        """The four prior maps of ``SGTADetector._get_further_dt_pnp_inputs`` (sgta_detector.py:486-497) for a
        batch, rendered on the device: prev_kp_projs / next_kp_projs [B,K,2] raw-image projections (the PnP outputs),
        metas = list of the reference's ``meta`` dicts (trans_input, trans_output, inp_/out_ sizes, width, height).
        -> (pre_hm [B,1,H,W], repro_hm, pre_hm_cls [B,K,h,w], repro_hm_cls)."""
        import numpy as np
        m0 = metas[0]
        t_in = np.stack([m["trans_input"] for m in metas])
        t_out = np.stack([m["trans_output"] for m in metas])
        wh = np.array([[m["width"], m["height"]] for m in metas], np.int32)
        dev = next(self.model.parameters()).device
        kw = dict(has_prior=has_prior, device=dev)
        iw, ih, ow, oh = m0["inp_width"], m0["inp_height"], m0["out_width"], m0["out_height"]
        return (prepost.get_prev_hm_wo_noise(prev_kp_projs, t_in, iw, ih, wh, **kw),
                prepost.get_prev_hm_wo_noise(next_kp_projs, t_in, iw, ih, wh, **kw),
                prepost.get_prev_hm_wo_noise_cls(prev_kp_projs, t_out, ow, oh, wh, **kw),
                prepost.get_prev_hm_wo_noise_cls(next_kp_projs, t_out, ow, oh, wh, **kw))

    def post_process_device(self, dets, metas):
        """``SGTADetector.post_process`` + ``_get_final_kps`` for a batch, on the device: dets = the device dict of
        process_device; -> dict(valid, ct, ct_wreg, reg, tracking, final_kps [B,K,2] fp64) of device tensors.
        ``final_kps`` feeds the next frame's PnP / prior rendering without leaving HBM."""
        import numpy as np
        t_inv = np.stack([prepost.get_affine_transform(m["c"], m["s"], 0, (m["out_width"], m["out_height"]), inv=1)
                          .astype(np.float32) for m in metas])
        return prepost.post_process_device(dets, t_inv, float(self.opt.out_thresh))

    def post_process(self, dets, metas):
        """Host-side view with the reference's structure: (list per sample of detection dicts, final_kps [B,K,2])."""
        pp = {k: v.cpu().numpy() for k, v in self.post_process_device(dets, metas).items()}
        dets_np = {k: v.detach().cpu().numpy() for k, v in dets.items()}
        return prepost.dream_generic_post_process(dets_np, pp, self.K), pp["final_kps"]

    @staticmethod
    def keypoints(dets):
        """[B,K,3] float32 (x + reg_x, y + reg_y, score) -- the tensor that is all-gathered across ranks."""
        return dets["kp"]                  # written by the decode kernel (sgta_peaks_decode_kp): no glue kernels

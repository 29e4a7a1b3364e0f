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

    def next_frame_priors(self, final_kps, prev_x3d, next_x3d, camera_K, metas):
        """The loop between two frames, on the device (``SGTADetector._get_further_dt_pnp_inputs``,
        sgta_detector.py:455-497): PnP on this frame's final key-points (missing ones skipped) against the robot's 3-D
        key-points, the pose applied to the next frame's 3-D key-points, and the four prior maps rendered from those
        projections.  final_kps [B,K,2] f64 (post_process_device), prev_x3d / next_x3d [B,K,3].
        -> (pre_hm, repro_hm, pre_hm_cls, repro_hm_cls, pnp dict)."""
        from . import pnp
        r = pnp.solve_pnp(prev_x3d, final_kps, camera_K, next_points=next_x3d)
        maps = self.render_inputs(final_kps, r["next_kp_projs_est"], metas)
        return maps + (r,)

    @staticmethod
    def keypoints(dets):
        """[B,K,3] float32 (x + reg_x, y + reg_y, score) -- the tensor that is all-gathered across ranks."""
        return dets["kp"]                  # written by the decode kernel (sgta_peaks_decode_kp): no glue kernels


class SequenceDetector(BatchedDetector):
    """Stateful video mode: frame t's image stem ``relu(bn(conv7x7(img_t)))`` is computed
    once, when t is the current frame, and reused at step t + 1 where t is the previous frame -- exact up to nothing: the
    two stems are separate conv+BN+ReLU branches whose outputs are only summed (dla.py:260-270, :325-328); nothing
    deeper can be reused because the summed heat-map stem differs (repro_hm vs pre_hm, dla.py:1506-1507).  Saves one 7x7
    image conv and one image read per frame pair on the fp32 engine.  On the 16-bit engines the stems, their sum and level0
    are ONE kernel that reads 16 bytes per pixel and keeps the stem map on the chip: a cached 32-byte-per-pixel stem would
    cost more than it saves, so there the step only carries the previous frame along (same interface, same results as the
    stateless detector)."""

    def __init__(self, model, opt=None):
        super().__init__(model, opt)
        self._stem = [None, None]              # ping-pong buffers: previous frame's / current frame's image stem
        self._have_prev = False

    def reset(self):
        self._have_prev = False

    @torch.no_grad()
    def step_device(self, image, pre_hm, repro_hm, pre_hm_cls, repro_hm_cls, pre_image=None, static=False):
        """One video step.  ``pre_image`` is only needed for the first step (or after ``reset``); afterwards the previous
        frame is the ``image`` of the previous call.  Returns (output, dets) like ``process_device``."""
        from .engine import CachedStem
        eng = self.model.engine()
        if getattr(eng, "dtype", "f32") != "f32":
            if not self._have_prev:
                if pre_image is None:
                    raise ValueError("the first step of a sequence needs pre_image")
                self._prev_image = pre_image
            out = self.process_device(image, self._prev_image, pre_hm, repro_hm, pre_hm_cls, repro_hm_cls, static=static)
            self._prev_image, self._have_prev = image, True      # self.pre_images = images (sgta_detector.py:203)
            return out
        B, _, H, W = image.shape
        shape = (B, 16, H, W)
        for k in (0, 1):
            if self._stem[k] is None or tuple(self._stem[k].shape) != shape:
                self._stem[k] = torch.empty(shape, dtype=torch.float32, device=image.device)
                self._have_prev = False
        if not self._have_prev:
            if pre_image is None:
                raise ValueError("the first step of a sequence needs pre_image")
            eng.image_stem(pre_image, self._stem[0])
        eng.image_stem(image, self._stem[1])
        out = eng.forward(CachedStem(self._stem[1]), CachedStem(self._stem[0]), pre_hm, repro_hm, pre_hm_cls, repro_hm_cls,
                          hm_act=ops.ACT_SIGMOID, static=static)
        self._stem.reverse()                   # the current frame's stem is the next step's previous-frame stem
        self._have_prev = True
        hm = out["hm"]
        ws = (eng.buf("dec_tmp", tuple(hm.shape)), eng.buf("dec_filt", tuple(hm.shape)))
        bufs = ops.decode_buffers(hm.shape[0], self.K, hm.device, "reg" in out, alloc=eng.buf) if static else None
        return out, dream_generic_decode(out, K=self.K, opt=self.opt, ws=ws, out=bufs)

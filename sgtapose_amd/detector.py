"""Batched detector step with the contract of ``SGTADetector.process`` (drop-in boundary #3,
sgtapose/lib/sgta_detector.py:881-927): forward -> sigmoid on 'hm' -> dream_generic_decode ->
numpy dets, but for B >= 1 frame pairs and entirely on the device until the final copy.
"""
import torch

from . import ops
from .decode import dream_generic_decode


class BatchedDetector:
    def __init__(self, model, opt=None):
        self.model = model.eval()
        self.opt = opt or model.opt
        self.K = int(self.opt.num_classes)

    @torch.no_grad()
    def process_device(self, images, pre_images, pre_hms, repro_hms, pre_hms_cls, repro_hms_cls, static=False):
        """Returns (output with post-sigmoid hm, dets as device tensors); nothing leaves the GPU."""
        eng = self.model.engine()
        out = eng.forward(images, pre_images, pre_hms, repro_hms, pre_hms_cls, repro_hms_cls,
                          hm_act=ops.ACT_SIGMOID, static=static)
        hm = out["hm"]
        ws = (eng.buf("dec_tmp", tuple(hm.shape)), eng.buf("dec_filt", tuple(hm.shape)))
        dets = dream_generic_decode(out, K=self.K, opt=self.opt, ws=ws)
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
        dets = {k: v.detach().cpu().numpy() for k, v in dets.items()}
        return (output, dets, forward_time, hms) if return_time else (output, dets)

    @staticmethod
    def keypoints(dets):
        """[B,K,3] float32 (x + reg_x, y + reg_y, score) -- the tensor that is all-gathered across ranks."""
        return torch.cat([dets["cts_wreg"].squeeze(-1), dets["scores"].unsqueeze(-1)], dim=2).float()

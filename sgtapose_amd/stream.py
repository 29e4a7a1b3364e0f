"""The reference's operating point: ONE camera (or B cameras in lock-step), frame after frame, closed loop
(``SGTADetector.run``, sgtapose/lib/sgta_detector.py:117-236, driven by ``inference_real``, sgtapose/inference.py:209-222).

Per frame the reference does, on the host with numpy / OpenCV and four host->device copies:

    pre_process (:368-399)  ->  priors from the PREVIOUS frame's detections: is_pnp + four rendered maps (:455-547; zeros
    on the first frame, :164-166, and when nothing was detected, :476-479)  ->  process = forward + sigmoid + decode
    (:881-927)  ->  post_process (:929-942)  ->  _get_final_kps (:608-651)  ->  self.pre_images = images (:203)

``VideoRunner.step`` is that loop with every array resident in HBM: the raw uint8 frame and the robot's 3-D key-points
of the frame go in, the key-point table comes out; nothing else crosses the bus and the host never waits inside a step.
A step is a fixed sequence of kernel launches on pre-allocated buffers, so the steady state is captured ONCE per buffer
parity in a HIP graph (``use_graph``): at batch 1 the ~170 launches of a step cost more host time than GPU time, and the
replay removes that.  All kernels of both parities run eagerly before the capture (one-time kernel attributes), and no
memset node is recorded (the prior maps are cleared by a fill kernel).
"""
import numpy as np
import torch

from . import _lib, ops, prepost
from .detector import BatchedDetector


class VideoRunner:
    def __init__(self, model, raw_hw, camera_K, batch=1, opt=None, use_graph=False):
        self.det = BatchedDetector(model, opt)
        self.opt = opt = self.det.opt
        self.eng = model.engine()
        self.dev = dev = self.eng.dev
        self.B, self.K = B, K = int(batch), int(opt.num_classes)
        Hr, Wr = int(raw_hw[0]), int(raw_hw[1])
        c, s, inp_w, inp_h = prepost.transform_scale(Hr, Wr, opt)
        self.inp_hw = (int(inp_h), int(inp_w))
        self.out_hw = (int(inp_h) // 4, int(inp_w) // 4)
        t_in = prepost.get_affine_transform(c, s, 0, [inp_w, inp_h])
        t_out = prepost.get_affine_transform(c, s, 0, [self.out_hw[1], self.out_hw[0]])
        t_inv = prepost.get_affine_transform(c, s, 0, (self.out_hw[1], self.out_hw[0]), inv=1).astype(np.float32)
        self.meta = {"c": c, "s": s, "height": Hr, "width": Wr, "out_height": self.out_hw[0], "out_width": self.out_hw[1],
                     "inp_height": self.inp_hw[0], "inp_width": self.inp_hw[1], "trans_input": t_in, "trans_output": t_out}
        rep = lambda a, dt: torch.from_numpy(np.stack([a] * B)).to(device=dev, dtype=dt).contiguous()      # noqa: E731
        self.t_in, self.t_out = rep(t_in, torch.float64), rep(t_out, torch.float64)
        self.t_inv = rep(t_inv, torch.float32)
        self.inv_in = rep(prepost.invert_affine(t_in), torch.float64)
        self.raw_wh = torch.tensor([[Wr, Hr]] * B, dtype=torch.int32, device=dev)
        self.mean = torch.tensor([0.5, 0.5, 0.5], dtype=torch.float32, device=dev)
        self.std = torch.tensor([0.5, 0.5, 0.5], dtype=torch.float32, device=dev)
        self.camK = torch.as_tensor(np.asarray(camera_K, np.float64).reshape(9)).to(dev)
        self.table = prepost._gauss_table(dev)
        e = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)                                       # noqa: E731
        H, W = self.inp_hw
        h, w = self.out_hw
        # static inputs of a step
        self.frame = e((B, Hr, Wr, 3), torch.uint8)
        self.frame_host = torch.zeros((B, Hr, Wr, 3), dtype=torch.uint8).pin_memory()
        self.x3d = [e((B, K, 3), torch.float64), e((B, K, 3), torch.float64)]        # [parity]: this frame's / the previous one's
        self.x3d_host = torch.zeros((B, K, 3), dtype=torch.float64).pin_memory()
        # state and intermediates
        self.img = [e((B, 3, H, W), torch.float32), e((B, 3, H, W), torch.float32)]
        self.pre_hm, self.repro_hm = e((B, 1, H, W), torch.float32), e((B, 1, H, W), torch.float32)
        self.pre_cls, self.repro_cls = e((B, K, h, w), torch.float32), e((B, K, h, w), torch.float32)
        self.has = e((B,), torch.uint8)
        self.pnp_ok = e((B,), torch.uint8)
        self.pnp_t, self.pnp_q = e((B, 3), torch.float64), e((B, 4), torch.float64)
        self.next_proj = e((B, K, 2), torch.float64)
        self.final_kps = torch.full((B, K, 2), prepost.MISSING, dtype=torch.float64, device=dev)
        self.pp = dict(valid=e((B, K), torch.uint8), ct=e((B, K, 2), torch.float32), ct_wreg=e((B, K, 2), torch.float32),
                       reg=e((B, K, 2), torch.float32), tracking=e((B, K, 2), torch.float32))
        self.kps_host = torch.zeros((B, K, 2), dtype=torch.float64).pin_memory()
        self.frame_idx = 0
        self.use_graph = bool(use_graph)
        self.graphs = [None, None]
        self._eager_done = [False, False]
        self.last = None

    # ------------------------------------------------------------------------------------------------------------
    def reset(self):
        self.frame_idx = 0
        self.final_kps.fill_(prepost.MISSING)

    def seed_keypoints(self, kps_raw, x3d):
        """Start (or re-start) the loop from known detections instead of a blank first frame -- the reference's
        ``tracker.init_track(meta['pre_dets'])`` (sgta_detector.py:161-162): ``kps_raw`` [B,K,2] raw-image key-points
        (missing ones at -999.999*4) and the 3-D key-points ``x3d`` [B,K,3] of the frame they belong to.  The next step then
        renders its priors from them; it still needs that frame's image as ``pre_image`` (the next step's own frame
        stands in, as on the reference's first frame)."""
        self.final_kps.copy_(torch.as_tensor(np.asarray(kps_raw, np.float64)).to(self.dev))
        p = self.frame_idx & 1
        self.x3d[1 - p].copy_(torch.as_tensor(np.asarray(x3d, np.float64)).to(self.dev))
        self._seeded = True

    def _render(self, kps, trans, out, per_class):
        B, K = self.B, self.K
        _lib.call("sgta_render_priors", kps.data_ptr(), trans.data_ptr(), self.raw_wh.data_ptr(), self.has.data_ptr(),
                  self.table.data_ptr(), out.data_ptr(), B, K, out.shape[2], out.shape[3], int(per_class), ops._stream())

    def _enqueue(self, p, first):
        """The kernels of one step on the current stream; p = buffer parity of this frame, first = no previous frame."""
        B, K = self.B, self.K
        st = ops._stream()
        Hr, Wr = self.frame.shape[1:3]
        _lib.call("sgta_pre_process", self.frame.data_ptr(), self.inv_in.data_ptr(), self.mean.data_ptr(), self.std.data_ptr(),
                  self.img[p].data_ptr(), B, Hr, Wr, self.inp_hw[0], self.inp_hw[1], st)
        # priors from the previous frame's detections (zeros for a sample without any, as the reference returns)
        _lib.call("sgta_kps_any_valid", self.final_kps.data_ptr(), self.has.data_ptr(), B, K, st)
        _lib.call("sgta_pnp_batched", self.x3d[1 - p].data_ptr(), self.final_kps.data_ptr(), None, self.camK.data_ptr(),
                  self.x3d[p].data_ptr(), self.pnp_ok.data_ptr(), self.pnp_t.data_ptr(), self.pnp_q.data_ptr(),
                  self.next_proj.data_ptr(), B, K, st)
        self._render(self.final_kps, self.t_in, self.pre_hm, False)
        self._render(self.next_proj, self.t_in, self.repro_hm, False)
        self._render(self.final_kps, self.t_out, self.pre_cls, True)
        self._render(self.next_proj, self.t_out, self.repro_cls, True)
        pre_img = self.img[p] if (first or self.frame_idx == 0) else self.img[1 - p]     # first frame: pre_images = images (:157-159)
        out, dets = self.det.process_device(self.img[p], pre_img, self.pre_hm, self.repro_hm, self.pre_cls, self.repro_cls,
                                            static=True)
        pp = self.pp
        _lib.call("sgta_post_process", dets["scores"].data_ptr(), dets["cts"].data_ptr(), dets["cts_wreg"].data_ptr(),
                  dets["regs"].data_ptr(), dets["tracking"].data_ptr(), self.t_inv.data_ptr(), float(self.opt.out_thresh),
                  pp["valid"].data_ptr(), pp["ct"].data_ptr(), pp["ct_wreg"].data_ptr(), pp["reg"].data_ptr(),
                  pp["tracking"].data_ptr(), self.final_kps.data_ptr(), B, K, st)
        self.last = (out, dets)

    @torch.no_grad()
    def step(self, frame_u8, x3d, sync=True):
        """One frame: ``frame_u8`` uint8 [B,Hr,Wr,3] (host numpy / tensor, or a device tensor), ``x3d`` [B,K,3] = the robot's
        3-D key-points of THIS frame (the reference reads them from the frame's json).  -> the frame's key-point table
        [B,K,2] float64 in raw-image pixels (-999.999*4 = not detected): a host array when ``sync`` (the call then waits for
        the frame), else the device tensor (valid once the stream reaches it; overwritten by the next step)."""
        p = self.frame_idx & 1
        first = self.frame_idx == 0 and not getattr(self, "_seeded", False)
        f = frame_u8 if torch.is_tensor(frame_u8) else torch.from_numpy(np.ascontiguousarray(frame_u8))
        if f.dtype != torch.uint8 or tuple(f.shape) != tuple(self.frame.shape):
            raise ValueError("frame must be uint8 %s" % (tuple(self.frame.shape),))
        if f.is_cuda:
            self.frame.copy_(f, non_blocking=True)
        else:
            self.frame_host.copy_(f)
            self.frame.copy_(self.frame_host, non_blocking=True)
        self.x3d_host.copy_(torch.as_tensor(np.asarray(x3d, np.float64)).reshape(self.B, self.K, 3))
        self.x3d[p].copy_(self.x3d_host, non_blocking=True)
        if first:
            self.final_kps.fill_(prepost.MISSING)                    # no previous detections: zero priors
        if self.use_graph and self.frame_idx >= 1 and self._eager_done[p]:
            if self.graphs[p] is None:
                # every kernel of this parity has run eagerly (one-time attributes are set): record the launches
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._enqueue(p, False)
                self.graphs[p] = g          # capturing does not execute: replay below
            self.graphs[p].replay()
        else:
            self._enqueue(p, first)
            if self.frame_idx >= 1:
                self._eager_done[p] = True
        self._seeded = False
        self.frame_idx += 1
        if not sync:
            return self.final_kps
        self.kps_host.copy_(self.final_kps, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self.kps_host.numpy().copy()

    def pose(self):
        """The PnP pose of the PREVIOUS frame computed by the last step (it feeds this frame's priors): dict(ok u8 [B],
        translation f64 [B,3], quaternion xyzw f64 [B,4]) of device tensors."""
        return dict(ok=self.pnp_ok, translation=self.pnp_t, quaternion=self.pnp_q)

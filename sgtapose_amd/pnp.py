"""Batched pose step on the device: the reference's ``solve_pnp`` / ``is_pnp``
(sgtapose/geometric_vision.py:43-116, :283-310) and Gauss-Newton refinement ``register_GN_C`` (sgtapose/rf_tools/LM.py:256-266)
for every frame of a batch, same names and argument meaning, arrays staying in HBM (the final key-points of
``BatchedDetector.post_process_device`` feed it directly, its projections feed ``render_inputs``)."""
import numpy as np
import torch

from . import _lib


def _t(a, dtype, dev):
    t = a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))
    return t.to(device=dev, dtype=dtype).contiguous()


def solve_pnp(canonical_points, projections, camera_K, valid=None, next_points=None, device=None):
    """canonical_points [B,K,3], projections [B,K,2] (raw-image pixels; entries at the detector's missing marker
    -999.999*4 are skipped unless ``valid`` [B,K] is given), camera_K [3,3].  -> dict(retval u8 [B], translation f64
    [B,3], quaternion f64 [B,4] xyzw[, next_kp_projs_est f64 [B,K,2] = is_pnp's projections of ``next_points``])."""
    dev = device or (projections.device if torch.is_tensor(projections) and projections.is_cuda else
                     torch.device("cuda", torch.cuda.current_device()))
    x3d, x2d = _t(canonical_points, torch.float64, dev), _t(projections, torch.float64, dev)
    B, K, _ = x2d.shape
    if tuple(x3d.shape) != (B, K, 3):
        raise ValueError("canonical_points must be [B,K,3]")
    cam = _t(np.asarray(camera_K, np.float64).reshape(9) if not torch.is_tensor(camera_K) else camera_K.reshape(9), torch.float64, dev)
    v = None if valid is None else _t(valid, torch.uint8, dev)
    nx = None if next_points is None else _t(next_points, torch.float64, dev)
    out = dict(retval=torch.empty(B, dtype=torch.uint8, device=dev), translation=torch.empty(B, 3, dtype=torch.float64, device=dev),
               quaternion=torch.empty(B, 4, dtype=torch.float64, device=dev))
    npj = torch.empty(B, K, 2, dtype=torch.float64, device=dev) if next_points is not None else None
    _lib.call("sgta_pnp_batched", x3d.data_ptr(), x2d.data_ptr(), None if v is None else v.data_ptr(), cam.data_ptr(),
              None if nx is None else nx.data_ptr(), out["retval"].data_ptr(), out["translation"].data_ptr(),
              out["quaternion"].data_ptr(), None if npj is None else npj.data_ptr(), B, K,
              torch.cuda.current_stream(dev).cuda_stream)
    if npj is not None:
        out["next_kp_projs_est"] = npj
    return out


def is_pnp(prev_kp_pos, prev_kp_projs, next_kp_pos, camera_K, valid=None):
    """geometric_vision.is_pnp for a batch: -> (prev_kp_projs, next_kp_projs_est [B,K,2]) on the device."""
    r = solve_pnp(prev_kp_pos, prev_kp_projs, camera_K, valid=valid, next_points=next_kp_pos)
    return _t(prev_kp_projs, torch.float64, r["retval"].device), r["next_kp_projs_est"]


def register_GN_C(x2d_input, x3d_input, quat_init, T_init, weights, camera, valid=None, max_iter=200):
    """rf_tools/LM.py:256-266 for a batch: x2d [B,K,2], x3d [B,K,3], quat_init [B,4] (w,x,y,z), T_init [B,3], weights
    [B,K,2] or None -> (quat [B,4] wxyz, T [B,3], iterations int32 [B])."""
    dev = torch.device("cuda", torch.cuda.current_device())
    x2d, x3d = _t(x2d_input, torch.float64, dev), _t(x3d_input, torch.float64, dev)
    B, K, _ = x2d.shape
    v0 = torch.cat([_t(quat_init, torch.float64, dev).reshape(B, 4), _t(T_init, torch.float64, dev).reshape(B, 3)], 1).contiguous()
    w = None if weights is None else _t(weights, torch.float64, dev)
    v = None if valid is None else _t(valid, torch.uint8, dev)
    cam = _t(np.asarray(camera, np.float64).reshape(9), torch.float64, dev)
    out = torch.empty(B, 7, dtype=torch.float64, device=dev)
    it = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.call("sgta_gn_refine", v0.data_ptr(), x2d.data_ptr(), x3d.data_ptr(), None if v is None else v.data_ptr(),
              None if w is None else w.data_ptr(), cam.data_ptr(), out.data_ptr(), it.data_ptr(), B, K, int(max_iter),
              torch.cuda.current_stream(dev).cuda_stream)
    return out[:, :4], out[:, 4:], it

"""ctypes binding of libsgta_hip.so (the C ABI declared in include/sgta_hip.h).

The library is the product: there is NO fallback.  If it is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsgta_hip.so")

_p = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_long
_f = ctypes.c_float

# name -> argtypes (return type is always int); must mirror include/sgta_hip.h
SIGNATURES = {
    "sgta_version": [],
    "sgta_check_device": [],
    "sgta_conv2d_f32": [_p, _l, _p, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_conv3x3s1_f32": [_p, _l, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_conv_c16_f32": [_p, _l, _p, _l, _p, _p, _p, _l, _i, _i, _i, _i, _p],
    "sgta_conv3x3_wino_f32": [_p, _l, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _l, _p],
    "sgta_conv3x3_wino4_f32": [_p, _l, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_conv1x1_f32": [_p, _l, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_conv1x1_skinny_f32": [_p, _l, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _p],
    "sgta_conv3x3s2_f32": [_p, _l, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_conv3x3s2_pool_f32": [_p, _l, _p, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_dcnv2_fwd_f32": [_p, _l, _p, _l, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_dcnv2_lds_f32": [_p, _l, _p, _l, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_maxpool2_f32": [_p, _l, _p, _l, _i, _i, _i, _i, _p],
    "sgta_upsample_add_f32": [_p, _l, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _p],
    "sgta_sigmoid_f32": [_p, _p, _l, _p],
    "sgta_prior_argmax": [_p, _p, _p, _i, _i, _i, _i, _p],
    "sgta_window_index": [_p, _p, _p, _i, _i, _f, _i, _i, _i, _p],
    "sgta_window_gather": [_p, _l, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_window_scatter": [_p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p],
    "sgta_linear_f32": [_p, _i, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "sgta_linear_ws_f32": [_p, _i, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _l, _p],
    "sgta_ffn_ln_f32": [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "sgta_mhca_fwd_f32": [_p, _i, _p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    "sgta_mhca_tiled_fwd_f32": [_p, _i, _p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    "sgta_peaks_decode": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "sgta_peaks_decode_full": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "sgta_peaks_decode_kp": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "sgta_peaks_decode_kp_static": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "sgta_peaks_decode_static_max_elems": [],
    "sgta_render_priors": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "sgta_kps_any_valid": [_p, _p, _i, _i, _p],
    "sgta_post_process": [_p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "sgta_conv_bf16": [_p, _l, _l, _p, _l, _p, _p, _l, _p, _l] + [_i] * 14 + [_p],
    "sgta_dcnv2_fwd_bf16": [_p, _l, _p, _i, _p, _l, _p, _p, _l] + [_i] * 9 + [_p],
    "sgta_stem_bf16": [_p, _l, _p, _l, _p, _p, _p, _i, _i, _i, _i, _p],
    "sgta_heads_bf16": [_p, _l, _l, _p, _l, _p, _p, _p, _p, _p, _p, _p] + [_i] * 12 + [_p],
    "sgta_stem_level0_bf16": [_p, _l, _p, _l, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p],
    "sgta_maxpool2_bf16": [_p, _l, _p, _l, _i, _i, _i, _i, _p],
    "sgta_upsample_add_bf16": [_p, _l, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _p],
    "sgta_window_gather_bf16": [_p, _l, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_window_scatter_bf16": [_p, _i, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p],
    # the IEEE-half twins of the 16-bit data path: same arguments as their ``_bf16`` counterparts
    "sgta_conv_f16": [_p, _l, _l, _p, _l, _p, _p, _l, _p, _l] + [_i] * 14 + [_p],
    "sgta_dcnv2_fwd_f16": [_p, _l, _p, _i, _p, _l, _p, _p, _l] + [_i] * 9 + [_p],
    "sgta_stem_f16": [_p, _l, _p, _l, _p, _p, _p, _i, _i, _i, _i, _p],
    "sgta_heads_f16": [_p, _l, _l, _p, _l, _p, _p, _p, _p, _p, _p, _p] + [_i] * 12 + [_p],
    "sgta_stem_level0_f16": [_p, _l, _p, _l, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p],
    "sgta_maxpool2_f16": [_p, _l, _p, _l, _i, _i, _i, _i, _p],
    "sgta_upsample_add_f16": [_p, _l, _p, _p, _l, _p, _l, _i, _i, _i, _i, _i, _p],
    "sgta_window_gather_f16": [_p, _l, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_window_scatter_f16": [_p, _i, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p],
    # fp32 emulated on the 16-bit matrix pipe by operand splitting (two IEEE-half planes per activation tensor)
    "sgta_conv_x3": [_p, _l, _l, _l, _p, _l, _p, _p, _l, _l, _p, _l, _l] + [_i] * 14 + [_p],
    "sgta_heads_x3": [_p, _l, _l, _l, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p] + [_i] * 12 + [_p],
    "sgta_maxpool2_x3": [_p, _l, _l, _p, _l, _l, _i, _i, _i, _i, _p],
    "sgta_window_gather_x3": [_p, _l, _l, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_window_scatter_x3": [_p, _i, _p, _p, _p, _l, _l, _i, _i, _i, _i, _i, _p],
    "sgta_nchw_to_x3": [_p, _l, _p, _l, _l, _i, _i, _i, _i, _p],
    "sgta_x3_to_nchw": [_p, _l, _l, _p, _l, _i, _i, _i, _i, _p],
    "sgta_stem_level0_x3": [_p, _l, _p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _l, _i, _i, _i, _i, _p],
    "sgta_dcnv2_x3": [_p, _l, _p, _l, _p, _l, _p, _p, _l, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "sgta_pre_process": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "sgta_pnp_batched": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "sgta_gn_refine": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "sgta_softargmax": [_p, _p, _i, _i, _i, _i, _f, _f, _p],
}

_lib = None


class SgtaHipError(RuntimeError):
    pass


def load():
    """dlopen the library once; raises SgtaHipError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SgtaHipError(
            "libsgta_hip.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C sgtapose_amd/csrc).  There is no CPU / PyTorch fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = header / library out of sync: loud
        fn.argtypes = argtypes
        fn.restype = ctypes.c_long if name.endswith("_max_elems") else ctypes.c_int
    _lib = lib
    return lib


# When a list, every C-ABI call is bracketed by events on the launch stream and appended as
# (name, meta, start_event, end_event) -- bench.py's live per-kernel timing (never on by default).
PROFILE = None


def call(name, *args, meta=None):
    if PROFILE is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(load(), name)(*args)
        e1.record()
        PROFILE.append((name, meta, e0, e1))
    else:
        rc = getattr(load(), name)(*args)
    if rc != 0:
        kind = ("size beyond the entry point's stated maximum" if rc == -2 else "bad argument") if rc < 0 else "hipError_t"
        raise SgtaHipError("%s failed: %s %d" % (name, kind, rc))
    return rc

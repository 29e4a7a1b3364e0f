"""sgtapose_amd: MI355X-native (gfx950) SGTAPose inference hot path behind the reference's API.

    from sgtapose_amd.model import create_model, load_model, default_opt
    model = create_model("dlapawdl3new_34", heads, head_conv, opt).cuda().eval()
    out = model(x, pre_img, pre_hm, repro_hm, pre_hm_cls, repro_hm_cls)[-1]     # HIP kernels

The compute path is libsgta_hip.so (sgtapose_amd/csrc, C ABI in include/sgta_hip.h); importing this
package does not load it, calling any op without it raises.
"""
__version__ = "0.1.0"

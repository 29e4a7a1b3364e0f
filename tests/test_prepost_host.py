"""Host logic of the pre/post-processing mirror (no GPU): affine matrices and the reference-shaped assembly."""
import numpy as np

import cases
from util import golden


def test_get_affine_transform_matches_fixture_and_inverts():
    from sgtapose_amd import prepost
    g = golden("prepost_ops.npz")
    cs = cases.prepost_case()
    for b in range(len(cs["raws"])):
        t_in = prepost.get_affine_transform(cs["c"][b], cs["s"][b], 0, [cs["inp_w"], cs["inp_h"]])
        t_out = prepost.get_affine_transform(cs["c"][b], cs["s"][b], 0, [cs["out_w"], cs["out_h"]])
        t_inv = prepost.get_affine_transform(cs["c"][b], cs["s"][b], 0, (cs["out_w"], cs["out_h"]), inv=1)
        assert t_in.dtype == np.float64 and t_in.shape == (2, 3)
        assert np.array_equal(t_in, g["trans_input"][b]) and np.array_equal(t_out, g["trans_output"][b])
        assert np.array_equal(t_inv, g["trans_inv"][b])
        # inv really inverts: a centre maps to the middle of the output and back
        mid = t_out @ np.array([cs["c"][b][0], cs["c"][b][1], 1.0])
        assert np.allclose(mid, [cs["out_w"] / 2.0, cs["out_h"] / 2.0], atol=1e-9)
        assert np.allclose(t_inv @ np.array([mid[0], mid[1], 1.0]), cs["c"][b], atol=1e-6)


def test_reference_shaped_assembly_from_device_outputs():
    """prepost.dream_generic_post_process builds the reference's list-of-dicts from the kernel outputs: fed with the
    golden arrays it reproduces the reference's detections and (through the oracle) its key-point table."""
    from oracle import prepost as op
    from sgtapose_amd import prepost
    g = golden("prepost_ops.npz")
    cs = cases.prepost_case()
    pp = {k: g["pp_" + k] for k in ("valid", "ct", "ct_wreg", "reg", "tracking")}
    items = prepost.dream_generic_post_process(cs["dets"], pp, cs["K"])
    ref = op.dream_generic_post_process(cs["dets"], g["trans_inv"], cs["out_thresh"])
    assert [len(x) for x in items] == [len(x) for x in ref]
    for a, b in zip(items, ref):
        for x, y in zip(a, b):
            assert x["class"] == y["class"] and x["score"] == y["score"]
            for k in ("ct", "ct_wreg", "reg", "tracking", "ct_wreg_output", "reg_output", "ct_output"):
                assert np.array_equal(np.asarray(x[k]), np.asarray(y[k])), k
    for b in range(len(items)):
        assert np.array_equal(op.get_final_kps(items[b], cs["K"]), g["final_kps"][b])
    assert prepost.MISSING == -999.999 * 4

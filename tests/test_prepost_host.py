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


def test_warp_affine_restatement_is_a_bilinear_warp_and_matches_host_mirror():
    """oracle.prepost.warp_affine_u8 (OpenCV's published 8-bit warpAffine, unpinned: OpenCV absent) against scipy's float
    bilinear interpolation: within half a grey level (the 1/32-pixel coordinate grid + integer rounding); and the
    product's host-side matrix inversion equals the oracle's."""
    from scipy.ndimage import gaussian_filter, map_coordinates
    from oracle import prepost as op
    from sgtapose_amd import prepost as hp
    rng = np.random.RandomState(0)
    img = np.clip(gaussian_filter(rng.randint(0, 256, (120, 160, 3)).astype(np.float64), (3, 3, 0)), 0, 255).astype(np.uint8)
    for (c, s, size, tol) in ((np.array([80., 60.], np.float32), 160.0, (128, 96), 0.75),
                              (np.array([80., 60.], np.float32), 100.0, (64, 64), 0.75),
                              (np.array([60., 70.], np.float32), 400.0, (96, 128), 2.5)):
        M = op.get_affine_transform(c, s, 0, list(size))
        w = op.warp_affine_u8(img, M, size)
        iM = op.invert_affine(M)
        assert np.array_equal(iM, hp.invert_affine(M))
        ys, xs = np.mgrid[0:size[1], 0:size[0]]
        sx = iM[0, 0] * xs + iM[0, 1] * ys + iM[0, 2]
        sy = iM[1, 0] * xs + iM[1, 1] * ys + iM[1, 2]
        ref = np.stack([map_coordinates(img[..., k].astype(np.float64), [sy, sx], order=1, mode="grid-constant", cval=0)
                        for k in range(3)], -1)
        assert np.abs(w.astype(np.float64) - ref).max() <= tol

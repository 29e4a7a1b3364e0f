"""The allocation-free decode entry point (sgta_peaks_decode_kp_static) against sgta_peaks_decode_kp, through the C ABI:
two calls in a row on one GPU, the second with a smaller input (the library's scratch array keeps the first call's
contents), bit-exact; and a size beyond the stated maximum is refused with a clear error."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _maps(B, K, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    hm = torch.zeros(B, K, H, W)          # a flat background: the filtered map has no spurious maxima above 0.01
    for b in range(B):
        for k in range(K):
            if (b + k) % 5 == 4:
                continue                                   # a map without a peak
            y, x = int(torch.randint(3, H - 3, (1,), generator=g)), int(torch.randint(3, W - 3, (1,), generator=g))
            yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
            hm[b, k] += 0.9 * torch.exp(-((yy - y) ** 2 + (xx - x) ** 2) / 8.0)
            if (b + k) % 7 == 3:                           # a second, nearly equal peak: the multi-peak rule rejects the map
                hm[b, k] += 0.85 * torch.exp(-((yy - (H - 1 - y)) ** 2 + (xx - (W - 1 - x)) ** 2) / 8.0)
    reg = torch.rand(B, 2, H, W, generator=g)
    trk = torch.randn(B, 2, H, W, generator=g)
    return hm.clamp_(0, 1), reg, trk


def test_static_decode_matches_workspace_decode_twice_second_call_smaller():
    from sgtapose_amd import _lib, ops
    dev = torch.device("cuda", 0)
    for B, H, W, seed in ((6, 120, 160, 1), (2, 32, 40, 2)):       # second call smaller: stale scratch must not matter
        hm, reg, trk = [t.to(dev) for t in _maps(B, 7, H, W, seed)]
        ref = ops.peaks_decode(hm, reg, trk)
        out = ops.decode_buffers(B, 7, dev, True)
        for t in out.values():
            t.fill_(-7)
        ops.peaks_decode_static(hm, reg, trk, out)
        torch.cuda.synchronize()
        assert int((ref["scores"] >= 0).sum()) > 0
        for k in ref:
            assert torch.equal(ref[k], out[k]), (k, B, H, W)
    lim = _lib.load().sgta_peaks_decode_static_max_elems()
    assert lim == 4718592
    big = torch.zeros(1, 7, 1024, 1024, device=dev)                 # 7.3 M elements > the stated maximum
    with pytest.raises(_lib.SgtaHipError, match="exceeds"):
        ops.peaks_decode_static(big, None, None, ops.decode_buffers(1, 7, dev, False))
    # and through the raw C ABI: the library itself refuses, with its own code
    import ctypes
    rc = _lib.load().sgta_peaks_decode_kp_static(big.data_ptr(), None, None, ops.gauss_taps(dev).data_ptr(), *([None] * 10),
                                                  1, 7, 1024, 1024, 0, None)
    assert rc == -2

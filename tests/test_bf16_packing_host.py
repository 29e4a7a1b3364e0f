"""Host-side packers of the fused bf16 kernels (no GPU): the MFMA fragment images that ops_bf16.PackedStemL0BF16 and
PackedHeadsBF16 build are contracted here exactly as the kernels' MFMAs contract them (A[row][k] . B[k][col] with the k index
split over the lane halves / quarters) and compared with plain torch convolutions on the same bf16-rounded weights."""
import pytest
import torch
import torch.nn.functional as F

from sgtapose_amd import ops_bf16 as bops

PERM = (0, 2, 1, 3)


def _bn(g, c=16):
    return (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1,
            torch.rand(c, generator=g) + 0.5)


def test_stem_level0_fragments_contract_to_the_convolutions():
    g = torch.Generator().manual_seed(1)
    wi, wh = torch.randn(16, 3, 7, 7, generator=g) * 0.1, torch.randn(16, 1, 7, 7, generator=g) * 0.1
    wl = torch.randn(16, 16, 3, 3, generator=g) * 0.1
    pk = bops.PackedStemL0BF16(wi, _bn(g), wh, _bn(g), wl, _bn(g))
    assert tuple(pk.w_stem.shape) == (7, 2, 64, 8) and tuple(pk.w_l0.shape) == (5, 64, 8)
    img = torch.randn(3, 7, 8, generator=g)                     # one pixel's 7 x 8 window (the 8th column meets zero weights)
    hm = torch.randn(1, 7, 8, generator=g)
    x4 = torch.cat([img, hm], 0)                                # [4 channels][ky][kx]
    ws = pk.w_stem.float()
    lane = torch.arange(64)
    m, gq = lane & 15, lane >> 4
    out = torch.zeros(2, 16)                                    # [stem][MFMA row]
    for ky in range(7):
        for st in range(2):
            for j in range(8):
                kx, ch = 2 * gq + (j >> 2), j & 3
                out[st].index_add_(0, m, ws[ky, st, :, j] * x4[ch, ky, kx])
    cout = torch.tensor([4 * PERM[r >> 2] + (r & 3) for r in range(16)])
    ref_i = (pk.w_img_ref.float() * img[None, :, :, :7]).sum((1, 2, 3))
    ref_h = (pk.w_hm_ref.float() * hm[None, :, :, :7]).sum((1, 2, 3))
    assert torch.allclose(out[0], ref_i[cout], atol=1e-5) and torch.allclose(out[1], ref_h[cout], atol=1e-5)
    # level0: k-step s, lane group g: tap 2s + (g >> 1), channels 8 (g & 1) .. + 7 of the 3x3 stem neighbourhood
    nb = torch.randn(16, 3, 3, generator=g)
    wf = pk.w_l0.float()
    m2, g2 = lane & 15, lane >> 4
    acc = torch.zeros(16)
    for s in range(5):
        for j in range(8):
            t = 2 * s + (g2 >> 1)
            ch = 8 * (g2 & 1) + j
            v = torch.where(t < 9, nb.reshape(16, 9)[ch, t.clamp(max=8)], torch.zeros(()))
            acc.index_add_(0, m2, wf[s, :, j] * v)
    ref_l = (pk.w_l0_ref.float() * nb[None]).sum((1, 2, 3))
    assert torch.allclose(acc, ref_l[cout], atol=1e-5)


@pytest.mark.parametrize("rows", [(7, 2, 2), (7, 2, 2, 2)])
def test_head_fragments_contract_to_the_block_diagonal_product(rows):
    g = torch.Generator().manual_seed(2)
    heads = [(torch.randn(256, 64, 3, 3, generator=g) * 0.05, torch.randn(256, generator=g) * 0.1,
              torch.randn(n, 256, 1, 1, generator=g) * 0.1, torch.randn(n, generator=g)) for n in rows]
    pk = bops.PackedHeadsBF16(heads)
    chid = 256 * len(rows)
    assert pk.Chid == chid and tuple(pk.w2.shape) == (chid // 128, 4, 2, 64, 8)
    hidden = torch.randn(chid, generator=g)                     # one pixel's hidden activations
    lane = torch.arange(64)
    m, h = lane & 31, lane >> 5
    out = torch.zeros(32)
    w2 = pk.w2.float()
    for ct in range(chid // 128):
        for i in range(4):
            for t in range(2):
                for j in range(8):
                    ch = 128 * ct + 32 * i + 8 * (2 * t + (j >> 2)) + 4 * h + (j & 3)
                    out.index_add_(0, m, w2[ct, i, t, :, j] * hidden[ch])
    ref, r0 = torch.zeros(32), 0
    for k, (_, _, w, _) in enumerate(heads):
        n = w.shape[0]
        ref[r0:r0 + n] = w.reshape(n, 256).to(torch.bfloat16).float() @ hidden[256 * k:256 * (k + 1)]
        r0 += n
    assert torch.allclose(out, ref, atol=1e-4)
    assert torch.equal(pk.b2[:r0], torch.cat([hd[3] for hd in heads])) and float(pk.b2[r0:].abs().max() if r0 < 16 else 0) == 0


def test_heads_packer_rejects_what_the_kernel_cannot_hold():
    g = torch.Generator().manual_seed(3)
    mk = lambda hid, n: (torch.randn(hid, 64, 3, 3, generator=g), torch.zeros(hid), torch.randn(n, hid, 1, 1, generator=g), torch.zeros(n))  # noqa: E731
    from sgtapose_amd._lib import SgtaHipError
    with pytest.raises(SgtaHipError):
        bops.PackedHeadsBF16([mk(256, 9), mk(256, 9)])           # 18 output rows > 16
    with pytest.raises(SgtaHipError):
        bops.PackedHeadsBF16([mk(96, 2)])                        # hidden width not a multiple of 128 (and na != 4)

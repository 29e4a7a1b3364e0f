"""Seeded INPUTS of the per-operator golden cases (shared by make_golden.py and the tests, so the
fixtures only need to store the reference's outputs).  Contains no reference code."""
import numpy as np
import torch

LEVEL_SCALE = (4, 2, 1, 1 / 2, 1 / 4, 1 / 8)
LEVEL_KS = (12, 6, 3, 1, 1, 1)


def window_case():
    """Prior maps on a 32x32 grid (sample 0: corners / borders / overlaps / a tie; sample 1: random,
    one all-zero class) and random feature pyramids [2, 16*2^i, 128>>i, 128>>i]."""
    g = torch.Generator().manual_seed(11)
    h = 32
    pre_cls = torch.zeros(2, 7, h, h)
    cur_cls = torch.zeros(2, 7, h, h)
    pts = [(0, 0), (31, 31), (0, 17), (30, 1), (15, 15), (16, 15), (5, 31)]
    for k, (x, y) in enumerate(pts):
        pre_cls[0, k, y, x] = 1.0
        cur_cls[0, k, (y + 1) % h, x] = 0.7
    cur_cls[0, 3, 4, 9] = 0.7                    # tie with the later blob -> lowest flat index wins
    pre_cls[1] = torch.rand(7, h, h, generator=g)
    cur_cls[1] = torch.rand(7, h, h, generator=g)
    cur_cls[1, 6] = 0                            # all-zero map -> index 0
    levels = []
    for i in range(6):
        C, H = 16 * 2 ** i, (4 * h) >> i
        pre_f = torch.randn(2, C, H, H, generator=g)
        cur_f = torch.randn(2, C, H, H, generator=g)
        n = 7 * (1 + 2 * (LEVEL_KS[i] // 2)) ** 2
        tok = torch.randn(2, n, C, generator=g)
        levels.append((pre_f, cur_f, tok))
    # non-square maps (reference needs python -O): 24x40 grid
    pcn = torch.zeros(1, 7, 24, 40)
    for k, (x, y) in enumerate([(39, 23), (0, 0), (25, 3), (38, 22), (13, 13), (23, 23), (30, 7)]):
        pcn[0, k, y, x] = 1.0
    fn = torch.randn(1, 8, 24, 40, generator=g)
    fn2 = torch.randn(1, 8, 12, 20, generator=g)
    return dict(pre_cls=pre_cls, cur_cls=cur_cls, levels=levels, ns_cls=pcn, ns_f=fn, ns_f2=fn2)


def encoder_case():
    g = torch.Generator().manual_seed(23)
    out = []
    for i, n in enumerate((1183, 343, 63)):
        C = 16 * 2 ** i
        B = 2 if i else 1
        q = torch.randn(B, n, C, generator=g)
        k = torch.randn(B, n, C, generator=g)
        out.append((q, k))
    return out


def decode_case():
    """Crafted post-sigmoid maps [2,7,40,48] + random reg / tracking heads."""
    from sgtapose_amd.synthetic import peaked_heatmaps
    H, W = 40, 48
    hm, _ = peaked_heatmaps(2, H, W, seed=5)
    m = hm.numpy().copy()
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)

    def blob(cx, cy, a):
        return a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / 8.0)

    m[0, 1] = 0.002 + blob(10.3, 9.6, 0.8) + blob(35.2, 28.1, 0.7)    # two peaks, |gap| < 0.25 -> missing
    m[0, 2] = 0.002 + blob(12.5, 30.2, 0.9) + blob(33.0, 8.0, 0.3)    # larger-y peak is the strong one -> kept
    m[0, 3] = 0.002 + blob(12.5, 8.2, 0.9) + blob(33.0, 30.0, 0.3)    # larger-y peak is the weak one -> missing
    m[0, 4] = 0.002                                                    # flat below threshold -> no peak
    m[0, 5] = 0.002 + blob(0.2, 0.4, 0.9)                              # border peak
    m[0, 6] = 0.002 + blob(46.8, 38.9, 0.9)                            # far-corner peak
    m[1, 0] = 0.0105                                                   # flat above threshold: every pixel a peak
    m[1, 1] = 0.0
    m[1, 1, 20, 20] = 0.05                                             # lone spike: filtered max < threshold
    hm = torch.from_numpy(m.astype(np.float32))
    g = torch.Generator().manual_seed(7)
    reg = torch.rand(2, 2, H, W, generator=g)
    trk = torch.randn(2, 2, H, W, generator=g)
    return hm, reg, trk


def prepost_case():
    """Inputs of the prior-rendering / post-processing fixtures (numpy, seeded): B samples with different raw image
    sizes, key-points inside / outside the raw image, near the map border (patch skipped), on integer coordinates,
    overlapping pairs, and a sample without priors; detections with scores around the threshold."""
    import numpy as np
    rng = np.random.RandomState(2023)
    K, inp_w, inp_h, out_w, out_h = 7, 160, 128, 40, 32
    raws = [(640, 480), (1280, 720), (160, 128), (333, 257), (640, 480)]
    B = len(raws)
    c = np.array([[w / 2.0, h / 2.0] for (w, h) in raws], np.float32)
    s = np.array([max(w, h) * 1.0 for (w, h) in raws], np.float32)
    prev = np.zeros((B, K, 2), np.float64)
    nxt = np.zeros((B, K, 2), np.float64)
    for b, (w, h) in enumerate(raws):
        prev[b] = rng.uniform([0, 0], [w, h], size=(K, 2))
        nxt[b] = prev[b] + rng.normal(0, 3.0, size=(K, 2))
    prev[0, 0] = [-5.0, 100.0]                     # outside the raw image -> (0, 0) -> not drawn
    prev[0, 1] = [639.999, 479.999]               # inside, lands near the map corner -> patch skipped
    prev[0, 2] = [320.0, 240.0]                   # integer coordinates
    prev[0, 3] = [322.0, 241.0]                   # overlaps key-point 2
    prev[1, 0] = [1280.0, 10.0]                   # x == raw_width -> outside
    prev[2, 4] = [20.0, 20.0]
    prev[2, 5] = [4.2, 60.0]                      # x - radius >= 0 holds exactly / fails
    prev[2, 6] = [3.9, 60.0]
    nxt[3, :] = -3999.996                          # "missing" key-points of the previous step
    has_prior = np.array([1, 1, 1, 1, 0], np.uint8)
    scores = rng.uniform(0.0, 1.0, size=(B, K)).astype(np.float32)
    scores[1, 2] = 0.01
    scores[3, :] = -1.0                            # decode's "no peak" marker
    xs = rng.randint(0, out_w, size=(B, K)).astype(np.int64)
    ys = rng.randint(0, out_h, size=(B, K)).astype(np.int64)
    regs = rng.uniform(0, 1, size=(B, K, 2, 1)).astype(np.float32)
    trk = rng.normal(0, 2, size=(B, K, 2)).astype(np.float32)
    dets = {"scores": scores, "clses": np.tile(np.arange(K, dtype=np.int64), (B, 1)),
            "xs": xs, "ys": ys, "cts": np.stack([xs, ys], -1),
            "cts_wreg": (np.stack([xs, ys], -1).astype(np.float32)[..., None] + regs).astype(np.float32),
            "regs": regs, "tracking": trk}
    return dict(K=K, inp_w=inp_w, inp_h=inp_h, out_w=out_w, out_h=out_h, raws=raws, c=c, s=s, prev=prev, nxt=nxt,
                has_prior=has_prior, dets=dets, out_thresh=0.1)


def pnp_case():
    """Seeded pose problems for the PnP / Gauss-Newton fixtures: a 7-key-point 'robot' (metres), a camera like the
    reference's (fx = fy = 615.5, 640x480), poses in front of the camera, pixel noise, samples with missing key-points."""
    rng = np.random.RandomState(77)
    K = np.array([[615.5, 0.0, 320.0], [0.0, 615.5, 240.0], [0.0, 0.0, 1.0]])
    x3d = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.33], [0.1, -0.05, 0.65], [0.3, 0.0, 0.75], [0.45, 0.1, 0.6],
                    [0.55, 0.0, 0.45], [0.6, -0.1, 0.3]])
    cases_ = []
    for i in range(12):
        ax = rng.randn(3)
        ax /= np.linalg.norm(ax)
        ang = rng.uniform(0.2, 2.6)
        rvec = ax * ang
        t = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(1.0, 2.5)])
        noise = rng.randn(7, 2) * (0.0 if i < 3 else 0.7)
        valid = np.ones(7, bool)
        if i % 4 == 3:
            valid[rng.choice(7, 2, replace=False)] = False
        if i == 10:
            valid[:] = False
            valid[[0, 3, 5, 6]] = True                       # exactly four points
        cases_.append(dict(rvec=rvec, t=t, noise=noise, valid=valid))
    return K, x3d, cases_

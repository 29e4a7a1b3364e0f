"""CPU oracle (numpy, fp64) for the pose step downstream of the key-points  --  TEST INFRASTRUCTURE ONLY.

  * ``solve_pnp``      sgtapose/geometric_vision.py:43-116   cv2.solvePnP(EPNP) then cv2.solvePnP(ITERATIVE,
                       useExtrinsicGuess) -> (retval, translation, quaternion xyzw)
  * ``is_pnp``         sgtapose/geometric_vision.py:283-310  pose of the previous frame -> projections of the next
                       frame's 3-D key-points (what the prior heat-maps are rendered from)
  * ``GN`` / ``fun`` / ``dfun``  sgtapose/rf_tools/LM.py:128-234 (the python twin of ``extern "C" LM``, LM.py:256-266):
                       weighted Gauss-Newton on (qw,qx,qy,qz,tx,ty,tz)

Parity status
-------------
* ``gn_refine`` (fun / dfun / GN) is PINNED: tests/golden/pnp_ops.npz holds the outputs of the reference's own python
  functions, loaded from sgtapose/rf_tools/LM.py in the build container (its sympy-generated expressions are the
  derivatives of the closed forms below).
* ``solve_pnp`` is UNPINNED: OpenCV is neither in the image nor vendored.  EPnP is restated from the published
  algorithm (Lepetit, Moreno-Noguer, Fua, IJCV 2009; the structure of OpenCV's modules/calib3d/src/epnp.cpp: PCA
  control points, 12x12 null space, three beta approximations + 5 Gauss-Newton steps, Horn alignment, best of three
  by reprojection error, on K-normalised image points) and the ITERATIVE refinement as a Levenberg-Marquardt descent
  of the pixel reprojection error over (rvec, tvec) from that guess -- both paths end in the same least-squares
  minimum, which is what "identical pose" can mean without the library.  ``Quaternion.matrix33`` (pyrr, absent) is
  taken to be the standard rotation matrix of the unit quaternion.
"""
import numpy as np

PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))


# ---- small helpers ------------------------------------------------------------------------------------------------
def rodrigues(rvec):
    th = float(np.sqrt(rvec @ rvec))
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def rvec_from_R(R):
    """Inverse Rodrigues (rotation matrix -> axis * angle)."""
    c = (np.trace(R) - 1.0) / 2.0
    c = min(1.0, max(-1.0, c))
    th = np.arccos(c)
    if th < 1e-10:
        return np.zeros(3)
    if np.pi - th < 1e-6:                           # near pi: axis from the symmetric part
        A = (R + np.eye(3)) / 2.0
        ax = np.sqrt(np.maximum(np.diag(A), 0))
        i = int(np.argmax(ax))
        ax = A[i] / ax[i]
        return ax / np.linalg.norm(ax) * th
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(th))
    return w * th


def quat_xyzw_from_rvec(rvec):
    """convert_rvec_to_quaternion (geometric_vision.py:20-27): Quaternion.from_axis_rotation(axis, theta) -> xyzw."""
    th = float(np.sqrt(rvec @ rvec))
    if th < 1e-12:
        return np.array([0.0, 0.0, 0.0, 1.0])
    ax = rvec / th
    return np.concatenate([ax * np.sin(th / 2), [np.cos(th / 2)]])


def R_from_quat_xyzw(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


# ---- EPnP -----------------------------------------------------------------------------------------------------------
def _horn(pcs, pws):
    pc0, pw0 = pcs.mean(0), pws.mean(0)
    abt = (pcs - pc0).T @ (pws - pw0)
    U, _, Vt = np.linalg.svd(abt)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R[2] *= -1
    return R, pc0 - R @ pw0


def _rep_error(R, t, pws, uv):
    Xc = pws @ R.T + t
    return float(np.mean(np.sqrt((uv[:, 0] - Xc[:, 0] / Xc[:, 2]) ** 2 + (uv[:, 1] - Xc[:, 1] / Xc[:, 2]) ** 2)))


def epnp(pws, uv):
    """pws [n,3] object points, uv [n,2] K-normalised image points (n >= 4) -> (R, t)."""
    n = len(pws)
    c0 = pws.mean(0)
    P0 = pws - c0
    dc, vec = np.linalg.eigh(P0.T @ P0)
    order = np.argsort(-dc)
    cws = np.vstack([c0] + [c0 + np.sqrt(max(dc[i], 0) / n) * vec[:, i] for i in order])
    CC = (cws[1:] - cws[0]).T
    a123 = np.linalg.solve(CC, (pws - cws[0]).T).T
    alphas = np.hstack([1 - a123.sum(1, keepdims=True), a123])                     # [n,4]
    M = np.zeros((2 * n, 12))
    for j in range(4):
        M[0::2, 3 * j] = alphas[:, j]
        M[0::2, 3 * j + 2] = alphas[:, j] * (-uv[:, 0])
        M[1::2, 3 * j + 1] = alphas[:, j]
        M[1::2, 3 * j + 2] = alphas[:, j] * (-uv[:, 1])
    ev, evec = np.linalg.eigh(M.T @ M)
    v = [evec[:, i].reshape(4, 3) for i in np.argsort(ev)[:4]]                      # smallest eigenvalue first
    L = np.zeros((6, 10))
    rho = np.zeros(6)
    for r_, (a, b) in enumerate(PAIRS):
        dv = [v[i][a] - v[i][b] for i in range(4)]
        L[r_] = [dv[0] @ dv[0], 2 * dv[0] @ dv[1], dv[1] @ dv[1], 2 * dv[0] @ dv[2], 2 * dv[1] @ dv[2], dv[2] @ dv[2],
                 2 * dv[0] @ dv[3], 2 * dv[1] @ dv[3], 2 * dv[2] @ dv[3], dv[3] @ dv[3]]
        rho[r_] = np.sum((cws[a] - cws[b]) ** 2)

    def approx1():
        b4 = np.linalg.lstsq(L[:, [0, 1, 3, 6]], rho, rcond=None)[0]
        b0 = np.sqrt(abs(b4[0]))
        s = -1.0 if b4[0] < 0 else 1.0
        return np.array([b0, s * b4[1] / b0, s * b4[2] / b0, s * b4[3] / b0])

    def approx2():
        b3 = np.linalg.lstsq(L[:, [0, 1, 2]], rho, rcond=None)[0]
        if b3[0] < 0:
            be = [np.sqrt(-b3[0]), np.sqrt(-b3[2]) if b3[2] < 0 else 0.0]
        else:
            be = [np.sqrt(b3[0]), np.sqrt(b3[2]) if b3[2] > 0 else 0.0]
        if b3[1] < 0:
            be[0] = -be[0]
        return np.array([be[0], be[1], 0.0, 0.0])

    def approx3():
        b5 = np.linalg.lstsq(L[:, [0, 1, 2, 3, 4]], rho, rcond=None)[0]
        if b5[0] < 0:
            be = [np.sqrt(-b5[0]), np.sqrt(-b5[2]) if b5[2] < 0 else 0.0]
        else:
            be = [np.sqrt(b5[0]), np.sqrt(b5[2]) if b5[2] > 0 else 0.0]
        if b5[1] < 0:
            be[0] = -be[0]
        return np.array([be[0], be[1], b5[3] / be[0], 0.0])

    best = None
    for init in (approx1, approx2, approx3):
        be = init()
        for _ in range(5):                                                   # gauss_newton on the four betas
            b0, b1, b2, b3 = be
            A = np.stack([2 * L[:, 0] * b0 + L[:, 1] * b1 + L[:, 3] * b2 + L[:, 6] * b3,
                          L[:, 1] * b0 + 2 * L[:, 2] * b1 + L[:, 4] * b2 + L[:, 7] * b3,
                          L[:, 3] * b0 + L[:, 4] * b1 + 2 * L[:, 5] * b2 + L[:, 8] * b3,
                          L[:, 6] * b0 + L[:, 7] * b1 + L[:, 8] * b2 + 2 * L[:, 9] * b3], 1)
            quad = np.array([b0 * b0, b0 * b1, b1 * b1, b0 * b2, b1 * b2, b2 * b2, b0 * b3, b1 * b3, b2 * b3, b3 * b3])
            be = be + np.linalg.lstsq(A, rho - L @ quad, rcond=None)[0]
        ccs = sum(be[k] * v[k] for k in range(4))                                 # [4,3] camera-frame control points
        pcs = alphas @ ccs
        if pcs[0, 2] < 0:
            pcs = -pcs
        R, t = _horn(pcs, pws)
        err = _rep_error(R, t, pws, uv)
        if best is None or err < best[0]:
            best = (err, R, t)
    return best[1], best[2]


# ---- iterative refinement (cv2.SOLVEPNP_ITERATIVE with useExtrinsicGuess) ------------------------------------------------
def project(K, R, t, pws):
    Xc = pws @ R.T + t
    return np.stack([K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]], 1)


def refine_lm(K, rvec, t, pws, x2d, iters=60):
    """Levenberg-Marquardt on the pixel reprojection error over (rvec, t); central-difference Jacobian in fp64."""
    p = np.concatenate([rvec, t]).astype(np.float64)

    def resid(q):
        return (project(K, rodrigues(q[:3]), q[3:], pws) - x2d).reshape(-1)

    lam = 1e-3
    r = resid(p)
    for _ in range(iters):
        J = np.zeros((len(r), 6))
        for k in range(6):
            h = 1e-6 * max(1.0, abs(p[k]))
            d = np.zeros(6)
            d[k] = h
            J[:, k] = (resid(p + d) - resid(p - d)) / (2 * h)
        JtJ, g = J.T @ J, J.T @ r
        for _try in range(10):
            step = np.linalg.solve(JtJ + lam * np.diag(np.diag(JtJ)), -g)
            r2 = resid(p + step)
            if r2 @ r2 < r @ r:
                p, r, lam = p + step, r2, max(lam * 0.1, 1e-12)
                break
            lam *= 10.0
        else:
            break
        if np.abs(step).max() < 1e-13:
            break
    return p[:3], p[3:]


def solve_pnp(canonical_points, projections, camera_K):
    """geometric_vision.py:43-116 (refinement=True, no distortion).  Rows holding None / NaN are dropped as the
    reference drops them; < 4 points (or a failed solve) -> (False, None, None).  -> (retval, translation [3],
    quaternion xyzw [4])."""
    pw, px = [], []
    for c, p in zip(canonical_points, projections):
        if c is None or p is None or len(c) == 0 or len(p) == 0 or c[0] is None or c[1] is None or p[0] is None or p[1] is None:
            continue
        pw.append(c)
        px.append(p)
    if len(pw) < 4:
        return False, None, None
    pw, px = np.asarray(pw, np.float64), np.asarray(px, np.float64)
    K = np.asarray(camera_K, np.float64)
    uv = np.stack([(px[:, 0] - K[0, 2]) / K[0, 0], (px[:, 1] - K[1, 2]) / K[1, 1]], 1)
    try:
        R, t = epnp(pw, uv)
        rvec, t = refine_lm(K, rvec_from_R(R), t, pw, px)
    except np.linalg.LinAlgError:
        return False, None, None
    if not (np.all(np.isfinite(rvec)) and np.all(np.isfinite(t))):
        return False, None, None
    return True, t, quat_xyzw_from_rvec(rvec)


def is_pnp(prev_kp_pos, prev_kp_projs, next_kp_pos, prev_kp_projs_all, camera_K):
    """geometric_vision.py:283-310 -> (prev_kp_projs_all, next_kp_projs_est [K,2]); on failure both are prev_kp_projs_all."""
    ok, t, q = solve_pnp(prev_kp_pos, prev_kp_projs, camera_K)
    if not ok:
        return prev_kp_projs_all, prev_kp_projs_all
    T = np.eye(4)
    T[:3, :3] = R_from_quat_xyzw(q)
    T[:3, 3] = t
    hom = np.hstack([np.asarray(next_kp_pos, np.float64), np.ones((len(next_kp_pos), 1))])
    aligned = (T @ hom.T).T[:, :3]
    pr = (np.asarray(camera_K, np.float64) @ aligned.T).T
    pr[:, 0] /= pr[:, 2]
    pr[:, 1] /= pr[:, 2]
    return prev_kp_projs_all, pr[:, :2]


# ---- rf_tools/LM.py: weighted Gauss-Newton refinement (the python twin of the C ``LM``) ----------------------------------
def _rot_point(q, p):
    """q (x) p (x) conj(q) for a NON-normalised quaternion q = (qw,qx,qy,qz): |q|^2 * R(q) p  (LM.py:141-142 expands it)."""
    qw, qx, qy, qz = q
    x, y, z = p
    a = qw * x + qy * z - qz * y           # the four terms of q (x) p, as the reference groups them
    b = qw * y - qx * z + qz * x
    c = qw * z + qx * y - qy * x
    d = -qx * x - qy * y - qz * z
    return np.array([qw * a - qx * d + qy * c - qz * b,
                     qw * b - qx * c - qy * d + qz * a,
                     qw * c + qx * b - qy * a - qz * d])


def gn_fun(value, x2d, x3d, weights, camera):
    """LM.py:128-158: f = [wx^2 (x2d - u)^2, wy^2 (y2d - v)^2 per point ..., 2e8 (|q|^2 - 1)^2]."""
    fx, cx, fy, cy = camera[0][0], camera[0][2], camera[1][1], camera[1][2]
    q, t = value[:4], value[4:]
    f = []
    for i in range(len(x2d)):
        X = _rot_point(q, x3d[i]) + t
        u = (cx * X[2] + fx * X[0]) / X[2]
        v = (cy * X[2] + fy * X[1]) / X[2]
        f.append(weights[i][0] ** 2 * (x2d[i][0] - u) ** 2)
        f.append(weights[i][1] ** 2 * (x2d[i][1] - v) ** 2)
    n2 = q @ q - 1
    f.append(1e8 * n2 ** 2 + 1e8 * n2 ** 2)
    return np.array(f)


def gn_dfun(value, x2d, x3d, weights, camera):
    """LM.py:163-221: the Jacobian of ``gn_fun`` (closed form of the reference's sympy expressions), INCLUDING the
    reference's slip: in every y row the d/dqz entry is a copy of the d/dqy entry (LM.py:196-197 are the same expression;
    the fixture generated from the reference's own ``dfun`` pins it)."""
    fx, cx, fy, cy = camera[0][0], camera[0][2], camera[1][1], camera[1][2]
    q, t = np.asarray(value[:4], np.float64), np.asarray(value[4:], np.float64)
    qw, qx, qy, qz = q
    rows = []
    for i in range(len(x2d)):
        x, y, z = x3d[i]
        X = _rot_point(q, x3d[i]) + t
        # d(rotated point)/dq: 3 x 4
        a = qw * x + qy * z - qz * y
        b = qw * y - qx * z + qz * x
        c = qw * z + qx * y - qy * x
        d = -qx * x - qy * y - qz * z
        dX = np.array([[2 * a, -2 * d, 2 * c, -2 * b],
                       [2 * b, -2 * c, -2 * d, 2 * a],
                       [2 * c, 2 * b, -2 * a, -2 * d]])
        for (f_, c_, k, w_, m) in ((fx, cx, 0, weights[i][0], x2d[i][0]), (fy, cy, 1, weights[i][1], x2d[i][1])):
            num = c_ * X[2] + f_ * X[k]
            pr = num / X[2]
            dnum_q = c_ * dX[2] + f_ * dX[k]
            dpr_q = dnum_q / X[2] - num * dX[2] / X[2] ** 2
            g = w_ ** 2 * (m - pr)
            row = list(g * (-2 * dpr_q))
            if k == 1:
                row[3] = row[2]            # the reference's y rows repeat the d/dqy entry in the d/dqz slot (LM.py:196-197): kept
            dt = [0.0, 0.0, 0.0]
            dt[k] = -2 * f_ * g / X[2]
            dt[2] = g * (-2 * c_ / X[2] + 2 * num / X[2] ** 2)
            rows.append(row + dt)
    n2 = q @ q - 1
    rows.append([8e8 * qw * n2, 8e8 * qx * n2, 8e8 * qy * n2, 8e8 * qz * n2, 0, 0, 0])
    return np.array(rows)


def gn_refine(value, x2d, x3d, weights, camera, max_iter=200):
    """LM.py:224-234 ``GN``: value <- value - (float32(J^T J) + 1e-4 I)^-1 J^T f while sum|delta| > 1e-4, <= 200 times."""
    value = np.asarray(value, np.float64)
    i = 0
    delta = np.ones(7) * 100
    while np.sum(np.abs(delta)) > 1e-4 and i < max_iter:
        df = gn_dfun(value, x2d, x3d, weights, camera)
        f = gn_fun(value, x2d, x3d, weights, camera)
        H = np.array(df.T @ df, dtype="float32") + 1e-4 * np.identity(7)
        value1 = value - np.linalg.inv(H) @ df.T @ f
        delta = value1 - value
        value = value1
        i += 1
    return value

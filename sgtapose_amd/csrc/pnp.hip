// Batched pose step downstream of the key-points, one thread per frame, fp64:
//   * solve_pnp   sgtapose/geometric_vision.py:43-116  = cv2.solvePnP(EPNP) + cv2.solvePnP(ITERATIVE, useExtrinsicGuess)
//   * is_pnp      sgtapose/geometric_vision.py:283-310 = that pose applied to the next frame's 3-D key-points
//   * GN          sgtapose/rf_tools/LM.py:128-234      = the weighted Gauss-Newton the reference also ships as extern "C" LM
// OpenCV is not in the image: EPnP follows the published algorithm (Lepetit et al. 2009; PCA control points, 12x12 null
// space by cyclic Jacobi, three beta initialisations + 5 Gauss-Newton steps, Horn alignment, best of three by
// reprojection error, on K-normalised points) and the ITERATIVE stage is a Levenberg-Marquardt descent of the pixel
// reprojection error over (rvec, tvec) from that guess -- the same least-squares minimum the library converges to.
// A 7-point problem is microseconds of arithmetic; the batch dimension is the parallelism (B threads), which keeps the
// pose of every frame of a batch on the device between the decode and the next frame's prior rendering.
#include "common.h"

#define PNP_MAXK 16

namespace {

template <int N>
__device__ void jacobi_eig(double A[N][N], double V[N][N]) {          // A symmetric -> diag(A) eigenvalues, columns of V
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < N; ++i) {
      diag += A[i][i] * A[i][i];
      for (int j = i + 1; j < N; ++j) off += A[i][j] * A[i][j];
    }
    if (off <= 1e-30 * diag || off == 0.0) break;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < N; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < N; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
}

template <int N>
__device__ bool solve_n(double A[N][N], double b[N]) {                   // Gaussian elimination, partial pivoting; b <- x
  for (int c = 0; c < N; ++c) {
    int piv = c;
    for (int r = c + 1; r < N; ++r)
      if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
    if (A[piv][c] == 0.0 || !isfinite(A[piv][c])) return false;
    if (piv != c) {
      for (int k = 0; k < N; ++k) { const double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
      const double t = b[c]; b[c] = b[piv]; b[piv] = t;
    }
    for (int r = c + 1; r < N; ++r) {
      const double f = A[r][c] / A[c][c];
      for (int k = c; k < N; ++k) A[r][k] -= f * A[c][k];
      b[r] -= f * b[c];
    }
  }
  for (int c = N - 1; c >= 0; --c) {
    double s = b[c];
    for (int k = c + 1; k < N; ++k) s -= A[c][k] * b[k];
    b[c] = s / A[c][c];
  }
  return true;
}

// least squares min |A x - b| for a 6 x M system through the normal equations (fp64, M <= 5)
template <int M>
__device__ bool lstsq6(const double A[6][M], const double b[6], double x[M]) {
  double N_[M][M], r[M];
  for (int i = 0; i < M; ++i) {
    r[i] = 0.0;
    for (int k = 0; k < 6; ++k) r[i] += A[k][i] * b[k];
    for (int j = 0; j < M; ++j) {
      double s = 0.0;
      for (int k = 0; k < 6; ++k) s += A[k][i] * A[k][j];
      N_[i][j] = s;
    }
    N_[i][i] += 1e-18 * (N_[i][i] + 1e-300);
  }
  if (!solve_n<M>(N_, r)) return false;
  for (int i = 0; i < M; ++i) x[i] = r[i];
  return true;
}

__device__ void rodrigues(const double r[3], double R[3][3]) {
  const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (th < 1e-12) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R[i][j] = i == j;
    return;
  }
  const double k[3] = {r[0] / th, r[1] / th, r[2] / th}, s = sin(th), c = 1.0 - cos(th);
  const double Kx[3][3] = {{0, -k[2], k[1]}, {k[2], 0, -k[0]}, {-k[1], k[0], 0}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double k2 = 0.0;
      for (int m = 0; m < 3; ++m) k2 += Kx[i][m] * Kx[m][j];
      R[i][j] = (i == j) + s * Kx[i][j] + c * k2;
    }
}

__device__ void rvec_from_R(const double R[3][3], double r[3]) {
  double c = (R[0][0] + R[1][1] + R[2][2] - 1.0) / 2.0;
  c = fmin(1.0, fmax(-1.0, c));
  const double th = acos(c);
  if (th < 1e-10) { r[0] = r[1] = r[2] = 0.0; return; }
  if (M_PI - th < 1e-6) {
    double A[3][3], ax[3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A[i][j] = (R[i][j] + (i == j)) / 2.0;
    int im = 0;
    for (int i = 0; i < 3; ++i) { ax[i] = sqrt(fmax(A[i][i], 0.0)); if (ax[i] > ax[im]) im = i; }
    double n = 0.0;
    for (int j = 0; j < 3; ++j) { r[j] = A[im][j] / ax[im]; n += r[j] * r[j]; }
    n = sqrt(n);
    for (int j = 0; j < 3; ++j) r[j] = r[j] / n * th;
    return;
  }
  const double f = th / (2.0 * sin(th));
  r[0] = (R[2][1] - R[1][2]) * f; r[1] = (R[0][2] - R[2][0]) * f; r[2] = (R[1][0] - R[0][1]) * f;
}

// Horn / Kabsch: R, t with pcs ~ R pws + t
__device__ void horn(const double (*pcs)[3], const double (*pws)[3], int n, double R[3][3], double t[3]) {
  double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 3; ++j) { pc0[j] += pcs[i][j]; pw0[j] += pws[i][j]; }
  for (int j = 0; j < 3; ++j) { pc0[j] /= n; pw0[j] /= n; }
  double H[3][3] = {};
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) H[j][k] += (pcs[i][j] - pc0[j]) * (pws[i][k] - pw0[k]);
  // H = U S V^T: V, S^2 from H^T H; U = H V / S (third column by the cross product -> proper handling of det and rank 2)
  double HtH[3][3], V[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += H[k][i] * H[k][j];
      HtH[i][j] = s;
    }
  jacobi_eig<3>(HtH, V);
  int o[3] = {0, 1, 2};
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (HtH[o[b]][o[b]] > HtH[o[a]][o[a]]) { const int tmp = o[a]; o[a] = o[b]; o[b] = tmp; }
  double v[3][3], u[3][3];                                    // rows = singular vectors, descending singular value
  for (int a = 0; a < 3; ++a)
    for (int k = 0; k < 3; ++k) v[a][k] = V[k][o[a]];
  for (int a = 0; a < 2; ++a) {
    double n2 = 0.0;
    for (int k = 0; k < 3; ++k) {
      u[a][k] = H[k][0] * v[a][0] + H[k][1] * v[a][1] + H[k][2] * v[a][2];
      n2 += u[a][k] * u[a][k];
    }
    n2 = sqrt(n2);
    for (int k = 0; k < 3; ++k) u[a][k] /= n2;
  }
  {                                                             // re-orthogonalise u1 against u0
    double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2], n2 = 0.0;
    for (int k = 0; k < 3; ++k) { u[1][k] -= d * u[0][k]; n2 += u[1][k] * u[1][k]; }
    n2 = sqrt(n2);
    for (int k = 0; k < 3; ++k) u[1][k] /= n2;
  }
  // third pair chosen so that det(U) det(V) = +1: R = U V^T is a proper rotation (equals flipping row 2 of the reference form)
  u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1]; u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2]; u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  v[2][0] = v[0][1] * v[1][2] - v[0][2] * v[1][1]; v[2][1] = v[0][2] * v[1][0] - v[0][0] * v[1][2]; v[2][2] = v[0][0] * v[1][1] - v[0][1] * v[1][0];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = u[0][i] * v[0][j] + u[1][i] * v[1][j] + u[2][i] * v[2][j];
  for (int i = 0; i < 3; ++i) t[i] = pc0[i] - (R[i][0] * pw0[0] + R[i][1] * pw0[1] + R[i][2] * pw0[2]);
}

__device__ double rep_error(const double R[3][3], const double t[3], const double (*pws)[3], const double (*uv)[2], int n) {
  double e = 0.0;
  for (int i = 0; i < n; ++i) {
    double X[3];
    for (int k = 0; k < 3; ++k) X[k] = R[k][0] * pws[i][0] + R[k][1] * pws[i][1] + R[k][2] * pws[i][2] + t[k];
    const double du = uv[i][0] - X[0] / X[2], dv = uv[i][1] - X[1] / X[2];
    e += sqrt(du * du + dv * dv);
  }
  return e / n;
}

__device__ bool epnp(const double (*pws)[3], const double (*uv)[2], int n, double Rb[3][3], double tb[3]) {
  double c0[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) c0[k] += pws[i][k];
  for (int k = 0; k < 3; ++k) c0[k] /= n;
  double C[3][3] = {}, Vc[3][3];
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) C[a][b] += (pws[i][a] - c0[a]) * (pws[i][b] - c0[b]);
  jacobi_eig<3>(C, Vc);
  double cws[4][3];
  for (int k = 0; k < 3; ++k) cws[0][k] = c0[k];
  for (int j = 0; j < 3; ++j) {
    const double s = sqrt(fmax(C[j][j], 0.0) / n);
    for (int k = 0; k < 3; ++k) cws[j + 1][k] = c0[k] + s * Vc[k][j];
  }
  // barycentric coordinates
  double alphas[PNP_MAXK][4];
  for (int i = 0; i < n; ++i) {
    double CC[3][3], b[3];
    for (int r = 0; r < 3; ++r) {
      for (int j = 0; j < 3; ++j) CC[r][j] = cws[j + 1][r] - cws[0][r];
      b[r] = pws[i][r] - cws[0][r];
    }
    if (!solve_n<3>(CC, b)) return false;
    alphas[i][0] = 1.0 - b[0] - b[1] - b[2];
    for (int j = 0; j < 3; ++j) alphas[i][j + 1] = b[j];
  }
  // M^T M (12 x 12) accumulated row by row
  double MtM[12][12] = {}, Vm[12][12];
  for (int i = 0; i < n; ++i) {
    double r1[12], r2[12];
    for (int j = 0; j < 4; ++j) {
      r1[3 * j] = alphas[i][j]; r1[3 * j + 1] = 0.0; r1[3 * j + 2] = -alphas[i][j] * uv[i][0];
      r2[3 * j] = 0.0; r2[3 * j + 1] = alphas[i][j]; r2[3 * j + 2] = -alphas[i][j] * uv[i][1];
    }
    for (int a = 0; a < 12; ++a)
      for (int b = 0; b < 12; ++b) MtM[a][b] += r1[a] * r1[b] + r2[a] * r2[b];
  }
  jacobi_eig<12>(MtM, Vm);
  int idx[4];                                                   // the four smallest eigenvalues, ascending
  {
    bool used[12] = {};
    for (int q = 0; q < 4; ++q) {
      int best = -1;
      for (int j = 0; j < 12; ++j)
        if (!used[j] && (best < 0 || MtM[j][j] < MtM[best][best])) best = j;
      used[best] = true;
      idx[q] = best;
    }
  }
  double v[4][4][3];                                            // v[q][control point][xyz]
  for (int q = 0; q < 4; ++q)
    for (int j = 0; j < 12; ++j) v[q][j / 3][j % 3] = Vm[j][idx[q]];
  const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
  double L[6][10], rho[6];
  for (int r = 0; r < 6; ++r) {
    double dv[4][3];
    for (int q = 0; q < 4; ++q)
      for (int k = 0; k < 3; ++k) dv[q][k] = v[q][pa[r]][k] - v[q][pb[r]][k];
    auto dot = [&](int a, int b) { return dv[a][0] * dv[b][0] + dv[a][1] * dv[b][1] + dv[a][2] * dv[b][2]; };
    L[r][0] = dot(0, 0); L[r][1] = 2 * dot(0, 1); L[r][2] = dot(1, 1); L[r][3] = 2 * dot(0, 2); L[r][4] = 2 * dot(1, 2);
    L[r][5] = dot(2, 2); L[r][6] = 2 * dot(0, 3); L[r][7] = 2 * dot(1, 3); L[r][8] = 2 * dot(2, 3); L[r][9] = dot(3, 3);
    rho[r] = 0.0;
    for (int k = 0; k < 3; ++k) rho[r] += (cws[pa[r]][k] - cws[pb[r]][k]) * (cws[pa[r]][k] - cws[pb[r]][k]);
  }
  double best = -1.0;
  for (int method = 0; method < 3; ++method) {
    double be[4] = {0, 0, 0, 0};
    if (method == 0) {
      double A[6][4], x[4];
      const int cols[4] = {0, 1, 3, 6};
      for (int r = 0; r < 6; ++r)
        for (int j = 0; j < 4; ++j) A[r][j] = L[r][cols[j]];
      if (!lstsq6<4>(A, rho, x)) continue;
      const double b0 = sqrt(fabs(x[0])), s = x[0] < 0 ? -1.0 : 1.0;
      be[0] = b0; be[1] = s * x[1] / b0; be[2] = s * x[2] / b0; be[3] = s * x[3] / b0;
    } else if (method == 1) {
      double A[6][3], x[3];
      for (int r = 0; r < 6; ++r)
        for (int j = 0; j < 3; ++j) A[r][j] = L[r][j];
      if (!lstsq6<3>(A, rho, x)) continue;
      if (x[0] < 0) { be[0] = sqrt(-x[0]); be[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
      else { be[0] = sqrt(x[0]); be[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
      if (x[1] < 0) be[0] = -be[0];
    } else {
      double A[6][5], x[5];
      for (int r = 0; r < 6; ++r)
        for (int j = 0; j < 5; ++j) A[r][j] = L[r][j];
      if (!lstsq6<5>(A, rho, x)) continue;
      if (x[0] < 0) { be[0] = sqrt(-x[0]); be[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
      else { be[0] = sqrt(x[0]); be[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
      if (x[1] < 0) be[0] = -be[0];
      be[2] = x[3] / be[0];
    }
    bool okb = true;
    for (int it = 0; it < 5 && okb; ++it) {
      double A[6][4], bb[6], dx[4];
      for (int r = 0; r < 6; ++r) {
        const double* l = L[r];
        A[r][0] = 2 * l[0] * be[0] + l[1] * be[1] + l[3] * be[2] + l[6] * be[3];
        A[r][1] = l[1] * be[0] + 2 * l[2] * be[1] + l[4] * be[2] + l[7] * be[3];
        A[r][2] = l[3] * be[0] + l[4] * be[1] + 2 * l[5] * be[2] + l[8] * be[3];
        A[r][3] = l[6] * be[0] + l[7] * be[1] + l[8] * be[2] + 2 * l[9] * be[3];
        bb[r] = rho[r] - (l[0] * be[0] * be[0] + l[1] * be[0] * be[1] + l[2] * be[1] * be[1] + l[3] * be[0] * be[2] +
                          l[4] * be[1] * be[2] + l[5] * be[2] * be[2] + l[6] * be[0] * be[3] + l[7] * be[1] * be[3] +
                          l[8] * be[2] * be[3] + l[9] * be[3] * be[3]);
      }
      okb = lstsq6<4>(A, bb, dx);
      if (okb)
        for (int j = 0; j < 4; ++j) be[j] += dx[j];
    }
    if (!okb || !isfinite(be[0] + be[1] + be[2] + be[3])) continue;
    double ccs[4][3], pcs[PNP_MAXK][3];
    for (int j = 0; j < 4; ++j)
      for (int k = 0; k < 3; ++k) ccs[j][k] = be[0] * v[0][j][k] + be[1] * v[1][j][k] + be[2] * v[2][j][k] + be[3] * v[3][j][k];
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < 3; ++k)
        pcs[i][k] = alphas[i][0] * ccs[0][k] + alphas[i][1] * ccs[1][k] + alphas[i][2] * ccs[2][k] + alphas[i][3] * ccs[3][k];
    if (pcs[0][2] < 0)
      for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) pcs[i][k] = -pcs[i][k];
    double R[3][3], t[3];
    horn(pcs, pws, n, R, t);
    const double e = rep_error(R, t, pws, uv, n);
    if (isfinite(e) && (best < 0 || e < best)) {
      best = e;
      for (int i = 0; i < 3; ++i) { tb[i] = t[i]; for (int j = 0; j < 3; ++j) Rb[i][j] = R[i][j]; }
    }
  }
  return best >= 0;
}

__device__ void resid6(const double K[4], const double p[6], const double (*pws)[3], const double (*x2d)[2], int n, double* r) {
  double R[3][3];
  rodrigues(p, R);
  for (int i = 0; i < n; ++i) {
    double X[3];
    for (int k = 0; k < 3; ++k) X[k] = R[k][0] * pws[i][0] + R[k][1] * pws[i][1] + R[k][2] * pws[i][2] + p[3 + k];
    r[2 * i] = K[0] * X[0] / X[2] + K[2] - x2d[i][0];
    r[2 * i + 1] = K[1] * X[1] / X[2] + K[3] - x2d[i][1];
  }
}

// Levenberg-Marquardt on the pixel reprojection error over (rvec, t): central-difference Jacobian, as oracle/pnp.py
__device__ void refine_lm(const double K[4], double p[6], const double (*pws)[3], const double (*x2d)[2], int n) {
  double r[2 * PNP_MAXK], r2[2 * PNP_MAXK], rp[2 * PNP_MAXK], rm[2 * PNP_MAXK], J[2 * PNP_MAXK][6];
  double lam = 1e-3;
  resid6(K, p, pws, x2d, n, r);
  double cost = 0.0;
  for (int i = 0; i < 2 * n; ++i) cost += r[i] * r[i];
  for (int it = 0; it < 60; ++it) {
    for (int k = 0; k < 6; ++k) {
      const double h = 1e-6 * fmax(1.0, fabs(p[k]));
      double q[6];
      for (int j = 0; j < 6; ++j) q[j] = p[j];
      q[k] = p[k] + h; resid6(K, q, pws, x2d, n, rp);
      q[k] = p[k] - h; resid6(K, q, pws, x2d, n, rm);
      for (int i = 0; i < 2 * n; ++i) J[i][k] = (rp[i] - rm[i]) / (2 * h);
    }
    double JtJ[6][6], g[6];
    for (int a = 0; a < 6; ++a) {
      g[a] = 0.0;
      for (int i = 0; i < 2 * n; ++i) g[a] += J[i][a] * r[i];
      for (int b = 0; b < 6; ++b) {
        double s = 0.0;
        for (int i = 0; i < 2 * n; ++i) s += J[i][a] * J[i][b];
        JtJ[a][b] = s;
      }
    }
    bool moved = false;
    double stepmax = 0.0;
    for (int tr = 0; tr < 10; ++tr) {
      double A[6][6], st[6], q[6];
      for (int a = 0; a < 6; ++a) {
        st[a] = -g[a];
        for (int b = 0; b < 6; ++b) A[a][b] = JtJ[a][b] + (a == b ? lam * JtJ[a][a] : 0.0);
      }
      if (!solve_n<6>(A, st)) { lam *= 10.0; continue; }
      for (int j = 0; j < 6; ++j) q[j] = p[j] + st[j];
      resid6(K, q, pws, x2d, n, r2);
      double c2 = 0.0;
      for (int i = 0; i < 2 * n; ++i) c2 += r2[i] * r2[i];
      if (c2 < cost) {
        stepmax = 0.0;
        for (int j = 0; j < 6; ++j) { p[j] = q[j]; stepmax = fmax(stepmax, fabs(st[j])); }
        for (int i = 0; i < 2 * n; ++i) r[i] = r2[i];
        cost = c2;
        lam = fmax(lam * 0.1, 1e-12);
        moved = true;
        break;
      }
      lam *= 10.0;
    }
    if (!moved || stepmax < 1e-13) break;
  }
}

}  // namespace

// one thread per frame
__global__ __launch_bounds__(64) void pnp_kernel(const double* __restrict__ x3d, const double* __restrict__ x2d,
                                                 const unsigned char* __restrict__ valid, const double* __restrict__ camK,
                                                 const double* __restrict__ next_x3d, unsigned char* __restrict__ ok,
                                                 double* __restrict__ t_out, double* __restrict__ q_out,
                                                 double* __restrict__ next_proj, int B, int K) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double pws[PNP_MAXK][3], px[PNP_MAXK][2], uv[PNP_MAXK][2];
  const double fx = camK[0], cx = camK[2], fy = camK[4], cy = camK[5];
  int n = 0;
  for (int k = 0; k < K; ++k) {
    const double u = x2d[((long)b * K + k) * 2], v = x2d[((long)b * K + k) * 2 + 1];
    // _get_further_dt_pnp_inputs (sgta_detector.py:469-472): rows with a coordinate above the missing marker
    const bool good = valid ? valid[(long)b * K + k] != 0 : (u > -999.999 * 4 || v > -999.999 * 4);
    if (!good) continue;
    for (int j = 0; j < 3; ++j) pws[n][j] = x3d[((long)b * K + k) * 3 + j];
    px[n][0] = u; px[n][1] = v;
    uv[n][0] = (u - cx) / fx; uv[n][1] = (v - cy) / fy;
    ++n;
  }
  bool good = n >= 4;
  double R[3][3], t[3], p[6] = {0, 0, 0, 0, 0, 0};
  if (good) good = epnp(pws, uv, n, R, t);
  if (good) {
    rvec_from_R(R, p);
    for (int j = 0; j < 3; ++j) p[3 + j] = t[j];
    const double Kc[4] = {fx, fy, cx, cy};
    refine_lm(Kc, p, pws, px, n);
    for (int j = 0; j < 6; ++j) good = good && isfinite(p[j]);
  }
  ok[b] = good ? 1 : 0;
  double q[4] = {0, 0, 0, 1};
  if (good) {
    const double th = sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    if (th >= 1e-12) {
      const double s = sin(th / 2) / th;
      q[0] = p[0] * s; q[1] = p[1] * s; q[2] = p[2] * s; q[3] = cos(th / 2);
    }
  }
  for (int j = 0; j < 3; ++j) t_out[(long)b * 3 + j] = good ? p[3 + j] : 0.0;
  for (int j = 0; j < 4; ++j) q_out[(long)b * 4 + j] = q[j];
  if (!next_proj) return;
  // is_pnp (geometric_vision.py:283-310): on failure the detections themselves stand in for the projections
  double Rq[3][3];
  if (good) rodrigues(p, Rq);
  for (int k = 0; k < K; ++k) {
    double* o = next_proj + ((long)b * K + k) * 2;
    if (!good || !next_x3d) { o[0] = x2d[((long)b * K + k) * 2]; o[1] = x2d[((long)b * K + k) * 2 + 1]; continue; }
    const double* X = next_x3d + ((long)b * K + k) * 3;
    double Y[3];
    for (int j = 0; j < 3; ++j) Y[j] = Rq[j][0] * X[0] + Rq[j][1] * X[1] + Rq[j][2] * X[2] + p[3 + j];
    const double z = Y[2];                                      // camera_K @ aligned, then / z
    o[0] = (fx * Y[0] + cx * z) / z;
    o[1] = (fy * Y[1] + cy * z) / z;
  }
}

extern "C" int sgta_pnp_batched(const double* x3d, const double* x2d, const unsigned char* valid, const double* camK,
                                const double* next_x3d, unsigned char* ok, double* t_out, double* quat_xyzw,
                                double* next_proj, int B, int K, hipStream_t st) {
  if (!x3d || !x2d || !camK || !ok || !t_out || !quat_xyzw || B <= 0 || K < 4 || K > PNP_MAXK) return SGTA_EBADARG;
  hipLaunchKernelGGL(pnp_kernel, dim3((B + 63) / 64), dim3(64), 0, st, x3d, x2d, valid, camK, next_x3d, ok, t_out, quat_xyzw,
                     next_proj, B, K);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

// ---- rf_tools/LM.py GN (fun :128-158, dfun :163-221, GN :224-234), one thread per problem ---------------------------------
__device__ void gn_eval(const double v[7], const double (*x2d)[2], const double (*x3d)[3], const double (*w)[2], int n,
                        const double K[4], double* f, double (*J)[7]) {
  const double qw = v[0], qx = v[1], qy = v[2], qz = v[3];
  const double fxy[2] = {K[0], K[1]}, cxy[2] = {K[2], K[3]};
  for (int i = 0; i < n; ++i) {
    const double x = x3d[i][0], y = x3d[i][1], z = x3d[i][2];
    const double a = qw * x + qy * z - qz * y, b = qw * y - qx * z + qz * x, c = qw * z + qx * y - qy * x,
                 d = -qx * x - qy * y - qz * z;
    const double X[3] = {qw * a - qx * d + qy * c - qz * b + v[4], qw * b - qx * c - qy * d + qz * a + v[5],
                         qw * c + qx * b - qy * a - qz * d + v[6]};
    const double dX[3][4] = {{2 * a, -2 * d, 2 * c, -2 * b}, {2 * b, -2 * c, -2 * d, 2 * a}, {2 * c, 2 * b, -2 * a, -2 * d}};
    for (int k = 0; k < 2; ++k) {
      const double num = cxy[k] * X[2] + fxy[k] * X[k], pr = num / X[2];
      const double wk = w[i][k] * w[i][k], g = wk * (x2d[i][k] - pr);
      f[2 * i + k] = wk * (x2d[i][k] - pr) * (x2d[i][k] - pr);
      double* row = J[2 * i + k];
      for (int m = 0; m < 4; ++m) {
        const double dnum = cxy[k] * dX[2][m] + fxy[k] * dX[k][m];
        row[m] = g * (-2.0 * (dnum / X[2] - num * dX[2][m] / (X[2] * X[2])));
      }
      if (k == 1) row[3] = row[2];                                // the reference's y rows repeat d/dqy in the d/dqz slot
      row[4] = row[5] = 0.0;
      row[4 + k] = -2.0 * fxy[k] * g / X[2];
      row[6] = g * (-2.0 * cxy[k] / X[2] + 2.0 * num / (X[2] * X[2]));
    }
  }
  const double n2 = qw * qw + qx * qx + qy * qy + qz * qz - 1.0;
  f[2 * n] = 1e8 * n2 * n2 + 1e8 * n2 * n2;
  double* row = J[2 * n];
  row[0] = 8e8 * qw * n2; row[1] = 8e8 * qx * n2; row[2] = 8e8 * qy * n2; row[3] = 8e8 * qz * n2;
  row[4] = row[5] = row[6] = 0.0;
}

__global__ __launch_bounds__(64) void gn_kernel(const double* __restrict__ value0, const double* __restrict__ x2d,
                                                const double* __restrict__ x3d, const unsigned char* __restrict__ valid,
                                                const double* __restrict__ weights, const double* __restrict__ camK,
                                                double* __restrict__ out, int* __restrict__ iters, int B, int K, int max_iter) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double p2[PNP_MAXK][2], p3[PNP_MAXK][3], w[PNP_MAXK][2], f[2 * PNP_MAXK + 1], J[2 * PNP_MAXK + 1][7];
  int n = 0;
  for (int k = 0; k < K; ++k) {
    if (valid && !valid[(long)b * K + k]) continue;
    for (int j = 0; j < 2; ++j) { p2[n][j] = x2d[((long)b * K + k) * 2 + j]; w[n][j] = weights ? weights[((long)b * K + k) * 2 + j] : 1.0; }
    for (int j = 0; j < 3; ++j) p3[n][j] = x3d[((long)b * K + k) * 3 + j];
    ++n;
  }
  const double Kc[4] = {camK[0], camK[4], camK[2], camK[5]};
  double v[7];
  for (int j = 0; j < 7; ++j) v[j] = value0[(long)b * 7 + j];
  double dsum = 700.0;
  int it = 0;
  while (dsum > 1e-4 && it < max_iter) {
    gn_eval(v, p2, p3, w, n, Kc, f, J);
    const int m = 2 * n + 1;
    double H[7][7], Hi[7][7];
    for (int a = 0; a < 7; ++a)
      for (int c = 0; c < 7; ++c) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += J[i][a] * J[i][c];
        H[a][c] = (double)(float)s + (a == c ? 1e-4 : 0.0);       // np.array(J^T J, dtype='float32') + 1e-4 I
      }
    // inverse by Gauss-Jordan (np.linalg.inv), then (H^-1 J^T) f
    for (int a = 0; a < 7; ++a)
      for (int c = 0; c < 7; ++c) Hi[a][c] = a == c;
    bool sing = false;
    for (int c = 0; c < 7 && !sing; ++c) {
      int piv = c;
      for (int r = c + 1; r < 7; ++r)
        if (fabs(H[r][c]) > fabs(H[piv][c])) piv = r;
      if (H[piv][c] == 0.0 || !isfinite(H[piv][c])) { sing = true; break; }
      for (int k = 0; k < 7; ++k) {
        double tmp = H[c][k]; H[c][k] = H[piv][k]; H[piv][k] = tmp;
        tmp = Hi[c][k]; Hi[c][k] = Hi[piv][k]; Hi[piv][k] = tmp;
      }
      const double d = H[c][c];
      for (int k = 0; k < 7; ++k) { H[c][k] /= d; Hi[c][k] /= d; }
      for (int r = 0; r < 7; ++r) {
        if (r == c) continue;
        const double fct = H[r][c];
        for (int k = 0; k < 7; ++k) { H[r][k] -= fct * H[c][k]; Hi[r][k] -= fct * Hi[c][k]; }
      }
    }
    dsum = 0.0;
    for (int a = 0; a < 7; ++a) {
      double s = 0.0;
      if (sing) s = NAN;                                           // numpy raises on a singular matrix: reported as NaN
      else
        for (int i = 0; i < m; ++i) {
          double hj = 0.0;
          for (int c = 0; c < 7; ++c) hj += Hi[a][c] * J[i][c];
          s += hj * f[i];
        }
      v[a] -= s;
      dsum += fabs(s);
    }
    ++it;
    if (!(dsum == dsum)) break;                                    // NaN: the python loop condition fails, too
  }
  for (int j = 0; j < 7; ++j) out[(long)b * 7 + j] = v[j];
  if (iters) iters[b] = it;
}

extern "C" int sgta_gn_refine(const double* value0, const double* x2d, const double* x3d, const unsigned char* valid,
                              const double* weights, const double* camK, double* out, int* iters, int B, int K,
                              int max_iter, hipStream_t st) {
  if (!value0 || !x2d || !x3d || !camK || !out || B <= 0 || K < 1 || K > PNP_MAXK || max_iter < 0) return SGTA_EBADARG;
  hipLaunchKernelGGL(gn_kernel, dim3((B + 63) / 64), dim3(64), 0, st, value0, x2d, x3d, valid, weights, camK, out, iters, B, K,
                     max_iter);
  SGTA_LAUNCH_CHECK();
  return SGTA_OK;
}

"""Robust two-view geometry on the host without OpenCV (numpy, batched minimal solvers).

The reference calls OpenCV for the pose half of its metric and for the demo's geometry filter:
  * tools/metrics.py:77-103   cv2.findEssentialMat(RANSAC, threshold, prob) + cv2.recoverPose(E, ..., 1e9, mask)
  * demo.py:514-517           cv2.findFundamentalMat(USAC_MAGSAC, 1.0 px, 0.999999, 10000)
OpenCV (`opencv-python`, a pip dependency of the reference, not vendored under /root/reference) is absent from the build
container and from the GPU boxes, so `gim_amd.zeb.estimate_pose` / `gim_amd.demo` could not run their robust-fitting step at
all.  north_star keeps RANSAC on the host; this module is that host step when cv2 does not import (cv2 stays the first choice
when it does: `GIM_POSE_BACKEND=auto|cv2|numpy`).

What is restated (OpenCV 4.x, modules/calib3d/src/five-point.cpp and ptsetreg.cpp, published algorithms):
  * five-point essential matrix (Nister 2004 in the Stewenius-Engels-Nister 2006 formulation): null space of the 5 x 9
    epipolar constraints, the ten cubic constraints det(E) = 0 and 2 E E^T E - tr(E E^T) E = 0 on E = xX + yY + zZ + W,
    Gauss-Jordan on the 10 x 20 coefficient matrix, 10 x 10 action matrix of multiplication by x, real eigenvectors
    -> up to 10 candidates;
  * RANSAC as `RANSACPointSetRegistrator::run`: 5-point samples, Sampson error (x1^T E x0)^2 / (|E x0|_xy^2 + |E^T x1|_xy^2)
    against threshold^2, the model with the most inliers wins, the iteration bound shrinks with
    log(1 - conf) / log(1 - w^5) (at most 1000 iterations, cv2.findEssentialMat's default maxIters), NO final re-fit;
  * `recoverPose`: SVD decomposition into (R1 | R2, +-t), DLT triangulation of the masked points, the candidate with the most
    points in front of both cameras (depth < distanceThresh) wins; its count is returned.
  * seven-point fundamental matrix inside the same RANSAC loop for the demo (plain RANSAC with the Sampson distance in pixels --
    OpenCV's USAC_MAGSAC scoring is NOT restated; the inlier mask is RANSAC's, which is what the demo prints and draws).

Parity: UNPINNED against cv2 (no OpenCV here to record vectors from; sampling is random in both).  tests/test_pose_cpu.py pins
the solvers on exact synthetic geometry (every ground-truth E / F is among the candidates to 1e-9, recovered poses within 1e-6
of the truth on noise-free data, sub-degree on noisy data with 50 % outliers) and compares with cv2 under `importorskip`.
Batched numpy: 250 samples (2 500 candidate models) are solved and scored per step; a 2 000-match pair takes ~0.2-0.5 s.
"""
import itertools
import os

import numpy as np

# ---- polynomial bookkeeping for the five-point constraints -------------------------------------------------------------
# monomials of degree <= 3 in (x, y, z): the ten cubic ones first (eliminated by Gauss-Jordan), then the basis of the quotient ring
_MONO = [(3, 0, 0), (2, 1, 0), (2, 0, 1), (1, 2, 0), (1, 1, 1), (1, 0, 2), (0, 3, 0), (0, 2, 1), (0, 1, 2), (0, 0, 3),
         (2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]
_IDX = {m: i for i, m in enumerate(_MONO)}


# degree <= 2 monomials (products of two linear polynomials), in _MONO's order of them
_MONO2 = [m for m in _MONO if sum(m) <= 2]
_IDX2 = {m: i for i, m in enumerate(_MONO2)}
_LIN1 = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]                          # a linear polynomial is (x, y, z, 1) coefficients


def _tables():
    t12 = np.zeros((4, 4, len(_MONO2)))                                      # linear x linear -> degree <= 2
    for (i, a), (j, b) in itertools.product(enumerate(_LIN1), repeat=2):
        t12[i, j, _IDX2[(a[0] + b[0], a[1] + b[1], a[2] + b[2])]] = 1.0
    t23 = np.zeros((len(_MONO2), 4, 20))                                     # degree <= 2 x linear -> degree <= 3
    for (i, a), (j, b) in itertools.product(enumerate(_MONO2), enumerate(_LIN1)):
        t23[i, j, _IDX[(a[0] + b[0], a[1] + b[1], a[2] + b[2])]] = 1.0
    return t12.reshape(16, -1), t23.reshape(4 * len(_MONO2), 20)


_T12, _T23 = _tables()


def _mul11(a, b):
    """linear x linear (coefficients over x, y, z, 1) -> coefficients over _MONO2; batched over leading axes"""
    return (a[..., :, None] * b[..., None, :]).reshape(*a.shape[:-1], 16) @ _T12


def _mul21(a, b):
    """degree <= 2 (over _MONO2) x linear -> coefficients over _MONO"""
    return (a[..., :, None] * b[..., None, :]).reshape(*a.shape[:-1], a.shape[-1] * 4) @ _T23


def _nullspace(A, k):
    """last k right singular vectors of every matrix of the batch A [N, r, 9] (r < 9: full_matrices gives the whole basis)"""
    _, _, vt = np.linalg.svd(A, full_matrices=True)
    return vt[:, 9 - k:, :]


def _epipolar_rows(x0, x1):
    """rows of the linear system x1^T M x0 = 0 in the row-major entries of M; x0, x1 [..., 2] (homogeneous 1 appended)"""
    o = np.ones_like(x0[..., :1])
    h0 = np.concatenate([x0, o], -1)
    h1 = np.concatenate([x1, o], -1)
    return (h1[..., :, None] * h0[..., None, :]).reshape(*x0.shape[:-1], 9)


def five_point(x0, x1):
    """Essential matrices through 5 correspondences.  x0, x1: [N, 5, 2] normalised image points (x1^T E x0 = 0).
    -> (E [N, 10, 3, 3], valid [N, 10]): up to ten real solutions per sample, each scaled to unit Frobenius norm."""
    x0 = np.asarray(x0, dtype=np.float64)
    x1 = np.asarray(x1, dtype=np.float64)
    n = x0.shape[0]
    basis = _nullspace(_epipolar_rows(x0, x1), 4).reshape(n, 4, 3, 3)       # X, Y, Z, W
    # E = x X + y Y + z Z + W: every entry a linear polynomial, coefficients (x, y, z, 1)
    Ep = np.moveaxis(basis, 1, -1)                                           # [N, 3, 3, 4]
    e = lambda i, j: Ep[:, i, j]   # noqa: E731

    def m2(a, b, c, d):   # a b - c d
        return _mul11(a, b) - _mul11(c, d)
    det = (_mul21(m2(e(1, 1), e(2, 2), e(1, 2), e(2, 1)), e(0, 0)) -
           _mul21(m2(e(1, 0), e(2, 2), e(1, 2), e(2, 0)), e(0, 1)) +
           _mul21(m2(e(1, 0), e(2, 1), e(1, 1), e(2, 0)), e(0, 2)))
    # E E^T (degree 2), its trace, then 2 (E E^T) E - tr(E E^T) E
    EEt = [[None] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(i, 3):
            EEt[i][j] = EEt[j][i] = _mul11(e(i, 0), e(j, 0)) + _mul11(e(i, 1), e(j, 1)) + _mul11(e(i, 2), e(j, 2))
    tr = EEt[0][0] + EEt[1][1] + EEt[2][2]
    rows = [det]
    for i in range(3):
        for j in range(3):
            sij = _mul21(EEt[i][0], e(0, j)) + _mul21(EEt[i][1], e(1, j)) + _mul21(EEt[i][2], e(2, j))
            rows.append(2.0 * sij - _mul21(tr, e(i, j)))
    M = np.stack(rows, 1)                                                   # [N, 10, 20]
    # Gauss-Jordan on the cubic block: [I | B]
    A, Bm = M[:, :, :10], M[:, :, 10:]
    ok = np.ones(n, dtype=bool)
    try:
        B = np.linalg.solve(A, Bm)
    except np.linalg.LinAlgError:
        B = np.zeros_like(Bm)
        for s in range(n):
            try:
                B[s] = np.linalg.solve(A[s], Bm[s])
            except np.linalg.LinAlgError:
                ok[s] = False
    ok &= np.isfinite(B).all((1, 2))
    B = np.where(ok[:, None, None], B, 0.0)
    # action matrix of multiplication by x on the basis [x^2, xy, xz, y^2, yz, z^2, x, y, z, 1]
    Act = np.zeros((n, 10, 10))
    Act[:, :6] = -B[:, :6]
    Act[:, 6, 0] = Act[:, 7, 1] = Act[:, 8, 2] = Act[:, 9, 6] = 1.0
    ev, vec = np.linalg.eig(Act)                                            # columns of vec: the basis evaluated at a solution
    real = (np.abs(ev.imag) < 1e-9 * (1.0 + np.abs(ev.real))) & ok[:, None]
    v = vec.real
    w = v[:, 9, :]
    real &= np.abs(w) > 1e-14
    w = np.where(real, w, 1.0)
    xyz = np.stack([v[:, 6, :] / w, v[:, 7, :] / w, v[:, 8, :] / w], -1)    # [N, 10, 3]
    E = (xyz[..., 0, None, None] * basis[:, None, 0] + xyz[..., 1, None, None] * basis[:, None, 1] +
         xyz[..., 2, None, None] * basis[:, None, 2] + basis[:, None, 3])
    nrm = np.linalg.norm(E.reshape(n, 10, 9), axis=-1)
    real &= np.isfinite(nrm) & (nrm > 0)
    E = np.where(real[..., None, None], E / np.where(real, nrm, 1.0)[..., None, None], 0.0)
    return E, real


def seven_point(x0, x1):
    """Fundamental matrices through 7 correspondences.  x0, x1: [N, 7, 2] (any units; x1^T F x0 = 0).
    -> (F [N, 3, 3, 3], valid [N, 3]): the real roots of det(a F1 + (1 - a) F2) = 0."""
    x0 = np.asarray(x0, dtype=np.float64)
    x1 = np.asarray(x1, dtype=np.float64)
    n = x0.shape[0]
    ns = _nullspace(_epipolar_rows(x0, x1), 2).reshape(n, 2, 3, 3)
    F1, F2 = ns[:, 0], ns[:, 1]
    # det(a F1 + (1 - a) F2) is a cubic in a: interpolate it at four abscissae
    aa = np.array([0.0, 1.0, -1.0, 2.0])
    d = np.stack([np.linalg.det(a * F1 + (1.0 - a) * F2) for a in aa], -1)  # [N, 4]
    V = np.vander(aa, 4, increasing=True)                                   # d = V c
    c = np.linalg.solve(V, d.T).T                                           # c0 + c1 a + c2 a^2 + c3 a^3
    lead = c[:, 3]
    good = np.abs(lead) > 1e-14 * (np.abs(c).max(1) + 1e-300)
    cm = np.zeros((n, 3, 3))
    ld = np.where(good, lead, 1.0)
    cm[:, 0, 2] = -c[:, 0] / ld
    cm[:, 1, 2] = -c[:, 1] / ld
    cm[:, 2, 2] = -c[:, 2] / ld
    cm[:, 1, 0] = cm[:, 2, 1] = 1.0
    roots = np.linalg.eigvals(cm)                                           # [N, 3]
    real = (np.abs(roots.imag) < 1e-9 * (1.0 + np.abs(roots.real))) & good[:, None]
    a = roots.real
    F = a[..., None, None] * F1[:, None] + (1.0 - a)[..., None, None] * F2[:, None]
    nrm = np.linalg.norm(F.reshape(n, 3, 9), axis=-1)
    real &= np.isfinite(nrm) & (nrm > 0)
    F = np.where(real[..., None, None], F / np.where(real, nrm, 1.0)[..., None, None], 0.0)
    return F, real


def sampson_error(M, x0, x1):
    """(x1^T M x0)^2 / (|M x0|_xy^2 + |M^T x1|_xy^2) for models M [..., 3, 3] and points [P, 2] -> [..., P]
    (EMEstimatorCallback::computeError / FMEstimatorCallback::computeError)"""
    o = np.ones((x0.shape[0], 1))
    h0t = np.concatenate([x0, o], 1).T                                      # [3, P]
    h1t = np.concatenate([x1, o], 1).T
    Mx0 = M @ h0t                                                           # [..., 3, P]
    Mtx1 = np.swapaxes(M, -1, -2) @ h1t
    num = (Mx0 * h1t).sum(-2) ** 2
    den = Mx0[..., 0, :] ** 2 + Mx0[..., 1, :] ** 2 + Mtx1[..., 0, :] ** 2 + Mtx1[..., 1, :] ** 2
    return num / np.maximum(den, 1e-300)


def _count_inliers(Ms, valid, x0, x1, thr2, chunk=128):
    """inlier counts of the models Ms [K, 3, 3] over all points, in chunks whose temporaries stay cache-resident"""
    out = np.zeros(Ms.shape[0], dtype=np.int64)
    for c0 in range(0, Ms.shape[0], chunk):
        out[c0:c0 + chunk] = (sampson_error(Ms[c0:c0 + chunk], x0, x1) <= thr2).sum(1)
    return np.where(valid, out, 0)


def _ransac(x0, x1, solver, m, thr, conf, max_iters, rng, s0=None, s1=None, batch=250):
    """RANSACPointSetRegistrator::run with `solver` on samples of m points; the error is evaluated on (x0, x1), the solver sees
    (s0, s1) when given (normalised copies of the same points).  -> (model [3,3] or None, mask [P] bool)"""
    P = x0.shape[0]
    if P < m:
        return None, np.zeros(P, dtype=bool)
    s0 = x0 if s0 is None else s0
    s1 = x1 if s1 is None else s1
    thr2 = float(thr) ** 2
    lconf = np.log(max(1.0 - conf, 1e-300))
    best_n, best_M = 0, None
    niters, done = int(max_iters), 0
    while done < niters:
        nb = min(batch, niters - done)
        # m distinct indices per sample: the m smallest of P random keys
        idx = np.argsort(rng.random((nb, P)), axis=1)[:, :m] if P <= 4096 else \
            np.stack([rng.choice(P, m, replace=False) for _ in range(nb)])
        Ms, valid = solver(s0[idx], s1[idx])
        k = Ms.shape[1]
        Ms = Ms.reshape(-1, 3, 3)
        per = _count_inliers(Ms, valid.reshape(-1), x0, x1, thr2).reshape(nb, k)
        # walk the batch in sample order so that the shrinking iteration bound behaves like the sequential loop
        for s in range(nb):
            j = int(per[s].argmax())
            c = int(per[s, j])
            if c > max(best_n, m - 1):
                best_n, best_M = c, Ms[s * k + j]
                den = 1.0 - (c / P) ** m
                if den < 1e-12:
                    niters = min(niters, done + s + 1)
                else:
                    niters = min(niters, max(done + s + 1, int(np.ceil(lconf / np.log(den)))))
            if done + s + 1 >= niters:
                break
        done += nb
    if best_M is None:
        return None, np.zeros(P, dtype=bool)
    return best_M, sampson_error(best_M, x0, x1) <= thr2


def find_essential_mat(x0, x1, threshold, prob=0.999, max_iters=1000, seed=0):
    """cv2.findEssentialMat(x0, x1, eye(3), method=RANSAC, prob, threshold) on normalised points -> (E [3,3] | None, mask [P])"""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    return _ransac(x0, x1, five_point, 5, threshold, prob, max_iters, np.random.default_rng(seed))


def find_fundamental_mat(p0, p1, threshold=1.0, prob=0.999999, max_iters=10000, seed=0):
    """Plain RANSAC over seven-point samples with the Sampson distance in pixels (the demo's geometry filter, demo.py:514-517;
    the reference's USAC_MAGSAC scoring is not restated).  The solver works on Hartley-normalised points, the candidates are
    mapped back to pixel units for scoring.  -> (F [3,3] | None, mask [P])"""
    p0 = np.ascontiguousarray(p0, dtype=np.float64)
    p1 = np.ascontiguousarray(p1, dtype=np.float64)
    if p0.shape[0] < 7:
        return None, np.zeros(p0.shape[0], dtype=bool)

    def norm_T(p):
        c = p.mean(0)
        s = np.sqrt(2.0) / max(np.sqrt(((p - c) ** 2).sum(1)).mean(), 1e-12)
        return np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])

    T0, T1 = norm_T(p0), norm_T(p1)
    q0 = p0 * T0[0, 0] + T0[:2, 2]
    q1 = p1 * T1[0, 0] + T1[:2, 2]

    def solver(a, b):
        F, v = seven_point(a, b)
        Fp = T1.T @ F @ T0
        nrm = np.linalg.norm(Fp.reshape(*Fp.shape[:-2], 9), axis=-1)
        return Fp / np.where(nrm > 0, nrm, 1.0)[..., None, None], v

    return _ransac(p0, p1, solver, 7, threshold, prob, max_iters, np.random.default_rng(seed), s0=q0, s1=q1)


def decompose_essential(E):
    """cv::decomposeEssentialMat -> (R1, R2, t)"""
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    return U @ W @ Vt, U @ W.T @ Vt, U[:, 2].copy()


def _triangulate(P0, P1, x0, x1):
    """DLT triangulation (cv::triangulatePoints) -> homogeneous points [P, 4]"""
    A = np.stack([x0[:, 0, None] * P0[2] - P0[0], x0[:, 1, None] * P0[2] - P0[1],
                  x1[:, 0, None] * P1[2] - P1[0], x1[:, 1, None] * P1[2] - P1[1]], 1)      # [P, 4, 4]
    _, _, vt = np.linalg.svd(A)
    return vt[:, -1, :]


def recover_pose(E, x0, x1, distance_thresh=1e9, mask=None):
    """cv2.recoverPose(E, x0, x1, eye(3), distanceThresh, mask=mask) on normalised points -> (n_good, R, t, mask_out)"""
    x0 = np.asarray(x0, dtype=np.float64)
    x1 = np.asarray(x1, dtype=np.float64)
    m_in = np.ones(x0.shape[0], dtype=bool) if mask is None else np.asarray(mask).ravel() > 0
    R1, R2, t = decompose_essential(np.asarray(E, dtype=np.float64))
    P0 = np.eye(3, 4)
    cands = [(R1, t), (R2, t), (R1, -t), (R2, -t)]
    goods = []
    for R, tt in cands:
        P1 = np.concatenate([R, tt[:, None]], 1)
        Q = _triangulate(P0, P1, x0, x1)
        w = np.where(np.abs(Q[:, 3]) > 1e-300, Q[:, 3], 1e-300)
        X = Q[:, :3] / w[:, None]
        z0 = X[:, 2]
        z1 = (X @ R.T + tt)[:, 2]
        goods.append(m_in & (z0 > 0) & (z0 < distance_thresh) & (z1 > 0) & (z1 < distance_thresh))
    n = [int(g.sum()) for g in goods]
    k = 0 if (n[0] >= n[1] and n[0] >= n[2] and n[0] >= n[3]) else 1 if (n[1] >= n[2] and n[1] >= n[3]) else 2 if n[2] >= n[3] else 3
    R, tt = cands[k]
    return n[k], R, tt, goods[k]


def backend():
    """'cv2' when OpenCV imports (and GIM_POSE_BACKEND does not say 'numpy'), else 'numpy'"""
    want = os.environ.get("GIM_POSE_BACKEND", "auto")
    if want not in ("auto", "cv2", "numpy"):
        raise ValueError(f"GIM_POSE_BACKEND={want!r}: auto, cv2 or numpy")
    if want == "numpy":
        return "numpy"
    try:
        import cv2  # noqa: F401
        return "cv2"
    except ImportError:
        if want == "cv2":
            raise
        return "numpy"

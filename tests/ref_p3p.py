"""A second, structurally independent statement of the P3P hypothesis and its inlier scoring, used
only by tests to pin the HIP kernels and the C oracle from outside their shared transliteration.

It follows the reference's mathematics (utils/exe7_Ransac/p3p.m:41-217, ransacLocalization.m:52-101,
utils/exe1/projectPoints.m:1-16) but none of the other two statements' structure:
  * the quartic p3p.m:146-179 is solved by numpy.roots (companion-matrix eigenvalues), not by the
    Ferrari closed form of solveQuartic.m;
  * everything downstream of the roots runs in numpy complex arithmetic and only the real part is
    taken at the end, which is what MATLAB does (p3p.m:189-192 takes real(x); sqrt(1 - x^2) of a root
    outside [-1, 1] is imaginary and ransacLocalization.m:64-65 keeps real(R), real(C));
  * scoring divides by depth and compares the squared pixel error with 100, as the reference does,
    instead of the kernels' division-free form.
numpy.roots returns the roots in another order than solveQuartic.m, so the four solutions of one
hypothesis are compared as a set.
"""
import numpy as np


def p3p_roots(world, bearings):
    """world, bearings: 3x3, one point per COLUMN (p3p.m's convention).  Returns a list of up to 4
    (C, R_W_C) pairs (real parts), or [] for collinear world points (p3p.m:56-58)."""
    P = [world[:, k].astype(np.float64) for k in range(3)]
    f = [bearings[:, k].astype(np.float64) for k in range(3)]
    if np.linalg.norm(np.cross(P[1] - P[0], P[2] - P[0])) == 0:
        return []

    def camera_frame(f1, f2):
        e3 = np.cross(f1, f2)
        e3 = e3 / np.linalg.norm(e3)
        return np.stack([f1, np.cross(e3, f1), e3])

    T = camera_frame(f[0], f[1])
    if (T @ f[2])[2] > 0:                                   # p3p.m:78-98
        f[0], f[1] = f[1], f[0]
        P[0], P[1] = P[1], P[0]
        T = camera_frame(f[0], f[1])
    f3 = T @ f[2]
    n1 = (P[1] - P[0]) / np.linalg.norm(P[1] - P[0])
    n3 = np.cross(n1, P[2] - P[0])
    n3 = n3 / np.linalg.norm(n3)
    N = np.stack([n1, np.cross(n3, n1), n3])
    p = N @ (P[2] - P[0])
    d = np.linalg.norm(P[1] - P[0])
    f1, f2, p1, p2 = f3[0] / f3[2], f3[1] / f3[2], p[0], p[1]
    cb = float(f[0] @ f[1])
    b = np.sqrt(1.0 / (1.0 - cb * cb) - 1.0) * (-1.0 if cb < 0 else 1.0)
    # p3p.m:146-179, grouped by powers (the same polynomial, not the same expression tree)
    a4 = -(p2 ** 4) * (f2 * f2 + f1 * f1 + 1.0)
    a3 = 2.0 * (p2 ** 3) * d * (b + f2 * f2 * b - f1 * f2)
    a2 = (p2 * p2) * (-(f2 * f2) * (p1 * p1 + d * d * b * b + d * d - p2 * p2) + p2 * p2 * f1 * f1 + 2.0 * p1 * d
                      + 2.0 * f1 * f2 * p1 * d * b - p1 * p1 * f1 * f1 + 2.0 * p1 * f2 * f2 * d - d * d * b * b
                      - 2.0 * p1 * p1)
    a1 = 2.0 * p2 * d * (p1 * p1 * b + f1 * f2 * p2 * p2 - f2 * f2 * p2 * p2 * b - p1 * d * b)
    a0 = (-2.0 * f1 * f2 * p2 * p2 * p1 * d * b + f2 * f2 * p2 * p2 * d * d * (1.0 + b * b) + 2.0 * p1 ** 3 * d
          - p1 * p1 * d * d + p2 * p2 * p1 * p1 * (f2 * f2 + f1 * f1) - p1 ** 4 - 2.0 * f2 * f2 * p2 * p2 * p1 * d)
    out = []
    for x in np.roots([a4, a3, a2, a1, a0]):
        x = complex(x).real                                 # p3p.m:189-192
        cot = (-f1 * p1 / f2 - x * p2 + d * b) / (-f1 * x * p2 / f2 + p1 - d)
        cos_t = x
        sin_t = np.sqrt(complex(1.0 - x * x))
        sin_a = np.sqrt(1.0 / (cot * cot + 1.0))
        cos_a = np.sqrt(1.0 - sin_a * sin_a) * (-1.0 if cot < 0 else 1.0)
        C = d * (sin_a * b + cos_a) * np.array([cos_a, cos_t * sin_a, sin_t * sin_a], dtype=complex)
        C = P[0] + N.T @ C
        R = np.array([[-cos_a, -sin_a * cos_t, -sin_a * sin_t],
                      [sin_a, -cos_a * cos_t, -cos_a * sin_t],
                      [0.0, -sin_t, cos_t]], dtype=complex)
        R = N.T @ R.T @ T
        out.append((C.real, R.real))
    return out


def hypothesis_counts(kp_uv, X, K, sample, tol_px=10.0):
    """ransacLocalization.m:46-101 for one sample triple: kp_uv n x 2 (u, v), X n x 3.
    Returns (counts, sq_errors, poses): the inlier count and the n squared errors of each solution, and the
    poses as (R_C_W, t_C_W)."""
    idx = np.asarray(sample, dtype=np.int64)
    world = X[idx].T
    hom = np.vstack([kp_uv[idx].T, np.ones(3)])
    bear = np.linalg.solve(K, hom)
    bear = bear / np.linalg.norm(bear, axis=0)
    counts, errs, poses = [], [], []
    for C, R_W_C in p3p_roots(world, bear):
        R = R_W_C.T
        t = -R @ C
        pc = X @ R.T + t
        with np.errstate(all="ignore"):
            proj = (pc @ K.T)
            uv = proj[:, :2] / proj[:, 2:3]
            e = ((kp_uv - uv) ** 2).sum(1)
        counts.append(int(np.sum(e < tol_px * tol_px)))
        errs.append(e)
        poses.append((R, t))
    return counts, errs, poses

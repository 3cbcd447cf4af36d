"""Host-side mirrors of the reference's small geometry functions (same names and argument
meaning); the arithmetic runs in libvohip.so on the GPU.

twist2HomogMatrix / HomogMatrix2twist   utils/exe9_BA/twist2HomogMatrix.m:1, HomogMatrix2twist.m:1
Cross2Matrix / Matrix2Cross             utils/exe9_BA/Cross2Matrix.m:14, Matrix2Cross.m:14
linearTriangulation                     utils/exe6_8point/linearTriangulation.m:1
linearTriangulation_refinement          utils/linearTriangulation_refinement.m:1
"""
import numpy as np

from . import _lib
from ._lib import as_f64, colmajor, dptr


def twist2HomogMatrix(twist, ctx=None):
    """H = twist2HomogMatrix(twist): 6-vector [v; w] -> 4x4 rigid transform."""
    ctx = ctx or _lib.default_context()
    t = as_f64(np.asarray(twist).reshape(6))
    H = np.zeros(16)
    ctx.check(_lib.load().vohip_twist2homog(ctx.handle, dptr(t), 1, dptr(H)))
    return H.reshape(4, 4).T.copy()


def twist2HomogMatrix_many(twists, ctx=None):
    """[twist2HomogMatrix(t) for t in twists] in one call (twists n x 6): n x 4 x 4."""
    ctx = ctx or _lib.default_context()
    t = as_f64(np.ascontiguousarray(np.asarray(twists, dtype=np.float64).reshape(-1, 6)))
    n = t.shape[0]
    H = np.zeros(16 * max(n, 1))
    if n:
        ctx.check(_lib.load().vohip_twist2homog(ctx.handle, dptr(t), n, dptr(H)))
    return H[:16 * n].reshape(n, 4, 4).transpose(0, 2, 1).copy()


def HomogMatrix2twist(H, ctx=None):
    """twist = HomogMatrix2twist(H): 4x4 rigid transform -> 6x1 [v; w], |w| in [0, pi]."""
    ctx = ctx or _lib.default_context()
    Hc = colmajor(H)
    t = np.zeros(6)
    ctx.check(_lib.load().vohip_homog2twist(ctx.handle, dptr(Hc), 1, dptr(t)))
    return t.reshape(6, 1)


def Cross2Matrix(x):
    """M = Cross2Matrix(x): antisymmetric matrix with M*y = cross(x, y) (pure indexing)."""
    x = np.asarray(x, dtype=np.float64).reshape(3)
    return np.array([[0.0, -x[2], x[1]], [x[2], 0.0, -x[0]], [-x[1], x[0], 0.0]])


def Matrix2Cross(M):
    """x = Matrix2Cross(M) = [-M(2,3); M(1,3); -M(1,2)] (pure indexing)."""
    M = np.asarray(M, dtype=np.float64)
    return np.array([[-M[1, 2]], [M[0, 2]], [-M[0, 1]]])


def _proj(M):
    M = np.asarray(M, dtype=np.float64)
    if M.ndim == 2:
        if M.shape != (3, 4):
            raise ValueError("Projection matrices should be of size 3x4")
        return np.ascontiguousarray(M.T), 0
    if M.shape[1:] != (3, 4):
        raise ValueError("Projection matrices should be of size 3x4")
    return np.ascontiguousarray(np.transpose(M, (0, 2, 1))), 1


def linearTriangulation(p1, p2, M1, M2, ctx=None):
    """P = linearTriangulation(p1, p2, M1, M2): p 3xN homogeneous image points, M 3x4 (or
    N x 3 x 4, one matrix per point) -> P 4xN homogeneous, dehomogenised."""
    ctx = ctx or _lib.default_context()
    p1 = np.asarray(p1, dtype=np.float64)
    p2 = np.asarray(p2, dtype=np.float64)
    if p1.shape != p2.shape:
        raise ValueError("Size mismatch of input points")
    if p1.shape[0] != 3:
        raise ValueError("Arguments x1, x2 should be 3xN matrices (homogeneous coords)")
    n = p1.shape[1]
    a, b = colmajor(p1), colmajor(p2)
    m1, f1 = _proj(M1)
    m2, f2 = _proj(M2)
    P = np.zeros((max(n, 1), 4))
    ctx.check(_lib.load().vohip_triangulate(ctx.handle, dptr(a), dptr(b), dptr(m1), dptr(m2), f1, f2, n, dptr(P)))
    return P[:n].T.copy()


def linearTriangulation_refinement(P_init, p1, p2, M1, M2, ctx=None):
    """P = linearTriangulation_refinement(P_init, p1, p2, M1, M2): P_init 4xN, p >= 2 x N."""
    ctx = ctx or _lib.default_context()
    Pi = colmajor(P_init)
    a, b = colmajor(p1), colmajor(p2)
    n = Pi.shape[0]
    if a.shape[1] != b.shape[1]:
        raise ValueError("p1 and p2 differ in row count")
    m1, _ = _proj(M1)
    m2, _ = _proj(M2)
    P = np.zeros((max(n, 1), 4))
    ctx.check(_lib.load().vohip_triangulate_refine(ctx.handle, dptr(Pi), dptr(a), dptr(b), a.shape[1], dptr(m1),
                                                   dptr(m2), n, dptr(P)))
    return P[:n].T.copy()


def candidate_filter(new_xy, C_rc, P_rc, r, ctx=None):
    """keep = redundancy test of update_candidate.m:12-27.  new_xy (n_new, 2) single (x, y);
    C_rc 2xM, P_rc 2xN double (row; col).  Boolean array of length n_new."""
    ctx = ctx or _lib.default_context()
    xy = np.ascontiguousarray(np.asarray(new_xy, dtype=np.float32).reshape(-1, 2).T)
    n_new = xy.shape[1]
    Cc = colmajor(np.asarray(C_rc, dtype=np.float64).reshape(2, -1))
    Pc = colmajor(np.asarray(P_rc, dtype=np.float64).reshape(2, -1))
    keep = np.zeros(max(n_new, 1), dtype=np.uint8)
    ctx.check(_lib.load().vohip_candidate_filter(ctx.handle, xy.ctypes.data_as(_lib.c_float_p), n_new, dptr(Cc),
                                                 Cc.shape[0], dptr(Pc), Pc.shape[0], float(r),
                                                 keep.ctypes.data_as(_lib.c_uint8_p)))
    return keep[:n_new].astype(bool)


def candidate_triangulate(cur_rc, first_rc, T_first, T_W_C, K, angle_threshold, min_depth, max_depth, is_BA,
                          ctx=None):
    """Angle test + triangulation + depth filter of update_landmarks.m:27-97.
    Returns status (n,) uint8 {0 keep, 1 new landmark, 2 dropped}, X 3xn, angles (n,) degrees."""
    ctx = ctx or _lib.default_context()
    c = colmajor(np.asarray(cur_rc, dtype=np.float64).reshape(2, -1))
    f = colmajor(np.asarray(first_rc, dtype=np.float64).reshape(2, -1))
    Tf = colmajor(np.asarray(T_first, dtype=np.float64).reshape(12, -1))
    n = c.shape[0]
    status = np.zeros(max(n, 1), dtype=np.uint8)
    X = np.zeros((max(n, 1), 3))
    ang = np.zeros(max(n, 1))
    ctx.check(_lib.load().vohip_candidate_triangulate(
        ctx.handle, dptr(c), dptr(f), dptr(Tf), n, dptr(colmajor(T_W_C)), dptr(colmajor(K)), float(angle_threshold),
        float(min_depth), float(max_depth), int(bool(is_BA)), status.ctypes.data_as(_lib.c_uint8_p), dptr(X),
        dptr(ang)))
    return status[:n], X[:n].T.copy(), ang[:n]
